"""MI355X-native hot path of Listening-to-Sound-of-Silence speech denoising.

Mirrors the reference's Python module API on top of hand-written HIP kernels
reached through the C ABI in include/sos_hip.h (libsos_hip.so, loaded with ctypes):

  transform            fast_stft / fast_istft / batch_fast_icRM_sigmoid / ...   (M*/transform.py)
  detector.networks    get_network() -> AudioVisualNet                          (M1/networks.py)
  denoiser.networks    get_network(config) -> JointModel                        (M2/networks.py)
  tools                convert_bitstreammask_to_audiomask, add_signals          (M2/tools.py)
  pipeline             on-device detector -> mask -> denoiser -> ISTFT chain    (M1+M2 predict.py)

There is no CPU fallback: every compute entry point raises if libsos_hip.so is missing or the
tensors are not on a GPU.
"""
__version__ = "0.1.0"

_PRECISION = "bf16"


def set_precision(mode):
    """'bf16'  : bf16 activations/weights on MFMA, fp32 accumulate (throughput mode, the
                 BASELINE.json dtype).
    'bf16x3': every activation and weight carried as a (hi, lo) bf16 pair and contracted as
              hi*hi + hi*lo + lo*hi on the same MFMA kernel (~fp32 accuracy, 3x the MACs);
              used to separate algorithmic from precision error in parity tests."""
    global _PRECISION
    if mode not in ("bf16", "bf16x3"):
        raise ValueError("precision must be 'bf16' or 'bf16x3'")
    _PRECISION = mode


def get_precision():
    return _PRECISION
