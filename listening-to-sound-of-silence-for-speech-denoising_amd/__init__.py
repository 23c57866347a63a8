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
              used to separate algorithmic from precision error in parity tests.
    'fp16'  : IEEE half activations/weights on the same MFMA kernels at the same rate (libsos_hip_f16.so, the
              -DSOS_F16 build of the same sources): 11 significand bits instead of 8, i.e. ~8x less rounding
              noise than 'bf16' at equal cost; training keeps the activation gradients in half's exponent
              range with a power-of-two loss scale chosen on the device (engine.GradScale).
    A forward pass and its backward pass must run in the same mode."""
    global _PRECISION
    if mode not in ("bf16", "bf16x3", "fp16"):
        raise ValueError("precision must be 'bf16', 'bf16x3' or 'fp16'")
    _PRECISION = mode


def get_precision():
    return _PRECISION
