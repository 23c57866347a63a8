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

import threading as _threading

_PRECISION = "bf16"
_TLS = _threading.local()
MODES = ("bf16", "bf16x3", "fp16", "mixed")


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
    'mixed' : the DETECTOR -- the only producer of thresholded values (M1/predict.py:117-119: the per-frame
              silent / non-silent decisions) -- decides in 'bf16x3', everything else runs in 'fp16'.  In training, and
              in inference with SOS_MIXED_TWO_PASS=0, the whole detector runs in bf16x3 (3x its MACs, 11 % of the
              algorithmic FLOPs): logits 1-4e-5 from the f32 reference's, decisions equal unless a logit lies within
              that of the threshold.  The inference pipeline's default is the TWO-PASS detector (pipeline.detect): every
              clip in fp16, and only the clips with a frame whose |logit| is below 9e-3 x max(1, max_t |logit|,
              max_t (|W2| a_t + |b2|)) -- three times the fp16 pass's asserted error, relative to the size of what the
              last layer sums -- again in bf16x3.  The DECISIONS are the parity detector's; the logit VALUES returned for
              unmarked clips are the fp16 pass's (1.5-3e-3 of the logit range, not 1e-5).
    A forward pass and its backward pass must run in the same mode (the networks record the mode on their tape)."""
    global _PRECISION
    if mode not in MODES:
        raise ValueError("precision must be one of " + ", ".join(repr(m) for m in MODES))
    _PRECISION = mode


def get_mode():
    """The mode set_precision() chose (may be 'mixed')."""
    return _PRECISION


def get_precision():
    """The EFFECTIVE mode of the code that asks: the innermost precision_scope of this host thread, else the global
    mode ('mixed' reads as 'fp16' outside the detector)."""
    ov = getattr(_TLS, "override", None)
    if ov is not None:
        return ov
    return "fp16" if _PRECISION == "mixed" else _PRECISION


def detector_precision():
    """Mode the silent-interval detector runs in: 'bf16x3' under 'mixed', else the effective mode."""
    if getattr(_TLS, "override", None) is None and _PRECISION == "mixed":
        return "bf16x3"
    return get_precision()


class precision_scope:
    """`with precision_scope('bf16x3'):` -- everything this host thread enqueues inside runs in that mode (thread
    local, nests); None = no change."""

    def __init__(self, mode):
        if mode is not None and mode not in ("bf16", "bf16x3", "fp16"):
            raise ValueError("precision_scope takes 'bf16', 'bf16x3', 'fp16' or None")
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_TLS, "override", None)
        if self.mode is not None:
            _TLS.override = self.mode
        return self

    def __exit__(self, *exc):
        _TLS.override = self.prev
        return False
