"""ctypes binding of libsos_hip.so (C ABI declared in include/sos_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# the two builds of the same sources (csrc/Makefile): storage type bfloat16 / IEEE half.  SOS_HIP_LIB overrides the
# bf16 one (A/B timing of two builds)
LIB_PATH = os.environ.get("SOS_HIP_LIB", os.path.join(_HERE, "libsos_hip.so"))
LIB_PATH_F16 = os.environ.get("SOS_HIP_LIB_F16", os.path.join(_HERE, "libsos_hip_f16.so"))

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_SIGMOID = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
DT_BF16, DT_BF16X3, DT_F32 = 0, 1, 2


class ConvDesc(C.Structure):
    """struct sos_conv_desc (include/sos_hip.h)."""
    _fields_ = [
        ("in_", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("in_cs", C.c_int32), ("cin_off", C.c_int32), ("cin", C.c_int32),
        ("in_nseg", C.c_int32), ("in_seg_stride", C.c_int32),
        ("w_gather", C.c_void_p), ("Wl", C.c_int32),
        ("wgt", C.c_void_p),
        ("kh", C.c_int32), ("kw", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32),
        ("stride", C.c_int32), ("dil_h", C.c_int32), ("dil_w", C.c_int32),
        ("pad_top", C.c_int32), ("pad_left", C.c_int32), ("pad_mode", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("out", C.c_void_p), ("out_dtype", C.c_int32),
        ("out_sb", C.c_int64), ("out_sh", C.c_int64), ("out_sw", C.c_int64), ("out_sc", C.c_int64),
        ("out_c_off", C.c_int32), ("cout_store", C.c_int32), ("out_third", C.c_int64),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("act", C.c_int32), ("act_param", C.c_void_p), ("accumulate", C.c_int32),
        ("stats", C.c_void_p), ("stats_c", C.c_int32),
        ("wl_tab", C.c_void_p), ("wo_tab", C.c_void_p), ("w_gather_stride", C.c_int32),
        ("t_frames", C.c_int32), ("t_taps", C.c_int32), ("t_pad", C.c_int32),
        ("fold_pad_out", C.c_void_p),
        ("fold_pad", C.c_int32), ("fold_H", C.c_int32), ("fold_W", C.c_int32), ("fold_sy", C.c_int32), ("fold_oy", C.c_int32),
        ("fold_sx", C.c_int32), ("fold_ox", C.c_int32), ("fold_row", C.c_int32), ("fold_third", C.c_int64),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
    ]


class View(C.Structure):
    """struct sos_view."""
    _fields_ = [("ptr", C.c_void_p), ("npix", C.c_int64), ("row", C.c_int32), ("c_off", C.c_int32),
                ("C", C.c_int32), ("x3", C.c_int32), ("third", C.c_int64)]


class WgradDesc(C.Structure):
    """struct sos_wgrad_desc."""
    _fields_ = [("g", C.c_void_p), ("B", C.c_int32), ("Hg", C.c_int32), ("Wg", C.c_int32), ("g_cs", C.c_int32),
                ("g_off", C.c_int32), ("x", C.c_void_p), ("Hx", C.c_int32), ("Wx", C.c_int32), ("x_cs", C.c_int32),
                ("x_off", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("stride", C.c_int32), ("dil_h", C.c_int32), ("dil_w", C.c_int32), ("pad_top", C.c_int32),
                ("pad_left", C.c_int32), ("pad_mode", C.c_int32), ("ksplit", C.c_int32), ("partial", C.c_void_p),
                ("dw", C.c_void_p), ("accumulate", C.c_int32), ("scale", C.c_float), ("scale_dev", C.c_void_p),
                ("t_frames", C.c_int32), ("t_taps", C.c_int32), ("t_pad", C.c_int32), ("t_cin", C.c_int32)]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
ADAM_CHUNK = 16384          # SOS_ADAM_CHUNK of include/sos_hip.h
GUARD_FLOATS = 5            # SOS_GUARD_FLOATS
EXPECTED_ABI = 10           # sos_abi_version() of the library these argument lists were written for

# name -> argtypes, exactly the prototypes of include/sos_hip.h
SIGNATURES = {
    "sos_abi_version": [],
    "sos_struct_size": [_I],
    "sos_stft_matrix_bytes": [_I, _I, _I],
    "sos_stft_pack_matrix": [_I, _I, _I, _P, _P],
    "sos_stft_f32": [_P, _L, _L, _L, _P, _P, _I, _I, _I, _P, _L, _P, _P],
    "sos_istft_matrix_bytes": [_I, _I, _I],
    "sos_istft_pack_matrix": [_I, _I, _I, _P, _P, _P],
    "sos_istft_f32": [_P, _L, _L, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P],
    "sos_power_law_f32": [_P, _L, _F, _P, _P],
    "sos_crm_apply_f32": [_P, _P, _P, _L, _L, _F, _F, _P],
    "sos_crm_apply_bwd_f32": [_P, _P, _P, _P, _L, _L, _F, _P],
    "sos_crm_target_f32": [_P, _P, _P, _L, _L, _F, _F, _P],
    "sos_bits_to_mask": [_P, _L, _L, _D, _L, _P, _P, _P, _P, _P, _P],
    "sos_threshold_bits": [_P, _L, _F, _P, _P, _P],
    "sos_logit_band_mark": [_P, _P, _L, _L, _P, _F, _P, _P, _I, _P, _P, _P],
    "sos_add_signals_f32": [_P, _P, _P, _L, _I, _L, _F, _P, _P, _P, _P],
    "sos_storage_dtype": [],
    "sos_pack_nchw_to_nhwc": [_P, _L, _I, _L, _L, _P, _I, _I, _P, _P],
    "sos_pack_nchw_wtaps": [_P, _L, _I, _L, _L, _I, _I, _I, _P, _P, _I, _I, _P, _P],
    "sos_conv2d_fwd": [C.POINTER(ConvDesc), _P],
    "sos_conv2d_tune": [C.POINTER(ConvDesc), _I, _I, C.POINTER(C.c_float), _P],
    "sos_conv2d_tune_save": [C.c_char_p],
    "sos_conv2d_tune_load": [C.c_char_p],
    "sos_conv2d_tile_count": [C.POINTER(ConvDesc)],
    "sos_lstm_pack_bytes": [_I, _I],
    "sos_lstm_pack_whh": [_P, _I, _P, _P, _P, _P, _P],
    "sos_lstm_bidir_fwd": [_P, _P, _P, _L, _L, _I, _P, _I, _I, _L, _P, _P, _P, _P],
    "sos_bn_bwd": [C.POINTER(View), C.POINTER(View), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, C.POINTER(View), _P, _P],
    "sos_act_bwd_from_y": [C.POINTER(View), C.POINTER(View), _I, C.POINTER(View), _P],
    "sos_pack_grad_f32": [_P, _P, _I, _L, _L, _I, _L, _L, _L, C.POINTER(View), _P, _P],
    "sos_feat_to_nhwc": [C.POINTER(View), _I, _I, _I, _I, _P, _P, C.POINTER(View), _P],
    "sos_reflect_fold": [C.POINTER(View), _I, _I, _I, C.POINTER(View), _I, _P],
    "sos_reflect_fold_border": [C.POINTER(View), _I, _I, _I, C.POINTER(View), _P],
    "sos_copy_crop": [C.POINTER(View), _I, _I, C.POINTER(View), _I, _I, _P],
    "sos_lstm_bidir_bwd": [_P, _I, _I, _L, _P, _P, _P, _P, _L, _L, _I, _P, _P],
    "sos_mse_loss": [_P, _P, _L, _F, _P, _P, _P, _P],
    "sos_bce_logits_loss": [_P, _P, _L, _F, _P, _P, _P, _P],
    "sos_grad_guard": [_P, _L, _P, _I, _P],
    "sos_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P, _P],
    "sos_adam_multi_step": [_P, _I, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P, _P],
    "sos_gather_pack_multi": [_P, _I, _P, _L, _P],
    "sos_amax_f32": [_P, _L, _P, _P],
    "sos_loss_scale": [_P, _F, _P, _P, _P],
    "sos_scale_f32": [_P, _L, _P, _P],
    "sos_bn_stats_blocks": [_L],
    "sos_bn_stats": [C.POINTER(View), _P, _P],
    "sos_bn_finalize": [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "sos_bn_act_apply": [C.POINTER(View), _P, _P, _I, _P, C.POINTER(View), _I, _I, _I, _P, _P],
    "sos_wgrad_workspace_bytes": [C.POINTER(WgradDesc)],
    "sos_conv2d_wgrad": [C.POINTER(WgradDesc), _P],
    "sos_conv2d_wgrad_partial": [C.POINTER(WgradDesc), _P],
    "sos_conv2d_wgrad_reduce": [C.POINTER(WgradDesc), _P],
    "sos_wgrad_tune": [C.POINTER(WgradDesc), _I, C.POINTER(C.c_float), _P],
    "sos_wgrad_tune_save": [C.c_char_p],
    "sos_wgrad_tune_load": [C.c_char_p],
    "sos_pcm_to_mono_f32": [_P, _I, _I, _L, _P, _P],
    "sos_resample_f32": [_P, _L, _D, _P, _I, _I, _P, _L, _P],
    "sos_resample_time_segments": [_D, _L, _P, _P, _P, _I],
    "sos_time_stack": [_P, _L, _I, _L, _I, _I, _I, _I, _P, _I, _P],
    "sos_spatial_mean": [_P, _L, _L, _I, _I, _I, _P, _L, _I, _I, _P],
    "sos_metric_totals": [_P, _P, _L, _P, _P],
    "sos_metric_frame_energy": [_P, _P, _L, _I, _I, _L, _P, _P, _P],
    "sos_metric_compact": [_P, _P, _L, _F, _P, _P, _P, _P],
    "sos_metric_llr": [_P, _P, _L, _I, _I, _L, _P, _I, _P, _P],
    "sos_metric_wss": [_P, _P, _L, _I, _I, _L, _P, _I, _P, _D, _P, _P],
    "sos_metric_l1": [_P, _L, _P, _L, _P, _P],
    "sos_time_unstack": [_P, _L, _I, _L, _I, _I, _I, _I, _P, _I, _P],
    "sos_spatial_mean_bwd": [_P, _L, _L, _I, _L, _I, _I, _I, _P, _I, _P],
}

_libs = {}


def _load(path, want_dtype):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    import torch  # noqa: F401  -- load torch's bundled HIP runtime FIRST so libsos_hip binds to the same one
    h = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in ("sos_wgrad_workspace_bytes", "sos_lstm_pack_bytes", "sos_conv2d_tile_count",
                                             "sos_stft_matrix_bytes", "sos_istft_matrix_bytes") else C.c_int
    if h.sos_abi_version() != EXPECTED_ABI:
        raise ImportError(f"{path} exports ABI version {h.sos_abi_version()}, this binding was written for {EXPECTED_ABI} "
                          "(the argument lists differ: rebuild the library, or drop the SOS_HIP_LIB / SOS_HIP_LIB_F16 override)")
    h.sos_last_error.restype = C.c_char_p
    h.sos_last_error.argtypes = []
    h.sos_storage_dtype.restype = C.c_char_p
    got = h.sos_storage_dtype().decode()
    if got != want_dtype:
        raise ImportError(f"{path} computes on {got} storage, expected {want_dtype}")
    for which, mirror in ((0, View), (1, ConvDesc), (2, WgradDesc)):
        if h.sos_struct_size(which) != C.sizeof(mirror):
            raise ImportError(f"{path}: {mirror.__name__} is {C.sizeof(mirror)} bytes here but {h.sos_struct_size(which)} in the "
                              "library (include/sos_hip.h and _lib.py have drifted apart, or the library is stale: rebuild it)")
    return h


def lib():
    """The library of the current precision mode (libsos_hip.so: 'bf16' / 'bf16x3'; libsos_hip_f16.so: 'fp16'), loaded
    on first use, or a loud failure (there is no fallback path)."""
    from . import get_precision
    which = "fp16" if get_precision() == "fp16" else "bf16"
    h = _libs.get(which)
    if h is None:
        h = _libs[which] = _load(LIB_PATH_F16 if which == "fp16" else LIB_PATH, which)
    return h


def check(rc, what=""):
    if rc != 0:
        msg = lib().sos_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libsos_hip {what} failed (rc={rc}): {msg}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sos_amd: tensors must live on an MI355X (no CPU fallback); got device "
                               f"{t.device}")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
