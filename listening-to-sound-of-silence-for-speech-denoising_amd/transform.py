"""Mirror of the reference `transform.py` (M1/transform.py == M2/transform.py) on HIP kernels.

Same public names and argument meaning: N_FFT/HOP_LENGTH/WIN_LENGTH (:6-8), fast_stft (:188-193),
fast_istft (:196-202), batch_fast_icRM_sigmoid (:156-169, differentiable), fast_icRM_sigmoid
(:141-153), fast_cRM_sigmoid (:130-138), real_imag_expand/shrink (:10-33).  numpy in -> numpy out
like the reference; the `*_batch` functions are the device-resident forms the pipeline uses.
"""
import math

import numpy as np
import torch

from . import _lib as L

N_FFT = 510
HOP_LENGTH = 158
WIN_LENGTH = 400

_tables = {}


def _front_tables(device, n_fft, hop, win_length, inverse=False):
    """The packed windowed-DFT matrix of the STFT (or the synthesis matrix + squared window of the ISTFT) in MFMA
    fragment order, hi and lo half-precision parts: packed once on the host by the library (sos_*_pack_matrix) and
    kept on the device."""
    key = ("istft" if inverse else "stft", str(device), n_fft, hop, win_length)
    if key not in _tables:
        h = L.lib()
        nbytes = (h.sos_istft_matrix_bytes if inverse else h.sos_stft_matrix_bytes)(n_fft, hop, win_length)
        if nbytes < 0:
            raise ValueError((h.sos_last_error() or b"").decode())
        hi, lo = np.empty(nbytes // 2, dtype=np.uint16), np.empty(nbytes // 2, dtype=np.uint16)
        if inverse:
            wsq = np.empty(win_length, dtype=np.float32)
            L.check(h.sos_istft_pack_matrix(n_fft, hop, win_length, hi.ctypes.data, lo.ctypes.data, wsq.ctypes.data),
                    "sos_istft_pack_matrix")
            _tables[key] = (torch.from_numpy(hi.view(np.int16)).to(device), torch.from_numpy(lo.view(np.int16)).to(device),
                            torch.from_numpy(wsq).to(device))
        else:
            L.check(h.sos_stft_pack_matrix(n_fft, hop, win_length, hi.ctypes.data, lo.ctypes.data), "sos_stft_pack_matrix")
            _tables[key] = (torch.from_numpy(hi.view(np.int16)).to(device), torch.from_numpy(lo.view(np.int16)).to(device))
    return _tables[key]


def stft_batch(wave, n_fft=N_FFT, hop_length=HOP_LENGTH, win_length=WIN_LENGTH, clip_samples=None):
    """wave f32 (B, N) on the GPU -> (B, 2, n_fft//2+1, 1+N//hop) f32 (a1, fused with the
    [F,T,2] -> [2,F,T] transpose of M2/dataset.py:255).  clip_samples: optional int32 device (B,), ragged batch: clip b
    has clip_samples[b] <= N samples (frames past its own 1 + n//hop are left unwritten)."""
    L.require_cuda(wave)
    if wave.dim() != 2 or wave.dtype != torch.float32:
        raise ValueError("stft_batch expects a float32 (B, N) tensor")
    wave = wave.contiguous()
    B, N = wave.shape
    T = 1 + N // hop_length
    mhi, mlo = _front_tables(wave.device, n_fft, hop_length, win_length)
    out = torch.empty((B, 2, n_fft // 2 + 1, T), dtype=torch.float32, device=wave.device)
    L.check(L.lib().sos_stft_f32(L.ptr(wave), B, N, N, L.ptr(mhi), L.ptr(mlo), n_fft, hop_length, win_length,
                                 L.ptr(out), T, L.ptr(clip_samples), L.stream_ptr()), "sos_stft_f32")
    return out


def istft_batch(spec, hop_length=HOP_LENGTH, win_length=WIN_LENGTH, clip_frames=None):
    """spec f32 (B, 2, F, T) on the GPU -> (B, hop*(T-1)) f32 (a2).  clip_frames: optional int32 device (B,), ragged
    batch: clip b has clip_frames[b] <= T frames and gets hop*(clip_frames[b]-1) samples."""
    L.require_cuda(spec)
    if spec.dim() != 4 or spec.shape[1] != 2 or spec.dtype != torch.float32:
        raise ValueError("istft_batch expects a float32 (B, 2, F, T) tensor")
    spec = spec.contiguous()
    B, _, F, T = spec.shape
    n_fft = 2 * (F - 1)
    mhi, mlo, wsq = _front_tables(spec.device, n_fft, hop_length, win_length, inverse=True)
    # the window-sum-square normalisation is computed inside the kernel (the <= 3 frames covering a sample)
    n_out = hop_length * (T - 1)
    out = torch.empty((B, n_out), dtype=torch.float32, device=spec.device)
    L.check(L.lib().sos_istft_f32(L.ptr(spec), B, T, L.ptr(mhi), L.ptr(mlo), L.ptr(wsq), n_fft, hop_length,
                                  win_length, L.ptr(out), n_out, L.ptr(clip_frames), L.stream_ptr()), "sos_istft_f32")
    return out


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("sos_amd.transform needs an MI355X: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def real_imag_expand(c_data, dim='new'):
    """M1/transform.py:10-22 (pure layout glue on host arrays)."""
    if dim == 'new':
        return np.stack([np.real(c_data), np.imag(c_data)], axis=-1).astype(np.float64)
    if dim == 'same':
        D = np.zeros((c_data.shape[0], c_data.shape[1] * 2))
        D[:, ::2] = np.real(c_data)
        D[:, 1::2] = np.imag(c_data)
        return D


def real_imag_shrink(F, dim='new'):
    """M1/transform.py:25-33."""
    if dim == 'new':
        return F[:, :, 0] + F[:, :, 1] * 1j
    if dim == 'same':
        return F[:, ::2] + F[:, 1::2] * 1j


def power_law_batch(x, power=0.3):
    """power_law, M1/transform.py:178-185, on a GPU tensor: sign(x) |x|^power."""
    L.require_cuda(x)
    x = x.contiguous().float()
    out = torch.empty_like(x)
    L.check(L.lib().sos_power_law_f32(L.ptr(x), x.numel(), float(power), L.ptr(out), L.stream_ptr()), "sos_power_law_f32")
    return out


def power_law(data, power=0.3):
    """M1/transform.py:178-185: numpy in, float64 out like the reference (its mask array is float64)."""
    x = torch.as_tensor(np.asarray(data, dtype=np.float32), device=_device())
    return power_law_batch(x, power).cpu().numpy().astype(np.float64)


def fast_stft(data, power=False, n_fft=N_FFT, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """M1/transform.py:188-193: 1-D waveform -> ndarray [F, T, 2].  power=True: A**0.3 companding first (:191-192)."""
    w = torch.as_tensor(np.asarray(data, dtype=np.float32), device=_device()).reshape(1, -1)
    if power:
        w = power_law_batch(w, 0.3)
    S = stft_batch(w, n_fft, hop_length, win_length)[0]          # (2, F, T)
    return S.permute(1, 2, 0).cpu().numpy().astype(np.float64)


def fast_istft(F, power=False, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """M1/transform.py:196-202: [F, T, 2] -> 1-D float32 of length hop*(T-1) (float64 with power=True, like the reference)."""
    S = torch.as_tensor(np.asarray(F, dtype=np.float32), device=_device()).permute(2, 0, 1).unsqueeze(0)
    y = istft_batch(S, hop_length, win_length)
    if power:                                                   # :200-201, the inverse companding
        return power_law_batch(y, 1.0 / 0.3)[0].cpu().numpy().astype(np.float64)
    return y[0].cpu().numpy()


class _CrmApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, crm, a, b):
        L.require_cuda(Y, crm)
        Yc, cc = Y.contiguous().float(), crm.contiguous().float()
        B = Yc.shape[0]
        plane = Yc[0, 0].numel()
        rec = torch.empty_like(Yc)
        L.check(L.lib().sos_crm_apply_f32(L.ptr(Yc), L.ptr(cc), L.ptr(rec), B, plane, float(a), float(b),
                                          L.stream_ptr()), "sos_crm_apply_f32")
        ctx.save_for_backward(Yc, cc)
        ctx.a = float(a)
        return rec

    @staticmethod
    def backward(ctx, g):
        Yc, cc = ctx.saved_tensors
        g = g.contiguous().float()
        gc = torch.empty_like(cc)
        L.check(L.lib().sos_crm_apply_bwd_f32(L.ptr(Yc), L.ptr(cc), L.ptr(g), L.ptr(gc), Yc.shape[0],
                                              Yc[0, 0].numel(), ctx.a, L.stream_ptr()), "sos_crm_apply_bwd_f32")
        return None, gc, None, None


def batch_fast_icRM_sigmoid(Y, crm, a=0.1, b=0):
    """M1/transform.py:156-169: Y, crm (B, 2, F, T) -> rec (B, 2, F, T); grad flows to crm."""
    if Y.dim() != 4 or Y.shape[1] != 2 or Y.shape != crm.shape:
        raise ValueError("batch_fast_icRM_sigmoid expects two (B, 2, F, T) tensors")
    return _CrmApply.apply(Y, crm, a, b)


def fast_icRM_sigmoid(Y, crm):
    """M1/transform.py:141-153: numpy [F, T, 2] x2 -> [F, T, 2]."""
    dev = _device()
    Yt = torch.as_tensor(np.asarray(Y, dtype=np.float32), device=dev).permute(2, 0, 1).unsqueeze(0)
    ct = torch.as_tensor(np.asarray(crm, dtype=np.float32), device=dev).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        rec = batch_fast_icRM_sigmoid(Yt, ct)
    return rec[0].permute(1, 2, 0).cpu().numpy().astype(np.float64)


def fast_cRM_sigmoid(Fclean, Fmix):
    """M1/transform.py:130-138: target mask sigma(0.1 * S conj(Y) / (abs(Y)^2 + 1e-8))."""
    dev = _device()
    S = torch.as_tensor(np.asarray(Fclean, dtype=np.float32), device=dev).permute(2, 0, 1).unsqueeze(0).contiguous()
    Y = torch.as_tensor(np.asarray(Fmix, dtype=np.float32), device=dev).permute(2, 0, 1).unsqueeze(0).contiguous()
    out = torch.empty_like(S)
    L.check(L.lib().sos_crm_target_f32(L.ptr(S), L.ptr(Y), L.ptr(out), 1, S[0, 0].numel(), 0.1, 0.0,
                                       L.stream_ptr()), "sos_crm_target_f32")
    return out[0].permute(1, 2, 0).cpu().numpy().astype(np.float64)
