"""Mirror of the on-path pieces of the reference `tools.py` (M2/tools.py:217-303,340-362)."""
import numpy as np
import torch

from . import _lib as L


def bits_to_mask_batch(bits, ratio, n_samples, sig=None, clip_frames=None, clip_samples=None):
    """bits uint8 (B, n_frames) on the GPU (1 = non-silent) -> mask f32 (B, n_samples)
    (1 on silent samples) and, if `sig` is given, sig*mask (M2/predict.py:310,317).  clip_frames / clip_samples:
    optional int32 device (B,) each, ragged batch (clip b uses its own frame and sample counts)."""
    L.require_cuda(bits, sig)
    bits = bits.contiguous()
    if bits.dtype != torch.uint8 or bits.dim() != 2:
        raise ValueError("bits must be a uint8 (B, n_frames) tensor")
    B, nfr = bits.shape
    mask = torch.empty((B, n_samples), dtype=torch.float32, device=bits.device)
    masked = None
    if sig is not None:
        sig = sig.contiguous()
        if sig.shape != mask.shape or sig.dtype != torch.float32:
            raise ValueError("sig must be float32 (B, n_samples)")
        masked = torch.empty_like(sig)
    L.check(L.lib().sos_bits_to_mask(L.ptr(bits), B, nfr, float(ratio), n_samples, L.ptr(mask), L.ptr(sig),
                                     L.ptr(masked), L.ptr(clip_frames), L.ptr(clip_samples), L.stream_ptr()),
            "sos_bits_to_mask")
    return (mask, masked) if sig is not None else mask


def convert_bitstreammask_to_audiomask(ref_audio_signal, frames_to_audiosample_ratio, bitstream):
    """M2/tools.py:340-362 (string bits) / M1/tools.py:770-792 (int bits): same arguments, same
    RuntimeError on an invalid bit, same dtype as `ref_audio_signal`."""
    vals = []
    for bit in bitstream:
        if bit in ('0', 0):
            vals.append(0)
        elif bit in ('1', 1):
            vals.append(1)
        else:
            print('Invalid bit?')
            raise RuntimeError
    if not torch.cuda.is_available():
        raise RuntimeError("sos_amd.tools needs an MI355X: there is no CPU fallback")
    bits = torch.tensor(vals, dtype=torch.uint8, device="cuda").reshape(1, -1)
    n = len(ref_audio_signal)
    mask = bits_to_mask_batch(bits, frames_to_audiosample_ratio, n)
    return mask[0].cpu().numpy().astype(np.asarray(ref_audio_signal).dtype)


def threshold_bits(logits, threshold=0.5):
    """M1/predict.py:117-119: bit = sigmoid(logit) >= threshold (1 = non-silent); returns
    (bits uint8, confidence f32) with the shape of `logits`."""
    L.require_cuda(logits)
    lg = logits.contiguous().float()
    bits = torch.empty(lg.shape, dtype=torch.uint8, device=lg.device)
    conf = torch.empty_like(lg)
    L.check(L.lib().sos_threshold_bits(L.ptr(lg), lg.numel(), float(threshold), L.ptr(bits), L.ptr(conf),
                                       L.stream_ptr()), "sos_threshold_bits")
    return bits, conf


def add_signals_batch(signal, noises, snr, norm=0.5):
    """Batched, on-device `add_signals`: signal f32 (B, n), noises f32 (B, K, n) or (B, n), snr scalar / sequence / tensor
    of B values (dB) -> (mixed (B, n), signal (B, n), noises like the input), GPU tensors."""
    L.require_cuda(signal, noises)
    signal = signal.contiguous().float()
    squeeze = noises.dim() == 2
    nz = (noises[:, None] if squeeze else noises).contiguous().float()
    B, n = signal.shape
    if nz.shape[0] != B or nz.shape[2] != n or not 1 <= nz.shape[1] <= 8:
        raise ValueError("noises must be (B, K <= 8, n) matching signal (B, n)")
    snr_t = torch.as_tensor(snr, dtype=torch.float32).reshape(-1)
    snr_t = (snr_t.expand(B) if snr_t.numel() == 1 else snr_t).contiguous().to(signal.device)
    if snr_t.numel() != B:
        raise ValueError("snr must be a scalar or one value per clip")
    mixed, s_out, n_out = torch.empty_like(signal), torch.empty_like(signal), torch.empty_like(nz)
    L.check(L.lib().sos_add_signals_f32(L.ptr(signal), L.ptr(nz), L.ptr(snr_t), B, nz.shape[1], n, float(norm or 0.0),
                                        L.ptr(mixed), L.ptr(s_out), L.ptr(n_out), L.stream_ptr()), "sos_add_signals_f32")
    return mixed, s_out, (n_out[:, 0] if squeeze else n_out)


def add_signals(signal, noises, snr, norm=0.5):
    """M2/tools.py:217-276 with its arguments and its return triple (mixed, signal, [noises]): numpy in, numpy out, the
    arithmetic in sos_add_signals_f32."""
    if not torch.cuda.is_available():
        raise RuntimeError("sos_amd.tools needs an MI355X: there is no CPU fallback")
    single = not isinstance(noises, (list, tuple))
    stack = np.stack([np.asarray(x, dtype=np.float32) for x in ([noises] if single else noises)])
    sig = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32)).cuda()[None]
    m, s, nz = add_signals_batch(sig, torch.from_numpy(stack).cuda()[None], float(snr), norm)
    dt = np.asarray(signal).dtype if np.issubdtype(np.asarray(signal).dtype, np.floating) else np.float32
    return m[0].cpu().numpy().astype(dt), s[0].cpu().numpy().astype(dt), [x.cpu().numpy().astype(dt) for x in nz[0]]


def trim_unknown_frames(bits):
    """M1/tools.py:270-274,306-311 (`truncate`): the first two runs of '2' (unlabelled frames) bound the labelled part
    of a bit-stream; a stream without two such runs is used whole.  Returns (first, end) frame indices.  One helper
    for the file data loader and the on-disk hand-off."""
    from itertools import groupby
    runs = [len(list(g)) for k, g in groupby(bits) if k == '2']
    if len(runs) >= 2:
        return runs[0], len(bits) - runs[1]
    return 0, len(bits)
