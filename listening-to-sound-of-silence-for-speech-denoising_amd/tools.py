"""Mirror of the on-path pieces of the reference `tools.py` (M2/tools.py:217-303,340-362)."""
import numpy as np
import torch

from . import _lib as L


def bits_to_mask_batch(bits, ratio, n_samples, sig=None):
    """bits uint8 (B, n_frames) on the GPU (1 = non-silent) -> mask f32 (B, n_samples)
    (1 on silent samples) and, if `sig` is given, sig*mask (M2/predict.py:310,317)."""
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
                                     L.ptr(masked), L.stream_ptr()), "sos_bits_to_mask")
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
