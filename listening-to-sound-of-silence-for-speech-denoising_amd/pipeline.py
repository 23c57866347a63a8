"""On-device inference chain: STFT -> detector -> threshold -> bits->sample-mask -> noise-interval
STFT -> JointModel -> mask apply -> ISTFT, with no host round trip.

Restates M1/predict.py:38-233 (detector pass) + M2/predict.py:255-326,377-447 (denoiser pass);
the reference hands results over through JSON/WAV files on disk, here the hand-off is the
`bits` tensor in HBM."""
import torch

from . import tools
from . import transform

SR = 14000
FPS = 30.0
SIGMOID_THRESHOLD = 0.5     # M1/predict.py:30


def n_video_frames(n_samples, sr=SR, fps=FPS):
    return int(round(n_samples / sr * fps))


@torch.no_grad()
def denoise(detector, denoiser, mixed, sr=SR, fps=FPS, bits=None, return_all=False):
    """mixed f32 (B, N) on the GPU -> denoised f32 (B, hop*(T-1)).  `bits` (uint8 (B, n_frames),
    1 = non-silent) overrides the detector (M2/predict.py's `recovered_prediction` input)."""
    B, N = mixed.shape
    S_mixed = transform.stft_batch(mixed)
    logits = None
    if bits is None:
        logits = detector(s=S_mixed, v_num_frames=n_video_frames(N, sr, fps))
        bits, _ = tools.threshold_bits(logits, SIGMOID_THRESHOLD)
    mask, noise_sig = tools.bits_to_mask_batch(bits, float(sr) / fps, N, mixed)
    S_noise = transform.stft_batch(noise_sig)
    n_pred, crm = denoiser(S_mixed, S_noise)
    S_out = transform.batch_fast_icRM_sigmoid(S_mixed, crm)
    out = transform.istft_batch(S_out)
    if return_all:
        return dict(out=out, logits=logits, bits=bits, mask=mask, n_pred=n_pred, crm=crm, S_mixed=S_mixed,
                    S_noise=S_noise, S_out=S_out)
    return out


@torch.no_grad()
def denoise_ragged(detector, denoiser, clips, sr=SR, fps=FPS, max_batch=64):
    """Variable-length inference (BASELINE configs[3]): `clips` = list of 1-D f32 GPU tensors of ANY lengths.
    The reference denoises one file at a time at its own length (M2/predict.py:377-447: no padding, so reflect
    padding, frame count and video-frame count follow the clip); clips of equal length are bucketed into one
    batch (<= max_batch; same result as one by one up to the summation order of the tuned conv tilings) and the
    outputs come back in input order, each hop*(T-1) samples long."""
    order = {}
    for i, c in enumerate(clips):
        if c.dim() != 1:
            raise ValueError("denoise_ragged expects 1-D waveforms")
        order.setdefault(int(c.numel()), []).append(i)
    outs = [None] * len(clips)
    for n, idx in sorted(order.items()):
        for j in range(0, len(idx), max_batch):
            part = idx[j:j + max_batch]
            y = denoise(detector, denoiser, torch.stack([clips[i] for i in part]).contiguous(), sr, fps)
            for k, i in enumerate(part):
                outs[i] = y[k]
    return outs
