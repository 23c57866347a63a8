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
