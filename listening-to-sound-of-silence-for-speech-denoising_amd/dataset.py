"""Synthetic-clip counterpart of the reference data layer (M1/dataset.py, M2/dataset.py).

The reference corpora (AVSpeech / DEMAND / AudioSet) are not shipped, so clips are synthesised
(SURVEY.md 8-d): seeded band-limited "speech" bursts gated by a per-video-frame bit-stream,
coloured noise mixed at SNR in {-10,-7,-3,0,3,7,10} dB with add_signals semantics
(M2/tools.py:217-276), peak-normalised to 0.5.  Waveform synthesis is host-side data prep
(numpy), exactly where the reference does it (DataLoader workers); every transform of the hot
path (bits->mask, STFT, cRM) runs on the GPU through the HIP kernels.

`get_dataloader` keeps the reference signature (M1/dataset.py:54-68, M2/dataset.py:44-50) and
emits the same batch dicts (M1/dataset.py:348-352, M2/dataset.py:311-320).
"""
import numpy as np
import torch
from scipy.signal import lfilter

from . import tools
from . import transform

DATA_REQUIRED_SR = 14000          # M1/dataset.py:38
CLIP_SECONDS = 2.0
FPS = 30.0
SNRS = [-10, -7, -3, 0, 3, 7, 10]
PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION = 'training', 'testing', 'pred'


def power_of_signal(signal):
    return np.sum(np.abs(signal ** 2))


def add_signals(signal, noises, snr, norm=0.5):
    """M2/tools.py:217-276 (same return triple: mixed, clean, [noises])."""
    if not isinstance(noises, list):
        noises = [noises]
    signal_power = power_of_signal(signal)
    pn = signal_power / np.power(10, snr / 10)
    new_noises = []
    ret_signal = np.copy(signal)
    for noise in noises:
        if signal_power == 0:
            new_noise = noise
        else:
            ratio = np.sqrt(power_of_signal(noise)) / np.sqrt(pn)
            new_noise = noise if ratio == 0 else noise / ratio
        new_noises.append(new_noise)
        ret_signal = ret_signal + new_noise
    if norm:
        scale = np.max(np.abs(ret_signal)) / norm
        if scale != 0:
            return ret_signal / scale, signal / scale, [x / scale for x in new_noises]
    return ret_signal, signal, new_noises


def synth_bits(rng, n_frames, p_silent=0.3, min_run=5):
    """Per-video-frame labels, 1 = non-silent, runs of at least `min_run` frames."""
    bits = []
    while len(bits) < n_frames:
        run = int(min_run + rng.integers(0, 12))
        bits += [0 if rng.random() < p_silent else 1] * run
    return np.array(bits[:n_frames], dtype=np.uint8)


def synth_clip(i, n_samples=int(DATA_REQUIRED_SR * CLIP_SECONDS), sr=DATA_REQUIRED_SR, fps=FPS, snr=None):
    """Returns dict(mixed, clean, full_noise: f32 [n], bits: u8 [n_frames], snr)."""
    rng = np.random.default_rng(1234 + i)
    n_frames = int(round(n_samples / sr * fps))
    bits = synth_bits(rng, n_frames)
    env = np.repeat(bits.astype(np.float32), int(np.ceil(sr / fps)))[:n_samples]
    env = np.pad(env, (0, n_samples - len(env)))
    # band-limited "speech": white noise through a 2-pole resonator-ish smoothing
    s = rng.standard_normal(n_samples).astype(np.float32)
    s = np.convolve(s, np.hanning(9) / np.hanning(9).sum(), mode="same").astype(np.float32) * env
    z = rng.standard_normal(n_samples).astype(np.float32)
    a = 0.85                                   # 1-pole low-pass colouring
    noise = lfilter([1 - a], [1, -a], z).astype(np.float32)
    snr = SNRS[i % len(SNRS)] if snr is None else snr
    mixed, clean, noises = add_signals(s, [noise], snr, norm=0.5)
    return dict(mixed=mixed.astype(np.float32), clean=clean.astype(np.float32),
                full_noise=noises[0].astype(np.float32), bits=bits, snr=snr)


def synth_batch(start, batch, **kw):
    clips = [synth_clip(start + i, **kw) for i in range(batch)]
    out = {k: np.stack([c[k] for c in clips]) for k in ("mixed", "clean", "full_noise", "bits")}
    out["snr"] = [c["snr"] for c in clips]
    return out


class _SyntheticLoader:
    def __init__(self, model, phase, batch_size, n_batches, device):
        self.model, self.phase, self.batch_size, self.n_batches, self.device = model, phase, batch_size, n_batches, device

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        off = {PHASE_TRAINING: 0, PHASE_TESTING: 10 ** 6, PHASE_PREDICTION: 2 * 10 ** 6}[self.phase]
        for b in range(self.n_batches):
            yield make_batch(self.model, off + b * self.batch_size, self.batch_size, self.device)


def make_batch(model, start, batch_size, device="cuda"):
    raw = synth_batch(start, batch_size)
    n = raw["mixed"].shape[1]
    mixed = torch.from_numpy(raw["mixed"]).to(device)
    bits = torch.from_numpy(raw["bits"]).to(device)
    if model == "detector":
        # M1/dataset.py:348-352: label 1 = non-silent, audio = STFT of the mixed clip
        return {"label": bits.float(), "audio": transform.stft_batch(mixed)}
    clean = torch.from_numpy(raw["clean"]).to(device)
    full_noise = torch.from_numpy(raw["full_noise"]).to(device)
    mask, noise_sig = tools.bits_to_mask_batch(bits, DATA_REQUIRED_SR / FPS, n, mixed)   # M2/dataset.py:193,229
    clean = clean * (1 - mask)                                                          # silent intervals truly silent
    S = transform.stft_batch(torch.cat([mixed, clean, noise_sig, full_noise], dim=0))
    B = batch_size
    mixed_s, clean_s, noise_s, full_s = S[:B], S[B:2 * B], S[2 * B:3 * B], S[3 * B:]
    target = torch.empty_like(mixed_s)
    from . import _lib as L
    L.check(L.lib().sos_crm_target_f32(L.ptr(clean_s.contiguous()), L.ptr(mixed_s.contiguous()), L.ptr(target), B,
                                       mixed_s[0, 0].numel(), 0.1, 0.0, L.stream_ptr()), "sos_crm_target_f32")
    return {"mixed": mixed_s, "clean": clean_s, "noise": noise_s, "full_noise": full_s, "mask": target,
            "start": list(range(start, start + B)), "bitstream": ["".join(map(str, r)) for r in raw["bits"]]}


def get_dataloader(phase, batch_size=4, num_workers=4, snr_idx=None, dataset_json=None, clean_audio=True,
                   model="denoiser", n_batches=8, device="cuda"):
    """Reference signature + `model` ('detector' -> M1 schema, 'denoiser' -> M2 schema)."""
    assert phase in (PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION)
    return _SyntheticLoader(model, phase, batch_size, n_batches, device)
