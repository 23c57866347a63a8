"""Counterpart of the reference data layer (M1/dataset.py, M2/dataset.py): synthetic clips by default, clips cut from
real recordings when `get_dataloader` is given a dataset JSON and noise files (_FileLoader).

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
    return batch_from_raw(model, synth_batch(start, batch_size), list(range(start, start + batch_size)), device)


def batch_from_raw(model, raw, starts, device="cuda"):
    """raw: dict of host arrays mixed / clean / full_noise (B, n) f32 and bits (B, n_frames) u8 -> the reference's
    batch dict, every transform on the GPU."""
    batch_size = len(raw["mixed"])
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
            "start": list(starts), "bitstream": ["".join(map(str, r)) for r in raw["bits"]]}


class _FileLoader:
    """Clips cut from real recordings (the reference's corpora layout): dataset JSON (PP/tools.py:28-31) + noise WAVE
    files.  Items as the reference makes them -- denoiser: 2 s windows every second of the labelled part of a file
    (M2/tools.py:134-177); detector: 60-frame windows every 30 frames, cut / zero-padded to 28 000 samples
    (M1/tools.py:297-332, M1/dataset.py:231-236) -- the clean clip silenced on its labelled silent intervals, a
    random crop of a random noise file mixed in at a random (or the `snr_idx`-th) SNR with add_signals semantics,
    peak 0.5 (M2/dataset.py:155-208).  Files are decoded and resampled once (GPU) and kept in host memory; draws
    come from a seeded numpy generator (the reference uses unseeded worker RNGs)."""

    def __init__(self, model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed):
        import json
        import os
        from . import audio_io
        self.model, self.phase, self.batch_size, self.device, self.snr_idx = model, phase, batch_size, device, snr_idx
        with open(dataset_json) as fp:
            ds = json.load(fp)
        root = ds.get("dataset_path", "")
        fix = (lambda p: os.path.join(data_root, os.path.relpath(p, root))) if data_root and root else (lambda p: p)
        self.audio, self.items = [], []
        n_clip = int(CLIP_SECONDS * DATA_REQUIRED_SR)
        for i, f in enumerate(ds["files"]):
            y, _ = audio_io.load(fix(f["audio_path"]), sr=DATA_REQUIRED_SR)
            self.audio.append(y)
            bits, fps = f["bit_stream"], float(f["framerate"])
            runs = [(k, len(list(g))) for k, g in __import__("itertools").groupby(bits)]
            i1 = runs[0][1] if runs and runs[0][0] == "2" else 0
            i2 = len(bits) - (runs[-1][1] if len(runs) > 1 and runs[-1][0] == "2" else 0)
            lab = bits[i1:i2]
            if model == "detector":
                nfr = int(round(CLIP_SECONDS * FPS))
                for x in range(0, len(lab) + 1 - nfr, nfr // 2):
                    f0 = i1 + x
                    self.items.append((i, int(f0 / fps * DATA_REQUIRED_SR), int((f0 + nfr) / fps * DATA_REQUIRED_SR), lab[x:x + nfr], fps))
            else:
                start_sec, end_sec = i1 / fps, i2 / fps
                dur = min(float(f["duration"]), len(y) / DATA_REQUIRED_SR, end_sec) - start_sec
                if dur < CLIP_SECONDS:
                    continue
                for k in range(int((dur - CLIP_SECONDS) // 1.0) + 1):
                    x = start_sec + k * 1.0
                    cb = lab[int((x - start_sec) * fps):int((x - start_sec + CLIP_SECONDS) * fps)]
                    self.items.append((i, int(x * DATA_REQUIRED_SR), int(x * DATA_REQUIRED_SR) + n_clip, cb, fps))
        self.noises = [audio_io.load(p, sr=DATA_REQUIRED_SR)[0] for p in noise_files]
        if not self.items or not self.noises:
            raise RuntimeError("no clips of %g s in %s (or no noise files)" % (CLIP_SECONDS, dataset_json))
        if any(len(nz) < n_clip for nz in self.noises):
            raise ValueError("noise files must be at least one clip long")
        self.rng = np.random.default_rng(seed)
        self.n_clip = n_clip

    def __len__(self):
        return (len(self.items) + self.batch_size - 1) // self.batch_size

    def _clip(self, item):
        fi, a, b, bits, fps = item
        nfr = int(round(CLIP_SECONDS * FPS))
        audio = self.audio[fi][a:b][:self.n_clip].astype(np.float32)
        if len(audio) < self.n_clip:
            audio = np.concatenate((audio, np.zeros(self.n_clip - len(audio), dtype=np.float32)))
        b8 = np.array([1 if c != "0" else 0 for c in bits][:nfr] + [1] * max(0, nfr - len(bits)), dtype=np.uint8)
        return audio, b8

    def __iter__(self):
        order = np.arange(len(self.items))
        if self.phase == PHASE_TRAINING:
            self.rng.shuffle(order)
        for s0 in range(0, len(order), self.batch_size):
            idx = order[s0:s0 + self.batch_size]
            clips = [self._clip(self.items[k]) for k in idx]
            audio = np.stack([c[0] for c in clips])
            bits = np.stack([c[1] for c in clips])
            # silent intervals truly silent before mixing (M2/dataset.py:160-183): the sample mask comes from the GPU kernel
            mask = tools.bits_to_mask_batch(torch.from_numpy(bits).to(self.device), DATA_REQUIRED_SR / FPS, self.n_clip).cpu().numpy()
            raw = dict(mixed=[], clean=[], full_noise=[], bits=bits, snr=[])
            for a, m in zip(audio, mask):
                snr = SNRS[self.snr_idx] if self.snr_idx is not None else SNRS[int(self.rng.integers(len(SNRS)))]
                nz = self.noises[int(self.rng.integers(len(self.noises)))]
                st = int(self.rng.integers(0, len(nz) - self.n_clip + 1))
                mixed, clean, fn = add_signals(a * (1 - m), [nz[st:st + self.n_clip]], snr, norm=0.5)
                raw["mixed"].append(mixed); raw["clean"].append(clean); raw["full_noise"].append(fn[0]); raw["snr"].append(snr)
            for k in ("mixed", "clean", "full_noise"):
                raw[k] = np.stack(raw[k]).astype(np.float32)
            batch = batch_from_raw(self.model, raw, [self.items[k][1] for k in idx], self.device)
            batch["_raw"] = raw
            yield batch


def get_dataloader(phase, batch_size=4, num_workers=4, snr_idx=None, dataset_json=None, clean_audio=True,
                   model="denoiser", n_batches=8, device="cuda", noise_files=None, data_root=None, seed=0):
    """Reference signature + `model` ('detector' -> M1 schema, 'denoiser' -> M2 schema).  With `dataset_json` and
    `noise_files` the clips come from real recordings (see _FileLoader); otherwise they are synthesised."""
    assert phase in (PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION)
    if dataset_json is not None and noise_files:
        return _FileLoader(model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed)
    return _SyntheticLoader(model, phase, batch_size, n_batches, device)
