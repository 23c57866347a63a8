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


def synth_bits(rng, n_frames, p_silent=0.3, min_run=5):
    """Per-video-frame labels, 1 = non-silent, runs of at least `min_run` frames."""
    bits = []
    while len(bits) < n_frames:
        run = int(min_run + rng.integers(0, 12))
        bits += [0 if rng.random() < p_silent else 1] * run
    return np.array(bits[:n_frames], dtype=np.uint8)


def _synth_raw(i, n_samples, sr, fps, snr):
    """Host-side draws of clip i (seeded): un-gated "speech", coloured noise, per-video-frame labels, SNR."""
    rng = np.random.default_rng(1234 + i)
    n_frames = int(round(n_samples / sr * fps))
    bits = synth_bits(rng, n_frames)
    # band-limited "speech": white noise through a short smoothing window
    s = rng.standard_normal(n_samples).astype(np.float32)
    s = np.convolve(s, np.hanning(9) / np.hanning(9).sum(), mode="same").astype(np.float32)
    z = rng.standard_normal(n_samples).astype(np.float32)
    a = 0.85                                   # 1-pole low-pass colouring
    noise = lfilter([1 - a], [1, -a], z).astype(np.float32)
    return s, noise, bits, (SNRS[i % len(SNRS)] if snr is None else snr)


def synth_batch(start, batch, n_samples=int(DATA_REQUIRED_SR * CLIP_SECONDS), sr=DATA_REQUIRED_SR, fps=FPS, snr=None,
                device="cuda"):
    """`batch` synthetic clips start .. start+batch-1 -> dict of host arrays mixed / clean / full_noise (B, n) f32,
    bits (B, n_frames) u8, snr list.  The draws are host numpy (seeded per clip); the speech is gated with the SAME
    per-sample mask the pipeline derives from the bits (sos_bits_to_mask: silent samples are exactly zero, so
    mixed == clean + full_noise sample for sample, M2/dataset.py:183 then :217), and the mix at the SNR + peak
    normalisation is sos_add_signals_f32."""
    draws = [_synth_raw(start + i, n_samples, sr, fps, snr) for i in range(batch)]
    bits = np.stack([d[2] for d in draws])
    sp = torch.from_numpy(np.stack([d[0] for d in draws])).to(device)
    nz = torch.from_numpy(np.stack([d[1] for d in draws])).to(device)
    mask = tools.bits_to_mask_batch(torch.from_numpy(bits).to(device), float(sr) / fps, n_samples)
    snrs = [d[3] for d in draws]
    mixed, clean, noise = tools.add_signals_batch(sp * (1 - mask), nz, snrs, norm=0.5)
    return dict(mixed=mixed.cpu().numpy(), clean=clean.cpu().numpy(), full_noise=noise.cpu().numpy(), bits=bits, snr=snrs)


def synth_clip(i, **kw):
    """One synthetic clip: dict(mixed, clean, full_noise: f32 [n], bits: u8 [n_frames], snr)."""
    b = synth_batch(i, 1, **kw)
    return dict(mixed=b["mixed"][0], clean=b["clean"][0], full_noise=b["full_noise"][0], bits=b["bits"][0], snr=b["snr"][0])


def shard_indices(n_items, rank, world_size):
    """The item indices rank `rank` of `world_size` processes owns: every rank gets ceil(n / world) of them (the tail
    wraps around, like torch's DistributedSampler, so that all ranks run the same number of steps and the gradient
    all-reduce never waits for a missing partner)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank must be in [0, world_size)")
    per = (n_items + world_size - 1) // world_size
    return [(rank + k * world_size) % n_items for k in range(per)] if n_items else []


class _SyntheticLoader:
    """Global batch b of the epoch = clips [b*G, (b+1)*G) with G = batch_size * world_size; rank r takes the r-th
    slice of `batch_size` clips (disjoint across ranks, identical for every world size)."""

    def __init__(self, model, phase, batch_size, n_batches, device, rank=0, world_size=1):
        self.model, self.phase, self.batch_size, self.n_batches, self.device = model, phase, batch_size, n_batches, device
        self.rank, self.world_size = rank, world_size

    def __len__(self):
        return self.n_batches

    def starts(self):
        off = {PHASE_TRAINING: 0, PHASE_TESTING: 10 ** 6, PHASE_PREDICTION: 2 * 10 ** 6}[self.phase]
        return [off + (b * self.world_size + self.rank) * self.batch_size for b in range(self.n_batches)]

    def __iter__(self):
        for st in self.starts():
            yield make_batch(self.model, st, self.batch_size, self.device)


def make_batch(model, start, batch_size, device="cuda"):
    return batch_from_raw(model, synth_batch(start, batch_size, device=device), list(range(start, start + batch_size)), device)


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
    # the clean clip was silenced on its silent intervals BEFORE mixing (M2/dataset.py:183 then :217: synth_batch and
    # _FileLoader do that), so mixed == clean + full_noise and nothing is re-gated here
    mask, noise_sig = tools.bits_to_mask_batch(bits, DATA_REQUIRED_SR / FPS, n, mixed)   # M2/dataset.py:193,229
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

    def __init__(self, model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed, rank=0,
                 world_size=1):
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
            i1, i2 = tools.trim_unknown_frames(bits)       # the reference's truncate(): same rule as the hand-off
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
        # the shuffle is drawn from `seed` alone (every rank computes the same permutation, then takes its own shard of
        # it); the noise / SNR draws come from a per-rank stream
        self.order_rng = np.random.default_rng(seed)
        self.rng = np.random.default_rng([seed, rank])
        self.n_clip = n_clip
        self.rank, self.world_size = rank, world_size

    def __len__(self):
        per_rank = (len(self.items) + self.world_size - 1) // self.world_size
        return (per_rank + self.batch_size - 1) // self.batch_size

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
            self.order_rng.shuffle(order)
        order = order[shard_indices(len(order), self.rank, self.world_size)]
        for s0 in range(0, len(order), self.batch_size):
            idx = order[s0:s0 + self.batch_size]
            clips = [self._clip(self.items[k]) for k in idx]
            audio = np.stack([c[0] for c in clips])
            bits = np.stack([c[1] for c in clips])
            # silent intervals truly silent before mixing (M2/dataset.py:160-183): the sample mask comes from the GPU kernel
            mask = tools.bits_to_mask_batch(torch.from_numpy(bits).to(self.device), DATA_REQUIRED_SR / FPS, self.n_clip).cpu().numpy()
            snrs, crops = [], []
            for _ in idx:
                snrs.append(SNRS[self.snr_idx] if self.snr_idx is not None else SNRS[int(self.rng.integers(len(SNRS)))])
                nz = self.noises[int(self.rng.integers(len(self.noises)))]
                st = int(self.rng.integers(0, len(nz) - self.n_clip + 1))
                crops.append(nz[st:st + self.n_clip].astype(np.float32))
            mixed, clean, fn = tools.add_signals_batch(torch.from_numpy(audio * (1 - mask)).to(self.device),
                                                       torch.from_numpy(np.stack(crops)).to(self.device), snrs, norm=0.5)
            raw = dict(mixed=mixed.cpu().numpy(), clean=clean.cpu().numpy(), full_noise=fn.cpu().numpy(), bits=bits, snr=snrs)
            batch = batch_from_raw(self.model, raw, [self.items[k][1] for k in idx], self.device)
            batch["_raw"] = raw
            yield batch


def get_dataloader(phase, batch_size=4, num_workers=4, snr_idx=None, dataset_json=None, clean_audio=True,
                   model="denoiser", n_batches=8, device="cuda", noise_files=None, data_root=None, seed=0, rank=None,
                   world_size=None):
    """Reference signature + `model` ('detector' -> M1 schema, 'denoiser' -> M2 schema).  With `dataset_json` and
    `noise_files` the clips come from real recordings (see _FileLoader); otherwise they are synthesised.
    `batch_size` is PER RANK; `rank` / `world_size` (default: torch.distributed's, else 0 / 1) shard the clips so that
    the ranks of a data-parallel job see disjoint data (the reference's single-process DataParallel scatters one
    loader's batch over the GPUs, M2/dataset.py:44-50 + M2/agent.py:151)."""
    assert phase in (PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION)
    if rank is None or world_size is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank = (dist.get_rank() if on else 0) if rank is None else rank
        world_size = (dist.get_world_size() if on else 1) if world_size is None else world_size
    if dataset_json is not None and noise_files:
        return _FileLoader(model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed, rank,
                           world_size)
    return _SyntheticLoader(model, phase, batch_size, n_batches, device, rank, world_size)
