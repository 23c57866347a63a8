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

Round 6: the loaders are ASYNCHRONOUS -- the reference overlaps data preparation with the GPU through
`DataLoader(num_workers=70, pin_memory=True)` (M2/dataset.py:44-50, M2/common.py:52-53); here `num_workers` worker PROCESSES make
the host draws (_HostPool: numpy / scipy only), a producer thread copies them into PINNED staging buffers, uploads them on a side
HIP stream and runs the device half of the batch (bits -> mask, mix at the SNR, the four STFTs, the cRM target) there, `prefetch`
batches ahead; the consumer's stream only waits for the batch's event.  No device -> host copy anywhere on the way (the round-5
loaders went device -> host -> device twice per batch).  num_workers = 0 is the synchronous path (same values, bit for bit)."""
import queue
import threading

import numpy as np
import torch

from . import tools
from . import transform
from ._synth_worker import SNRS, synth_bits, synth_chunk, synth_raw as _synth_raw      # noqa: F401  (re-exported)

DATA_REQUIRED_SR = 14000          # M1/dataset.py:38
CLIP_SECONDS = 2.0
FPS = 30.0
PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION = 'training', 'testing', 'pred'


def host_draws(start, batch, n_samples=int(DATA_REQUIRED_SR * CLIP_SECONDS), sr=DATA_REQUIRED_SR, fps=FPS, snr=None):
    """The host half of `batch` synthetic clips start .. start + batch - 1 (seeded per clip): dict of numpy arrays speech /
    noise (B, n) f32 (un-gated speech, coloured noise), bits (B, n_frames) u8, snr list."""
    sp, nz, bits, snrs = synth_chunk((start, batch, n_samples, sr, fps, snr))
    return dict(speech=sp, noise=nz, bits=bits, snr=snrs)


def synth_batch_device(start, batch, n_samples=int(DATA_REQUIRED_SR * CLIP_SECONDS), sr=DATA_REQUIRED_SR, fps=FPS, snr=None,
                       device="cuda", draws=None):
    """`batch` synthetic clips -> dict of DEVICE tensors mixed / clean / full_noise (B, n) f32 + host bits (B, n_frames) u8 and the
    snr list.  draws: host_draws() output (numpy, or already-uploaded tensors for speech / noise / bits_dev).  The speech is
    gated with the SAME per-sample mask the pipeline derives from the bits (sos_bits_to_mask: silent samples are exactly zero,
    so mixed == clean + full_noise sample for sample, M2/dataset.py:183 then :217); the mix at the SNR + peak normalisation is
    sos_add_signals_f32."""
    d = host_draws(start, batch, n_samples, sr, fps, snr) if draws is None else draws
    up = lambda a: a if torch.is_tensor(a) else torch.from_numpy(a).to(device)      # noqa: E731
    sp, nz = up(d["speech"]), up(d["noise"])
    bits_dev = up(d["bits_dev"] if "bits_dev" in d else d["bits"])
    mask = tools.bits_to_mask_batch(bits_dev, float(sr) / fps, n_samples)
    # (snr_dev: the SNRs already on the device -- the asynchronous loaders upload them with the waveforms; a Python list here
    # is a pageable, i.e. synchronising, host -> device copy)
    mixed, clean, noise = tools.add_signals_batch(sp * (1 - mask), nz, d.get("snr_dev", d["snr"]), norm=0.5)
    return dict(mixed=mixed, clean=clean, full_noise=noise, bits=d["bits"], bits_dev=bits_dev, snr=d["snr"])


def synth_batch(start, batch, n_samples=int(DATA_REQUIRED_SR * CLIP_SECONDS), sr=DATA_REQUIRED_SR, fps=FPS, snr=None,
                device="cuda"):
    """`batch` synthetic clips start .. start+batch-1 -> dict of HOST arrays mixed / clean / full_noise (B, n) f32,
    bits (B, n_frames) u8, snr list (the host view of synth_batch_device: tests, the oracle and the CPU baseline read it)."""
    r = synth_batch_device(start, batch, n_samples, sr, fps, snr, device)
    return dict(mixed=r["mixed"].cpu().numpy(), clean=r["clean"].cpu().numpy(), full_noise=r["full_noise"].cpu().numpy(),
                bits=r["bits"], snr=r["snr"])


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


class _HostPool:
    """`num_workers` worker PROCESSES (spawn: no forked HIP state, workers import numpy / scipy only) making the host draws of
    synthetic batches -- the counterpart of the reference's DataLoader workers (M2/dataset.py:44-50).  0 workers: in-process.
    (Threads do not help here: the draws are many small numpy calls that hold the GIL -- 1.4 ms per clip on one thread, 1.6 ms
    on four.)"""

    def __init__(self, num_workers):
        self.n = max(0, int(num_workers))
        self.ex = None

    def _pool(self):
        if self.ex is None:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            self.ex = ProcessPoolExecutor(self.n, mp_context=mp.get_context("spawn"))
        return self.ex

    def submit(self, start, batch, n_samples, sr, fps, snr):
        """Start the draws of clips start .. start + batch - 1; returns a callable that waits for them: host_draws() dict."""
        if self.n == 0:
            return lambda: host_draws(start, batch, n_samples, sr, fps, snr)
        per = (batch + self.n - 1) // self.n
        futs = [self._pool().submit(synth_chunk, (start + o, min(per, batch - o), n_samples, sr, fps, snr)) for o in range(0, batch, per)]

        def wait():
            parts = [f.result() for f in futs]
            return dict(speech=np.concatenate([q[0] for q in parts]), noise=np.concatenate([q[1] for q in parts]),
                        bits=np.concatenate([q[2] for q in parts]), snr=[v for q in parts for v in q[3]])
        return wait

    def close(self):
        if self.ex is not None:
            self.ex.shutdown(wait=False, cancel_futures=True)
            self.ex = None


class _Prefetcher:
    """Producer thread of an asynchronous loader.  For batch k = 0 .. n - 1: host(k) -> dict of numpy arrays (blocking: worker
    results, clip cutting); the arrays named in `upload` are copied into PINNED staging buffers (a ring of depth + 2 sets, a set is
    reused only after its copies' event has completed) and uploaded with non-blocking copies on `side`; device(k, dict) builds the
    batch dict on `side`; (batch, event) goes into a queue of `depth` entries.  The consumer makes its current stream wait for the
    event and records the batch's tensors on it (they were allocated from the side stream's pool)."""

    def __init__(self, n, host, device_fn, upload, device, depth=2):
        self.n, self.host, self.device_fn, self.upload, self.depth = n, host, device_fn, upload, max(1, int(depth))
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.side = torch.cuda.Stream(self.device)
        self.q = queue.Queue(maxsize=self.depth)
        self.stop = threading.Event()
        self.pinned = [dict() for _ in range(self.depth + 2)]
        self.copied = [None] * (self.depth + 2)
        self.thread = threading.Thread(target=self._run, name="sos-loader", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            for k in range(self.n):
                if self.stop.is_set():
                    return
                raw = self.host(k)
                slot = k % len(self.pinned)
                if self.copied[slot] is not None:
                    self.copied[slot].synchronize()             # the uploads that last used this staging set are done
                with torch.cuda.stream(self.side):
                    dev = dict(raw)
                    for key in self.upload:
                        a = raw[key]
                        buf = self.pinned[slot].get(key)
                        if buf is None or buf.shape != a.shape or buf.dtype != torch.from_numpy(a).dtype:
                            buf = self.pinned[slot][key] = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True)
                        buf.numpy()[...] = a
                        dev[key + "_dev"] = buf.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.copied[slot] = ev
                    batch = self.device_fn(k, dev)
                    ready = torch.cuda.Event()
                    ready.record(self.side)
                while not self.stop.is_set():
                    try:
                        self.q.put((batch, ready), timeout=0.1)
                        break
                    except queue.Full:
                        pass
        except BaseException as e:                               # handed to the consumer, which re-raises it
            self.q.put((e, None))

    def __iter__(self):
        try:
            for _ in range(self.n):
                batch, ready = self.q.get()
                if ready is None:
                    raise batch
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)
                for v in batch.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                yield batch
        finally:
            self.close()

    def close(self):
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=10)


class _LazyHost(dict):
    """dict whose device tensors become numpy arrays on first access (the file loader's `_raw` view of a batch: tests and
    diagnostics read it, the training loop never does -- no device -> host copy unless somebody asks)."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if torch.is_tensor(v):
            v = v.cpu().numpy()
            dict.__setitem__(self, k, v)
        return v


class _SyntheticLoader:
    """Global batch b of the epoch = clips [b*G, (b+1)*G) with G = batch_size * world_size; rank r takes the r-th
    slice of `batch_size` clips (disjoint across ranks, identical for every world size)."""

    def __init__(self, model, phase, batch_size, n_batches, device, rank=0, world_size=1, num_workers=0, prefetch=2):
        self.model, self.phase, self.batch_size, self.n_batches, self.device = model, phase, batch_size, n_batches, device
        self.rank, self.world_size = rank, world_size
        self.num_workers, self.prefetch = max(0, int(num_workers)), prefetch
        self.pool = _HostPool(self.num_workers) if self.num_workers else None

    def __len__(self):
        return self.n_batches

    def starts(self):
        off = {PHASE_TRAINING: 0, PHASE_TESTING: 10 ** 6, PHASE_PREDICTION: 2 * 10 ** 6}[self.phase]
        return [off + (b * self.world_size + self.rank) * self.batch_size for b in range(self.n_batches)]

    def __iter__(self):
        starts = self.starts()
        if not self.num_workers or not torch.cuda.is_available():
            for st in starts:
                yield make_batch(self.model, st, self.batch_size, self.device)
            return
        n = int(DATA_REQUIRED_SR * CLIP_SECONDS)
        # the draws of the first `prefetch + 1` batches are requested at once, then one more whenever one is taken
        ahead = min(len(starts), self.prefetch + 1)
        waits = [self.pool.submit(st, self.batch_size, n, DATA_REQUIRED_SR, FPS, None) for st in starts[:ahead]]

        def host(k):
            d = waits[k]()
            waits[k] = None
            if k + ahead < len(starts):
                waits.append(self.pool.submit(starts[k + ahead], self.batch_size, n, DATA_REQUIRED_SR, FPS, None))
            d["snr_arr"] = np.asarray(d["snr"], dtype=np.float32)
            return d

        def device_fn(k, d):
            r = synth_batch_device(starts[k], self.batch_size, device=self.device,
                                   draws=dict(speech=d["speech_dev"], noise=d["noise_dev"], bits=d["bits"], bits_dev=d["bits_dev"],
                                              snr=d["snr"], snr_dev=d["snr_arr_dev"]))
            return batch_from_raw(self.model, r, list(range(starts[k], starts[k] + self.batch_size)), self.device)

        yield from _Prefetcher(len(starts), host, device_fn, ("speech", "noise", "bits", "snr_arr"), self.device, self.prefetch)

    def close(self):
        if self.pool is not None:
            self.pool.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_batch(model, start, batch_size, device="cuda"):
    return batch_from_raw(model, synth_batch_device(start, batch_size, device=device), list(range(start, start + batch_size)), device)


def batch_from_raw(model, raw, starts, device="cuda"):
    """raw: dict of mixed / clean / full_noise (B, n) f32 -- host arrays or device tensors -- and bits (B, n_frames) u8 on the
    host (+ optionally bits_dev, the same on the device) -> the reference's batch dict, every transform on the GPU.  The
    denoiser's dict also carries `_bits` (u8 (B, n_frames) on the device: the labels a detector step on the same clips takes)."""
    up = lambda a: a if torch.is_tensor(a) else torch.from_numpy(a).to(device)      # noqa: E731
    batch_size = len(raw["mixed"])
    n = raw["mixed"].shape[1]
    mixed = up(raw["mixed"])
    bits = up(raw["bits_dev"]) if "bits_dev" in raw else up(raw["bits"])
    if model == "detector":
        # M1/dataset.py:348-352: label 1 = non-silent, audio = STFT of the mixed clip
        return {"label": bits.float(), "audio": transform.stft_batch(mixed)}
    clean = up(raw["clean"])
    full_noise = up(raw["full_noise"])
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
            "start": list(starts), "bitstream": ["".join(map(str, r)) for r in raw["bits"]], "_bits": bits}


class _FileLoader:
    """Clips cut from real recordings (the reference's corpora layout): dataset JSON (PP/tools.py:28-31) + noise WAVE
    files.  Items as the reference makes them -- denoiser: 2 s windows every second of the labelled part of a file
    (M2/tools.py:134-177); detector: 60-frame windows every 30 frames, cut / zero-padded to 28 000 samples
    (M1/tools.py:297-332, M1/dataset.py:231-236) -- the clean clip silenced on its labelled silent intervals, a
    random crop of a random noise file mixed in at a random (or the `snr_idx`-th) SNR with add_signals semantics,
    peak 0.5 (M2/dataset.py:155-208).  Files are decoded and resampled once (GPU) and kept in host memory; draws
    come from a seeded numpy generator (the reference uses unseeded worker RNGs).  Cutting is slicing (cheap): with
    num_workers > 0 it runs in the producer thread and the device half on the side stream (_Prefetcher)."""

    def __init__(self, model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed, rank=0,
                 world_size=1, num_workers=0, prefetch=2):
        import json
        import os
        from . import audio_io
        self.model, self.phase, self.batch_size, self.device, self.snr_idx = model, phase, batch_size, device, snr_idx
        self.num_workers, self.prefetch = max(0, int(num_workers)), prefetch
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

    def _host(self, idx):
        """The host half of one batch: cut the clips, draw the noise crops and the SNRs (in batch order: the draws of a seeded
        run do not depend on num_workers)."""
        clips = [self._clip(self.items[k]) for k in idx]
        snrs, crops = [], []
        for _ in idx:
            snrs.append(SNRS[self.snr_idx] if self.snr_idx is not None else SNRS[int(self.rng.integers(len(SNRS)))])
            nz = self.noises[int(self.rng.integers(len(self.noises)))]
            st = int(self.rng.integers(0, len(nz) - self.n_clip + 1))
            crops.append(nz[st:st + self.n_clip].astype(np.float32))
        return dict(audio=np.stack([c[0] for c in clips]), bits=np.stack([c[1] for c in clips]), crops=np.stack(crops), snr=snrs,
                    snr_arr=np.asarray(snrs, dtype=np.float32), starts=[self.items[k][1] for k in idx])

    def _device(self, d):
        """The device half: silent intervals truly silent before mixing (M2/dataset.py:160-183: the sample mask comes from the GPU
        kernel), the mix at the SNR, the batch dict.  d: _host() output, `*_dev` = already-uploaded copies if present."""
        up = lambda k: d[k + "_dev"] if k + "_dev" in d else torch.from_numpy(d[k]).to(self.device)      # noqa: E731
        audio, crops, bits_dev = up("audio"), up("crops"), up("bits")
        mask = tools.bits_to_mask_batch(bits_dev, DATA_REQUIRED_SR / FPS, self.n_clip)
        mixed, clean, fn = tools.add_signals_batch(audio * (1 - mask), crops, d.get("snr_arr_dev", d["snr"]), norm=0.5)
        raw = dict(mixed=mixed, clean=clean, full_noise=fn, bits=d["bits"], bits_dev=bits_dev, snr=d["snr"])
        batch = batch_from_raw(self.model, raw, d["starts"], self.device)
        batch["_raw"] = _LazyHost(mixed=mixed, clean=clean, full_noise=fn, bits=d["bits"], snr=d["snr"])
        return batch

    def __iter__(self):
        order = np.arange(len(self.items))
        if self.phase == PHASE_TRAINING:
            self.order_rng.shuffle(order)
        order = order[shard_indices(len(order), self.rank, self.world_size)]
        parts = [order[s0:s0 + self.batch_size] for s0 in range(0, len(order), self.batch_size)]
        if not self.num_workers or not torch.cuda.is_available():
            for idx in parts:
                yield self._device(self._host(idx))
            return
        yield from _Prefetcher(len(parts), lambda k: self._host(parts[k]), lambda k, d: self._device(d), ("audio", "crops", "bits", "snr_arr"),
                               self.device, self.prefetch)


def get_dataloader(phase, batch_size=4, num_workers=4, snr_idx=None, dataset_json=None, clean_audio=True,
                   model="denoiser", n_batches=8, device="cuda", noise_files=None, data_root=None, seed=0, rank=None,
                   world_size=None, prefetch=2):
    """Reference signature + `model` ('detector' -> M1 schema, 'denoiser' -> M2 schema).  With `dataset_json` and
    `noise_files` the clips come from real recordings (see _FileLoader); otherwise they are synthesised.
    `batch_size` is PER RANK; `rank` / `world_size` (default: torch.distributed's, else 0 / 1) shard the clips so that
    the ranks of a data-parallel job see disjoint data (the reference's single-process DataParallel scatters one
    loader's batch over the GPUs, M2/dataset.py:44-50 + M2/agent.py:151).
    `num_workers` (M2/dataset.py:44-50: DataLoader workers): worker processes of the host draws; > 0 also moves the batch
    construction onto a producer thread + side stream, `prefetch` batches ahead of the consumer (pinned staging buffers,
    non-blocking uploads: the reference's pin_memory=True).  0 = synchronous, in the caller's thread and stream.  The batches
    are the same either way, bit for bit."""
    assert phase in (PHASE_TRAINING, PHASE_TESTING, PHASE_PREDICTION)
    if rank is None or world_size is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank = (dist.get_rank() if on else 0) if rank is None else rank
        world_size = (dist.get_world_size() if on else 1) if world_size is None else world_size
    if dataset_json is not None and noise_files:
        return _FileLoader(model, phase, batch_size, dataset_json, noise_files, snr_idx, data_root, device, seed, rank,
                           world_size, num_workers, prefetch)
    return _SyntheticLoader(model, phase, batch_size, n_batches, device, rank, world_size, num_workers, prefetch)
