"""Host-side draws of the synthetic clips -- numpy / scipy only, NO torch import: this is the module the data
loader's worker PROCESSES import (dataset._HostPool, the counterpart of the reference's DataLoader(num_workers=...) workers,
M2/dataset.py:44-50), and a spawned worker must neither initialise HIP nor pay torch's import time."""
import numpy as np
from scipy.signal import lfilter

SNRS = [-10, -7, -3, 0, 3, 7, 10]


def synth_bits(rng, n_frames, p_silent=0.3, min_run=5):
    """Per-video-frame labels, 1 = non-silent, runs of at least `min_run` frames."""
    bits = []
    while len(bits) < n_frames:
        run = int(min_run + rng.integers(0, 12))
        bits += [0 if rng.random() < p_silent else 1] * run
    return np.array(bits[:n_frames], dtype=np.uint8)


def synth_raw(i, n_samples, sr, fps, snr):
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


def synth_chunk(args):
    """(first clip, count, n_samples, sr, fps, snr) -> stacked draws of the clips first .. first + count - 1: speech (count, n)
    f32, noise (count, n) f32, bits (count, n_frames) u8, snr list.  One call per worker task: the arrays cross the process boundary
    once per chunk."""
    first, count, n_samples, sr, fps, snr = args
    d = [synth_raw(first + k, n_samples, sr, fps, snr) for k in range(count)]
    return (np.stack([x[0] for x in d]), np.stack([x[1] for x in d]), np.stack([x[2] for x in d]), [x[3] for x in d])
