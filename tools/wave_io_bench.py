"""Wave front door micro-benchmark (run on the GPU box): 16-bit stereo 44.1 kHz -> mono f32 -> 14 kHz.
Prints per-kernel time, the HBM rate of the conversion (its roofline) and the tap rate of the resampler
(VALU/LDS bound: 2 x ~202 taps of one LDS pair read + one input read + 2 FMAs per output sample)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sos_amd  # noqa: E402,F401
from sos_amd import audio_io  # noqa: E402


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for minutes in (0.36, 5, 60):      # 21.5 s = the reference's sample files; an hour-long recording
    n = int(44100 * 60 * minutes)
    pcm = torch.randint(-32768, 32767, (n, 2), dtype=torch.int16, device="cuda")
    t1 = timed(lambda: audio_io.pcm_to_mono_device(pcm, "s16"))
    mono = audio_io.pcm_to_mono_device(pcm, "s16")
    t2 = timed(lambda: audio_io.resample_device(mono, 44100, 14000))
    n_out = int(np.ceil(n * 14000 / 44100))
    taps = 2 * ((32769) // 162)
    print(f"{minutes:6.2f} min  pcm->mono {t1 * 1e3:8.1f} us ({n * 8 / t1 / 1e6:7.1f} GB/s)   resample {t2 * 1e3:8.1f} us "
          f"({n_out * taps / t2 / 1e6:7.1f} Gtap/s, {n / 44100 / (t1 + t2) * 1e3:9.0f} x real time)")
