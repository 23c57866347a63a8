#!/usr/bin/env python3
"""Golden vectors of the objective measures: the reference's M2/metrics.py functions that run
without pypesq / pystoi / soundfile (stubbed: never called) on closed-form signals, with the oracle asserted equal.
Usage:  python tests/golden/make_goldens_metrics.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import M2, OUT, _stub, hashed, install_stubs, load  # noqa: E402
from oracle import metrics as om  # noqa: E402


def signals(idx, n, sr):
    t = np.arange(n) / sr
    env = (np.sin(2 * np.pi * 0.9 * t + 0.4) > -0.3).astype(np.float64)
    clean = env * (0.3 * np.sin(2 * np.pi * 210 * t * (1 + 0.2 * np.sin(2 * np.pi * 2.5 * t))) + 0.1 * np.sin(2 * np.pi * 1900 * t))
    clean = clean + 0.002 * hashed(idx, (n,))
    noisy = 0.9 * clean + 0.04 * hashed(idx + 1, (n,)) + 0.01 * np.sin(2 * np.pi * 50 * t)
    return clean.astype(np.float32), noisy.astype(np.float32)


def main():
    install_stubs()
    _stub("soundfile"); _stub("pypesq", pesq=None); _stub("pystoi"); _stub("pystoi.stoi", stoi=None)
    ref = load(os.path.join(M2, "metrics.py"), "ref_metrics")
    g = {}
    for tag, n, sr in (("a", 16000 * 2 + 37, 16000), ("b", 8000 * 3, 8000)):
        clean, noisy = signals(900 + ord(tag), n, sr)
        r = dict(l1=ref.metrics_L1(noisy[:n - 500], clean), ssnr=ref.metrics_ssnr(clean, noisy, srate=sr),
                 ssnr0=ref.metrics_ssnr(clean, noisy, srate=sr, min_snr=0, eps=1e-20),
                 shift=ref.metrics_ssnr_shift(clean, noisy, srate=sr), exsi=ref.metrics_ssnr_exclude_silence(clean, noisy, srate=sr),
                 llr=np.asarray(ref.llr(clean, noisy, sr), dtype=np.float64), wss=np.asarray(ref.wss(clean, noisy, sr), dtype=np.float64))
        o = dict(l1=om.metrics_L1(noisy[:n - 500], clean), ssnr=om.metrics_ssnr(clean, noisy, sr),
                 ssnr0=om.metrics_ssnr(clean, noisy, sr, min_snr=0, eps=1e-20), shift=om.metrics_ssnr_shift(clean, noisy, sr),
                 exsi=om.metrics_ssnr_exclude_silence(clean, noisy, sr), llr=om.llr(clean, noisy, sr), wss=np.asarray(om.wss(clean, noisy, sr)))
        for k in r:
            a, b = np.asarray(r[k], dtype=np.float64), np.asarray(o[k], dtype=np.float64)
            assert a.shape == b.shape and np.max(np.abs(a - b)) <= 2e-5 * (np.max(np.abs(a)) + 1e-12), (tag, k, np.max(np.abs(a - b)))
            g[f"{tag}_{k}"] = a
        g[f"{tag}_idx"] = np.array([900 + ord(tag), n, sr])
        print(tag, {k: (np.round(np.asarray(v, dtype=np.float64).reshape(-1)[:2], 4)) for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **g)
    print("metrics ok")


if __name__ == "__main__":
    main()
