"""Objective measures: oracle vs goldens from the reference's M2/metrics.py, HIP vs both.
Tolerances: oracle 2e-5; HIP scalars 1e-4 relative, per-frame LLR 2e-3 absolute (f32 quadratic forms of nearly
singular Toeplitz matrices, as in the reference), per-frame WSS 2e-3 relative (f32 direct DFT vs f64 FFT)."""
import numpy as np
import pytest

from oracle import metrics as om
from util import hashed


def signals(idx, n, sr):
    t = np.arange(n) / sr
    env = (np.sin(2 * np.pi * 0.9 * t + 0.4) > -0.3).astype(np.float64)
    clean = env * (0.3 * np.sin(2 * np.pi * 210 * t * (1 + 0.2 * np.sin(2 * np.pi * 2.5 * t))) + 0.1 * np.sin(2 * np.pi * 1900 * t))
    clean = clean + 0.002 * hashed(idx, (n,))
    noisy = 0.9 * clean + 0.04 * hashed(idx + 1, (n,)) + 0.01 * np.sin(2 * np.pi * 50 * t)
    return clean.astype(np.float32), noisy.astype(np.float32)


def _close(a, b, rel):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.max(np.abs(a - b)) <= rel * (np.max(np.abs(b)) + 1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_goldens(golden, tag):
    g = golden("metrics")
    idx, n, sr = [int(v) for v in g[f"{tag}_idx"]]
    clean, noisy = signals(idx, n, sr)
    assert _close(om.metrics_L1(noisy[:n - 500], clean), g[f"{tag}_l1"], 2e-5)
    assert _close(om.metrics_ssnr(clean, noisy, sr), g[f"{tag}_ssnr"], 2e-5)
    assert _close(om.metrics_ssnr(clean, noisy, sr, min_snr=0, eps=1e-20), g[f"{tag}_ssnr0"], 2e-5)
    assert _close(om.metrics_ssnr_shift(clean, noisy, sr), g[f"{tag}_shift"], 2e-5)
    assert _close(om.metrics_ssnr_exclude_silence(clean, noisy, sr), g[f"{tag}_exsi"], 2e-5)
    assert _close(om.llr(clean, noisy, sr), g[f"{tag}_llr"], 2e-5)
    assert _close(om.wss(clean, noisy, sr), g[f"{tag}_wss"], 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_metrics_match_goldens(golden, tag):
    import torch
    from sos_amd import metrics as M
    g = golden("metrics")
    idx, n, sr = [int(v) for v in g[f"{tag}_idx"]]
    clean, noisy = signals(idx, n, sr)
    assert _close(M.metrics_L1(noisy[:n - 500], clean), g[f"{tag}_l1"], 1e-5)
    assert _close(M.metrics_ssnr(clean, noisy, srate=sr), g[f"{tag}_ssnr"], 1e-4)
    assert _close(M.metrics_ssnr(clean, noisy, srate=sr, min_snr=0, eps=1e-20), g[f"{tag}_ssnr0"], 1e-4)
    assert _close(M.metrics_ssnr_shift(clean, noisy, srate=sr), g[f"{tag}_shift"], 1e-4)
    assert _close(M.metrics_ssnr_exclude_silence(clean, noisy, srate=sr), g[f"{tag}_exsi"], 1e-4)
    l = M.llr(torch.from_numpy(clean).cuda(), torch.from_numpy(noisy).cuda(), sr)      # GPU tensors work too
    assert l.shape == g[f"{tag}_llr"].shape and np.max(np.abs(l - g[f"{tag}_llr"])) < 2e-3
    w = np.asarray(M.wss(clean, noisy, sr))
    assert _close(w, g[f"{tag}_wss"], 2e-3)
    # composite: PESQ-free parts always, regression outputs when a PESQ value is supplied
    ref = om.composite(clean, noisy, sr, eps=1e-20, pesq_raw=2.7)
    c = M.CompositeEval(clean, noisy, sr, eps=1e-20, pesq_raw=2.7)
    assert _close([c[0], c[1], c[2]], [ref["csig"], ref["cbak"], ref["covl"]], 2e-3) and c[3] == 2.7
    assert _close([c[4], c[5]], [ref["segSNR"], ref["overall_snr"]], 1e-4)
    m = M.evaluate_metrics(noisy, clean, sr=sr)
    assert list(m) == ["l1", "stoi", "csig", "cbak", "covl", "pesq", "ssnr_regular", "ssnr_shift", "ssnr_clip", "ssnr_exsi", "overall_snr"]
    assert m["pesq"] is None and m["stoi"] is None and m["csig"] is None and np.isfinite(m["ssnr_exsi"])
    with pytest.raises(AssertionError):
        M.llr(clean, noisy[:-1], sr)
