"""Objective measures of the reference's M2/metrics.py with the same names and arguments:
metrics_L1 (:40-45), metrics_ssnr (:86-130), metrics_ssnr_shift (:132-176), metrics_ssnr_exclude_silence (:178-244),
llr (:561-623), wss (:404-558), CompositeEval (:346-402), evaluate_metrics (:16-33).

The per-sample / per-frame work runs in HIP kernels (csrc/metrics.hip); the per-frame results (a few thousand numbers)
are finalised here on the host exactly like the reference does (log10, clamp, means, the 95 % trimmed means and the
composite regression formulas).  PESQ and STOI come from third-party packages (pypesq, pystoi) that are absent:
`evaluate_metrics` / `CompositeEval` take them as arguments and return None for everything that depends on a
missing one.  Signals: 1-D numpy arrays or GPU tensors.  No CPU fallback."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L

CENT_FREQ = [50., 120, 190, 260, 330, 400, 470, 540, 617.372, 703.378, 798.717, 904.128, 1020.38, 1148.30, 1288.72, 1442.54,
             1610.70, 1794.16, 1993.93, 2211.08, 2446.71, 2701.97, 2978.04, 3276.17, 3597.63]
BANDWIDTH = [70., 70, 70, 70, 70, 70, 70, 77.3724, 86.0056, 95.3398, 105.411, 116.256, 127.914, 140.423, 153.823, 168.154,
             183.457, 199.776, 217.153, 235.631, 255.255, 276.072, 298.126, 321.465, 346.136]
_tables = {}


def _dev(x):
    if not torch.cuda.is_available():
        raise RuntimeError("sos_amd.metrics needs an MI355X: there is no CPU fallback")
    if torch.is_tensor(x):
        L.require_cuda(x)
        return x.detach().reshape(-1).float().contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.float32)).cuda()


def _frame_setup(n, srate, win_ms=30):
    winlength = int(np.round(win_ms * srate / 1000))
    skip = winlength // 4
    num_frames = int(n / skip - (winlength / skip))
    key = ("win", winlength)
    if key not in _tables:
        time = np.linspace(1, winlength, winlength) / (winlength + 1)
        _tables[key] = torch.from_numpy(0.5 * (1 - np.cos(2 * np.pi * time))).cuda()
    return winlength, skip, max(num_frames, 0), _tables[key]


def _totals(ref, deg):
    out = torch.empty(3, dtype=torch.float64, device=ref.device)
    L.check(L.lib().sos_metric_totals(L.ptr(ref), L.ptr(deg), ref.numel(), L.ptr(out), L.stream_ptr()), "sos_metric_totals")
    return out.cpu().numpy()


def _frame_energies(ref, deg, srate):
    w, s, nf, win = _frame_setup(ref.numel(), srate)
    if nf < 1:
        return np.zeros((0, 2))
    out = torch.empty((nf, 2), dtype=torch.float64, device=ref.device)
    L.check(L.lib().sos_metric_frame_energy(L.ptr(ref), L.ptr(deg), ref.numel(), w, s, nf, L.ptr(win), L.ptr(out), L.stream_ptr()),
            "sos_metric_frame_energy")
    return out.cpu().numpy()


def _segmental(en, min_snr, max_snr, eps, inner):
    if len(en) == 0:
        return float("nan")
    seg = 10 * np.log10(en[:, 0] / (en[:, 1] + eps) + inner)
    return float(np.nanmean(np.minimum(np.maximum(seg, min_snr), max_snr)))


def _same_length(ref, deg):
    if ref.numel() != deg.numel():
        raise AssertionError(ref.numel())
    return ref, deg


def metrics_L1(output, target):
    o, t = _dev(output), _dev(target)
    res = torch.empty(1, dtype=torch.float64, device=o.device)
    L.check(L.lib().sos_metric_l1(L.ptr(o), o.numel(), L.ptr(t), t.numel(), L.ptr(res), L.stream_ptr()), "sos_metric_l1")
    return float(res.cpu()[0])


def metrics_ssnr(ref_wav, deg_wav, srate=16000, win_len=30, min_snr=-10, max_snr=35, eps=1e-10):
    ref, deg = _same_length(_dev(ref_wav), _dev(deg_wav))
    tot = _totals(ref, deg)
    overall = 10 * np.log10(tot[0] / (tot[1] + eps))
    return float(overall), _segmental(_frame_energies(ref, deg, srate), min_snr, max_snr, eps, eps)


def metrics_ssnr_shift(ref_wav, deg_wav, srate=16000, win_len=30, min_snr=-10, max_snr=35, eps=1e-10):
    ref, deg = _same_length(_dev(ref_wav), _dev(deg_wav))
    tot = _totals(ref, deg)
    overall = 10 * np.log10(tot[0] / (tot[1] + eps))
    return float(overall), _segmental(_frame_energies(ref, deg, srate), min_snr, max_snr, eps, 1.0)


def metrics_ssnr_exclude_silence(ref_wav, deg_wav, srate=16000, win_len=30, min_snr=-10, max_snr=35, eps=1e-10):
    ref, deg = _same_length(_dev(ref_wav), _dev(deg_wav))
    tot = _totals(ref, deg)
    overall = 10 * np.log10(tot[0] / (tot[1] + eps))
    nc, npz = torch.empty_like(ref), torch.empty_like(deg)
    cnt = torch.zeros(1, dtype=torch.int64, device=ref.device)
    thr = np.float32(np.float32(tot[2]) * np.float32(0.03))
    L.check(L.lib().sos_metric_compact(L.ptr(ref), L.ptr(deg), ref.numel(), float(thr), L.ptr(nc), L.ptr(npz), L.ptr(cnt),
                                       L.stream_ptr()), "sos_metric_compact")
    k = int(cnt.cpu()[0])
    return float(overall), _segmental(_frame_energies(nc[:k].contiguous(), npz[:k].contiguous(), srate), min_snr, max_snr, eps, eps)


def llr(ref_wav, deg_wav, srate):
    ref, deg = _same_length(_dev(ref_wav), _dev(deg_wav))
    w, s, nf, win = _frame_setup(ref.numel(), srate)
    P = 10 if srate < 10000 else 16
    out = torch.empty(max(nf, 1), dtype=torch.float32, device=ref.device)
    if nf:
        L.check(L.lib().sos_metric_llr(L.ptr(ref), L.ptr(deg), ref.numel(), w, s, nf, L.ptr(win), P, L.ptr(out), L.stream_ptr()),
                "sos_metric_llr")
    return out[:nf].cpu().numpy()


def _crit_filters(srate, n_fft, device):
    key = ("crit", srate, n_fft, str(device))
    if key not in _tables:
        half = n_fft // 2
        max_freq = srate / 2
        min_factor = np.exp(-30. / (2 * 2.303))
        j = np.arange(half)
        cf = np.zeros((25, half))
        for i in range(25):
            f0 = np.floor((CENT_FREQ[i] / max_freq) * half)
            bw = (BANDWIDTH[i] / max_freq) * half
            row = np.exp(-11 * (((j - f0) / bw) ** 2) + np.log(BANDWIDTH[0]) - np.log(BANDWIDTH[i]))
            cf[i] = row * (row > min_factor)
        _tables[key] = torch.from_numpy(cf.astype(np.float32)).to(device)
    return _tables[key]


def wss(ref_wav, deg_wav, srate, eps=1e-10):
    ref, deg = _same_length(_dev(ref_wav), _dev(deg_wav))
    w, s, nf, win = _frame_setup(ref.numel(), srate)
    n_fft = int(2 ** np.ceil(np.log(2 * w) / np.log(2)))
    out = torch.empty(max(nf, 1), dtype=torch.float32, device=ref.device)
    if nf:
        cf = _crit_filters(srate, n_fft, ref.device)
        L.check(L.lib().sos_metric_wss(L.ptr(ref), L.ptr(deg), ref.numel(), w, s, nf, L.ptr(win), n_fft, L.ptr(cf), float(eps),
                                       L.ptr(out), L.stream_ptr()), "sos_metric_wss")
    return [float(v) for v in out[:nf].cpu().numpy()]


def CompositeEval(ref_wav, deg_wav, srate=16000, eps=1e-10, pesq_raw=None):
    """(Csig, Cbak, Covl, pesq_raw, segSNR, overall_snr); the first four are None without a PESQ value
    (the reference calls pypesq here, M2/metrics.py:377)."""
    ref, deg = _dev(ref_wav), _dev(deg_wav)
    n = min(ref.numel(), deg.numel())
    ref, deg = ref[:n].contiguous(), deg[:n].contiguous()
    wv = sorted(wss(ref, deg, srate, eps=eps))
    wss_dist = float(np.nanmean(wv[:int(round(len(wv) * 0.95))]))
    lv = sorted(llr(ref, deg, srate))
    llr_mean = float(np.nanmean(lv[:round(len(lv) * 0.95)]))
    overall_snr, segSNR = metrics_ssnr(ref, deg, srate=srate, min_snr=0, eps=eps)
    if pesq_raw is None:
        return None, None, None, None, segSNR, overall_snr

    def trim_mos(val):
        return min(max(val, 1), 5)
    Csig = trim_mos(3.093 - 1.029 * llr_mean + 0.603 * pesq_raw - 0.009 * wss_dist)
    Cbak = trim_mos(1.634 + 0.478 * pesq_raw - 0.007 * wss_dist + 0.063 * segSNR)
    Covl = trim_mos(1.594 + 0.805 * pesq_raw - 0.512 * llr_mean - 0.007 * wss_dist)
    return Csig, Cbak, Covl, pesq_raw, segSNR, overall_snr


def evaluate_metrics(noisy, clean, sr=16000, eps=1e-20, pesq=None, stoi=None):
    """Same keys and order as the reference (M2/metrics.py:16-33); `pesq` / `stoi`: values computed elsewhere
    (pypesq.pesq(clean, noisy, sr), pystoi.stoi(clean, noisy, sr)) or None."""
    csig, cbak, covl, pesq_raw, ssnr, overall_snr = CompositeEval(clean, noisy, sr, eps=eps, pesq_raw=pesq)
    m = OrderedDict()
    m['l1'] = metrics_L1(noisy, clean)
    m['stoi'] = stoi
    m['csig'], m['cbak'], m['covl'], m['pesq'] = csig, cbak, covl, pesq_raw
    m['ssnr_regular'] = metrics_ssnr(clean, noisy, srate=sr, eps=eps)[1]
    m['ssnr_shift'] = metrics_ssnr_shift(clean, noisy, srate=sr, eps=eps)[1]
    m['ssnr_clip'] = ssnr
    m['ssnr_exsi'] = metrics_ssnr_exclude_silence(clean, noisy, srate=sr, eps=eps)[1]
    m['overall_snr'] = overall_snr
    return m
