"""TEST INFRASTRUCTURE ONLY -- numpy (f64) restatement of the objective measures of the reference's
M2/metrics.py that do not need absent third-party packages: metrics_L1 (:40-45), metrics_ssnr (:86-130),
metrics_ssnr_shift (:132-176), metrics_ssnr_exclude_silence (:178-244), llr (:561-623) with lpcoeff (:626-681),
wss (:404-558) and the composite formulas of CompositeEval (:346-402, PESQ supplied by the caller).
Pinned against the imported reference functions (tests/golden/make_goldens_metrics.py -> metrics.npz).
PESQ (pypesq) and STOI (pystoi) are absent: not restated."""
import numpy as np

CENT_FREQ = [50., 120, 190, 260, 330, 400, 470, 540, 617.372, 703.378, 798.717, 904.128, 1020.38, 1148.30, 1288.72, 1442.54,
             1610.70, 1794.16, 1993.93, 2211.08, 2446.71, 2701.97, 2978.04, 3276.17, 3597.63]
BANDWIDTH = [70., 70, 70, 70, 70, 70, 70, 77.3724, 86.0056, 95.3398, 105.411, 116.256, 127.914, 140.423, 153.823, 168.154,
             183.457, 199.776, 217.153, 235.631, 255.255, 276.072, 298.126, 321.465, 346.136]


def frame_setup(n, srate, win_ms=30):
    winlength = int(np.round(win_ms * srate / 1000))
    skip = winlength // 4
    num_frames = int(n / skip - (winlength / skip))
    time = np.linspace(1, winlength, winlength) / (winlength + 1)
    return winlength, skip, max(num_frames, 0), 0.5 * (1 - np.cos(2 * np.pi * time))


def _frames(x, winlength, skip, num_frames, window):
    idx = np.arange(num_frames)[:, None] * skip + np.arange(winlength)[None, :]
    return np.asarray(x, dtype=np.float64)[idx] * window[None, :]


def metrics_L1(output, target):
    output, target = np.asarray(output, np.float64), np.asarray(target, np.float64)
    steps = np.linspace(0, len(output) - 1, len(target))
    return float(np.mean(np.abs(np.interp(steps, np.arange(len(output)), output) - target)))


def _ssnr(ref, deg, frames_ref, frames_deg, srate, min_snr, max_snr, eps, inner):
    ref, deg = np.asarray(ref, np.float64), np.asarray(deg, np.float64)
    overall = 10 * np.log10(np.sum(ref ** 2) / (np.sum((ref - deg) ** 2) + eps))
    w, s, nf, win = frame_setup(len(frames_ref), srate)
    c, p = _frames(frames_ref, w, s, nf, win), _frames(frames_deg, w, s, nf, win)
    se, ne = np.sum(c ** 2, axis=1), np.sum((c - p) ** 2, axis=1)
    seg = np.clip(10 * np.log10(se / (ne + eps) + inner), min_snr, max_snr)
    return float(overall), float(np.nanmean(seg)) if nf else float("nan")


def metrics_ssnr(ref, deg, srate=16000, min_snr=-10, max_snr=35, eps=1e-10):
    return _ssnr(ref, deg, ref, deg, srate, min_snr, max_snr, eps, eps)


def metrics_ssnr_shift(ref, deg, srate=16000, min_snr=-10, max_snr=35, eps=1e-10):
    return _ssnr(ref, deg, ref, deg, srate, min_snr, max_snr, eps, 1.0)


def metrics_ssnr_exclude_silence(ref, deg, srate=16000, min_snr=-10, max_snr=35, eps=1e-10):
    ref, deg = np.asarray(ref), np.asarray(deg)
    keep = ~(np.abs(ref) < np.max(np.abs(ref)) * 0.03)
    return _ssnr(ref, deg, ref[keep], deg[keep], srate, min_snr, max_snr, eps, eps)


def lpcoeff(frame, order):
    n = len(frame)
    R = np.array([np.sum(frame[:n - k] * frame[k:]) for k in range(order + 1)])
    a = np.ones(order)
    E = R[0]
    for i in range(order):
        past = a[:i].copy()
        rc = (R[i + 1] - np.sum(past * R[i:0:-1])) / E
        a[i] = rc
        a[:i] = past - rc * past[::-1]
        E = (1 - rc * rc) * E
    return R.astype(np.float32), np.concatenate([[1.0], -a]).astype(np.float32)


def llr(ref, deg, srate):
    w, s, nf, win = frame_setup(len(ref), srate)
    P = 10 if srate < 10000 else 16
    c, p = _frames(ref, w, s, nf, win), _frames(deg, w, s, nf, win)
    out = np.zeros(nf, dtype=np.float32)
    ii = np.abs(np.arange(P + 1)[:, None] - np.arange(P + 1)[None, :])
    for f in range(nf):
        Rc, Ac = lpcoeff(c[f], P)
        _, Ap = lpcoeff(p[f], P)
        T = Rc[ii]
        out[f] = np.log((Ap[None] @ T @ Ap[:, None]) / (Ac[None] @ T @ Ac[:, None]))[0, 0]
    return out


def crit_filters(srate, n_fft):
    half = n_fft // 2
    max_freq = srate / 2
    min_factor = np.exp(-30. / (2 * 2.303))
    j = np.arange(half)
    out = np.zeros((25, half))
    for i in range(25):
        f0 = np.floor((CENT_FREQ[i] / max_freq) * half)
        bw = (BANDWIDTH[i] / max_freq) * half
        row = np.exp(-11 * (((j - f0) / bw) ** 2) + np.log(BANDWIDTH[0]) - np.log(BANDWIDTH[i]))
        out[i] = row * (row > min_factor)
    return out


def _wss_frame(en_c, en_p):
    def peaks(en):
        slope = en[1:] - en[:-1]
        pk = np.zeros(24)
        for i in range(24):
            n = i
            if slope[i] > 0:
                while n < 24 and slope[n] > 0:
                    n += 1
                pk[i] = en[n - 1]
            else:
                while n >= 0 and slope[n] <= 0:
                    n -= 1
                pk[i] = en[n + 1]
        return slope, pk
    sc, pc = peaks(en_c)
    sp, pp = peaks(en_p)
    Wc = (20 / (20 + en_c.max() - en_c[:24])) * (1 / (1 + pc - en_c[:24]))
    Wp = (20 / (20 + en_p.max() - en_p[:24])) * (1 / (1 + pp - en_p[:24]))
    W = (Wc + Wp) / 2
    return float(np.sum(W * (sc - sp) ** 2) / np.sum(W))


def wss(ref, deg, srate, eps=1e-10):
    w, s, nf, win = frame_setup(len(ref), srate)
    n_fft = int(2 ** np.ceil(np.log(2 * w) / np.log(2)))
    cf = crit_filters(srate, n_fft)
    c, p = _frames(ref, w, s, nf, win), _frames(deg, w, s, nf, win)
    half = n_fft // 2
    out = []
    for f in range(nf):
        sc = np.abs(np.fft.fft(c[f], n_fft))[:half] ** 2
        spp = np.abs(np.fft.fft(p[f], n_fft))[:half] ** 2
        out.append(_wss_frame(10 * np.log10(np.maximum(cf @ sc, eps)), 10 * np.log10(np.maximum(cf @ spp, eps))))
    return out


def composite(ref, deg, srate=16000, eps=1e-10, pesq_raw=None):
    """CompositeEval without the PESQ call: (wss_dist, llr_mean, segSNR(min 0), overall_snr) and, when a PESQ value is
    supplied, (Csig, Cbak, Covl) by the reference's regression formulas."""
    n = min(len(ref), len(deg))
    ref, deg = np.asarray(ref)[:n], np.asarray(deg)[:n]
    wv = sorted(wss(ref, deg, srate, eps))
    wss_dist = float(np.nanmean(wv[:int(round(len(wv) * 0.95))]))
    lv = sorted(llr(ref, deg, srate))
    llr_mean = float(np.nanmean(lv[:round(len(lv) * 0.95)]))
    overall, seg = metrics_ssnr(ref, deg, srate, min_snr=0, eps=eps)
    res = dict(wss_dist=wss_dist, llr_mean=llr_mean, segSNR=seg, overall_snr=overall)
    if pesq_raw is not None:
        trim = lambda v: min(max(v, 1), 5)   # noqa: E731
        res.update(csig=trim(3.093 - 1.029 * llr_mean + 0.603 * pesq_raw - 0.009 * wss_dist),
                   cbak=trim(1.634 + 0.478 * pesq_raw - 0.007 * wss_dist + 0.063 * seg),
                   covl=trim(1.594 + 0.805 * pesq_raw - 0.512 * llr_mean - 0.007 * wss_dist))
    return res
