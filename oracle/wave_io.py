"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy f64) of the wave front door the reference gets from
`librosa.load(path, sr=14000)` (M1/dataset.py:226; M2/predict.py:288,297,303).

PARITY UNPINNED: the arithmetic lives in third-party packages that are absent from /root/reference and from
this image (requirements.txt: librosa==0.7.1 -> soundfile, resampy>=0.2.2); the reference holds no golden
vectors for it.  Restated from their published algorithms:

  soundfile : integer PCM -> float32 by 1 / 2**(bits-1)
  librosa   : to_mono = mean over channels; resample = resampy.resample(y, sr_orig, sr, filter='kaiser_best')
              then util.fix_length(ceil(n * ratio))   (core/audio.py: load, to_mono, resample)
  resampy   : filters.sinc_window + interpn.resample_f (Smith's band-limited interpolation); `kaiser_best`
              = 64 zero crossings, 512 table points per crossing, Kaiser beta 14.769656459379492,
              roll-off 0.9475937167399596

Cross-checks that do exist (tests/test_oracle_wave_io.py): band-limited test signals against their analytic
resampling, and against scipy.signal.resample_poly."""
import numpy as np
import scipy.signal


def sinc_window(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596):
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample_f(x, sample_ratio, interp_win, num_table):
    """resampy interpn.resample_f, the loop over output samples vectorised (each output keeps the loop's
    tap order; the time register is the same running f64 sum)."""
    x = np.asarray(x, dtype=np.float64)
    n_orig = x.shape[0]
    n_out = int(n_orig * sample_ratio)
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, sample_ratio)
    time_increment = 1.0 / sample_ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    time_register = np.concatenate([[0.0], np.cumsum(np.full(max(n_out - 1, 0), time_increment))])[:n_out]
    n = time_register.astype(np.int64)
    y = np.zeros(n_out)
    frac = scale * (time_register - n)
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    i_max = np.minimum(n + 1, (nwin - offset) // index_step)
    for i in range(int(i_max.max()) if n_out else 0):
        live = i < i_max
        o = np.where(live, offset + i * index_step, 0)
        weight = interp_win[o] + eta * interp_delta[o]
        y += np.where(live, weight * x[np.where(live, n - i, 0)], 0.0)
    frac = scale - frac
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    k_max = np.minimum(n_orig - n - 1, (nwin - offset) // index_step)
    for k in range(int(k_max.max()) if n_out else 0):
        live = k < k_max
        o = np.where(live, offset + k * index_step, 0)
        weight = interp_win[o] + eta * interp_delta[o]
        y += np.where(live, weight * x[np.where(live, n + k + 1, 0)], 0.0)
    return y


def resample(y, orig_sr, target_sr, fix=True):
    """librosa.resample(res_type='kaiser_best')."""
    if orig_sr == target_sr:
        return np.asarray(y, dtype=np.float64)
    ratio = float(target_sr) / orig_sr
    win, num_table = sinc_window()
    if ratio < 1:
        win = win * ratio
    out = resample_f(y, ratio, win, num_table)
    if fix:
        n = int(np.ceil(len(y) * ratio))
        out = np.concatenate([out, np.zeros(n - len(out))]) if n > len(out) else out[:n]
    return out


def pcm_to_float(pcm):
    """soundfile's integer -> float scaling; (n_frames, channels) or (n_frames,)."""
    pcm = np.asarray(pcm)
    if pcm.dtype == np.uint8:
        return ((pcm.astype(np.float64) - 128.0) / 128.0).astype(np.float32)
    if pcm.dtype.kind == "i":
        return (pcm.astype(np.float64) / float(2 ** (8 * pcm.dtype.itemsize - 1))).astype(np.float32)
    return pcm.astype(np.float32)


def to_mono(x):
    """librosa.to_mono on soundfile's (n_frames, channels) layout."""
    x = np.asarray(x)
    return x if x.ndim == 1 else np.mean(x, axis=1, dtype=np.float32)


def load_from_pcm(pcm, sr_native, sr):
    """librosa.load semantics on already decoded samples -> float32 mono at `sr`."""
    y = to_mono(pcm_to_float(pcm))
    if sr is not None and sr != sr_native:
        y = resample(y, sr_native, sr)
    return np.ascontiguousarray(y, dtype=np.float32)
