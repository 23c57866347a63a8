"""oracle.frontend -- numpy float64 restatement of the reference front-end / glue.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Reference paths are relative
to /root/reference; M1 = model_1_silent_interval_detection/audioonly_model,
M2 = model_2_audio_denoising/audio_denoising_model.
"""
import numpy as np

N_FFT = 510        # M1/transform.py:6
HOP_LENGTH = 158   # M1/transform.py:7
WIN_LENGTH = 400   # M1/transform.py:8


def hann_periodic(win_length):
    """scipy.signal.get_window('hann', M, fftbins=True) == periodic hann."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def padded_window(n_fft=N_FFT, win_length=WIN_LENGTH):
    """librosa.util.pad_center(window, n_fft): zeros, window at (n_fft-win)//2."""
    w = np.zeros(n_fft, dtype=np.float64)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def stft_complex(y, n_fft=N_FFT, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """librosa 0.7.1 `stft(y, n_fft, hop, win)` with its defaults: hann (periodic),
    center=True, pad_mode='reflect', dtype=complex64.  Call site M1/transform.py:193.
    Returns complex64 [1+n_fft//2, 1+len(y)//hop]."""
    y = np.asarray(y, dtype=np.float64)
    w = padded_window(n_fft, win_length)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx] * w[:, None]
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def real_imag_expand(c):
    """M1/transform.py:10-17 (dim='new'): complex [F,T] -> float64 [F,T,2]."""
    d = np.zeros((c.shape[0], c.shape[1], 2))
    d[:, :, 0] = np.real(c)
    d[:, :, 1] = np.imag(c)
    return d


def real_imag_shrink(f):
    """M1/transform.py:25-33 (dim='new')."""
    return f[:, :, 0] + f[:, :, 1] * 1j


def power_law(data, power=0.3):
    """M1/transform.py:178-185."""
    data = np.asarray(data)
    mask = np.zeros(data.shape)
    mask[data >= 0] = 1
    mask[data < 0] = -1
    return np.power(np.abs(data), power) * mask


def fast_stft(data, n_fft=N_FFT, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """M1/transform.py:188-193 (power=False)."""
    return real_imag_expand(stft_complex(data, n_fft, hop_length, win_length))


def window_sumsquare(n_frames, n_fft=N_FFT, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """librosa 0.7.1 filters.window_sumsquare (norm=None)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float64)
    wsq = padded_window(n_fft, win_length) ** 2
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += wsq[:max(0, min(n_fft, n - s))]
    return x


def istft_complex(S, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """librosa 0.7.1 `istft(S, hop, win)` defaults: hann, center=True, length=None,
    dtype=float32.  Call site M1/transform.py:199.  Output length hop*(T-1)."""
    n_fft = 2 * (S.shape[0] - 1)
    w = padded_window(n_fft, win_length)
    n_frames = S.shape[1]
    n = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(n, dtype=np.float64)
    ytmp = w[:, None] * np.fft.irfft(np.asarray(S, dtype=np.complex128), n=n_fft, axis=0)
    for i in range(n_frames):
        y[i * hop_length:i * hop_length + n_fft] += ytmp[:, i]
    wss = window_sumsquare(n_frames, n_fft, hop_length, win_length)
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[n_fft // 2:-(n_fft // 2)]
    return y.astype(np.float32)


def fast_istft(F, hop_length=HOP_LENGTH, win_length=WIN_LENGTH):
    """M1/transform.py:196-202 (power=False): [F,T,2] -> float32 [hop*(T-1)]."""
    return istft_complex(real_imag_shrink(np.asarray(F)), hop_length, win_length)


def generate_cRM(Y, S):
    """M1/transform.py:36-54."""
    eps = 1e-8
    M = np.zeros(Y.shape)
    den = Y[:, :, 0] ** 2 + Y[:, :, 1] ** 2 + eps
    M[:, :, 0] = (Y[:, :, 0] * S[:, :, 0] + Y[:, :, 1] * S[:, :, 1]) / den
    M[:, :, 1] = (Y[:, :, 0] * S[:, :, 1] - Y[:, :, 1] * S[:, :, 0]) / den
    return M


def cRM_sigmoid_compress(M, a=0.1, b=0):
    """M1/transform.py:92-94."""
    return 1.0 / (1.0 + np.exp(-a * M + b))


def cRM_sigmoid_recover(O, a=0.1, b=0):
    """M1/transform.py:97-99."""
    return 1.0 / a * (np.log(O / (1 - O + 1e-8) + 1e-10) + b)


def fast_cRM_sigmoid(Fclean, Fmix):
    """M1/transform.py:130-138."""
    return cRM_sigmoid_compress(generate_cRM(Fmix, Fclean))


def fast_icRM_sigmoid(Y, crm):
    """M1/transform.py:141-153: [F,T,2] numpy float64."""
    M = cRM_sigmoid_recover(np.asarray(crm, dtype=np.float64))
    Y = np.asarray(Y, dtype=np.float64)
    S = np.zeros(M.shape)
    S[:, :, 0] = M[:, :, 0] * Y[:, :, 0] - M[:, :, 1] * Y[:, :, 1]
    S[:, :, 1] = M[:, :, 0] * Y[:, :, 1] + M[:, :, 1] * Y[:, :, 0]
    return S


def batch_fast_icRM_sigmoid(Y, crm, a=0.1, b=0):
    """M1/transform.py:156-169 restated in numpy, float32 arithmetic like the torch op.
    Y, crm: (B,2,F,T)."""
    Y = np.asarray(Y, dtype=np.float32)
    crm = np.asarray(crm, dtype=np.float32)
    one = np.float32(1.0)
    M = (one / np.float32(a)) * (np.log(crm / (one - crm + np.float32(1e-8)) + np.float32(1e-10)) + np.float32(b))
    r = M[:, 0] * Y[:, 0] - M[:, 1] * Y[:, 1]
    i = M[:, 0] * Y[:, 1] + M[:, 1] * Y[:, 0]
    return np.stack([r, i], axis=1)


def convert_bitstreammask_to_audiomask(ref_audio_signal, ratio, bitstream):
    """M2/tools.py:340-362 (string bits), M1/tools.py:770-792 (int bits),
    M2/predict.py:232-252 (anything else treated as non-silent when lenient).
    Sample mask: 1 on silent samples.  Integer index rule int(i*r) : int((i+1)*r - 1),
    then ONE pass over the original runs flipping runs shorter than 5 samples."""
    n = len(ref_audio_signal)
    mask = np.zeros(n, dtype=np.asarray(ref_audio_signal).dtype)
    for i, bit in enumerate(bitstream):
        lo = int(i * ratio)
        hi = int((i + 1) * ratio - 1)
        if bit in ("0", 0):
            mask[lo:hi] = 1
        elif bit in ("1", 1):
            mask[lo:hi] = 0
        else:
            raise RuntimeError("Invalid bit?")
    # run-length pass: itertools.groupby only ever reads positions >= the current run,
    # and the flip only writes the run just consumed, so runs are the ORIGINAL runs.
    pos = 0
    while pos < n:
        k = mask[pos]
        end = pos
        while end < n and mask[end] == k:
            end += 1
        if end - pos < 5:
            mask[pos:end] = 1 - k
        pos = end
    return mask


def power_of_signal(x):
    """M2/tools.py:213-214."""
    return np.sum(np.abs(x ** 2))


def add_signals(signal, noise, snr, norm=0.5):
    """M2/tools.py:217-276 with a single noise.  Returns mixed, clean, noise (scaled)."""
    sp = power_of_signal(signal)
    pn = sp / np.power(10, snr / 10)
    ret = np.copy(signal)
    if sp == 0:
        new_noise = noise
    else:
        ratio = np.sqrt(power_of_signal(noise)) / np.sqrt(pn)
        new_noise = noise if ratio == 0 else noise / ratio
    ret = ret + new_noise
    if norm:
        scale = np.max(np.abs(ret)) / norm
        if scale != 0:
            return ret / scale, signal / scale, new_noise / scale
    return ret, signal, new_noise


def nearest_index(in_size, out_size):
    """torch F.interpolate(mode='nearest') source index: min(floor(dst*scale), in-1)
    with scale = float32(in)/out (ATen UpSample.h nearest_idx).  Used at
    M1/networks.py:133 and M2/networks.py:199-203."""
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def si_sdr(est, ref):
    """Scale-invariant SDR in dB (new in the build)."""
    est = np.asarray(est, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    n = min(len(est), len(ref))
    est, ref = est[:n], ref[:n]
    alpha = np.dot(est, ref) / (np.dot(ref, ref) + 1e-30)
    t = alpha * ref
    return 10.0 * np.log10((np.dot(t, t) + 1e-30) / (np.dot(t - est, t - est) + 1e-30))
