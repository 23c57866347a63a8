"""Oracle (CPU restatement) vs committed golden vectors -- front-end and glue.
Goldens were produced by the imported reference (mask ops, bits->mask, add_signals,
nearest index) or by the oracle cross-checked with torch.stft/istft (STFT/ISTFT:
librosa 0.7.1 is absent -> 'parity unpinned', see oracle/__init__.py)."""
import numpy as np
import torch

from oracle import frontend as ofe
from util import rel_err


def test_stft_matches_golden_and_torch(golden):
    g = golden("frontend")
    for i in range(3):
        y = g[f"wave{i}"]
        S = ofe.fast_stft(y)
        assert S.shape == (256, 1 + len(y) // 158, 2)
        assert rel_err(S, g[f"stft{i}"]) < 1e-6
        St = torch.stft(torch.from_numpy(y), 510, 158, 400, window=torch.hann_window(400, periodic=True),
                        center=True, pad_mode="reflect", return_complex=True).numpy()
        assert np.max(np.abs(S[:, :, 0] + 1j * S[:, :, 1] - St)) < 5e-6


def test_istft_matches_golden_and_length(golden):
    g = golden("frontend")
    for i in range(3):
        yi = ofe.fast_istft(g[f"stft{i}"])
        assert len(yi) == 158 * (g[f"stft{i}"].shape[1] - 1)   # 27 966 for a 28 000-sample clip
        assert rel_err(yi, g[f"istft{i}"]) < 1e-6
        # round trip reproduces the waveform on the common length
        y = g[f"wave{i}"]
        assert np.max(np.abs(yi - y[:len(yi)])) < 1e-5
    assert rel_err(ofe.fast_istft(g["rand_spec"]), g["rand_spec_istft"]) < 1e-6


def test_mask_ops(golden):
    g = golden("maskops")
    assert rel_err(ofe.batch_fast_icRM_sigmoid(g["Y"], g["crm"]), g["rec"]) < 1e-5
    Y1 = g["Y"][0].transpose(1, 2, 0).astype(np.float64)
    c1 = g["crm"][0].transpose(1, 2, 0).astype(np.float64)
    assert np.allclose(ofe.fast_icRM_sigmoid(Y1, c1), g["rec1"], rtol=1e-12, atol=0)
    assert np.allclose(ofe.fast_cRM_sigmoid(g["S1"], Y1), g["tgt"], rtol=1e-12, atol=0)


def test_bits_to_mask_bit_exact(golden):
    g = golden("bitmask")
    for i in range(20):
        n = int(g[f"n{i}"])
        bits = "".join(str(int(b)) for b in g[f"bits{i}"])
        m = ofe.convert_bitstreammask_to_audiomask(np.zeros(n, np.float32), float(g[f"ratio{i}"]), bits)
        want = np.unpackbits(g[f"mask{i}"])[:n]
        assert np.array_equal(m.astype(np.uint8), want), i


def test_nearest_index(golden):
    g = golden("nearest")
    for k in g.files:
        a, b = (int(v) for v in k.split("_"))
        assert np.array_equal(ofe.nearest_index(a, b), g[k])


def test_add_signals(golden):
    g = golden("addsignals")
    for snr in (-10, 0, 7):
        m, c, n = ofe.add_signals(g["sig"], g["noi"], snr, 0.5)
        assert np.allclose(m, g[f"mixed_{snr}"], atol=1e-7)
        assert np.allclose(c, g[f"clean_{snr}"], atol=1e-7)
        assert np.allclose(n, g[f"noise_{snr}"], atol=1e-7)
        assert abs(np.max(np.abs(m)) - 0.5) < 1e-6


def test_si_sdr_properties():
    rng = np.random.default_rng(0)
    s = rng.standard_normal(4000)
    assert ofe.si_sdr(3.0 * s, s) > 100
    e = s + 0.1 * rng.standard_normal(4000)
    assert abs(ofe.si_sdr(e, s) - ofe.si_sdr(2 * e, s)) < 1e-9
    assert 18 < ofe.si_sdr(e, s) < 22


def test_stft_istft_against_scipy_signal():
    """Second, independent pin of the STFT / ISTFT restatement (librosa 0.7.1 itself is not installable here): scipy.signal
    implements the same transform with different conventions -- segments of nperseg = 400 zero-padded at the END to nfft =
    510 (librosa centres the window in n_fft: offset 55), 'even' boundary extension by nperseg/2 (librosa reflects by
    n_fft/2, the extra 55 samples per side meet a zero window), spectrum scaling 1/sum(w).  After undoing the scale and
    the 55-sample phase ramp both must agree on every frame; scipy's istft (NOLA normalisation by the window-sum-square,
    boundary trim) must invert to the same waveform and length."""
    import scipy.signal
    rng = np.random.default_rng(7)
    for n in (14000, 28000, 28123):
        x = (rng.standard_normal(n) * np.hanning(n) * 0.3).astype(np.float32)
        S = ofe.stft_complex(x.astype(np.float64))                              # (256, T)
        w = scipy.signal.get_window("hann", 400)                                # periodic (fftbins=True)
        f, t, Z = scipy.signal.stft(x.astype(np.float64), window=w, nperseg=400, noverlap=400 - 158, nfft=510,
                                    boundary="even", padded=False, return_onesided=True, scaling="spectrum")
        assert Z.shape == S.shape == (256, 1 + n // 158)
        ramp = np.exp(-2j * np.pi * np.arange(256) * 55 / 510.0)[:, None]       # window centred in n_fft
        Zl = Z * w.sum() * ramp
        assert np.max(np.abs(Zl - S)) < 2e-7 * max(1.0, np.max(np.abs(S)))       # the restatement returns complex64 like librosa
        # inverse: scipy on its own convention vs the restatement on librosa's
        y = ofe.fast_istft(ofe.real_imag_expand(S))
        _, ys = scipy.signal.istft(Z, window=w, nperseg=400, noverlap=400 - 158, nfft=510, input_onesided=True, boundary=True,
                                   scaling="spectrum")
        assert len(y) == 158 * (n // 158)
        m = min(len(y), len(ys))
        assert m >= len(y) - 400 and np.max(np.abs(ys[:m] - y[:m])) < 5e-6      # fast_istft returns float32
        assert np.max(np.abs(y - x[:len(y)])) < 5e-6                            # and both are the signal
