"""Wave front door on the GPU (csrc/wave_io.hip through the C ABI) against the oracle restatement of
librosa.load's arithmetic (oracle/wave_io.py; resampy/librosa absent: parity unpinned).

Tolerances: sample conversion + mix-down of 1/2-channel integer PCM is bit-exact; the resampler accumulates
~400 f32 products per sample with f32 table weights against the oracle's f64: <= 1e-5 of the signal peak."""
import numpy as np
import pytest
import scipy.io.wavfile
import torch

from oracle import wave_io as owio
from test_oracle_wave_io import snr_db, tones

pytestmark = pytest.mark.gpu
RESAMPLE_TOL = 1e-5


def peak_err(got, ref):
    return float(np.max(np.abs(np.asarray(got, np.float64) - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize("sr_in,sr_out,n", [(44100, 14000, 50001), (14000, 44100, 9000), (16000, 14000, 33333),
                                            (44100, 14000, 700), (48000, 14000, 20000), (8000, 14000, 12345)])
def test_resample_matches_oracle(sr_in, sr_out, n):
    from sos_amd import audio_io
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    ref = owio.resample(x, sr_in, sr_out)
    got = audio_io.resample_device(torch.from_numpy(x).cuda(), sr_in, sr_out).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    assert peak_err(got, ref) < RESAMPLE_TOL
    got2 = audio_io.resample(x, sr_in, sr_out, fix=False)
    assert len(got2) == int(n * sr_out / sr_in) and np.array_equal(got2, got[:len(got2)])


def test_resample_shortest_inputs_and_errors():
    from sos_amd import audio_io
    x = np.array([1.0, -2.0, 0.5, 0.25], dtype=np.float32)
    ref = owio.resample(x, 44100, 14000)
    got = audio_io.resample(x, 44100, 14000)
    assert got.shape == ref.shape == (2,) and peak_err(got, ref) < RESAMPLE_TOL
    with pytest.raises(ValueError):
        audio_io.resample(x[:3], 44100, 14000)                 # int(3 * ratio) == 0 output samples (resampy raises)
    with pytest.raises(ValueError):
        audio_io.resample(x, 44100, 14000, res_type="kaiser_fast")
    with pytest.raises(RuntimeError):
        audio_io.resample_device(torch.from_numpy(x), 44100, 14000)   # host tensor: no CPU fallback
    assert audio_io.resample(x, 14000, 14000) is x


def test_resample_is_linear_and_reproduces_tones_at_file_scale():
    """Size-independent properties on a 5-minute 44.1 kHz file (13.2 M samples)."""
    from sos_amd import audio_io
    n = 44100 * 300
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(n, device="cuda", generator=g)
    b = torch.randn(n, device="cuda", generator=g)
    ra, rb = audio_io.resample_device(a, 44100, 14000), audio_io.resample_device(b, 44100, 14000)
    rc = audio_io.resample_device(0.75 * a - 1.5 * b, 44100, 14000)
    assert ra.numel() == 14000 * 300
    lin = 0.75 * ra - 1.5 * rb
    assert float((rc - lin).abs().max() / lin.abs().max()) < 2e-5
    x = tones(44100 * 20, 44100).astype(np.float32)
    y = audio_io.resample_device(torch.from_numpy(x).cuda(), 44100, 14000).cpu().numpy().astype(np.float64)
    ref = tones(len(y), 14000)
    r, yy = ref[400:-400], y[400:-400]
    gain = np.dot(r, yy) / np.dot(r, r)
    assert 1.001 < gain < 1.004 and snr_db(gain * r, yy) > 70.0    # same bounds as the oracle's own test


@pytest.mark.parametrize("dtype,ch,exact", [(np.int16, 1, True), (np.int16, 2, True), (np.int32, 2, True), (np.uint8, 2, True),
                                            (np.float32, 2, True), (np.int16, 3, False), (np.float32, 6, False)])
def test_pcm_to_mono(dtype, ch, exact):
    from sos_amd import audio_io
    rng = np.random.default_rng(ch)
    n = 100003
    if np.issubdtype(dtype, np.floating):
        pcm = rng.uniform(-1, 1, (n, ch)).astype(dtype)
    else:
        info = np.iinfo(dtype)
        pcm = rng.integers(info.min, info.max, (n, ch), endpoint=True).astype(dtype)
    kind = {np.int16: "s16", np.int32: "s32", np.uint8: "u8", np.float32: "f32"}[dtype]
    got = audio_io.pcm_to_mono_device(torch.from_numpy(pcm).cuda(), kind).cpu().numpy()
    ref = owio.to_mono(owio.pcm_to_float(pcm))
    if exact:
        assert np.array_equal(got, ref)
    else:                                                     # x * (1/ch) vs x / ch and the summation order: 2 ulp
        assert np.max(np.abs(got - ref)) <= 2.5e-7


def test_load_matches_librosa_semantics(tmp_path):
    """A 44.1 kHz stereo 16-bit file like the reference's data/sounds_of_silence_audioonly/*.wav."""
    from sos_amd import audio_io
    sr0, n = 44100, 44100 * 3 + 17
    t = np.arange(n) / sr0
    rng = np.random.default_rng(11)
    left = 0.4 * np.sin(2 * np.pi * 330 * t) + 0.05 * rng.standard_normal(n)
    right = 0.3 * np.sin(2 * np.pi * 1200 * t + 0.5) + 0.05 * rng.standard_normal(n)
    pcm = np.clip(np.stack([left, right], axis=1) * 32768, -32768, 32767).astype(np.int16)
    p = tmp_path / "clip.wav"
    scipy.io.wavfile.write(p, sr0, pcm)
    y, sr = audio_io.load(str(p), sr=14000)
    ref = owio.load_from_pcm(pcm, sr0, 14000)
    assert sr == 14000 and y.dtype == np.float32 and y.shape == ref.shape == (int(np.ceil(n * 14000 / sr0)),)
    assert peak_err(y, ref.astype(np.float64)) < RESAMPLE_TOL
    y0, sr_native = audio_io.load(str(p), sr=None)            # native rate: conversion + mix-down only, bit-exact
    assert sr_native == sr0 and np.array_equal(y0, owio.load_from_pcm(pcm, sr0, None))
    y1, _ = audio_io.load(str(p), sr=14000, offset=0.5, duration=1.0)
    ref1 = owio.load_from_pcm(pcm[int(0.5 * sr0):int(0.5 * sr0) + int(1.0 * sr0)], sr0, 14000)
    assert y1.shape == ref1.shape == (14000,) and peak_err(y1, ref1.astype(np.float64)) < RESAMPLE_TOL
    yd, _ = audio_io.load_device(str(p), sr=14000)
    assert yd.is_cuda and np.array_equal(yd.cpu().numpy(), y)
    with pytest.raises(NotImplementedError):
        audio_io.load(str(p), sr=14000, mono=False)


def test_write_then_load_round_trip(tmp_path):
    from sos_amd import audio_io
    y = torch.randn(28000, device="cuda") * 0.1
    p = tmp_path / "denoised_output.wav"
    audio_io.write_wav(str(p), y, 14000)                      # M2/predict.py:521-522 hands a float32 signal
    back, sr = audio_io.load(str(p), sr=14000)
    assert sr == 14000 and np.array_equal(back, y.cpu().numpy())
