"""The wave front door: what the reference gets from `librosa.load(path, sr=14000)`
(M1/dataset.py:226, M2/predict.py:288,297,303) and hands to `librosa.output.write_wav(path, y, sr)`
(M2/predict.py:515-528), with the same names and argument meaning:

  load(path, sr=22050, mono=True, offset=0.0, duration=None, dtype=np.float32, res_type='kaiser_best') -> (y, sr)
  resample(y, orig_sr, target_sr, res_type='kaiser_best', fix=True, scale=False)
  to_mono(y)
  write_wav(path, y, sr, norm=False)

The RIFF container is parsed here on the host (bytes -> header fields + one frombuffer view, no per-sample
Python); sample conversion, channel mix-down and the band-limited resampler are HIP kernels
(csrc/wave_io.hip).  `load_device` keeps the result in HBM for the pipeline.  No CPU fallback."""
import os
import struct

import numpy as np
import torch

from . import _lib as L

# resampy 0.2.2 `kaiser_best`: 64 zero crossings, 2**9 table points per crossing, Kaiser beta, roll-off
KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)

_FMT = {"s16": 0, "s32": 1, "f32": 2, "u8": 3}
_filters = {}


class WaveFormatError(ValueError):
    pass


def sinc_window(num_zeros, precision, beta, rolloff):
    """Half of a Kaiser-windowed sinc low-pass, `2**precision` points per zero crossing (resampy
    filters.sinc_window with window=kaiser(beta)).  Returns (half_window f64, num_table)."""
    num_table = 2 ** precision
    n = num_table * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_table


def _filter_on(device, ratio, res_type):
    if res_type != "kaiser_best":
        raise ValueError("res_type %r is not supported (the reference uses librosa's default 'kaiser_best')" % (res_type,))
    key = (str(device), float(ratio))
    if key not in _filters:
        if "half" not in _filters:
            _filters["half"] = sinc_window(**KAISER_BEST)
        half, num_table = _filters["half"]
        w = half * ratio if ratio < 1 else half
        _filters[key] = (torch.from_numpy(w.astype(np.float32)).to(device), num_table)
    return _filters[key]


def resample_device(x, orig_sr, target_sr, res_type="kaiser_best", fix=True):
    """x f32 1-D on the GPU -> f32 1-D, ceil(n * ratio) samples (floor when fix=False)."""
    L.require_cuda(x)
    if x.dim() != 1 or x.dtype != torch.float32:
        raise ValueError("resample_device expects a 1-D float32 tensor")
    if orig_sr == target_sr:
        return x
    ratio = float(target_sr) / orig_sr
    n_in = x.numel()
    n_res = int(n_in * ratio)
    if n_res < 1:
        raise ValueError("Input signal length=%d is too small to resample from %s->%s" % (n_in, orig_sr, target_sr))
    n_out = int(np.ceil(n_in * ratio)) if fix else n_res
    win, num_table = _filter_on(x.device, ratio, res_type)
    x = x.contiguous()
    out = torch.empty(n_out, dtype=torch.float32, device=x.device)
    L.check(L.lib().sos_resample_f32(L.ptr(x), n_in, ratio, L.ptr(win), win.numel(), num_table, L.ptr(out), n_out,
                                     L.stream_ptr()), "sos_resample_f32")
    return out


def pcm_to_mono_device(pcm, fmt):
    """pcm: (n_frames, channels) device tensor of int16 / int32 / float32 / uint8 -> mono f32 (n_frames,)."""
    L.require_cuda(pcm)
    pcm = pcm.contiguous()
    n, ch = pcm.shape
    out = torch.empty(n, dtype=torch.float32, device=pcm.device)
    L.check(L.lib().sos_pcm_to_mono_f32(L.ptr(pcm), _FMT[fmt], ch, n, L.ptr(out), L.stream_ptr()), "sos_pcm_to_mono_f32")
    return out


# ------------------------------------------------------------------------------------ RIFF/WAVE container
def read_wave(path):
    """Parse a RIFF/WAVE file -> (samples ndarray (n_frames, channels) in a kernel format, fmt, sample rate).
    PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE wrappers of those."""
    with open(path, "rb") as fp:
        buf = bytearray(os.fstat(fp.fileno()).st_size)       # writable backing store: the sample view goes to torch
        fp.readinto(buf)
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise WaveFormatError("%s: not a RIFF/WAVE file" % path)
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack_from("<I", buf, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            if size < 16:
                raise WaveFormatError("%s: short fmt chunk" % path)
            tag, ch, rate, _, align, bits = struct.unpack_from("<HHIIHH", buf, body)
            if tag == 0xFFFE and size >= 26:                 # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID's first word
                tag = struct.unpack_from("<H", buf, body + 24)[0]
            fmt = (tag, ch, rate, align, bits)
        elif cid == b"data":
            data = memoryview(buf)[body:min(body + size, len(buf))]
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise WaveFormatError("%s: missing fmt or data chunk" % path)
    tag, ch, rate, align, bits = fmt
    if ch < 1:
        raise WaveFormatError("%s: no channels" % path)
    width = bits // 8
    n = len(data) // (width * ch)
    data = data[:n * width * ch]
    if tag == 1 and bits == 16:
        arr, kind = np.frombuffer(data, dtype="<i2"), "s16"
    elif tag == 1 and bits == 8:
        arr, kind = np.frombuffer(data, dtype=np.uint8), "u8"
    elif tag == 1 and bits == 32:
        arr, kind = np.frombuffer(data, dtype="<i4"), "s32"
    elif tag == 1 and bits == 24:                           # widen to the top three bytes of an int32
        b3 = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3)
        arr = np.zeros((b3.shape[0], 4), dtype=np.uint8)
        arr[:, 1:] = b3
        arr, kind = arr.view("<i4").reshape(-1), "s32"
    elif tag == 3 and bits == 32:
        arr, kind = np.frombuffer(data, dtype="<f4"), "f32"
    elif tag == 3 and bits == 64:
        arr, kind = np.frombuffer(data, dtype="<f8").astype(np.float32), "f32"
    else:
        raise WaveFormatError("%s: unsupported WAVE format tag %d with %d bits" % (path, tag, bits))
    return arr.reshape(n, ch), kind, rate


def load_device(path, sr=22050, mono=True, offset=0.0, duration=None, res_type="kaiser_best", device="cuda"):
    """`librosa.load` with the result left in HBM: (y f32 GPU tensor, sr).  mono=False is not on the
    reference's path (every call site takes the default) and raises."""
    if not mono:
        raise NotImplementedError("load(mono=False): the reference only loads mono (M1/dataset.py:226)")
    if not torch.cuda.is_available():
        raise RuntimeError("sos_amd.audio_io needs an MI355X: there is no CPU fallback")
    arr, kind, sr_native = read_wave(path)
    if offset:
        arr = arr[int(offset * sr_native):]
    if duration is not None:
        arr = arr[:int(duration * sr_native)]
    if arr.shape[0] == 0:
        return torch.zeros(0, dtype=torch.float32, device=device), (sr_native if sr is None else sr)
    pcm = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    y = pcm_to_mono_device(pcm, kind)
    if sr is not None and sr != sr_native:
        y = resample_device(y, sr_native, sr, res_type)
    else:
        sr = sr_native
    return y, sr


def load(path, sr=22050, mono=True, offset=0.0, duration=None, dtype=np.float32, res_type="kaiser_best"):
    """Drop-in for `librosa.load` (librosa 0.7.1 core/audio.py) on WAVE files: (y ndarray, sr)."""
    y, sr = load_device(path, sr, mono, offset, duration, res_type)
    return np.ascontiguousarray(y.cpu().numpy(), dtype=dtype), sr


def to_mono(y):
    """librosa.to_mono: (channels, n) -> (n,) mean; 1-D passes through."""
    y = np.asarray(y)
    if y.ndim == 1:
        return y
    pcm = torch.from_numpy(np.ascontiguousarray(y.T, dtype=np.float32)).cuda()
    return pcm_to_mono_device(pcm, "f32").cpu().numpy().astype(y.dtype, copy=False)


def resample(y, orig_sr, target_sr, res_type="kaiser_best", fix=True, scale=False):
    """librosa.resample (1-D): numpy in -> numpy out."""
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError("resample expects a 1-D signal")
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    out = resample_device(torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda(), orig_sr, target_sr,
                          res_type, fix).cpu().numpy()
    if scale:
        out = out / np.sqrt(ratio)
    return np.ascontiguousarray(out, dtype=y.dtype if np.issubdtype(y.dtype, np.floating) else np.float32)


def wave_bytes(y, sr):
    """Bytes of the file `scipy.io.wavfile.write(path, sr, y)` produces (what librosa.output.write_wav calls):
    int16/int32/uint8 -> PCM; float32/float64 -> IEEE float with an 18-byte fmt chunk and a `fact` chunk."""
    y = np.asarray(y)
    if y.ndim == 1:
        ch = 1
    elif y.ndim == 2:
        ch = y.shape[1]
    else:
        raise ValueError("wave data must be (n,) or (n, channels)")
    kind = y.dtype.kind
    if not (kind in "iu" and y.dtype.itemsize in (1, 2, 4, 8) or kind == "f" and y.dtype.itemsize in (4, 8)):
        raise ValueError("Unsupported data type '%s'" % y.dtype)
    if kind == "u" and y.dtype.itemsize != 1 or kind == "i" and y.dtype.itemsize == 1:
        raise ValueError("Unsupported data type '%s'" % y.dtype)
    tag = 3 if kind == "f" else 1
    bits = y.dtype.itemsize * 8
    align = ch * (bits // 8)
    fmt = struct.pack("<HHIIHH", tag, ch, int(sr), int(sr) * align, align, bits)
    if tag != 1:
        fmt += b"\x00\x00"
    head = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if tag != 1:
        head += b"fact" + struct.pack("<II", 4, y.shape[0])
    data = np.ascontiguousarray(y).astype(y.dtype.newbyteorder("<"), copy=False).tobytes()
    if len(head) + len(data) + 12 > 0xFFFFFFFF:
        raise ValueError("Data exceeds wave file size limit")
    body = b"WAVE" + head + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def write_wav(path, y, sr, norm=False):
    """librosa.output.write_wav (0.7.1 output.py): y mono (n,) or stereo (2, n); float data is written as
    32/64-bit float WAVE unchanged unless norm=True (peak-normalised to 1)."""
    if torch.is_tensor(y):
        y = y.detach().cpu().numpy()
    y = np.asarray(y)
    if not np.issubdtype(y.dtype, np.floating):
        raise ValueError("Audio data must be floating-point")          # librosa.util.valid_audio
    if y.ndim not in (1, 2) or not np.isfinite(y).all():
        raise ValueError("Audio buffer is not finite everywhere or has a bad shape")
    wav = y
    if norm:
        peak = np.max(np.abs(y)) if y.size else 0.0
        wav = y / peak if peak > np.finfo(y.dtype).tiny else y
    if wav.ndim > 1 and wav.shape[0] == 2:
        wav = wav.T
    with open(path, "wb") as fp:
        fp.write(wave_bytes(wav, sr))
