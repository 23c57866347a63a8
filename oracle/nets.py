"""oracle.nets -- torch-CPU float32 functional restatement of the reference networks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Everything is driven by a plain
dict of tensors that uses the reference's state_dict key names, so the same dict
can be loaded into the reference modules (tests/golden/make_goldens.py does that to
pin this file) and into the HIP-backed modules.

M1 = model_1_silent_interval_detection/audioonly_model/networks.py
M2 = model_2_audio_denoising/audio_denoising_model/networks.py
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# M1/networks.py:91-93 (detector audio branch), nf=48, outf=8
DET_KERNELS = [(1, 7), (7, 1)] + [(5, 5)] * 9
DET_DILATIONS = [(1, 1), (1, 1), (1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1), (1, 1), (2, 2), (4, 4)]
# M2/common.py:80-81
CTX_KERNELS = [(1, 7), (7, 1)] + [(5, 5)] * 12
CTX_DILATIONS = [(1, 1), (1, 1), (1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1),
                 (1, 1), (2, 2), (4, 4), (8, 8), (16, 16), (32, 32)]


# --------------------------------------------------------------------------- specs
def _bn_spec(prefix, c):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
            (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
            (prefix + ".num_batches_tracked", (), "nbt")]


def _lstm_spec(prefix, inp, hid):
    out = []
    for sfx in ("", "_reverse"):
        out += [(f"{prefix}.weight_ih_l0{sfx}", (4 * hid, inp), "lstm"),
                (f"{prefix}.weight_hh_l0{sfx}", (4 * hid, hid), "lstm"),
                (f"{prefix}.bias_ih_l0{sfx}", (4 * hid,), "lstm"),
                (f"{prefix}.bias_hh_l0{sfx}", (4 * hid,), "lstm")]
    return out


def _enc_spec(prefix, kernels, nf, outf):
    """Conv2dBlock stacks: M1/networks.py:120-128, M2/networks.py:72-80."""
    out = []
    cin = 2
    for i, k in enumerate(kernels):
        out.append((f"{prefix}.{i}.block.0.weight", (nf, cin, k[0], k[1]), "conv"))
        out += _bn_spec(f"{prefix}.{i}.block.1", nf)
        cin = nf
    i = len(kernels)
    out.append((f"{prefix}.{i}.block.0.weight", (outf, nf, 1, 1), "conv"))
    out += _bn_spec(f"{prefix}.{i}.block.1", outf)
    return out


def detector_spec(freq_bins=256, nf=48, kernels=None):
    kernels = DET_KERNELS if kernels is None else kernels
    spec = _enc_spec("encoder_audio", kernels, nf, 8)
    spec += _lstm_spec("lstm", 8 * freq_bins, 100)
    spec += [("fc1.0.weight", (100, 200), "lin"), ("fc1.0.bias", (100,), "bias"),
             ("fc1.2.weight", (1, 100), "lin"), ("fc1.2.bias", (1,), "bias")]
    return spec


def _down_spec(prefix, cin, cout, k, bn=True, act=True):
    out = [(prefix + ".block.1.weight", (cout, cin, k, k), "conv")]
    if not bn:
        out.append((prefix + ".block.1.bias", (cout,), "bias"))
    else:
        out += _bn_spec(prefix + ".block.2", cout)
    if act:
        out.append((prefix + ".block.3.weight", (1,), "prelu"))
    return out


def _up_spec(prefix, cin, cout, k):
    out = [(prefix + ".block.0.weight", (cin, cout, k, k), "convT")]
    out += _bn_spec(prefix + ".block.1", cout)
    out.append((prefix + ".block.2.weight", (1,), "prelu"))
    return out


def inpaint_spec(prefix="stage1", ch=(64, 128, 256)):
    """M2/networks.py:152-190 in registration order."""
    c1, c2, c3 = ch
    s = []
    s += _down_spec(f"{prefix}.down1.0", 2, c1, 5)
    s += _down_spec(f"{prefix}.down2.0", c1, c2, 5)
    s += _down_spec(f"{prefix}.down2.1", c2, c2, 5)
    s += _down_spec(f"{prefix}.down3.0", 2, c1, 5)
    s += _down_spec(f"{prefix}.down4.0", c1, c2, 5)
    s += _down_spec(f"{prefix}.down4.1", c2, c2, 5)
    s += _down_spec(f"{prefix}.mid.0", 2 * c2, c3, 3)
    for i in range(1, 8):
        s += _down_spec(f"{prefix}.mid.{i}", c3, c3, 3)
    s += _up_spec(f"{prefix}.mid.8", c3, c2, 3)
    s += _down_spec(f"{prefix}.up1.0", 2 * c2, c2, 3)
    s += _up_spec(f"{prefix}.up1.1", c2, c1, 3)
    s += _down_spec(f"{prefix}.up2.0", 2 * c1, c1, 3)
    s += _down_spec(f"{prefix}.up2.1", c1, 2, 3, bn=False, act=False)
    return s


def context_spec(prefix="stage2", freq_bins=256, nf=96, kernels=None, fc_hidden=600, lstm_hidden=200):
    kernels = CTX_KERNELS if kernels is None else kernels
    s = _enc_spec(f"{prefix}.encoder_x", kernels, nf, 8)
    s += _enc_spec(f"{prefix}.encoder_n", kernels, nf // 2, 4)
    s += _lstm_spec(f"{prefix}.lstm", 12 * freq_bins, lstm_hidden)
    s += [(f"{prefix}.fc.0.weight", (fc_hidden, 2 * lstm_hidden), "lin"), (f"{prefix}.fc.0.bias", (fc_hidden,), "bias"),
          (f"{prefix}.fc.2.weight", (fc_hidden, fc_hidden), "lin"), (f"{prefix}.fc.2.bias", (fc_hidden,), "bias"),
          (f"{prefix}.fc.4.weight", (2 * freq_bins, fc_hidden), "lin"), (f"{prefix}.fc.4.bias", (2 * freq_bins,), "bias")]
    return s


def joint_spec():
    return inpaint_spec("stage1") + context_spec("stage2")


# ------------------------------------------------------------- closed-form weights
# ---- audio-visual variant: the video branch the reference keeps as commented-out configuration
# (M1/networks.py:87-89 kernel sizes / strides, nf=128, outf=256) on its live classes Conv3dBlock (:54-77) and
# make_video_branch (:110-118); fusion per the commented lines of forward (:135-142).
VID_KERNELS = [(5, 7, 7), (5, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3), (1, 3, 3)]
VID_STRIDES = [(1, 2, 2), (1, 1, 1), (1, 2, 2), (1, 2, 2), (1, 2, 2), (1, 3, 3), (1, 3, 3)]


def _vid_spec(prefix, kernels, nf, outf):
    out = []
    cin = 3
    for i, k in enumerate(kernels):
        out.append((f"{prefix}.{i}.block.0.weight", (nf, cin, k[0], k[1], k[2]), "conv"))
        out += _bn_spec(f"{prefix}.{i}.block.1", nf)
        cin = nf
    i = len(kernels)
    out.append((f"{prefix}.{i}.block.0.weight", (outf, nf, 1, 1, 1), "conv"))
    out += _bn_spec(f"{prefix}.{i}.block.1", outf)
    return out


def audiovisual_spec(freq_bins=256, nf=48, nf_v=128, outf_v=256):
    """Registration order of the resurrected module: encoder_video is assigned after __init__ (so it follows fc1),
    the LSTM takes 8*freq_bins + outf_v features."""
    spec = _enc_spec("encoder_audio", DET_KERNELS, nf, 8)
    spec += _lstm_spec("lstm", 8 * freq_bins + outf_v, 100)
    spec += [("fc1.0.weight", (100, 200), "lin"), ("fc1.0.bias", (100,), "bias"),
             ("fc1.2.weight", (1, 100), "lin"), ("fc1.2.bias", (1,), "bias")]
    spec += _vid_spec("encoder_video", VID_KERNELS, nf_v, outf_v)
    return spec


def _hash_uniform(tensor_idx, n):
    """Deterministic, platform independent U(-1,1): splitmix64 finaliser on (idx, i)."""
    with np.errstate(over="ignore"):
        x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(tensor_idx + 1) * np.uint64(0xBF58476D1CE4E5B9))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return 2.0 * u - 1.0


def closed_form_state(spec, seed=0):
    """Fill every tensor of `spec` with a closed-form pattern keyed by its position."""
    sd = OrderedDict()
    for idx, (name, shape, kind) in enumerate(spec):
        n = int(np.prod(shape)) if len(shape) else 1
        u = _hash_uniform(idx + 1000 * seed, n)
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            v = u * np.sqrt(6.0 / fan_in)
        elif kind == "convT":
            # effective fan-in of the k3/s2 transposed conv is cin*k*k/4
            fan_in = shape[0] * shape[2] * shape[3] / 4.0
            v = u * np.sqrt(6.0 / fan_in)
        elif kind == "lin":
            v = u * np.sqrt(6.0 / shape[1])
        elif kind == "lstm":
            hid = shape[0] // 4
            v = u / np.sqrt(hid)
        elif kind == "bias":
            v = 0.1 * u
        elif kind == "bn_w":
            v = 1.0 + 0.1 * u
        elif kind == "bn_b":
            v = 0.1 * u
        elif kind == "bn_rm":
            v = 0.1 * u
        elif kind == "bn_rv":
            v = 1.0 + 0.2 * u
        elif kind == "prelu":
            v = 0.25 + 0.05 * u
        elif kind == "nbt":
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise KeyError(kind)
        sd[name] = torch.from_numpy(v.astype(np.float32).reshape(shape))
    return sd


# ----------------------------------------------------------------- building blocks
def batch_norm(x, sd, prefix, training, stats_out=None):
    """nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1).  Training: normalise with the
    biased batch variance, update running_var with the unbiased one."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() / x.shape[1]
            stats_out[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_mean"] + BN_MOMENTUM * mean.detach()
            stats_out[prefix + ".running_var"] = (1 - BN_MOMENTUM) * sd[prefix + ".running_var"] + BN_MOMENTUM * var.detach() * n / (n - 1)
            stats_out[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def conv_block(x, sd, prefix, dilation, training, stats_out=None):
    """Conv2dBlock/ConvBlock: M1/networks.py:28-51, M2/networks.py:28-51.
    zero pad (k-1)//2*dil, bias-free conv -> BN -> ReLU."""
    w = sd[prefix + ".block.0.weight"]
    kh, kw = w.shape[2], w.shape[3]
    pad = ((kh - 1) // 2 * dilation[0], (kw - 1) // 2 * dilation[1])
    y = F.conv2d(x, w, None, 1, pad, dilation)
    y = batch_norm(y, sd, prefix + ".block.1", training, stats_out)
    return torch.relu(y)


def encoder(x, sd, prefix, dilations, training, stats_out=None):
    for i, d in enumerate(dilations):
        x = conv_block(x, sd, f"{prefix}.{i}", d, training, stats_out)
    return conv_block(x, sd, f"{prefix}.{len(dilations)}", (1, 1), training, stats_out)


def nearest_resize_last(x, size):
    """F.interpolate(x, size) on the last dim (nearest), M1/networks.py:133."""
    from .frontend import nearest_index
    idx = torch.from_numpy(nearest_index(x.shape[-1], size))
    return x.index_select(-1, idx)


def nearest_resize_2d(x, size):
    """F.interpolate(x, (H,W)) nearest, M2/networks.py:199-203."""
    from .frontend import nearest_index
    ih = torch.from_numpy(nearest_index(x.shape[-2], size[0]))
    iw = torch.from_numpy(nearest_index(x.shape[-1], size[1]))
    return x.index_select(-2, ih).index_select(-1, iw)


def lstm_bidir(x, sd, prefix):
    """nn.LSTM(num_layers=1, bidirectional=True), seq-first x (T,B,I) -> (T,B,2H).
    Gate order i,f,g,o (cuDNN/torch)."""
    T, B, _ = x.shape
    outs = []
    for sfx in ("", "_reverse"):
        wih, whh = sd[f"{prefix}.weight_ih_l0{sfx}"], sd[f"{prefix}.weight_hh_l0{sfx}"]
        bias = sd[f"{prefix}.bias_ih_l0{sfx}"] + sd[f"{prefix}.bias_hh_l0{sfx}"]
        H = whh.shape[1]
        xp = x @ wih.t() + bias                       # (T,B,4H)
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        hs = [None] * T
        order = range(T) if sfx == "" else range(T - 1, -1, -1)
        for t in order:
            g = xp[t] + h @ whh.t()
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs, 0))
    return torch.cat(outs, dim=2)


def linear(x, sd, prefix):
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


# ------------------------------------------------------------------------ detector
def detector_forward(sd, s, v_num_frames=60, training=False, stats_out=None, dilations=None):
    """AudioVisualNet.forward, M1/networks.py:130-155.  s (B,2,F,T) -> logits (B,n)."""
    dilations = DET_DILATIONS if dilations is None else dilations
    f = encoder(s, sd, "encoder_audio", dilations, training, stats_out)
    f = f.reshape(f.shape[0], -1, f.shape[3])          # (B, 8*F, T)  index c*F+f
    f = nearest_resize_last(f, v_num_frames)
    m = f.permute(2, 0, 1)
    m = lstm_bidir(m, sd, "lstm").permute(1, 0, 2)     # (B,n,200)
    m = torch.relu(linear(m, sd, "fc1.0"))
    m = linear(m, sd, "fc1.2")
    return m.squeeze(2)


def conv3d_block(x, sd, prefix, stride, training=False, stats_out=None):
    """Conv3dBlock (M1/networks.py:54-77): Conv3d(pad (k-1)//2 per axis, no bias) -> BatchNorm3d -> ReLU.
    BatchNorm3d over (B, T, H, W) is BatchNorm2d of the tensor viewed as (B, C, T*H, W)."""
    w = sd[prefix + ".block.0.weight"]
    pad = tuple((k - 1) // 2 for k in w.shape[2:])
    y = F.conv3d(x, w, None, stride, pad)
    shp = y.shape
    y = batch_norm(y.reshape(shp[0], shp[1], shp[2] * shp[3], shp[4]), sd, prefix + ".block.1", training, stats_out)
    return torch.relu(y.reshape(shp))


def video_forward(sd, v, prefix="encoder_video", strides=None, taps=None, training=False, stats_out=None):
    """make_video_branch (M1/networks.py:110-118): v (B,3,T,H,W) -> (B,256,T,h,w).  `taps` collects block outputs."""
    strides = VID_STRIDES if strides is None else strides
    for i, st in enumerate(strides):
        v = conv3d_block(v, sd, f"{prefix}.{i}", st, training, stats_out)
        if taps is not None:
            taps.append(v)
    v = conv3d_block(v, sd, f"{prefix}.{len(strides)}", (1, 1, 1), training, stats_out)
    if taps is not None:
        taps.append(v)
    return v


def audiovisual_forward(sd, s, v, training=False, stats_out=None):
    """AudioVisualNet.forward with the commented fusion lines live (M1/networks.py:130-155): spatial mean of the
    video features, audio features resized to the video frame count, channel concat, BiLSTM, FC head."""
    f_s = encoder(s, sd, "encoder_audio", DET_DILATIONS, training, stats_out)
    f_s = f_s.reshape(f_s.shape[0], -1, f_s.shape[3])
    f_v = video_forward(sd, v, training=training, stats_out=stats_out).mean(dim=(-2, -1))          # (B, 256, T2)
    f_s = nearest_resize_last(f_s, f_v.shape[2])
    m = torch.cat([f_s, f_v], dim=1).permute(2, 0, 1)
    m = lstm_bidir(m, sd, "lstm").permute(1, 0, 2)
    m = torch.relu(linear(m, sd, "fc1.0"))
    return linear(m, sd, "fc1.2").squeeze(2)


# ------------------------------------------------------------------------ denoiser
def prelu(x, a):
    return torch.where(x >= 0, x, a * x)


def down_block(x, sd, prefix, k, stride, dilation, training, stats_out=None, bn=True, act=True):
    """DownConvBlock, M2/networks.py:97-117: ReflectionPad2d -> valid conv -> BN -> PReLU."""
    pad = (k - 1) // 2 * dilation
    x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    bias = None if bn else sd[prefix + ".block.1.bias"]
    y = F.conv2d(x, sd[prefix + ".block.1.weight"], bias, stride, 0, dilation)
    if bn:
        y = batch_norm(y, sd, prefix + ".block.2", training, stats_out)
    if act:
        y = prelu(y, sd[prefix + ".block.3.weight"])
    return y


def up_block(x, sd, prefix, training, stats_out=None):
    """UpConvBlock, M2/networks.py:120-149: ConvTranspose2d(k3,s2,p1,output_padding=1)
    (the positional `dilation` lands in output_padding, :130) -> BN -> PReLU."""
    y = F.conv_transpose2d(x, sd[prefix + ".block.0.weight"], None, 2, 1, 1)
    y = batch_norm(y, sd, prefix + ".block.1", training, stats_out)
    return prelu(y, sd[prefix + ".block.2.weight"])


def inpaint_forward(sd, x, y, training=False, stats_out=None, prefix="stage1"):
    """InpaintNet.forward(x, y), M2/networks.py:192-205.  x = noise-interval STFT,
    y = mixed STFT (JointModel calls stage1(n, x), :215)."""
    p = prefix
    d1 = down_block(x, sd, f"{p}.down1.0", 5, 1, 1, training, stats_out)
    d2 = down_block(d1, sd, f"{p}.down2.0", 5, 2, 1, training, stats_out)
    d2 = down_block(d2, sd, f"{p}.down2.1", 5, 1, 1, training, stats_out)
    d3 = down_block(y, sd, f"{p}.down3.0", 5, 1, 1, training, stats_out)
    d4 = down_block(d3, sd, f"{p}.down4.0", 5, 2, 1, training, stats_out)
    d4 = down_block(d4, sd, f"{p}.down4.1", 5, 1, 1, training, stats_out)
    o = torch.cat([d2, d4], dim=1)
    o = down_block(o, sd, f"{p}.mid.0", 3, 2, 1, training, stats_out)
    for i, dil in zip(range(1, 8), (1, 2, 4, 8, 16, 1, 1)):
        o = down_block(o, sd, f"{p}.mid.{i}", 3, 1, dil, training, stats_out)
    o = up_block(o, sd, f"{p}.mid.8", training, stats_out)
    if o.shape != d4.shape:
        o = nearest_resize_2d(o, d4.shape[-2:])
    o = down_block(torch.cat([o, d4], dim=1), sd, f"{p}.up1.0", 3, 1, 1, training, stats_out)
    o = up_block(o, sd, f"{p}.up1.1", training, stats_out)
    if o.shape != d3.shape:
        o = nearest_resize_2d(o, d3.shape[-2:])
    o = down_block(torch.cat([o, d3], dim=1), sd, f"{p}.up2.0", 3, 1, 1, training, stats_out)
    return down_block(o, sd, f"{p}.up2.1", 3, 1, 1, training, stats_out, bn=False, act=False)


def context_forward(sd, x, n, training=False, stats_out=None, prefix="stage2", dilations=None):
    """ContextAggNet.forward, M2/networks.py:82-94 -> sigmoid mask (B,2,F,T)."""
    dilations = CTX_DILATIONS if dilations is None else dilations
    fx = encoder(x, sd, f"{prefix}.encoder_x", dilations, training, stats_out)
    fx = fx.reshape(fx.shape[0], -1, fx.shape[3]).permute(2, 0, 1)
    fn = encoder(n, sd, f"{prefix}.encoder_n", dilations, training, stats_out)
    fn = fn.reshape(fn.shape[0], -1, fn.shape[3]).permute(2, 0, 1)
    h = lstm_bidir(torch.cat([fx, fn], dim=2), sd, f"{prefix}.lstm").permute(1, 0, 2)
    h = torch.relu(linear(h, sd, f"{prefix}.fc.0"))
    h = torch.relu(linear(h, sd, f"{prefix}.fc.2"))
    h = torch.sigmoid(linear(h, sd, f"{prefix}.fc.4"))      # (B,T,2F)
    return h.permute(0, 2, 1).reshape(h.shape[0], 2, -1, h.shape[1])


def joint_forward(sd, x, n, training=False, stats_out=None):
    """JointModel.forward(x=mixed, n=noise), M2/networks.py:214-217."""
    n_pred = inpaint_forward(sd, n, x, training, stats_out)
    out = context_forward(sd, x, n_pred, training, stats_out)
    return n_pred, out


# -------------------------------------------------------------------------- losses
def mask_apply(Y, crm, a=0.1, b=0.0):
    """batch_fast_icRM_sigmoid in torch, M1/transform.py:156-169."""
    M = 1.0 / a * (torch.log(crm / (1 - crm + 1e-8) + 1e-10) + b)
    r = M[:, 0] * Y[:, 0] - M[:, 1] * Y[:, 1]
    i = M[:, 0] * Y[:, 1] + M[:, 1] * Y[:, 0]
    return torch.stack([r, i], dim=1)


def denoiser_losses(sd, batch, training=True, stats_out=None):
    """MyAgent.forward, M2/agent.py:176-190: MSE(n_pred, full_noise) + MSE(rec, clean)."""
    n_pred, out = joint_forward(sd, batch["mixed"], batch["noise"], training, stats_out)
    rec = mask_apply(batch["mixed"], out)
    l1 = torch.mean((n_pred - batch["full_noise"]) ** 2)
    l2 = torch.mean((rec - batch["clean"]) ** 2)
    return (n_pred, out), {"stage1": l1, "stage2": l2}


def detector_loss(sd, batch, training=True, stats_out=None):
    """MyAgent.forward, M1/agent.py:189-206: BCEWithLogits(mean)."""
    logits = detector_forward(sd, batch["audio"], batch["label"].shape[1], training, stats_out)
    y = batch["label"]
    loss = torch.mean(torch.clamp(logits, min=0) - logits * y + torch.log1p(torch.exp(-logits.abs())))
    return logits, {"bce": loss}
