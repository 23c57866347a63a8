"""TEST INFRASTRUCTURE: the f32 oracle with NOTHING but round trips through the 16-bit storage type inserted where the HIP
path stores a tensor -- a model of "what the format costs", used to turn flat 16-bit tolerances into computed ones
(`the kernels add nothing beyond the format' as an assertion: the HIP result must lie within a small
factor of this model's own deviation from the f32 reference).

Storage points of the product path: the module input (boundary pack), every packed weight, the raw conv
output of a training-mode block (BatchNorm statistics are taken from the rounded values), every block output, the LSTM
feature matrix, the stored hidden state h_t and W_hh of the recurrence, the FC operands; f32: accumulators, BatchNorm
arithmetic, gate arithmetic and the cell state, the network outputs (logits, n_pred, mask).  Gradients: every activation
gradient is stored in 16 bits (scaled by the pass's power-of-two loss scale in IEEE half), parameter gradients are f32.

`with storage_model(torch.float16): onet.detector_forward(...)` patches oracle.nets' building blocks for the duration of
the block (and restores them: other tests see the plain oracle).
"""
import contextlib

import torch
import torch.nn.functional as F

from oracle import nets as onet

_STATE = {"dt": torch.float16, "scale": 1.0}


def q(x):
    return x.to(_STATE["dt"]).float()


class _Q(torch.autograd.Function):
    """Value rounded to the storage type in forward; its gradient rounded (under the loss scale) in backward."""

    @staticmethod
    def forward(ctx, x):
        return q(x)

    @staticmethod
    def backward(ctx, g):
        s = _STATE["scale"]
        return (g * s).to(_STATE["dt"]).float() / s


def Q(x):
    return _Q.apply(x) if x.requires_grad else q(x)


def loss_scale_for(*grads):
    """engine.GradScale's choice: S = 2^floor(log2(256 / max|g|)) in IEEE half, 1 in bfloat16."""
    if _STATE["dt"] != torch.float16:
        return 1.0
    amax = max(float(g.abs().max()) for g in grads if g is not None)
    if not (amax > 0.0) or amax != amax or amax == float("inf"):
        return 1.0
    import math
    return 2.0 ** max(-40, min(40, math.floor(math.log2(256.0 / amax))))


def _wq(sd, key):
    return Q(sd[key]) if sd[key].requires_grad else q(sd[key])


def conv_block(x, sd, prefix, dilation, training, stats_out=None):
    x = Q(x)                              # block inputs are stored tensors (idempotent on already rounded values)
    w = _wq(sd, prefix + ".block.0.weight")
    kh, kw = w.shape[2], w.shape[3]
    pad = ((kh - 1) // 2 * dilation[0], (kw - 1) // 2 * dilation[1])
    y = F.conv2d(x, w, None, 1, pad, dilation)
    if training:
        y = Q(y)                          # the raw conv output is stored; statistics are taken from the stored values
    y = onet.batch_norm(y, sd, prefix + ".block.1", training, stats_out)
    return Q(torch.relu(y))


def down_block(x, sd, prefix, k, stride, dilation, training, stats_out=None, bn=True, act=True):
    pad = (k - 1) // 2 * dilation
    x = F.pad(Q(x), (pad, pad, pad, pad), mode="reflect")
    bias = None if bn else sd[prefix + ".block.1.bias"]
    y = F.conv2d(x, _wq(sd, prefix + ".block.1.weight"), bias, stride, 0, dilation)
    if bn:
        if training:
            y = Q(y)
        y = onet.batch_norm(y, sd, prefix + ".block.2", training, stats_out)
    if act:
        y = onet.prelu(y, sd[prefix + ".block.3.weight"])
    return Q(y) if bn else y              # (the last block of stage 1 writes f32)


def up_block(x, sd, prefix, training, stats_out=None):
    y = F.conv_transpose2d(Q(x), _wq(sd, prefix + ".block.0.weight"), None, 2, 1, 1)
    if training:
        y = Q(y)
    y = onet.batch_norm(y, sd, prefix + ".block.1", training, stats_out)
    return Q(onet.prelu(y, sd[prefix + ".block.2.weight"]))


def conv3d_block(x, sd, prefix, stride, training=False, stats_out=None):
    """Conv3dBlock of the audio-visual variant: stored input frames, stored packed weight, stored (raw and) block output."""
    w = _wq(sd, prefix + ".block.0.weight")
    pad = tuple((k - 1) // 2 for k in w.shape[2:])
    y = F.conv3d(Q(x), w, None, stride, pad)
    if training:
        y = Q(y)
    shp = y.shape
    y = onet.batch_norm(y.reshape(shp[0], shp[1], shp[2] * shp[3], shp[4]), sd, prefix + ".block.1", training, stats_out)
    return Q(torch.relu(y.reshape(shp)))


def lstm_bidir(x, sd, prefix):
    T, B, _ = x.shape
    outs = []
    xq = Q(x)
    for sfx in ("", "_reverse"):
        wih, whh = _wq(sd, f"{prefix}.weight_ih_l0{sfx}"), _wq(sd, f"{prefix}.weight_hh_l0{sfx}")
        bias = sd[f"{prefix}.bias_ih_l0{sfx}"] + sd[f"{prefix}.bias_hh_l0{sfx}"]
        H = whh.shape[1]
        xp = xq @ wih.t() + bias
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        hs = [None] * T
        order = range(T) if sfx == "" else range(T - 1, -1, -1)
        for t in order:
            g = xp[t] + h @ whh.t()
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = Q(torch.sigmoid(o) * torch.tanh(c))
            hs[t] = h
        outs.append(torch.stack(hs, 0))
    return torch.cat(outs, dim=2)


def linear(x, sd, prefix):
    # (hidden FC outputs are re-rounded by the next layer's Q(x); the heads' outputs stay f32)
    return Q(x) @ _wq(sd, prefix + ".weight").t() + sd[prefix + ".bias"]


_PATCH = dict(conv_block=conv_block, down_block=down_block, up_block=up_block, lstm_bidir=lstm_bidir, linear=linear,
              conv3d_block=conv3d_block)


@contextlib.contextmanager
def storage_model(dtype, scale=1.0):
    """Patch oracle.nets' building blocks with the round-trip versions above; `dtype` = torch.float16 / torch.bfloat16,
    `scale` = the backward pass's loss scale (loss_scale_for)."""
    saved = {k: getattr(onet, k) for k in _PATCH}
    prev = dict(_STATE)
    _STATE["dt"], _STATE["scale"] = dtype, scale
    try:
        for k, v in _PATCH.items():
            setattr(onet, k, v)
        yield
    finally:
        for k, v in saved.items():
            setattr(onet, k, v)
        _STATE.update(prev)


def storage_dtype(precision):
    return {"fp16": torch.float16, "bf16": torch.bfloat16}[precision]
