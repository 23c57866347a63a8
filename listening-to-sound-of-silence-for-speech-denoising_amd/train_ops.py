"""Training-mode forward (batch-statistics BatchNorm) and the hand-written backward pass of the
network pieces, driven from torch.autograd.Function wrappers in the module mirrors.

Everything numeric runs in the HIP kernels of libsos_hip (conv / dgrad on the MFMA conv kernel,
wgrad, BN forward/backward, LSTM BPTT, activation grads); this file only sequences launches and
owns the tape (saved activations)."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from . import engine as E

_consts = {}


def ones_zeros(n, device):
    key = (n, str(device))
    if key not in _consts:
        _consts[key] = (torch.ones(n, dtype=torch.float32, device=device), torch.zeros(n, dtype=torch.float32, device=device))
    return _consts[key]


def colsum(act, c_off, C):
    """Column sums (bias gradients) of an Act slice: f32 [C]."""
    dev = act.t.device
    v = E.view(act, c_off, C)
    nblk = L.lib().sos_bn_stats_blocks(v.npix)
    partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=dev)
    L.check(L.lib().sos_bn_stats(ctypes.byref(v), L.ptr(partial), L.stream_ptr()), "sos_bn_stats")
    out = torch.empty(C, dtype=torch.float32, device=dev)
    scratch = torch.empty((2, C), dtype=torch.float32, device=dev)
    # finalize with count = 1: "mean" is the plain sum
    L.check(L.lib().sos_bn_finalize(L.ptr(partial), nblk, C, 1, None, None, 1.0, 0.0, None, None, None, L.ptr(scratch[0]),
                                    L.ptr(scratch[1]), L.ptr(out), None, L.stream_ptr()), "sos_bn_finalize(colsum)")
    return out


def pack_grad(g, y, act, outer, inner, C, so, st, sc, dst, c_off=0):
    v = E.view(dst, c_off, C)
    L.check(L.lib().sos_pack_grad_f32(L.ptr(g), L.ptr(y), act, outer, inner, C, so, st, sc, ctypes.byref(v),
                                      L.stream_ptr()), "sos_pack_grad_f32")


def bn_bwd(dy, dy_off, raw, raw_off, C, saved, gamma, act, slope, dx, dx_off=0):
    """Backward of BatchNorm(train)+activation (or bias+activation when saved has no mean).
    Returns (dgamma, dbeta, dslope)."""
    dev = raw.t.device
    dyv, xv, dxv = E.view(dy, dy_off, C), E.view(raw, raw_off, C), E.view(dx, dx_off, C)
    nblk = L.lib().sos_bn_stats_blocks(xv.npix)
    partial = torch.empty((nblk, 3, C), dtype=torch.float32, device=dev)
    coef = torch.empty((3, C), dtype=torch.float32, device=dev)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    dslope = torch.empty(1, dtype=torch.float32, device=dev) if slope is not None else None
    L.check(L.lib().sos_bn_bwd(ctypes.byref(dyv), ctypes.byref(xv), L.ptr(saved["scale"]), L.ptr(saved["shift"]),
                               L.ptr(saved.get("mean")), L.ptr(saved.get("invstd")), L.ptr(gamma), act, L.ptr(slope),
                               L.ptr(partial), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dslope), ctypes.byref(dxv),
                               L.stream_ptr()), "sos_bn_bwd")
    return dgamma, dbeta, dslope


def act_bwd_from_y(dy, y, act, dz, C):
    a, b, c = E.view(dy, 0, C), E.view(y, 0, C), E.view(dz, 0, C)
    L.check(L.lib().sos_act_bwd_from_y(ctypes.byref(a), ctypes.byref(b), act, ctypes.byref(c), L.stream_ptr()),
            "sos_act_bwd_from_y")


def dgrad_weight(w, x3):
    """Conv2d weight (O,I,kh,kw) -> packed weight of the data-gradient conv: taps flipped, channel
    roles swapped, contraction over the (padded) O channels."""
    wd = w.detach().float().flip(2, 3).transpose(0, 1).contiguous()        # (I, O, kh, kw)
    return E.pack_weight(wd, E.pad_to(w.shape[0], 16), x3)


# ------------------------------------------------------------------ zero-padded Conv2d+BN+ReLU stack
def encoder_train_plan(enc, x3):
    plan = []
    for blk in enc:
        conv, bn = blk.block[0], blk.block[1]
        cin_store = E.pad_to(conv.in_channels, 16)
        plan.append(dict(w=E.pack_weight(conv.weight, cin_store, x3), wd=dgrad_weight(conv.weight, x3), conv=conv, bn=bn,
                         kh=conv.kernel_size[0], kw=conv.kernel_size[1], dil=tuple(conv.dilation),
                         pad=tuple(conv.padding), cout=conv.out_channels, cin=conv.in_channels, cin_store=cin_store))
    return plan


def encoder_forward_train(plan, a, feat, x3):
    """blocks 0..n-2: conv(raw) -> BN(batch stats) -> ReLU -> Act; last block writes the LSTM feature
    matrix (feat dict, see engine.bn_apply).  Returns the tape."""
    dev = a.t.device
    B, H, W = a.B, a.H, a.W
    tape = []
    cur = a
    for i, lp in enumerate(plan):
        cs = E.pad_to(lp["cout"], 16)
        one, zero = ones_zeros(lp["w"].shape[1], dev)
        raw = E.Act(B, H, W, cs, x3, dev)
        E.conv_to_act(cur, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw,
                      cout_store=cs, dil=lp["dil"], pad=lp["pad"], Ho=H, Wo=W)
        last = i == len(plan) - 1
        y = None if last else E.Act(B, H, W, cs, x3, dev, zero=cs > E.pad_to(lp["cout"], 8))
        saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_RELU, None, y, 0, feat if last else None)
        tape.append(dict(inp=cur, raw=raw, saved=saved))
        cur = y
    return tape


def encoder_backward(plan, tape, dy, grads, prefix, x3, need_input_grad=False):
    """dy: Act grad of the last block's NHWC output (already converted from the feature layout).
    Fills grads[name] for conv weights and BN affine params."""
    dev = dy.t.device
    for i in range(len(plan) - 1, -1, -1):
        lp, tp = plan[i], tape[i]
        raw = tp["raw"]
        # the BN kernels write whole 8-channel runs: zero the buffer when its padding goes beyond
        # that (garbage * zero weight would still be NaN for NaN garbage)
        d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8))
        dgamma, dbeta, _ = bn_bwd(dy, 0, raw, 0, lp["cout"], tp["saved"], lp["bn"].weight, L.ACT_RELU, None, d_raw)
        grads[f"{prefix}.{i}.block.1.weight"] = dgamma
        grads[f"{prefix}.{i}.block.1.bias"] = dbeta
        dw = torch.empty_like(lp["conv"].weight, dtype=torch.float32)
        E.wgrad(d_raw, 0, lp["cout"], tp["inp"], 0, lp["cin"], lp["kh"], lp["kw"], dw, dil=lp["dil"], pad=lp["pad"])
        grads[f"{prefix}.{i}.block.0.weight"] = dw
        if i == 0 and not need_input_grad:
            break
        inp = tp["inp"]
        d_in = E.Act(inp.B, inp.H, inp.W, inp.cs, x3, dev)
        one, zero = ones_zeros(lp["wd"].shape[1], dev)
        # "same" convs: pad' = dil*(k-1) - pad == pad
        E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], lp["kh"], lp["kw"], lp["cin"], one, zero, L.ACT_NONE, d_in,
                      cout_store=inp.cs, dil=lp["dil"],
                      pad=(lp["dil"][0] * (lp["kh"] - 1) - lp["pad"][0], lp["dil"][1] * (lp["kw"] - 1) - lp["pad"][1]),
                      Ho=inp.H, Wo=inp.W)
        dy = d_in
    return dy


def gather_ranges(gather_np, W):
    """For the nearest-resize gather i -> gather[i] (monotonic): [lo[w], hi[w]) = the i with gather[i] == w."""
    lo = np.searchsorted(gather_np, np.arange(W), side="left").astype(np.int32)
    hi = np.searchsorted(gather_np, np.arange(W), side="right").astype(np.int32)
    return lo, hi


def feat_grad_to_nhwc(dfeat, row, third, c_off, C, B, H, W, Wo, x3, lo=None, hi=None):
    """d(feature matrix) [B][Wo][row] -> Act [B,H,W,16] grad of the last encoder block's output."""
    dev = dfeat.device
    out = E.Act(B, H, W, E.pad_to(C, 16), x3, dev, zero=True)
    fv = L.View()
    fv.ptr, fv.npix, fv.row, fv.c_off, fv.C, fv.x3, fv.third = dfeat.data_ptr(), 1, row, c_off, C, (1 if x3 else 0), third
    ov = E.view(out, 0, C)
    L.check(L.lib().sos_feat_to_nhwc(ctypes.byref(fv), B, H, W, Wo, L.ptr(lo), L.ptr(hi), ctypes.byref(ov), L.stream_ptr()),
            "sos_feat_to_nhwc")
    return out


# ------------------------------------------------------------------------------------------ LSTM
def lstm_train_plan(lstm, cin_store, x3):
    H = lstm.hidden_size
    w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], dim=0).detach()          # (8H, I)
    bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse]).detach()
    w = E.pack_weight(w_ih[:, :, None, None], cin_store, x3)
    wd = E.pack_weight(w_ih.t().contiguous()[:, :, None, None], E.pad_to(8 * H, 16), x3)      # (I, 8H)
    whh_t = torch.stack([lstm.weight_hh_l0.detach().t(), lstm.weight_hh_l0_reverse.detach().t()]).float().contiguous()
    whh = torch.stack([lstm.weight_hh_l0.detach(), lstm.weight_hh_l0_reverse.detach()]).float().contiguous()
    dev = w.device
    return dict(w=w, wd=wd, whh_t=whh_t, whh=whh, H=H, I=lstm.input_size, cin_store=cin_store,
                scale=E.pad_vec(torch.ones(8 * H, device=dev), w.shape[1], 1.0), shift=E.pad_vec(bias, w.shape[1]))


def lstm_forward_train(lp, feat_dims, B, T, x3, dev):
    H = lp["H"]
    xproj = torch.empty((B, T, 8 * H), dtype=torch.float32, device=dev)
    E.conv(None, 0, lp["cin_store"], lp["w"], 1, 1, 8 * H, lp["scale"], lp["shift"], L.ACT_NONE, out=xproj,
           out_dtype=L.DT_F32, sb=T * 8 * H, sh=0, sw=8 * H, sc=1, Ho=1, Wo=T, in_dims=feat_dims)
    h = E.Act(B, 1, T, E.pad_to(2 * H, 16), x3, dev, zero=True)
    gates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dev)
    csave = torch.empty((B, T, 2, H), dtype=torch.float32, device=dev)
    E.lstm(xproj, lp["whh_t"], B, T, H, h, gates, csave)
    return h, dict(gates=gates, csave=csave, h=h, feat_dims=feat_dims)


def lstm_backward(lp, tape, dh, grads, prefix, B, T, x3, dev):
    """dh: Act grad of the LSTM output.  Returns d(feature matrix) [B][T][nseg*I] bf16."""
    H, I = lp["H"], lp["I"]
    dgates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dev)
    L.check(L.lib().sos_lstm_bidir_bwd(L.ptr(dh.t), dh.nseg * dh.cs, dh.dtype_code, dh.cs, L.ptr(tape["gates"]),
                                       L.ptr(tape["csave"]), L.ptr(lp["whh"]), B, T, H, L.ptr(dgates), L.stream_ptr()),
            "sos_lstm_bidir_bwd")
    dga = E.Act(B, 1, T, E.pad_to(8 * H, 16), x3, dev, zero=(8 * H) % 16 != 0)
    pack_grad(dgates, None, L.ACT_NONE, B, T, 8 * H, T * 8 * H, 8 * H, 1, dga)
    # biases: both b_ih and b_hh receive sum over (b,t) of the gate grads
    db = colsum(dga, 0, 8 * H)
    grads[f"{prefix}.bias_ih_l0"] = db[:4 * H]
    grads[f"{prefix}.bias_hh_l0"] = db[:4 * H]
    grads[f"{prefix}.bias_ih_l0_reverse"] = db[4 * H:]
    grads[f"{prefix}.bias_hh_l0_reverse"] = db[4 * H:]
    # W_ih: (8H x I) = dgates^T @ feat
    ft, fB, fH, fW, fcs, fnseg = tape["feat_dims"]
    feat_act = E.Act.__new__(E.Act)
    feat_act.t, feat_act.B, feat_act.H, feat_act.W, feat_act.cs, feat_act.x3, feat_act.nseg = ft, fB, 1, fW, fcs, x3, fnseg
    dwih = torch.empty((8 * H, I), dtype=torch.float32, device=dev)
    E.wgrad(dga, 0, 8 * H, feat_act, 0, I, 1, 1, dwih)
    grads[f"{prefix}.weight_ih_l0"] = dwih[:4 * H]
    grads[f"{prefix}.weight_ih_l0_reverse"] = dwih[4 * H:]
    # W_hh: dgate[t] (x) h[t-1] (forward) / h[t+1] (reverse): a 1-tap "conv" shifted by +-1 frame
    hact = tape["h"]
    # (all 2H columns are contracted and the direction's H columns sliced out: keeps the 16-byte
    # channel alignment of the loads for any H)
    for d, sfx, pad_l in ((0, "", 1), (1, "_reverse", -1)):
        dwhh = torch.empty((4 * H, 2 * H), dtype=torch.float32, device=dev)
        E.wgrad(dga, d * 4 * H, 4 * H, hact, 0, 2 * H, 1, 1, dwhh, pad=(0, pad_l))
        grads[f"{prefix}.weight_hh_l0{sfx}"] = dwhh[:, d * H:(d + 1) * H]
    # input gradient
    nseg = 3 if x3 else 1
    dfeat = torch.empty((B, T, nseg * I), dtype=torch.bfloat16, device=dev)
    one, zero = ones_zeros(lp["wd"].shape[1], dev)
    E.conv(dga, 0, dga.cs, lp["wd"], 1, 1, I, one, zero, L.ACT_NONE, out=dfeat,
           out_dtype=L.DT_BF16X3 if x3 else L.DT_BF16, sb=T * nseg * I, sh=0, sw=nseg * I, sc=1, third=I, Ho=1, Wo=T)
    return dfeat


# ---------------------------------------------------------------------------------------- Linear
def linear_train_plan(lin, cin_store, x3):
    w = E.pack_weight(lin.weight[:, :, None, None], cin_store, x3)
    wd = E.pack_weight(lin.weight.detach().t().contiguous()[:, :, None, None], E.pad_to(lin.out_features, 16), x3)
    return dict(w=w, wd=wd, lin=lin, cout=lin.out_features, cin=lin.in_features, cin_store=cin_store,
                scale=E.pad_vec(torch.ones(lin.out_features, device=w.device), w.shape[1], 1.0),
                shift=E.pad_vec(lin.bias, w.shape[1]))


def linear_backward(lp, a_in, dz, grads, prefix, x3, dev):
    """dz: Act grad of the pre-activation output (rows [B,1,T,*]).  Returns Act grad of the input."""
    dw = torch.empty((lp["cout"], lp["cin"]), dtype=torch.float32, device=dev)
    E.wgrad(dz, 0, lp["cout"], a_in, 0, lp["cin"], 1, 1, dw)
    grads[f"{prefix}.weight"] = dw
    grads[f"{prefix}.bias"] = colsum(dz, 0, lp["cout"])
    d_in = E.Act(a_in.B, 1, a_in.W, a_in.cs, x3, dev)
    one, zero = ones_zeros(lp["wd"].shape[1], dev)
    E.conv_to_act(dz, 0, dz.cs, lp["wd"], 1, 1, lp["cin"], one, zero, L.ACT_NONE, d_in, cout_store=a_in.cs, Ho=1,
                  Wo=a_in.W)
    return d_in
