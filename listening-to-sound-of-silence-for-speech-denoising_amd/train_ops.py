"""Training-mode forward (batch-statistics BatchNorm) and the hand-written backward pass of the
network pieces, driven from torch.autograd.Function wrappers in the module mirrors.

Everything numeric runs in the HIP kernels of libsos_hip (conv / dgrad on the MFMA conv kernel,
wgrad, BN forward/backward, LSTM BPTT, activation grads); this file only sequences launches and
owns the tape (saved activations)."""
import ctypes
import os

import numpy as np
import torch

from . import _lib as L
from . import engine as E




# SOS_FUSED_STATS=0: separate sos_bn_stats pass instead of the statistics fused into the conv epilogue (A/B timing)
FUSED_STATS = __import__("os").environ.get("SOS_FUSED_STATS", "1") != "0"
# SOS_FUSED_STATS_UNET=1: the U-Net's DownConvBlocks take their statistics from the conv epilogue too (14 statistics passes per step
# less).  Measured +0.1 % on the step (531.9 -> 532.6, inside the noise): opt-in only
FUSED_STATS_UNET = __import__("os").environ.get("SOS_FUSED_STATS_UNET", "0") == "1"


def ones_zeros(n, device):
    """(scale, shift) of a conv with no affine epilogue: NULL for both (sos_conv_desc: y = act(acc))."""
    return None, None


def colsum(act, c_off, C):
    """Column sums (bias gradients) of an Act slice: f32 [C]."""
    dev = act.t.device
    v = E.view(act, c_off, C)
    nblk = L.lib().sos_bn_stats_blocks(v.npix)
    partial = torch.empty((2, C, nblk), dtype=torch.float32, device=dev)
    L.check(L.lib().sos_bn_stats(ctypes.byref(v), L.ptr(partial), L.stream_ptr()), "sos_bn_stats")
    out = torch.empty(C, dtype=torch.float32, device=dev)
    scratch = torch.empty((2, C), dtype=torch.float32, device=dev)
    # finalize with count = 1: "mean" is the plain sum
    L.check(L.lib().sos_bn_finalize(L.ptr(partial), nblk, C, 1, None, None, 1.0, 0.0, None, None, None, L.ptr(scratch[0]),
                                    L.ptr(scratch[1]), L.ptr(out), None, L.stream_ptr()), "sos_bn_finalize(colsum)")
    return E.cur_gs().unscale(out)          # parameter gradients leave in the caller's units (fp16 loss scale)


def pack_grad(g, y, act, outer, inner, C, so, st, sc, dst, c_off=0, scaled=False):
    """f32 gradient -> 16-bit rows.  scaled=False: `g` is in the caller's units and the pass's loss scale is multiplied
    in here; scaled=True: `g` was derived from already scaled gradients (the LSTM's dgates)."""
    v = E.view(dst, c_off, C)
    mul = None if scaled else E.cur_gs().mul
    L.check(L.lib().sos_pack_grad_f32(L.ptr(g), L.ptr(y), act, outer, inner, C, so, st, sc, ctypes.byref(v),
                                      L.ptr(mul), L.stream_ptr()), "sos_pack_grad_f32")


def bn_bwd(dy, dy_off, raw, raw_off, C, saved, gamma, act, slope, dx, dx_off=0):
    """Backward of BatchNorm(train)+activation (or bias+activation when saved has no mean).
    Returns (dgamma, dbeta, dslope)."""
    dev = raw.t.device
    dyv, xv, dxv = E.view(dy, dy_off, C), E.view(raw, raw_off, C), E.view(dx, dx_off, C)
    nblk = L.lib().sos_bn_stats_blocks(xv.npix)
    partial = torch.empty((3, C, nblk), dtype=torch.float32, device=dev)
    coef = torch.empty((4, C), dtype=torch.float32, device=dev)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    dslope = torch.empty(1, dtype=torch.float32, device=dev) if slope is not None else None
    L.check(L.lib().sos_bn_bwd(ctypes.byref(dyv), ctypes.byref(xv), L.ptr(saved["scale"]), L.ptr(saved["shift"]),
                               L.ptr(saved.get("mean")), L.ptr(saved.get("invstd")), L.ptr(gamma), act, L.ptr(slope),
                               L.ptr(partial), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dslope), ctypes.byref(dxv),
                               L.ptr(E.cur_gs().inv), L.stream_ptr()), "sos_bn_bwd")
    return dgamma, dbeta, dslope


def act_bwd_from_y(dy, y, act, dz, C):
    a, b, c = E.view(dy, 0, C), E.view(y, 0, C), E.view(dz, 0, C)
    L.check(L.lib().sos_act_bwd_from_y(ctypes.byref(a), ctypes.byref(b), act, ctypes.byref(c), L.stream_ptr()),
            "sos_act_bwd_from_y")


def dgrad_weight(w, x3):
    """Conv2d weight (O,I,kh,kw) -> packed weight of the data-gradient conv: taps flipped, channel
    roles swapped, contraction over the (padded) O channels."""
    # (I, O, kh, kw); a callable: not evaluated when the packed tensor is refreshed by the gather kernel (engine.PackRecorder)
    return E.pack_weight(lambda: w.detach().float().flip(2, 3).transpose(0, 1).contiguous(), E.pad_to(w.shape[0], 16), x3)


# ------------------------------------------------------------------ zero-padded Conv2d+BN+ReLU stack
WGRAD_SIDE = os.environ.get("SOS_WGRAD_SIDE", "0") == "1"      # A/B switch of the side stream below (measured: 479-483 vs 483-488 utt/s without: off)
_SIDE = {}


def _side_stream(dev, cur):
    key = (dev.index, cur.cuda_stream)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=dev)
    return s


def encoder_train_plan(enc, x3):
    plan = []
    for i, blk in enumerate(enc):
        conv, bn = blk.block[0], blk.block[1]
        wf = E.wfold_spec(conv, conv.padding[1], L.PAD_ZERO) if i == 0 else None
        if wf is not None:
            # first block: horizontal taps on the channel axis (engine.wfold_spec) -- forward conv and weight gradient see a
            # kh x 1 layer over kw * I channels; the data gradient (encoder_n only: the stage-1 prediction) keeps the layer's
            # own geometry (`dg`), it contracts the 48 output channels
            cin_store = E.pad_to(wf["kw"] * wf["I"], 16)
            plan.append(dict(w=E.pack_weight(lambda conv=conv, wf=wf: wf["fold"](conv.weight.detach().float()), cin_store, x3),
                             wd=dgrad_weight(conv.weight, x3), conv=conv, bn=bn, kh=conv.kernel_size[0], kw=1,
                             dil=(conv.dilation[0], 1), pad=(conv.padding[0], 0), cout=conv.out_channels,
                             cin=wf["kw"] * wf["I"], cin_store=cin_store, wtaps=wf["wtaps"], unfold=wf["unfold"],
                             dg=dict(kh=conv.kernel_size[0], kw=conv.kernel_size[1], dil=tuple(conv.dilation),
                                     pad=tuple(conv.padding), cin=conv.in_channels)))
            continue
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
        # the conv's epilogue also produces the BatchNorm partial sums of its (bf16) output tile
        st = E.conv_to_act(cur, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw,
                           cout_store=cs, dil=lp["dil"], pad=lp["pad"], Ho=H, Wo=W, stats_c=lp["cout"] if FUSED_STATS else 0)
        last = i == len(plan) - 1
        y = None if last else E.Act(B, H, W, cs, x3, dev, zero=cs > E.pad_to(lp["cout"], 8))
        saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_RELU, None, y, 0, feat if last else None, stats=st)
        tape.append(dict(inp=cur, raw=raw, saved=saved))
        cur = y
    return tape


def encoder_backward(plan, tape, dy, grads, prefix, x3, need_input_grad=False):
    """dy: Act grad of the last block's NHWC output (already converted from the feature layout).
    Fills grads[name] for conv weights and BN affine params."""
    dev = dy.t.device
    cur = torch.cuda.current_stream(dev)
    # side stream for the weight gradients: single-process training only (with a gradient sink the buckets' copies and
    # collectives are ordered on the main stream), never inside a stream capture
    side = _side_stream(dev, cur) if (WGRAD_SIDE and type(grads) is dict and not torch.cuda.is_current_stream_capturing()) else None
    gs = E.cur_gs()
    try:
        return _encoder_backward(plan, tape, dy, grads, prefix, x3, need_input_grad, dev, cur, side, gs)
    finally:
        if side is not None:
            cur.wait_stream(side)


def _encoder_backward(plan, tape, dy, grads, prefix, x3, need_input_grad, dev, cur, side, gs):
    for i in range(len(plan) - 1, -1, -1):
        lp, tp = plan[i], tape[i]
        raw = tp["raw"]
        # the BN kernels write whole 8-channel runs: zero the buffer when its padding goes beyond
        # that (garbage * zero weight would still be NaN for NaN garbage)
        d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8))
        dgamma, dbeta, _ = bn_bwd(dy, 0, raw, 0, lp["cout"], tp["saved"], lp["bn"].weight, L.ACT_RELU, None, d_raw)
        grads[f"{prefix}.{i}.block.1.weight"] = dgamma
        grads[f"{prefix}.{i}.block.1.bias"] = dbeta
        dw_shape = (lp["cout"], lp["cin"], lp["kh"], lp["kw"])       # (the folded first block: (O, kw * I, kh, 1), un-folded below)
        if side is not None and "unfold" not in lp:
            # the weight gradient (MFMA bound, one workgroup per CU) depends only on d_raw and the block's input: it runs on
            # a side stream under the data gradient + the BatchNorm backward passes of the block below (HBM bound).  A folded
            # first block stays on the main stream: its un-fold copy below reads dw right away (stream race)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dw = torch.empty(dw_shape, dtype=torch.float32, device=dev)
                E.wgrad(d_raw, 0, lp["cout"], tp["inp"], 0, lp["cin"], lp["kh"], lp["kw"], dw, dil=lp["dil"], pad=lp["pad"], gs=gs)
            d_raw.t.record_stream(side)
            dw.record_stream(cur)
        else:
            dw = torch.empty(dw_shape, dtype=torch.float32, device=dev)
            # (the reduce of the partial sums may trail on a side stream when nothing touches dw before the optimizer: engine.wgrad)
            E.wgrad(d_raw, 0, lp["cout"], tp["inp"], 0, lp["cin"], lp["kh"], lp["kw"], dw, dil=lp["dil"], pad=lp["pad"],
                    defer="unfold" not in lp and type(grads) is dict)
        grads[f"{prefix}.{i}.block.0.weight"] = lp["unfold"](dw) if "unfold" in lp else dw
        if i == 0 and not need_input_grad:
            break
        inp = tp["inp"]
        d_in = E.Act(inp.B, inp.H, inp.W, inp.cs, x3, dev)
        one, zero = ones_zeros(lp["wd"].shape[1], dev)
        dg = lp.get("dg", lp)           # geometry of the data gradient: the layer's own (a folded first block keeps it in `dg`)
        # "same" convs: pad' = dil*(k-1) - pad == pad
        E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], dg["kh"], dg["kw"], dg["cin"], one, zero, L.ACT_NONE, d_in,
                      cout_store=inp.cs, dil=dg["dil"],
                      pad=(dg["dil"][0] * (dg["kh"] - 1) - dg["pad"][0], dg["dil"][1] * (dg["kw"] - 1) - dg["pad"][1]),
                      Ho=inp.H, Wo=inp.W)
        dy = d_in
    return dy


# ------------------------------------------------------------------ video branch (Conv3dBlock stack) of the audio-visual variant
def _phase_dgrad_weights(w2, s, pad, cout_cs, x3):
    """Data gradient of a zero-padded stride-s conv as s*s stride-1 convs over dy, one per input phase (rh, rw):
    dx[s*j + r] = sum_m w[k0 + s*m] dy[j + c - m], k0 = (r + pad) % s, c = (r + pad - k0) / s  (per axis), i.e. a
    correlation with the flipped sub-kernel g[m'] = w[k0 + s*(M-1-m')] and left padding (M-1) - c.
    Returns {(rh, rw): (packed (I, O, Mh, Mw) weight, Mh, Mw, pad_h, pad_w)} (phases without taps are skipped:
    their gradient is zero)."""
    O, I, kh, kw = w2.shape
    wt = w2.detach().float().transpose(0, 1)                       # (I, O, kh, kw)

    def axis(r, k, p):
        k0 = (r + p) % s
        taps = list(range(k0, k, s))
        c = (r + p - k0) // s
        return taps, c

    phases = {}
    for rh in range(s):
        th, ch = axis(rh, kh, pad[0])
        for rw in range(s):
            tw, cw = axis(rw, kw, pad[1])
            if not th or not tw:
                continue
            sub = wt[:, :, th[::-1]][:, :, :, tw[::-1]].contiguous()
            phases[(rh, rw)] = (E.pack_weight(sub, cout_cs, x3), len(th), len(tw), len(th) - 1 - ch, len(tw) - 1 - cw)
    return phases


def video_train_plan(enc, x3):
    from .common_nets import NO_TEMPORAL_TAPS
    plan = []
    for blk in enc:
        conv, bn = blk.block[0], blk.block[1]
        kt, kh, kw = conv.kernel_size
        O, I, s = conv.out_channels, conv.in_channels, conv.stride[1]
        w2 = conv.weight.detach().permute(0, 2, 1, 3, 4).reshape(O, kt * I, kh, kw)
        cin_store = E.pad_to(kt * I, 16)
        pad = (conv.padding[1], conv.padding[2])
        lp = dict(w=E.pack_weight(w2, cin_store, x3), conv=conv, bn=bn, kt=kt, kh=kh, kw=kw, stride=s, pad=pad, cin=I,
                  cout=O, cin_store=cin_store)
        cout_cs = E.pad_to(O, 16)
        # temporal taps inside the kernels (no materialised time stack) for the blocks whose frames hold whole 128-channel
        # groups and whose weight gradient is one wgrad call (stride <= 2)
        lp["native"] = kt > 1 and I % 128 == 0 and O % 16 == 0 and s <= 2 and not NO_TEMPORAL_TAPS
        if lp["native"]:
            # data gradient: dx[t] = sum_dt corr(dy[t + tpad - dt], w[:, :, dt]) -- a conv over dy with temporal taps
            # s' = kt - 1 - dt, written as the "forward" weight (kt*O, I, kh, kw) whose data-gradient transform is taken
            w2n = conv.weight.detach().flip(2).permute(2, 0, 1, 3, 4).reshape(kt * O, I, kh, kw)
            if s == 1:
                lp["wd"] = dgrad_weight(w2n, x3)
            else:
                lp["wd_phases"] = _phase_dgrad_weights(w2n, s, pad, kt * cout_cs, x3)
        elif s == 1:
            lp["wd"] = dgrad_weight(w2, x3)
        else:
            lp["wd_phases"] = _phase_dgrad_weights(w2, s, pad, cout_cs, x3)
        plan.append(lp)
    return plan


def video_forward_train(plan, frames, B, T, feat, feat_row, feat_third, feat_c_off, x3):
    """Training forward of the Conv3dBlock stack: time stack -> conv (raw) -> BatchNorm3d with batch statistics over
    (B, T, H, W) -> ReLU; the spatial mean of the last block goes into the BiLSTM feature matrix."""
    dev = frames.t.device
    tape = []
    cur, C = frames, 3
    for lp in plan:
        H, W = cur.H, cur.W
        st = cur
        if lp["native"]:
            st = None
        elif lp["kt"] > 1 or cur.cs != lp["cin_store"]:
            st = E.Act(B * T, H, W, lp["cin_store"], x3, dev)
            L.check(L.lib().sos_time_stack(L.ptr(cur.t), B, T, H * W, C, cur.cs, cur.nseg, lp["kt"], L.ptr(st.t), st.cs,
                                           L.stream_ptr()), "sos_time_stack")
        s = lp["stride"]
        Ho = (H + 2 * lp["pad"][0] - lp["kh"]) // s + 1
        Wo = (W + 2 * lp["pad"][1] - lp["kw"]) // s + 1
        cs = E.pad_to(lp["cout"], 16)
        one, zero = ones_zeros(lp["w"].shape[1], dev)
        raw = E.Act(B * T, Ho, Wo, cs, x3, dev)
        if lp["native"]:
            E.conv_to_act(cur, 0, C, lp["w"], lp["kh"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw,
                          cout_store=cs, stride=s, pad=lp["pad"], Ho=Ho, Wo=Wo, temporal=(T, lp["kt"]))
        else:
            E.conv_to_act(st, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw,
                          cout_store=cs, stride=s, pad=lp["pad"], Ho=Ho, Wo=Wo)
        y = E.Act(B * T, Ho, Wo, cs, x3, dev, zero=cs > E.pad_to(lp["cout"], 8))
        saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_RELU, None, y, 0)
        tape.append(dict(src=cur, src_C=C, stacked=st, raw=raw, saved=saved))
        cur, C = y, lp["cout"]
    L.check(L.lib().sos_spatial_mean(L.ptr(cur.t), B * T, cur.H * cur.W, C, cur.cs, cur.nseg, L.ptr(feat), feat_row,
                                     feat_third, feat_c_off, L.stream_ptr()), "sos_spatial_mean")
    return dict(blocks=tape, out=cur)


def video_backward(plan, tape, dfeat, f_row, f_third, f_c_off, grads, prefix, B, T, x3):
    """BPTT side of the video branch: d(features) -> spatial-mean backward -> per block BN/ReLU backward, weight
    gradient against the time-stacked input (reshaped to the Conv3d weight), data gradient (flipped conv, or one conv
    per input phase for the strided blocks) folded back onto the frames."""
    dev = dfeat.device
    last = tape["out"]
    C = plan[-1]["cout"]
    dy = E.Act(last.B, last.H, last.W, last.cs, x3, dev)
    L.check(L.lib().sos_spatial_mean_bwd(L.ptr(dfeat), B * T, last.H * last.W, C, f_row, f_third, f_c_off, dy.nseg,
                                         L.ptr(dy.t), dy.cs, L.stream_ptr()), "sos_spatial_mean_bwd")
    for i in range(len(plan) - 1, -1, -1):
        lp, tp = plan[i], tape["blocks"][i]
        raw, st = tp["raw"], tp["stacked"]
        d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8))
        dgamma, dbeta, _ = bn_bwd(dy, 0, raw, 0, lp["cout"], tp["saved"], lp["bn"].weight, L.ACT_RELU, None, d_raw)
        grads[f"{prefix}.{i}.block.1.weight"], grads[f"{prefix}.{i}.block.1.bias"] = dgamma, dbeta
        kt, I, O = lp["kt"], lp["cin"], lp["cout"]
        dw2 = torch.empty((O, kt * I, lp["kh"], lp["kw"]), dtype=torch.float32, device=dev)
        if lp["native"]:
            src = tp["src"]
            E.wgrad(d_raw, 0, O, src, 0, kt * I, lp["kh"], lp["kw"], dw2, stride=lp["stride"], pad=lp["pad"],
                    temporal=(T, kt, I))
        elif lp["stride"] >= 3:
            # stride 3 (the two smallest feature maps): the wgrad kernel's pixel tile would need a 9x patch; gather the
            # pixels each tap touches (a strided view of the zero-padded input, layout plumbing) and run 1x1 wgrads
            sd_, (ph_, pw_) = lp["stride"], lp["pad"]
            xp = torch.nn.functional.pad(st.t, (0, 0, pw_, pw_ + sd_, ph_, ph_ + sd_))
            for a in range(lp["kh"]):
                for b_ in range(lp["kw"]):
                    sub = E.Act(st.B, raw.H, raw.W, st.cs, x3, dev)
                    sub.t.copy_(xp[:, a:a + sd_ * raw.H:sd_, b_:b_ + sd_ * raw.W:sd_, :])
                    dwa = torch.empty((O, kt * I, 1, 1), dtype=torch.float32, device=dev)
                    E.wgrad(d_raw, 0, O, sub, 0, kt * I, 1, 1, dwa)
                    dw2[:, :, a:a + 1, b_:b_ + 1] = dwa
        else:       # (7x7: the wgrad kernel divides the tap rows over workgroups of one launch)
            E.wgrad(d_raw, 0, O, st, 0, kt * I, lp["kh"], lp["kw"], dw2, stride=lp["stride"], pad=lp["pad"])
        grads[f"{prefix}.{i}.block.0.weight"] = dw2.reshape(O, kt, I, lp["kh"], lp["kw"]).permute(0, 2, 1, 3, 4).contiguous()
        if i == 0:
            break
        if lp["native"]:
            # data gradient straight onto the frames: conv over d_raw with (flipped) temporal taps
            src = tp["src"]
            dx = E.Act(src.B, src.H, src.W, src.cs, x3, dev, zero=lp["stride"] > 1)
            if lp["stride"] == 1:
                one, zero = ones_zeros(lp["wd"].shape[1], dev)
                E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], lp["kh"], lp["kw"], I, one, zero, L.ACT_NONE, dx,
                              cout_store=dx.cs, pad=(lp["kh"] - 1 - lp["pad"][0], lp["kw"] - 1 - lp["pad"][1]), Ho=src.H, Wo=src.W,
                              temporal=(T, kt))
            else:
                s = lp["stride"]
                row = dx.nseg * dx.cs
                for (rh, rw), (w, Mh, Mw, ph, pw) in lp["wd_phases"].items():
                    Hp, Wp = (src.H - rh + s - 1) // s, (src.W - rw + s - 1) // s
                    if Hp < 1 or Wp < 1:
                        continue
                    one, zero = ones_zeros(w.shape[1], dev)
                    E.conv(d_raw, 0, d_raw.cs, w, Mh, Mw, I, one, zero, L.ACT_NONE, out=dx.t, out_dtype=dx.dtype_code,
                           sb=dx.H * dx.W * row, sh=s * dx.W * row, sw=s * row, sc=1, cout_store=dx.cs, third=dx.cs,
                           pad=(ph, pw), Ho=Hp, Wo=Wp, out_elem_offset=(rh * dx.W + rw) * row, temporal=(T, kt))
            dy = dx
            continue
        d_st = E.Act(st.B, st.H, st.W, st.cs, x3, dev, zero=lp["stride"] > 1 or st.cs > E.pad_to(kt * I, 8))
        if lp["stride"] == 1:
            one, zero = ones_zeros(lp["wd"].shape[1], dev)
            E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], lp["kh"], lp["kw"], kt * I, one, zero, L.ACT_NONE, d_st,
                          cout_store=st.cs, pad=(lp["kh"] - 1 - lp["pad"][0], lp["kw"] - 1 - lp["pad"][1]), Ho=st.H, Wo=st.W)
        else:
            s = lp["stride"]
            row = d_st.nseg * d_st.cs
            for (rh, rw), (w, Mh, Mw, ph, pw) in lp["wd_phases"].items():
                Hp, Wp = (st.H - rh + s - 1) // s, (st.W - rw + s - 1) // s
                if Hp < 1 or Wp < 1:
                    continue
                one, zero = ones_zeros(w.shape[1], dev)
                E.conv(d_raw, 0, d_raw.cs, w, Mh, Mw, kt * I, one, zero, L.ACT_NONE, out=d_st.t, out_dtype=d_st.dtype_code,
                       sb=d_st.H * d_st.W * row, sh=s * d_st.W * row, sw=s * row, sc=1, cout_store=st.cs, third=d_st.cs,
                       pad=(ph, pw), Ho=Hp, Wo=Wp, out_elem_offset=(rh * d_st.W + rw) * row)
        src = tp["src"]
        if st is src:                                    # 1x1x1 block on an unstacked input
            dy = d_st
        else:
            dy = E.Act(src.B, src.H, src.W, src.cs, x3, dev)
            L.check(L.lib().sos_time_unstack(L.ptr(d_st.t), B, T, src.H * src.W, tp["src_C"], d_st.cs, d_st.nseg, kt,
                                             L.ptr(dy.t), dy.cs, L.stream_ptr()), "sos_time_unstack")


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
    perm, inv = E.lstm_gate_perm(H, lstm.weight_ih_l0.device)     # gate-interleaved projection rows (sos_hip.h)
    w_ih = lambda: torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], dim=0).detach()[perm]    # noqa: E731  (8H, I)
    bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse]).detach()[perm]
    w = E.pack_weight(lambda: w_ih()[:, :, None, None], cin_store, x3)
    wd = E.pack_weight(lambda: w_ih().t().contiguous()[:, :, None, None], E.pad_to(8 * H, 16), x3)      # (I, 8H)
    dev = w.device
    return dict(w=w, wd=wd, wpk=E.lstm_pack(lstm, x3), H=H, I=lstm.input_size, cin_store=cin_store, inv=inv,
                scale=E.pad_vec(torch.ones(8 * H, device=dev), w.shape[1], 1.0), shift=E.pad_vec(bias, w.shape[1]))


def lstm_forward_train(lp, feat_dims, B, T, x3, dev):
    H = lp["H"]
    xproj = torch.empty((B, T, 8 * H), dtype=torch.float32, device=dev)
    E.conv(None, 0, lp["cin_store"], lp["w"], 1, 1, 8 * H, lp["scale"], lp["shift"], L.ACT_NONE, out=xproj,
           out_dtype=L.DT_F32, sb=T * 8 * H, sh=0, sw=8 * H, sc=1, Ho=1, Wo=T, in_dims=feat_dims)
    h = E.Act(B, 1, T, E.pad_to(2 * H, 16), x3, dev, zero=True)
    Bp = (B + 15) // 16 * 16                       # kernel-native layout, 16-clip groups (sos_hip.h)
    gates = torch.empty((Bp, T, 2, 4 * H), dtype=torch.float32, device=dev)
    csave = torch.empty((Bp, T, 2, H), dtype=torch.float32, device=dev)
    E.lstm(xproj, lp["wpk"], B, T, H, h, gates, csave)
    return h, dict(gates=gates, csave=csave, h=h, feat_dims=feat_dims)


def lstm_backward(lp, tape, dh, grads, prefix, B, T, x3, dev):
    """dh: Act grad of the LSTM output.  Returns d(feature matrix) [B][T][nseg*I] bf16."""
    H, I = lp["H"], lp["I"]
    dgates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dev)
    L.check(L.lib().sos_lstm_bidir_bwd(L.ptr(dh.t), dh.nseg * dh.cs, dh.dtype_code, dh.cs, L.ptr(tape["gates"]),
                                       L.ptr(tape["csave"]), L.ptr(lp["wpk"]["bh"]), L.ptr(lp["wpk"]["bl"]), B, T, H, L.ptr(dgates), L.stream_ptr()),
            "sos_lstm_bidir_bwd")
    dga = E.Act(B, 1, T, E.pad_to(8 * H, 16), x3, dev, zero=(8 * H) % 16 != 0)
    pack_grad(dgates, None, L.ACT_NONE, B, T, 8 * H, T * 8 * H, 8 * H, 1, dga, scaled=True)
    # biases: both b_ih and b_hh receive sum over (b,t) of the gate grads
    inv = lp["inv"]                                # gate-interleaved rows -> torch's (gate, unit) order
    db = colsum(dga, 0, 8 * H)[inv]
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
    dwih = dwih[inv]
    grads[f"{prefix}.weight_ih_l0"] = dwih[:4 * H]
    grads[f"{prefix}.weight_ih_l0_reverse"] = dwih[4 * H:]
    # W_hh: dgate[t] (x) h[t-1] (forward) / h[t+1] (reverse): a 1-tap "conv" shifted by +-1 frame
    hact = tape["h"]
    # (all 2H columns are contracted and the direction's H columns sliced out: keeps the 16-byte
    # channel alignment of the loads for any H)
    for d, sfx, pad_l in ((0, "", 1), (1, "_reverse", -1)):
        dwhh = torch.empty((4 * H, 2 * H), dtype=torch.float32, device=dev)
        E.wgrad(dga, d * 4 * H, 4 * H, hact, 0, 2 * H, 1, 1, dwhh, pad=(0, pad_l))
        grads[f"{prefix}.weight_hh_l0{sfx}"] = dwhh[inv[:4 * H], d * H:(d + 1) * H]
    # input gradient
    nseg = 3 if x3 else 1
    dfeat = torch.empty((B, T, nseg * I), dtype=E.act_dtype(), device=dev)
    one, zero = ones_zeros(lp["wd"].shape[1], dev)
    E.conv(dga, 0, dga.cs, lp["wd"], 1, 1, I, one, zero, L.ACT_NONE, out=dfeat,
           out_dtype=L.DT_BF16X3 if x3 else L.DT_BF16, sb=T * nseg * I, sh=0, sw=nseg * I, sc=1, third=I, Ho=1, Wo=T)
    return dfeat


# ---------------------------------------------------------------------------------------- Linear
def linear_train_plan(lin, cin_store, x3):
    w = E.pack_weight(lambda: lin.weight[:, :, None, None], cin_store, x3)
    wd = E.pack_weight(lambda: lin.weight.detach().t().contiguous()[:, :, None, None], E.pad_to(lin.out_features, 16), x3)
    return dict(w=w, wd=wd, lin=lin, cout=lin.out_features, cin=lin.in_features, cin_store=cin_store,
                scale=E.pad_vec(torch.ones(lin.out_features, device=w.device), w.shape[1], 1.0),
                shift=E.pad_vec(lin.bias, w.shape[1]))


def linear_backward(lp, a_in, dz, grads, prefix, x3, dev):
    """dz: Act grad of the pre-activation output (rows [B,1,T,*]).  Returns Act grad of the input."""
    dw = torch.empty((lp["cout"], lp["cin"]), dtype=torch.float32, device=dev)
    E.wgrad(dz, 0, lp["cout"], a_in, 0, lp["cin"], 1, 1, dw, defer=type(grads) is dict)
    grads[f"{prefix}.weight"] = dw
    grads[f"{prefix}.bias"] = colsum(dz, 0, lp["cout"])
    d_in = E.Act(a_in.B, 1, a_in.W, a_in.cs, x3, dev)
    one, zero = ones_zeros(lp["wd"].shape[1], dev)
    E.conv_to_act(dz, 0, dz.cs, lp["wd"], 1, 1, lp["cin"], one, zero, L.ACT_NONE, d_in, cout_store=a_in.cs, Ho=1,
                  Wo=a_in.W)
    return d_in


# ----------------------------------------------------------- InpaintNet blocks (M2/networks.py:97-205)
def reflect_fold(padded, H, W, pad, dst, dst_off, C, accumulate=True):
    pv, ov = E.view(padded, 0, C), E.view(dst, dst_off, C)
    L.check(L.lib().sos_reflect_fold(ctypes.byref(pv), H, W, pad, ctypes.byref(ov), 1 if accumulate else 0, L.stream_ptr()),
            "sos_reflect_fold")


def copy_crop(src, src_off, dst, dst_off, C):
    sv, dv = E.view(src, src_off, C), E.view(dst, dst_off, C)
    L.check(L.lib().sos_copy_crop(ctypes.byref(sv), src.H, src.W, ctypes.byref(dv), dst.H, dst.W, L.stream_ptr()),
            "sos_copy_crop")


def down_train_plan(blk, x3, in_perm=None, first=False):
    """DownConvBlock: block.0 ReflectionPad2d, block.1 Conv2d, [block.2 BN, block.3 PReLU].  first=True: the block reads the
    2-channel module input and needs no input gradient -- its horizontal taps may sit on the channel axis (engine.wfold_spec):
    forward conv and weight gradient then see a k x 1 layer over kw * I channels (`kw` = 1, `pad_w` = 0, `wtaps` for the pack)."""
    import torch.nn as nn
    conv = blk.block[1]
    has_bn = len(blk.block) > 2 and isinstance(blk.block[2], nn.BatchNorm2d)
    prelu = blk.block[-1] if isinstance(blk.block[-1], nn.PReLU) else None
    k, s, d = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
    cin_store = E.pad_to(conv.in_channels, 16)
    cout_cs = E.pad_to(conv.out_channels, 16)
    wf = lambda: conv.weight.detach().float()                                    # noqa: E731
    wperm = lambda: wf() if in_perm is None else wf()[:, in_perm]                 # noqa: E731
    fold = E.wfold_spec(conv, (k - 1) // 2 * d, L.PAD_REFLECT) if (first and in_perm is None) else None
    if fold is not None:
        cin_store = E.pad_to(fold["kw"] * fold["I"], 16)
        return dict(w=E.pack_weight(lambda: fold["fold"](wf()), cin_store, x3), conv=conv, bn=blk.block[2] if has_bn else None,
                    prelu=prelu, k=k, kw=1, stride=s, dil=d, pad=(k - 1) // 2 * d, pad_w=0, cout=conv.out_channels,
                    cin=fold["kw"] * fold["I"], cin_store=cin_store, in_perm=None, wtaps=fold["wtaps"], unfold=fold["unfold"])
    plan = dict(w=E.pack_weight(wf, cin_store, x3, in_perm), conv=conv, bn=blk.block[2] if has_bn else None, prelu=prelu,
                k=k, kw=k, stride=s, dil=d, pad=(k - 1) // 2 * d, pad_w=(k - 1) // 2 * d, cout=conv.out_channels,
                cin=conv.in_channels, cin_store=cin_store, in_perm=in_perm)
    if s == 1:
        plan["wd"] = E.pack_weight(lambda: wperm().flip(2, 3).transpose(0, 1).contiguous(), cout_cs, x3)
    else:
        assert s == 2 and d == 1
        phases = {}
        for ph in (0, 1):
            for pw in (0, 1):
                Mh, Mw = (k + 1 - ph) // 2, (k + 1 - pw) // 2
                a = [ph + 2 * (Mh - 1 - t) for t in range(Mh)]
                b = [pw + 2 * (Mw - 1 - t) for t in range(Mw)]
                sub = lambda a=a, b=b: wperm()[:, :, a][:, :, :, b].transpose(0, 1).contiguous()     # noqa: E731  (I, O, Mh, Mw)
                phases[(ph, pw)] = (E.pack_weight(sub, cout_cs, x3), Mh, Mw)
        plan["wd_phases"] = phases
    return plan


def up_train_plan(blk, x3):
    """UpConvBlock: block.0 ConvTranspose2d(k3,s2,p1,op1), block.1 BN, block.2 PReLU."""
    ct, bn, prelu = blk.block[0], blk.block[1], blk.block[2]
    cin_store = E.pad_to(ct.in_channels, 16)
    w = lambda: ct.weight.detach().float()                                            # noqa: E731  (Cin, Cout, 3, 3)
    taps = {0: [1], 1: [2, 0]}
    phases = {}
    for ph in (0, 1):
        for pw in (0, 1):
            sub = lambda ph=ph, pw=pw: w()[:, :, taps[ph]][:, :, :, taps[pw]].permute(1, 0, 2, 3).contiguous()   # noqa: E731
            phases[(ph, pw)] = E.pack_weight(sub, cin_store, x3)
    # data gradient: d_in[hi] = sum_a d_raw[2hi - 1 + a] W[ci][co][a] -> stride-2 conv, weight (O=Cin, I=Cout)
    wd = E.pack_weight(w, E.pad_to(ct.out_channels, 16), x3)
    return dict(phases=phases, wd=wd, ct=ct, bn=bn, prelu=prelu, cout=ct.out_channels, cin=ct.in_channels,
                cin_store=cin_store)


def down_forward_train(lp, src, cin_off, dst, c_off, Ho, Wo, x3):
    dev = src.t.device
    cs = E.pad_to(lp["cout"], 16)
    one, zero = ones_zeros(lp["w"].shape[1], dev)
    raw = E.Act(src.B, Ho, Wo, cs, x3, dev)
    st = E.conv_to_act(src, cin_off, lp["cin_store"], lp["w"], lp["k"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw,
                       cout_store=cs, stride=lp["stride"], dil=(lp["dil"], lp["dil"]), pad=(lp["pad"], lp["pad_w"]),
                       pad_mode=L.PAD_REFLECT, Ho=Ho, Wo=Wo, stats_c=lp["cout"] if (FUSED_STATS and FUSED_STATS_UNET) else 0)
    saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_PRELU, lp["prelu"].weight, dst, c_off, stats=st)
    return dict(kind="down", lp=lp, src=src, cin_off=cin_off, dst=dst, c_off=c_off, raw=raw, saved=saved)


def up_forward_train(lp, src, dst, c_off, x3):
    dev = src.t.device
    cs = E.pad_to(lp["cout"], 16)
    raw = E.Act(src.B, 2 * src.H, 2 * src.W, cs, x3, dev, zero=cs > lp["cout"])
    row = raw.nseg * raw.cs
    for (ph, pw), w in lp["phases"].items():
        one, zero = ones_zeros(w.shape[1], dev)
        E.conv(src, 0, lp["cin_store"], w, 1 + ph, 1 + pw, lp["cout"], one, zero, L.ACT_NONE, out=raw.t,
               out_dtype=raw.dtype_code, sb=raw.H * raw.W * row, sh=2 * raw.W * row, sw=2 * row, sc=1,
               cout_store=lp["cout"], third=raw.cs, Ho=src.H, Wo=src.W, out_elem_offset=(ph * raw.W + pw) * row)
    # batch statistics over the UNCROPPED output (the reference resizes after the block), then crop
    if (dst.H, dst.W) == (raw.H, raw.W):            # nothing to crop (up1.1: 2 x 128 x 89 = 256 x 178): straight into the concat buffer
        saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_PRELU, lp["prelu"].weight, dst, c_off)
    else:
        yfull = E.Act(raw.B, raw.H, raw.W, cs, x3, dev, zero=cs > E.pad_to(lp["cout"], 8))
        saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_PRELU, lp["prelu"].weight, yfull, 0)
        copy_crop(yfull, 0, dst, c_off, lp["cout"])
    return dict(kind="up", lp=lp, src=src, dst=dst, c_off=c_off, raw=raw, saved=saved)


class GradBufs:
    """Gradient mirror of every activation buffer, created on first use.  The first producer of a channel slice STORES,
    later ones accumulate (skip connections fan in); channels no producer has written yet are zeroed on demand, so the
    buffers are never memset as a whole (3.5 GB of fills per step for the U-Net: 1 ms)."""

    def __init__(self, x3):
        self.x3, self.bufs, self.written = x3, {}, {}

    def of(self, act):
        k = id(act)
        if k not in self.bufs:
            self.bufs[k] = E.Act(act.B, act.H, act.W, act.cs, self.x3, act.t.device)
        return self.bufs[k]

    def _zero(self, act, lo, hi):
        g = self.of(act)
        g.t.view(g.B, g.H, g.W, g.nseg, g.cs)[..., lo:hi].zero_()

    def first_write(self, act, c_off, C):
        """True when no producer has written any of the channels [c_off, c_off + C) of grad(act) yet: that producer may STORE
        instead of accumulate (one read pass of the tensor less).  When only a part of the slice has been written, the rest
        is zeroed here and the producer accumulates.  The slice counts as written from now on."""
        iv = self.written.setdefault(id(act), [])
        lo, hi = c_off, c_off + C
        hit = [(max(a, lo), min(b, hi)) for a, b in iv if a < hi and lo < b]
        fresh = not hit
        if hit:
            pos = lo
            for a, b in sorted(hit):
                if a > pos:
                    self._zero(act, pos, a)
                pos = max(pos, b)
            if pos < hi:
                self._zero(act, pos, hi)
        iv.append((lo, hi))
        return fresh

    def read(self, act, c_off, C):
        """grad(act) for a consumer that READS channels [c_off, c_off + C): whatever no producer wrote is zero."""
        iv = self.written.setdefault(id(act), [])
        lo, hi = c_off, c_off + C
        pos = lo
        for a, b in sorted((max(a, lo), min(b, hi)) for a, b in iv if a < hi and lo < b):
            if a > pos:
                self._zero(act, pos, a)
            pos = max(pos, b)
        if pos < hi:
            self._zero(act, pos, hi)
            iv.append((lo, hi))
        return self.of(act)


def reflect_fold_border(padded, H, W, pad, dst, dst_off, C):
    pv, ov = E.view(padded, 0, C), E.view(dst, dst_off, C)
    L.check(L.lib().sos_reflect_fold_border(ctypes.byref(pv), H, W, pad, ctypes.byref(ov), L.stream_ptr()), "sos_reflect_fold_border")


FOLD_FUSED = os.environ.get("SOS_FOLD_FUSED", "1") != "0"      # A/B: 0 = padded-domain gradient + a full fold pass (rounds 1-2)


def _reflect_dgrad(lp, d_raw, src, cin_off, gb, x3):
    """Data gradient of ReflectionPad2d + (strided / dilated) valid conv: zero-padded full correlation onto the padded
    domain whose border cells are folded back onto the interior cells they mirror (accumulating into grad(src)).
    Round 3: the convolution writes the INTERIOR cells straight into grad(src) (sos_conv_desc.fold_*) and only the border
    cells into a padded scratch tensor, which sos_reflect_fold_border adds on: the fold no longer reads the whole padded
    gradient and re-writes the whole tensor (3.0 ms of HBM passes per training step at B = 64)."""
    dev = d_raw.t.device
    p, k, d = lp["pad"], lp["k"], lp["dil"]
    H, W = src.H, src.W
    cin_cs = E.pad_to(lp["cin"], 16)
    fused = FOLD_FUSED and lp["cin"] % 8 == 0 and p > 0
    dst = gb.of(src)
    accumulate = not gb.first_write(src, cin_off, lp["cin"])
    if not fused and not accumulate and lp["cin"] % 8:
        # sos_reflect_fold STORES whole 8-channel groups: a slice that is not a multiple of 8 inside a wider gradient buffer
        # would zero up to 7 neighbouring channels -- zero the slice and accumulate instead
        gb._zero(src, cin_off, cin_off + lp["cin"])
        accumulate = True
    dpad = E.Act(src.B, H + 2 * p, W + 2 * p, cin_cs, x3, dev, zero=(lp["stride"] != 1 and not fused))
    drow = dst.nseg * dst.cs
    if lp["stride"] == 1:
        one, zero = ones_zeros(lp["wd"].shape[1], dev)
        if fused:
            E.conv(d_raw, 0, d_raw.cs, lp["wd"], k, k, lp["cin"], one, zero, L.ACT_NONE, out=dst.t, out_dtype=dst.dtype_code,
                   sb=H * W * drow, sh=0, sw=drow, sc=1, c_off=cin_off, cout_store=lp["cin"], third=dst.cs, dil=(d, d),
                   pad=((k - 1) * d, (k - 1) * d), Ho=dpad.H, Wo=dpad.W, accumulate=accumulate, fold=(dpad, p, H, W, 1, 0, 1, 0))
        else:
            E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], k, k, lp["cin"], one, zero, L.ACT_NONE, dpad, cout_store=cin_cs,
                          dil=(d, d), pad=((k - 1) * d, (k - 1) * d), Ho=dpad.H, Wo=dpad.W)
    else:
        row = dpad.nseg * dpad.cs
        for (ph, pw), (w, Mh, Mw) in lp["wd_phases"].items():
            Ho, Wo = (dpad.H - ph + 1) // 2, (dpad.W - pw + 1) // 2
            one, zero = ones_zeros(w.shape[1], dev)
            if fused:
                E.conv(d_raw, 0, d_raw.cs, w, Mh, Mw, lp["cin"], one, zero, L.ACT_NONE, out=dst.t, out_dtype=dst.dtype_code,
                       sb=H * W * drow, sh=0, sw=drow, sc=1, c_off=cin_off, cout_store=lp["cin"], third=dst.cs,
                       pad=(Mh - 1, Mw - 1), Ho=Ho, Wo=Wo, accumulate=accumulate, fold=(dpad, p, H, W, 2, ph, 2, pw))
            else:
                E.conv(d_raw, 0, d_raw.cs, w, Mh, Mw, lp["cin"], one, zero, L.ACT_NONE, out=dpad.t, out_dtype=dpad.dtype_code,
                       sb=dpad.H * dpad.W * row, sh=2 * dpad.W * row, sw=2 * row, sc=1, cout_store=cin_cs, third=dpad.cs,
                       pad=(Mh - 1, Mw - 1), Ho=Ho, Wo=Wo, out_elem_offset=(ph * dpad.W + pw) * row)
    if fused:
        reflect_fold_border(dpad, H, W, p, dst, cin_off, lp["cin"])
    else:
        reflect_fold(dpad, H, W, p, dst, cin_off, lp["cin"], accumulate=accumulate)


_INV_PERM = {}           # (concat permutation, device) -> its inverse as a device index tensor


def down_backward(t, gb, grads, name, x3, need_src_grad=True):
    lp, raw = t["lp"], t["raw"]
    dev = raw.t.device
    d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8))
    dgamma, dbeta, dslope = bn_bwd(gb.read(t["dst"], t["c_off"], lp["cout"]), t["c_off"], raw, 0, lp["cout"], t["saved"],
                                   lp["bn"].weight, L.ACT_PRELU, lp["prelu"].weight, d_raw)
    grads[f"{name}.block.2.weight"], grads[f"{name}.block.2.bias"], grads[f"{name}.block.3.weight"] = dgamma, dbeta, dslope
    dw = torch.empty((lp["cout"], lp["cin"], lp["k"], lp["kw"]), dtype=torch.float32, device=dev)
    E.wgrad(d_raw, 0, lp["cout"], t["src"], t["cin_off"], lp["cin"], lp["k"], lp["kw"], dw, stride=lp["stride"],
            dil=(lp["dil"], lp["dil"]), pad=(lp["pad"], lp["pad_w"]), pad_mode=L.PAD_REFLECT,
            defer="unfold" not in lp and lp["in_perm"] is None and type(grads) is dict)
    if "unfold" in lp:                      # first block with its horizontal taps on the channel axis: back to (O, I, kh, kw)
        dw = lp["unfold"](dw)
    if lp["in_perm"] is not None:           # dw is in the stored channel order; undo the concat permutation
        key = (tuple(lp["in_perm"]), str(dw.device))
        inv = _INV_PERM.get(key)
        if inv is None:
            # built ONCE (the plan dicts are rebuilt every optimizer step): a pageable host -> device copy here blocks the
            # host until the stream has drained, i.e. until the whole stage-2 backward has run (107 ms of "enqueue" time
            # per step instead of ~25)
            inv = torch.empty(len(lp["in_perm"]), dtype=torch.long)
            inv[torch.tensor(lp["in_perm"])] = torch.arange(len(lp["in_perm"]))
            inv = _INV_PERM[key] = inv.to(dw.device)
        dw = dw[:, inv].contiguous()
    grads[f"{name}.block.1.weight"] = dw
    if need_src_grad:
        if "unfold" in lp:
            raise RuntimeError("down_backward: a block planned with first=True (folded horizontal taps) has no input gradient")
        _reflect_dgrad(lp, d_raw, t["src"], t["cin_off"], gb, x3)


def up_backward(t, gb, grads, name, x3):
    lp, raw, src = t["lp"], t["raw"], t["src"]
    dev = raw.t.device
    gdst = gb.read(t["dst"], t["c_off"], lp["cout"])
    if (gdst.H, gdst.W) == (raw.H, raw.W):          # nothing was cropped in the forward pass (up1.1: 2 x 128 x 89): no copy
        dy_full, dy_off = gdst, t["c_off"]
    else:                                           # zero-pad the cropped rows / columns back (mid.8: 128 x 90 -> 128 x 89)
        dy_full, dy_off = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8)), 0
        copy_crop(gdst, t["c_off"], dy_full, 0, lp["cout"])
    d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev, zero=raw.cs > E.pad_to(lp["cout"], 8))
    dgamma, dbeta, dslope = bn_bwd(dy_full, dy_off, raw, 0, lp["cout"], t["saved"], lp["bn"].weight, L.ACT_PRELU,
                                   lp["prelu"].weight, d_raw)
    grads[f"{name}.block.1.weight"], grads[f"{name}.block.1.bias"], grads[f"{name}.block.2.weight"] = dgamma, dbeta, dslope
    dw = torch.empty_like(lp["ct"].weight, dtype=torch.float32)                      # (Cin, Cout, 3, 3)
    E.wgrad(src, 0, lp["cin"], d_raw, 0, lp["cout"], 3, 3, dw, stride=2, pad=(1, 1), defer=type(grads) is dict)
    grads[f"{name}.block.0.weight"] = dw
    gsrc = gb.of(src)
    one, zero = ones_zeros(lp["wd"].shape[1], dev)
    E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], 3, 3, lp["cin"], one, zero, L.ACT_NONE, gsrc, cout_store=lp["cin"],
                  stride=2, pad=(1, 1), Ho=src.H, Wo=src.W, accumulate=not gb.first_write(src, 0, lp["cin"]))
