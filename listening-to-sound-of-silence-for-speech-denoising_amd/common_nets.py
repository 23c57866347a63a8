"""Pieces shared by the detector and the denoiser mirrors: the zero-padded dilated
Conv2d+BN+ReLU stack (Conv2dBlock M1/networks.py:28-51 == ConvBlock M2/networks.py:28-51),
the BiLSTM head and nn.Linear, all executed by the HIP kernels."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E


class Conv2dBlock(nn.Module):
    """Parameter container with the reference's layout: block.0 = Conv2d(bias=False, zero pad
    (k-1)//2*dil, dilation), block.1 = BatchNorm2d, block.2 = ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation):
        super().__init__()
        pad = ((kernel_size[0] - 1) // 2 * dilation[0], (kernel_size[1] - 1) // 2 * dilation[1])
        self.block = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size, 1, pad, dilation, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU())

    def forward(self, x):
        raise RuntimeError("Conv2dBlock is executed by its parent network through libsos_hip")


def make_encoder(kernel_sizes, dilations, nf, outf):
    """make_audio_branch (M1/networks.py:120-128) / make_enc (M2/networks.py:72-80)."""
    blocks = []
    for i, (k, d) in enumerate(zip(kernel_sizes, dilations)):
        blocks.append(Conv2dBlock(2 if i == 0 else nf, nf, k, d))
    blocks.append(Conv2dBlock(nf, outf, (1, 1), (1, 1)))
    return nn.Sequential(*blocks)


class Conv3dBlock(nn.Module):
    """Parameter container of the reference's Conv3dBlock (M1/networks.py:54-77): block.0 = Conv3d(bias=False,
    padding (k-1)//2 per axis, stride), block.1 = BatchNorm3d, block.2 = ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        pad = tuple((k - 1) // 2 for k in kernel_size)
        self.block = nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size, stride, pad, bias=False),
                                   nn.BatchNorm3d(out_channels), nn.ReLU())

    def forward(self, x):
        raise RuntimeError("Conv3dBlock is executed by its parent network through libsos_hip")


VIDEO_KERNEL_SIZES = [(5, 7, 7), (5, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3), (1, 3, 3)]   # M1/networks.py:87
VIDEO_STRIDES = [(1, 2, 2), (1, 1, 1), (1, 2, 2), (1, 2, 2), (1, 2, 2), (1, 3, 3), (1, 3, 3)]         # M1/networks.py:88


def make_video_branch(kernel_sizes, strides, nf=256, outf=256):
    """make_video_branch (M1/networks.py:110-118)."""
    blocks = [Conv3dBlock(3 if i == 0 else nf, nf, k, s) for i, (k, s) in enumerate(zip(kernel_sizes, strides))]
    blocks.append(Conv3dBlock(nf, outf, (1, 1, 1), (1, 1, 1)))
    return nn.Sequential(*blocks)


# SOS_VIDEO_STACK=1: materialise the time-stacked input of every Conv3dBlock (sos_time_stack) instead of the conv kernel's
# temporal taps (A/B timing, and the path the round-1 goldens were first checked on)
NO_TEMPORAL_TAPS = __import__("os").environ.get("SOS_VIDEO_STACK") == "1"


def video_plan(enc, x3):
    """Per Conv3dBlock: the (O, I, kt, kh, kw) weight as a 2-D conv weight over the time-stacked input
    (contraction index dt*I + i, csrc/video.hip), packed like every other conv weight; eval BatchNorm3d folded."""
    plan = []
    for blk in enc:
        conv, bn = blk.block[0], blk.block[1]
        kt, kh, kw = conv.kernel_size
        if conv.stride[0] != 1 or conv.stride[1] != conv.stride[2] or conv.dilation != (1, 1, 1):
            raise NotImplementedError("video branch: temporal stride 1, square spatial stride, no dilation")
        O, I = conv.out_channels, conv.in_channels
        w2 = conv.weight.detach().permute(0, 2, 1, 3, 4).reshape(O, kt * I, kh, kw)
        cin_store = E.pad_to(kt * I, 16)
        w = E.pack_weight(w2, cin_store, x3)
        scale, shift = E.fold_bn(bn, w.shape[1])
        plan.append(dict(w=w, scale=scale, shift=shift, kt=kt, kh=kh, kw=kw, stride=conv.stride[1], cin=I, cout=O,
                         cin_store=cin_store, pad=(conv.padding[1], conv.padding[2])))
    return plan


def run_video_branch(plan, frames, B, T, feat, feat_row, feat_third, feat_c_off, x3):
    """frames: Act [B*T, H, W, 16] (3 real channels).  Every block = one conv with the folded BN + ReLU epilogue whose
    contraction runs over the kt neighbouring frames inside the kernel (temporal taps of sos_conv_desc); only the first
    block (3 channels per frame: 5 frames packed into 16 channels) reads a materialised time stack.  The last block's
    output is averaged over (H, W) into the feature matrix feat[b][t][feat_c_off + c] (torch.mean(f_v, dim=(-2,-1)) +
    the channel concat, M1/networks.py:136,141)."""
    dev = frames.t.device
    cur, C = frames, 3
    for lp in plan:
        H, W = cur.H, cur.W
        native = lp["kt"] > 1 and C % 16 == 0 and cur.cs == C and lp["cin_store"] == lp["kt"] * C and not NO_TEMPORAL_TAPS
        if native:
            s = lp["stride"]
            Ho = (H + 2 * lp["pad"][0] - lp["kh"]) // s + 1
            Wo = (W + 2 * lp["pad"][1] - lp["kw"]) // s + 1
            dst = E.Act(B * T, Ho, Wo, E.pad_to(lp["cout"], 16), x3, dev)
            E.conv_to_act(cur, 0, C, lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU, dst,
                          cout_store=dst.cs, stride=s, pad=lp["pad"], Ho=Ho, Wo=Wo, temporal=(T, lp["kt"]))
            cur, C = dst, lp["cout"]
            continue
        if lp["kt"] > 1 or cur.cs != lp["cin_store"]:
            st = E.Act(B * T, H, W, lp["cin_store"], x3, dev)
            L.check(L.lib().sos_time_stack(L.ptr(cur.t), B, T, H * W, C, cur.cs, cur.nseg, lp["kt"], L.ptr(st.t),
                                           st.cs, L.stream_ptr()), "sos_time_stack")
            cur = st
        s = lp["stride"]
        Ho = (H + 2 * lp["pad"][0] - lp["kh"]) // s + 1
        Wo = (W + 2 * lp["pad"][1] - lp["kw"]) // s + 1
        dst = E.Act(B * T, Ho, Wo, E.pad_to(lp["cout"], 16), x3, dev)
        E.conv_to_act(cur, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"],
                      L.ACT_RELU, dst, cout_store=dst.cs, stride=s, pad=lp["pad"], Ho=Ho, Wo=Wo)
        cur, C = dst, lp["cout"]
    L.check(L.lib().sos_spatial_mean(L.ptr(cur.t), B * T, cur.H * cur.W, C, cur.cs, cur.nseg, L.ptr(feat), feat_row,
                                     feat_third, feat_c_off, L.stream_ptr()), "sos_spatial_mean")


def encoder_plan(enc, x3):
    """Packed weights + folded eval BN for every block of an encoder stack."""
    plan = []
    for i, blk in enumerate(enc):
        conv, bn = blk.block[0], blk.block[1]
        wf = E.wfold_spec(conv, conv.padding[1], L.PAD_ZERO) if i == 0 else None
        if wf is not None:      # horizontal taps on the channel axis (engine.wfold_spec): a kh x 1 layer over kw * I channels
            cin_store = E.pad_to(wf["kw"] * wf["I"], 16)
            w = E.pack_weight(lambda: wf["fold"](conv.weight.detach().float()), cin_store, x3)
            scale, shift = E.fold_bn(bn, w.shape[1])
            plan.append(dict(w=w, scale=scale, shift=shift, kh=conv.kernel_size[0], kw=1, dil=(conv.dilation[0], 1),
                             pad=(conv.padding[0], 0), cout=conv.out_channels, cin_store=cin_store, wtaps=wf["wtaps"]))
            continue
        cin_store = E.pad_to(conv.in_channels, 16)
        w = E.pack_weight(conv.weight, cin_store, x3)
        scale, shift = E.fold_bn(bn, w.shape[1])
        plan.append(dict(w=w, scale=scale, shift=shift, kh=conv.kernel_size[0], kw=conv.kernel_size[1],
                         dil=tuple(conv.dilation), pad=tuple(conv.padding), cout=conv.out_channels,
                         cin_store=cin_store))
    return plan


def pack_encoder_input(plan, x, x3, rag=None, mul=None):
    """The f32 NCHW module input of an encoder stack as the Act its first block reads (with the first block's horizontal
    taps on the channel axis when the plan says so)."""
    wt = plan[0].get("wtaps")
    return E.pack_input(x, x3, mul=mul, wtaps=wt, clip_w=(rag.level(0) if (rag is not None and wt is not None) else None))


def run_encoder(plan, a, feat, feat_row, feat_third, feat_c_off, x3, w_gather=None, T_out=None, rag=None, rag_out=None):
    """Run blocks 0..n-2 NHWC->NHWC (ping-pong) and the final 1x1 block straight into the LSTM
    feature matrix feat[b][t][c*F + f] (the reference's view(B,-1,T).permute(2,0,1),
    M1/networks.py:132,142 / M2/networks.py:84-87), optionally through a nearest-resize column
    gather (F.interpolate, M1/networks.py:133).  rag: engine.Ragged of a variable-length batch (every conv then
    takes the clips' own widths); rag_out: (per-clip output widths table, gather stride) of the last block when it
    resizes (w_gather is then a (B, T_out) table of per-clip nearest indices)."""
    B, H, W = a.B, a.H, a.W
    rk = rag.kw(0) if rag is not None else {}
    cur = a
    bufs = [None, None]
    for i, lp in enumerate(plan[:-1]):
        dst = bufs[i & 1]
        cs = E.pad_to(lp["cout"], 16)
        if dst is None or dst.cs != cs:
            dst = E.Act(B, H, W, cs, x3, a.t.device)
            bufs[i & 1] = dst
        E.conv_to_act(cur, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"],
                      L.ACT_RELU, dst, cout_store=cs, dil=lp["dil"], pad=lp["pad"], Ho=H, Wo=W, **rk)
        cur = dst
    lp = plan[-1]
    Wo = W if T_out is None else T_out
    if rag is not None and rag_out is not None:       # logical input columns = output columns = the clip's frame count
        rk = dict(wl_tab=rag_out[0], wo_tab=rag_out[0], wg_stride=rag_out[1], valid_cols=sum(rag.n_vframes))
    E.conv(cur, 0, lp["cin_store"], lp["w"], 1, 1, lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU,
           out=feat, out_dtype=L.DT_BF16X3 if x3 else L.DT_BF16, sb=Wo * feat_row, sh=1, sw=feat_row, sc=H,
           c_off=feat_c_off, third=feat_third, Ho=H, Wo=Wo, w_gather=w_gather, **rk)


def lstm_plan(lstm, cin_store, x3):
    H = lstm.hidden_size
    perm, _ = E.lstm_gate_perm(H, lstm.weight_ih_l0.device)       # gate-interleaved projection rows (sos_hip.h)
    w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], dim=0).detach()[perm]
    bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse]).detach()[perm]
    w = E.pack_weight(w_ih[:, :, None, None], cin_store, x3)
    return dict(w=w, scale=E.pad_vec(torch.ones(8 * H, device=w.device), w.shape[1], 1.0),
                shift=E.pad_vec(bias, w.shape[1]), wpk=E.lstm_pack(lstm, x3), H=H, cin_store=cin_store)


def run_lstm(lp, feat_dims, B, T, x3, device, lengths=None):
    """Input projection (1x1 conv on MFMA) + recurrent kernel -> Act [B,1,T,pad16(2H)].  lengths: int32 device (B,)
    frames per clip of a ragged batch."""
    H = lp["H"]
    xproj = torch.empty((B, T, 8 * H), dtype=torch.float32, device=device)
    E.conv(None, 0, lp["cin_store"], lp["w"], 1, 1, 8 * H, lp["scale"], lp["shift"], L.ACT_NONE,
           out=xproj, out_dtype=L.DT_F32, sb=T * 8 * H, sh=0, sw=8 * H, sc=1, Ho=1, Wo=T, in_dims=feat_dims)
    h = E.Act(B, 1, T, E.pad_to(2 * H, 16), x3, device, zero=True)
    E.lstm(xproj, lp["wpk"], B, T, H, h, lengths=lengths)
    return h


def linear_plan(lin, cin_store, x3, absolute=False):
    """absolute=True: |W|, |b| -- the layer that bounds the magnitude of the terms of `lin`'s dot products (pipeline.detect)."""
    wt, bias = (lin.weight.abs(), lin.bias.abs()) if absolute else (lin.weight, lin.bias)
    w = E.pack_weight(wt[:, :, None, None], cin_store, x3)
    return dict(w=w, scale=E.pad_vec(torch.ones(lin.out_features, device=w.device), w.shape[1], 1.0),
                shift=E.pad_vec(bias, w.shape[1]), cout=lin.out_features, cin_store=cin_store)


_nearest_tables = {}


def nearest_index(in_size, out_size, device):
    """Source column of F.interpolate(mode='nearest'): min(floor(dst * float32(in/out)), in-1)
    (ATen nearest_idx); int32 table on the device, uploaded once per (in, out, device) -- no host->device copy
    on later calls (none is allowed inside a hipGraph capture)."""
    key = (int(in_size), int(out_size), str(device))
    if key not in _nearest_tables:
        scale = np.float32(in_size) / np.float32(out_size)
        idx = np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1)
        _nearest_tables[key] = torch.from_numpy(idx.astype(np.int32)).to(device)
    return _nearest_tables[key]


_gather_range_tables = {}


def nearest_ranges(in_size, out_size, device):
    """Backward companion of nearest_index: for every source column w the half-open range [lo[w], hi[w]) of output
    columns that read it (the gather is monotonic).  Computed on the host from the same formula and uploaded once per
    (in, out, device): the backward pass neither synchronises nor copies in steady state."""
    key = (int(in_size), int(out_size), str(device))
    if key not in _gather_range_tables:
        scale = np.float32(in_size) / np.float32(out_size)
        idx = np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1)
        cols = np.arange(in_size)
        lo = np.searchsorted(idx, cols, side="left").astype(np.int32)
        hi = np.searchsorted(idx, cols, side="right").astype(np.int32)
        _gather_range_tables[key] = (torch.from_numpy(lo).to(device), torch.from_numpy(hi).to(device))
    return _gather_range_tables[key]


def nearest_index_ragged(rag, n_max, device):
    """(B, n_max) int32: row b = nearest_index(rag.T[b], rag.n_vframes[b]) padded with zeros (never read: the conv's
    per-clip logical width stops at n_vframes[b]); cached on the Ragged object."""
    key = ("nearest", n_max)
    rag = getattr(rag, "base", rag)             # (engine.MaskedRagged: the gather rows are its base geometry's, built once)
    t = rag._tabs.get(key)
    if t is None:
        rows = np.zeros((len(rag.T), n_max), dtype=np.int32)
        for b, (Tc, nc) in enumerate(zip(rag.T, rag.n_vframes)):
            scale = np.float32(Tc) / np.float32(nc)
            rows[b, :nc] = np.minimum(np.floor(np.arange(nc, dtype=np.float32) * scale).astype(np.int64), Tc - 1)
        t = rag._tabs[key] = E.upload(rows, torch.int32, device)
    return t
