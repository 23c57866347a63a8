"""Mirror of M2/networks.py (model_2_audio_denoising/audio_denoising_model/networks.py):
`get_network(config)` -> JointModel = InpaintNet (stage1) + ContextAggNet (stage2), same module
tree / state_dict keys, forward executed by the HIP kernels."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import get_precision, precision_scope
from .. import common_nets as CN
from .. import engine as E
from .. import train_ops as TO


def get_network(config):
    """M2/networks.py:8-9; reads only config.kernel_sizes / config.dilations (:212)."""
    return JointModel(config)


class DownConvBlock(nn.Module):
    """M2/networks.py:97-117: block.0 ReflectionPad2d, block.1 Conv2d(valid), block.2 BN, block.3 PReLU
    (norm_fn=None -> conv has a bias and there is no BN; act=None -> no PReLU)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation=1, norm_fn='bn', act='prelu'):
        super().__init__()
        pad = (kernel_size - 1) // 2 * dilation
        block = [nn.ReflectionPad2d(pad),
                 nn.Conv2d(in_channels, out_channels, kernel_size, stride, 0, dilation, bias=norm_fn is None)]
        if norm_fn == 'bn':
            block.append(nn.BatchNorm2d(out_channels))
        if act == 'prelu':
            block.append(nn.PReLU())
        self.block = nn.Sequential(*block)

    def forward(self, x):
        raise RuntimeError("DownConvBlock is executed by InpaintNet through libsos_hip")


class UpConvBlock(nn.Module):
    """M2/networks.py:120-149 ('upconv'): block.0 ConvTranspose2d(k, stride, pad, output_padding=1)
    -- the reference passes `dilation` in the output_padding slot (:130) --, block.1 BN, block.2 PReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation=1):
        super().__init__()
        pad = (kernel_size - 1) // 2 * dilation
        self.block = nn.Sequential(
            nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride, pad, dilation, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.PReLU())

    def forward(self, x):
        raise RuntimeError("UpConvBlock is executed by InpaintNet through libsos_hip")


def _down_plan(blk, x3, in_perm=None, first=False):
    """first=True: the block reads the 2-channel module input -- its horizontal taps may sit on the channel axis
    (engine.wfold_spec): the plan then describes a k x 1 layer and carries `wtaps` for the boundary pack."""
    conv = blk.block[1]
    k0 = conv.kernel_size[0]
    wf = E.wfold_spec(conv, (k0 - 1) // 2 * conv.dilation[0], L.PAD_REFLECT) if (first and in_perm is None) else None
    if wf is not None:
        cin_store = E.pad_to(wf["kw"] * wf["I"], 16)
        w = E.pack_weight(lambda: wf["fold"](conv.weight.detach().float()), cin_store, x3)
    else:
        cin_store = E.pad_to(conv.in_channels, 16)
        w = E.pack_weight(conv.weight, cin_store, x3, in_perm)
    has_bn = len(blk.block) > 2 and isinstance(blk.block[2], nn.BatchNorm2d)
    if has_bn:
        scale, shift = E.fold_bn(blk.block[2], w.shape[1])
    else:
        scale = E.pad_vec(torch.ones(conv.out_channels, device=w.device), w.shape[1], 1.0)
        shift = E.pad_vec(conv.bias, w.shape[1])
    prelu = blk.block[-1] if isinstance(blk.block[-1], nn.PReLU) else None
    k = conv.kernel_size[0]
    pad = (k - 1) // 2 * conv.dilation[0]
    return dict(w=w, scale=scale, shift=shift, k=k, kw=1 if wf is not None else k, stride=conv.stride[0], dil=conv.dilation[0],
                pad=pad, pad_w=0 if wf is not None else pad, cout=conv.out_channels, cin_store=cin_store,
                wtaps=wf["wtaps"] if wf is not None else None,
                slope=prelu.weight.detach().float() if prelu is not None else None,
                act=L.ACT_PRELU if prelu is not None else L.ACT_NONE)


# ConvTranspose2d(k3,s2,p1,op1): output row 2i+ph gathers input rows i+u with kernel row a:
#   ph=0: (u=0,a=1)        ph=1: (u=0,a=2), (u=1,a=0)        (same for columns)
_PHASE_TAPS = {0: [1], 1: [2, 0]}


def _up_plan(blk, x3):
    ct, bn, prelu = blk.block[0], blk.block[1], blk.block[2]
    assert ct.kernel_size == (3, 3) and ct.stride == (2, 2) and ct.padding == (1, 1) and ct.output_padding == (1, 1)
    cin_store = E.pad_to(ct.in_channels, 16)
    w = ct.weight.detach().float()                       # (Cin, Cout, 3, 3)
    phases = {}
    for ph in (0, 1):
        for pw in (0, 1):
            a, b = _PHASE_TAPS[ph], _PHASE_TAPS[pw]
            sub = w[:, :, a][:, :, :, b].permute(1, 0, 2, 3).contiguous()   # (Cout, Cin, len(a), len(b))
            phases[(ph, pw)] = E.pack_weight(sub, cin_store, x3)
    cout_pad = E.pad_to(ct.out_channels, 32)
    scale, shift = E.fold_bn(bn, cout_pad)
    return dict(phases=phases, scale=scale, shift=shift, cout=ct.out_channels, cin_store=cin_store,
                slope=prelu.weight.detach().float())


def _run_down(lp, src, cin_off, dst, c_off, Ho, Wo, rk=None):
    """rk: ragged-batch keywords (engine.Ragged.kw): the clips' own input / output widths."""
    E.conv_to_act(src, cin_off, lp["cin_store"], lp["w"], lp["k"], lp["kw"], lp["cout"], lp["scale"], lp["shift"],
                  lp["act"], dst, c_off=c_off, cout_store=lp["cout"], stride=lp["stride"], dil=(lp["dil"], lp["dil"]),
                  pad=(lp["pad"], lp["pad_w"]), pad_mode=L.PAD_REFLECT, slope=lp["slope"], Ho=Ho, Wo=Wo, **(rk or {}))


def _run_up(lp, src, dst, c_off, rag=None, lsrc=0, ldst=0):
    """Four output-parity phases of the transposed conv, each a small stride-1 conv written to the
    interleaved positions of `dst` (cropped to dst's size == the reference's nearest-resize fix-up
    F.interpolate(out, skip.size()) which drops the surplus last row/column, M2/networks.py:199-203)."""
    row = dst.nseg * dst.cs
    for (ph, pw), w in lp["phases"].items():
        Ho = min(src.H, (dst.H - ph + 1) // 2)
        Wo = min(src.W, (dst.W - pw + 1) // 2)
        if Ho <= 0 or Wo <= 0:
            continue
        rk = {}
        if rag is not None:     # per clip: its own source width, and the crop to ITS skip connection's width
            ws, wd = rag.widths(lsrc), rag.widths(ldst)
            wo = [min(a, (b - pw + 1) // 2) for a, b in zip(ws, wd)]
            rk = dict(wl_tab=rag.tab(ws), wo_tab=rag.tab(wo), valid_cols=sum(wo))
        E.conv(src, 0, lp["cin_store"], w, 1 + ph, 1 + pw, lp["cout"], lp["scale"], lp["shift"], L.ACT_PRELU,
               out=dst.t, out_dtype=dst.dtype_code, sb=dst.H * dst.W * row, sh=2 * dst.W * row, sw=2 * row, sc=1,
               c_off=c_off, cout_store=lp["cout"], third=dst.cs, slope=lp["slope"], Ho=Ho, Wo=Wo,
               out_elem_offset=(ph * dst.W + pw) * row, **rk)


class InpaintNet(nn.Module):
    """M2/networks.py:152-205."""

    def __init__(self):
        super().__init__()
        ch1, ch2, ch3 = 64, 128, 256
        self.down1 = nn.Sequential(DownConvBlock(2, ch1, 5, 1))
        self.down2 = nn.Sequential(DownConvBlock(ch1, ch2, 5, 2), DownConvBlock(ch2, ch2, 5, 1))
        self.down3 = nn.Sequential(DownConvBlock(2, ch1, 5, 1))
        self.down4 = nn.Sequential(DownConvBlock(ch1, ch2, 5, 2), DownConvBlock(ch2, ch2, 5, 1))
        self.mid = nn.Sequential(
            DownConvBlock(ch2 * 2, ch3, 3, 2),
            DownConvBlock(ch3, ch3, 3, 1),
            DownConvBlock(ch3, ch3, 3, 1, dilation=2),
            DownConvBlock(ch3, ch3, 3, 1, dilation=4),
            DownConvBlock(ch3, ch3, 3, 1, dilation=8),
            DownConvBlock(ch3, ch3, 3, 1, dilation=16),
            DownConvBlock(ch3, ch3, 3, 1),
            DownConvBlock(ch3, ch3, 3, 1),
            UpConvBlock(ch3, ch2, 3, 2))
        self.up1 = nn.Sequential(DownConvBlock(ch2 * 2, ch2, 3, 1), UpConvBlock(ch2, ch1, 3, 2))
        self.up2 = nn.Sequential(DownConvBlock(ch1 * 2, ch1, 3, 1), DownConvBlock(ch1, 2, 3, 1, norm_fn=None, act=None))

    def build_plan(self, x3):
        perm_up1 = list(range(128, 256)) + list(range(0, 128))   # buffer order [down4 | out] vs cat([out, down4])
        return dict(
            down1=_down_plan(self.down1[0], x3, first=True), down2_0=_down_plan(self.down2[0], x3),
            down2_1=_down_plan(self.down2[1], x3), down3=_down_plan(self.down3[0], x3, first=True),
            down4_0=_down_plan(self.down4[0], x3), down4_1=_down_plan(self.down4[1], x3),
            mid=[_down_plan(self.mid[i], x3) for i in range(8)], mid8=_up_plan(self.mid[8], x3),
            up1_0=_down_plan(self.up1[0], x3, perm_up1), up1_1=_up_plan(self.up1[1], x3),
            up2_0=_down_plan(self.up2[0], x3), up2_1=_down_plan(self.up2[1], x3))

    def run(self, plan, x, y, x3, rag=None):
        """forward(x, y) of M2/networks.py:192-205: x = noise-interval STFT, y = mixed STFT,
        both f32 (B,2,F,T); returns f32 (B,2,F,T).  rag: engine.Ragged of a variable-length batch (T = the longest
        clip; every block takes the clips' own widths at its resolution level, so reflect borders, stride-2 sizes and
        the crop after the transposed convs are each clip's own)."""
        dev = x.device
        B, _, H, W = x.shape
        H1, W1 = (H + 1) // 2, (W + 1) // 2          # after the 5x5 stride-2 reflect-padded convs
        H2, W2 = (H1 + 1) // 2, (W1 + 1) // 2        # after the 3x3 stride-2 one
        cw = rag.level(0) if rag is not None else None
        ax = E.pack_input(x, x3, wtaps=plan["down1"]["wtaps"], clip_w=cw if plan["down1"]["wtaps"] else None)
        ay = E.pack_input(y, x3, wtaps=plan["down3"]["wtaps"], clip_w=cw if plan["down3"]["wtaps"] else None)
        d1 = E.Act(B, H, W, 64, x3, dev)
        U2 = E.Act(B, H, W, 128, x3, dev)            # [up1.1 out | down3]
        X = E.Act(B, H1, W1, 384, x3, dev)           # [down2 | down4 | mid.8 out]
        t128 = E.Act(B, H1, W1, 128, x3, dev)
        k = (lambda a, b=None: rag.kw(a, b)) if rag is not None else (lambda a, b=None: None)
        _run_down(plan["down1"], ax, 0, d1, 0, H, W, k(0))
        _run_down(plan["down2_0"], d1, 0, t128, 0, H1, W1, k(0, 1))
        _run_down(plan["down2_1"], t128, 0, X, 0, H1, W1, k(1))
        _run_down(plan["down3"], ay, 0, U2, 64, H, W, k(0))
        _run_down(plan["down4_0"], U2, 64, t128, 0, H1, W1, k(0, 1))
        _run_down(plan["down4_1"], t128, 0, X, 128, H1, W1, k(1))
        m = [E.Act(B, H2, W2, 256, x3, dev), E.Act(B, H2, W2, 256, x3, dev)]
        _run_down(plan["mid"][0], X, 0, m[0], 0, H2, W2, k(1, 2))
        for i in range(1, 8):
            _run_down(plan["mid"][i], m[(i - 1) & 1], 0, m[i & 1], 0, H2, W2, k(2))
        _run_up(plan["mid8"], m[1], X, 256, rag, 2, 1)
        _run_down(plan["up1_0"], X, 128, t128, 0, H1, W1, k(1))
        _run_up(plan["up1_1"], t128, U2, 0, rag, 1, 0)
        _run_down(plan["up2_0"], U2, 0, d1, 0, H, W, k(0))
        lp = plan["up2_1"]
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        E.conv(d1, 0, lp["cin_store"], lp["w"], 3, 3, 2, lp["scale"], lp["shift"], L.ACT_NONE, out=out,
               out_dtype=L.DT_F32, sb=2 * H * W, sh=W, sw=1, sc=H * W, pad=(1, 1), pad_mode=L.PAD_REFLECT, Ho=H, Wo=W,
               **(k(0) or {}))
        return out

    # ------------------------------------------------------------------ training path
    _LAYERS = ["down1.0", "down2.0", "down2.1", "down3.0", "down4.0", "down4.1"] + [f"mid.{i}" for i in range(8)]

    def build_train_plan(self, x3):
        perm_up1 = list(range(128, 256)) + list(range(0, 128))
        P = TO.down_train_plan
        return dict(down1=P(self.down1[0], x3, first=True), down2_0=P(self.down2[0], x3), down2_1=P(self.down2[1], x3),
                    down3=P(self.down3[0], x3, first=True), down4_0=P(self.down4[0], x3), down4_1=P(self.down4[1], x3),
                    mid=[P(self.mid[i], x3) for i in range(8)], mid8=TO.up_train_plan(self.mid[8], x3),
                    up1_0=P(self.up1[0], x3, perm_up1), up1_1=TO.up_train_plan(self.up1[1], x3),
                    up2_0=P(self.up2[0], x3), up2_1=_down_plan(self.up2[1], x3),
                    up2_1_wd=E.pack_weight(lambda: self.up2[1].block[1].weight.detach().float().flip(2, 3).transpose(0, 1).contiguous(),
                                           16, x3))

    def forward_train(self, plan, x, y, x3):
        dev = x.device
        B, _, H, W = x.shape
        H1, W1 = (H + 1) // 2, (W + 1) // 2
        H2, W2 = (H1 + 1) // 2, (W1 + 1) // 2
        ax = E.pack_input(x, x3, wtaps=plan["down1"].get("wtaps"))
        ay = E.pack_input(y, x3, wtaps=plan["down3"].get("wtaps"))
        d1 = E.Act(B, H, W, 64, x3, dev)
        U2 = E.Act(B, H, W, 128, x3, dev)
        X = E.Act(B, H1, W1, 384, x3, dev)
        ta, tb = E.Act(B, H1, W1, 128, x3, dev), E.Act(B, H1, W1, 128, x3, dev)
        F_ = TO.down_forward_train
        tape = []
        tape.append(("down1.0", F_(plan["down1"], ax, 0, d1, 0, H, W, x3)))
        tape.append(("down2.0", F_(plan["down2_0"], d1, 0, ta, 0, H1, W1, x3)))
        tape.append(("down2.1", F_(plan["down2_1"], ta, 0, X, 0, H1, W1, x3)))
        tape.append(("down3.0", F_(plan["down3"], ay, 0, U2, 64, H, W, x3)))
        tape.append(("down4.0", F_(plan["down4_0"], U2, 64, tb, 0, H1, W1, x3)))
        tape.append(("down4.1", F_(plan["down4_1"], tb, 0, X, 128, H1, W1, x3)))
        prev, cin_off = X, 0
        for i in range(8):
            m = E.Act(B, H2, W2, 256, x3, dev)
            tape.append((f"mid.{i}", F_(plan["mid"][i], prev, cin_off, m, 0, H2, W2, x3)))
            prev = m
        tape.append(("mid.8", TO.up_forward_train(plan["mid8"], prev, X, 256, x3)))
        u128 = E.Act(B, H1, W1, 128, x3, dev)
        tape.append(("up1.0", F_(plan["up1_0"], X, 128, u128, 0, H1, W1, x3)))
        tape.append(("up1.1", TO.up_forward_train(plan["up1_1"], u128, U2, 0, x3)))
        u64 = E.Act(B, H, W, 64, x3, dev)
        tape.append(("up2.0", F_(plan["up2_0"], U2, 0, u64, 0, H, W, x3)))
        lp = plan["up2_1"]
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        E.conv(u64, 0, lp["cin_store"], lp["w"], 3, 3, 2, lp["scale"], lp["shift"], L.ACT_NONE, out=out,
               out_dtype=L.DT_F32, sb=2 * H * W, sh=W, sw=1, sc=H * W, pad=(1, 1), pad_mode=L.PAD_REFLECT, Ho=H, Wo=W)
        return out, dict(layers=tape, u64=u64, packed=(ax, ay))

    def backward(self, plan, tape, d_out, grads, x3, prefix="stage1"):
        """d_out: Act [B,H,W,16] holding the gradient of the 2-channel output."""
        dev = d_out.t.device
        gb = TO.GradBufs(x3)
        u64 = tape["u64"]
        # up2.1: conv + bias, no BN / activation
        grads[f"{prefix}.up2.1.block.1.bias"] = TO.colsum(d_out, 0, 2)
        dw = torch.empty((2, 64, 3, 3), dtype=torch.float32, device=dev)
        E.wgrad(d_out, 0, 2, u64, 0, 64, 3, 3, dw, pad=(1, 1), pad_mode=L.PAD_REFLECT, defer=type(grads) is dict)
        grads[f"{prefix}.up2.1.block.1.weight"] = dw
        last = dict(wd=plan["up2_1_wd"], pad=1, k=3, dil=1, stride=1, cin=64)
        TO._reflect_dgrad(last, d_out, u64, 0, gb, x3)
        no_src_grad = {"down1.0", "down3.0"}
        for name, t in reversed(tape["layers"]):
            full = f"{prefix}.{name}"
            if t["kind"] == "up":
                TO.up_backward(t, gb, grads, full, x3)
            else:
                TO.down_backward(t, gb, grads, full, x3, need_src_grad=name not in no_src_grad)

    def forward(self, x, y):
        raise RuntimeError("call InpaintNet through JointModel (libsos_hip path)")


class ContextAggNet(nn.Module):
    """M2/networks.py:54-94."""

    def __init__(self, kernel_sizes, dilations, freq_bins=256, nf=96):
        super().__init__()
        self.encoder_x = CN.make_encoder(kernel_sizes, dilations, nf, 8)
        self.encoder_n = CN.make_encoder(kernel_sizes, dilations, nf // 2, 4)
        self.lstm = nn.LSTM(input_size=8 * freq_bins + 4 * freq_bins, hidden_size=200, bidirectional=True)
        self.fc = nn.Sequential(nn.Linear(400, 600), nn.ReLU(True), nn.Linear(600, 600), nn.ReLU(True),
                                nn.Linear(600, freq_bins * 2), nn.Sigmoid())
        self.freq_bins = freq_bins

    def build_plan(self, x3):
        return dict(enc_x=CN.encoder_plan(self.encoder_x, x3), enc_n=CN.encoder_plan(self.encoder_n, x3),
                    lstm=CN.lstm_plan(self.lstm, 12 * self.freq_bins, x3),
                    fc0=CN.linear_plan(self.fc[0], 400, x3), fc2=CN.linear_plan(self.fc[2], E.pad_to(600, 16), x3),
                    fc4=CN.linear_plan(self.fc[4], E.pad_to(600, 16), x3))

    def start_x(self, plan, x, x3, rag, side):
        """encoder_x of forward(x, n) -- the half of stage 2 that does not depend on n (M2/networks.py:84) -- enqueued on `side`
        ahead of the rest; returns the handle run(..., started=) takes."""
        B, _, F, T = x.shape
        nseg = 3 if x3 else 1
        nfeat = 12 * F
        feat = (torch.empty if rag is None else torch.zeros)((B, T, nseg * nfeat), dtype=E.act_dtype(), device=x.device)
        cur = torch.cuda.current_stream(x.device)
        side.wait_stream(cur)                     # the input, the packed weights and `feat` exist
        with torch.cuda.stream(side):
            CN.run_encoder(plan["enc_x"], CN.pack_encoder_input(plan["enc_x"], x, x3, rag), feat, nseg * nfeat, nfeat, 0, x3, rag=rag)
        return dict(feat=feat, side=side, x=x, x3=x3, mode=get_precision())     # the effective mode the feature matrix was written in

    def run(self, plan, x, n, x3, rag=None, started=None):
        """forward(x, n) of M2/networks.py:82-94 -> sigmoid mask f32 (B,2,F,T).  rag: engine.Ragged of a variable-length
        batch.  started: handle of start_x() for this very x (encoder_x is already running on a side stream: joined before the
        BiLSTM)."""
        dev = x.device
        B, _, F, T = x.shape
        nseg = 3 if x3 else 1
        nfeat = 12 * F
        lengths = rag.level(0) if rag is not None else None
        if started is not None:
            feat = started["feat"]
        else:
            # ragged: rows of frames past a clip's end stay zero (finite) -- the FC head runs over all rows
            feat = (torch.empty if rag is None else torch.zeros)((B, T, nseg * nfeat), dtype=E.act_dtype(), device=dev)
            CN.run_encoder(plan["enc_x"], CN.pack_encoder_input(plan["enc_x"], x, x3, rag), feat, nseg * nfeat, nfeat, 0, x3, rag=rag)
        CN.run_encoder(plan["enc_n"], CN.pack_encoder_input(plan["enc_n"], n, x3, rag), feat, nseg * nfeat, nfeat, 8, x3, rag=rag)
        if started is not None:
            torch.cuda.current_stream(dev).wait_stream(started["side"])      # join: the feature matrix is complete
        h = CN.run_lstm(plan["lstm"], (feat, B, 1, T, nfeat, nseg), B, T, x3, dev, lengths=lengths)
        f0, f2, f4 = plan["fc0"], plan["fc2"], plan["fc4"]
        a0 = E.Act(B, 1, T, E.pad_to(600, 16), x3, dev)
        a1 = E.Act(B, 1, T, E.pad_to(600, 16), x3, dev)
        E.conv_to_act(h, 0, f0["cin_store"], f0["w"], 1, 1, 600, f0["scale"], f0["shift"], L.ACT_RELU, a0,
                      cout_store=a0.cs, Ho=1, Wo=T)
        E.conv_to_act(a0, 0, f2["cin_store"], f2["w"], 1, 1, 600, f2["scale"], f2["shift"], L.ACT_RELU, a1,
                      cout_store=a1.cs, Ho=1, Wo=T)
        out = torch.empty((B, 2, F, T), dtype=torch.float32, device=dev)
        # fc output index c*F+f of pixel (b,t) lands at out[b, c, f, t]  (permute(0,2,1).view, :92)
        E.conv(a1, 0, f4["cin_store"], f4["w"], 1, 1, 2 * F, f4["scale"], f4["shift"], L.ACT_SIGMOID, out=out,
               out_dtype=L.DT_F32, sb=2 * F * T, sh=0, sw=1, sc=T, Ho=1, Wo=T)
        return out

    # ------------------------------------------------------------------ training path
    def build_train_plan(self, x3):
        return dict(enc_x=TO.encoder_train_plan(self.encoder_x, x3), enc_n=TO.encoder_train_plan(self.encoder_n, x3),
                    lstm=TO.lstm_train_plan(self.lstm, 12 * self.freq_bins, x3),
                    fc0=TO.linear_train_plan(self.fc[0], 400, x3), fc2=TO.linear_train_plan(self.fc[2], E.pad_to(600, 16), x3),
                    fc4=TO.linear_train_plan(self.fc[4], E.pad_to(600, 16), x3))

    def forward_train(self, plan, x, n, x3, before_lstm=None, side=None):
        """n: the stage-1 prediction, or a zero-argument callable producing it (JointModel passes stage 1's training forward).
        side: optional HIP stream for the branch [n() -> encoder_n] -- it does not depend on encoder_x (M2/networks.py:214-217:
        only encoder_n consumes n_pred), so the U-Net and the 48-channel stack run beside the 96-channel stack and join at the
        BiLSTM.  Same kernels in the same order per branch: bit-identical to the one-stream schedule."""
        dev = x.device
        B, _, F, T = x.shape
        nseg = 3 if x3 else 1
        nfeat = 12 * F
        feat = torch.empty((B, T, nseg * nfeat), dtype=E.act_dtype(), device=dev)
        fs = dict(t=feat, row=nseg * nfeat, third=nfeat, H=F, W=T, Wo=T, gather=None, x3=x3)
        n_extra = None
        if side is not None:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)                     # the inputs, the refreshed weights and `feat` exist
            with torch.cuda.stream(side):
                if callable(n):
                    n, n_extra = n()
                tn = TO.encoder_forward_train(plan["enc_n"], CN.pack_encoder_input(plan["enc_n"], n, x3), dict(fs, c_off=8), x3)
            tx = TO.encoder_forward_train(plan["enc_x"], CN.pack_encoder_input(plan["enc_x"], x, x3), dict(fs, c_off=0), x3)
            cur.wait_stream(side)                     # join: the feature matrix is complete
        else:
            if callable(n):
                n, n_extra = n()
            tx = TO.encoder_forward_train(plan["enc_x"], CN.pack_encoder_input(plan["enc_x"], x, x3), dict(fs, c_off=0), x3)
            tn = TO.encoder_forward_train(plan["enc_n"], CN.pack_encoder_input(plan["enc_n"], n, x3), dict(fs, c_off=8), x3)
        if before_lstm is not None:    # agent.train_concurrent: the recurrence, the FC head and their backward leave the chip
            before_lstm()              # mostly idle (8 workgroups stepping through T frames): another model's forward may start
        h, tl = TO.lstm_forward_train(plan["lstm"], (feat, B, 1, T, nfeat, nseg), B, T, x3, dev)
        f0, f2, f4 = plan["fc0"], plan["fc2"], plan["fc4"]
        a0 = E.Act(B, 1, T, E.pad_to(600, 16), x3, dev)
        a1 = E.Act(B, 1, T, E.pad_to(600, 16), x3, dev)
        E.conv_to_act(h, 0, f0["cin_store"], f0["w"], 1, 1, 600, f0["scale"], f0["shift"], L.ACT_RELU, a0,
                      cout_store=a0.cs, Ho=1, Wo=T)
        E.conv_to_act(a0, 0, f2["cin_store"], f2["w"], 1, 1, 600, f2["scale"], f2["shift"], L.ACT_RELU, a1,
                      cout_store=a1.cs, Ho=1, Wo=T)
        out = torch.empty((B, 2, F, T), dtype=torch.float32, device=dev)
        E.conv(a1, 0, f4["cin_store"], f4["w"], 1, 1, 2 * F, f4["scale"], f4["shift"], L.ACT_SIGMOID, out=out,
               out_dtype=L.DT_F32, sb=2 * F * T, sh=0, sw=1, sc=T, Ho=1, Wo=T)
        return out, dict(tx=tx, tn=tn, tl=tl, h=h, a0=a0, a1=a1, out=out, dims=(B, F, T), n=n, n_extra=n_extra)

    def backward(self, plan, tape, g_out, grads, x3, prefix="stage2", side=None, tail=None, before_join=None):
        """g_out: f32 (B,2,F,T) gradient of the mask.  Returns the Act gradient of encoder_n's
        2-channel input (the stage-1 prediction).
        side: optional HIP stream for the branch [encoder_n backward -> tail(d_n)] (tail = the rest of the chain that hangs off
        encoder_n's input gradient: JointModel passes stage 1's backward); encoder_x's backward runs beside it on the current
        stream, which then calls before_join() and waits for the side stream before returning."""
        B, F, T = tape["dims"]
        dev = g_out.device
        nseg = 3 if x3 else 1
        dz4 = E.Act(B, 1, T, 2 * F, x3, dev)
        TO.pack_grad(g_out.contiguous().float(), tape["out"], L.ACT_SIGMOID, B, T, 2 * F, 2 * F * T, 1, T, dz4)
        d_a1 = TO.linear_backward(plan["fc4"], tape["a1"], dz4, grads, f"{prefix}.fc.4", x3, dev)
        dz2 = E.Act(B, 1, T, tape["a1"].cs, x3, dev, zero=True)
        TO.act_bwd_from_y(d_a1, tape["a1"], L.ACT_RELU, dz2, 600)
        d_a0 = TO.linear_backward(plan["fc2"], tape["a0"], dz2, grads, f"{prefix}.fc.2", x3, dev)
        dz0 = E.Act(B, 1, T, tape["a0"].cs, x3, dev, zero=True)
        TO.act_bwd_from_y(d_a0, tape["a0"], L.ACT_RELU, dz0, 600)
        dh = TO.linear_backward(plan["fc0"], tape["h"], dz0, grads, f"{prefix}.fc.0", x3, dev)
        dfeat = TO.lstm_backward(plan["lstm"], tape["tl"], dh, grads, f"{prefix}.lstm", B, T, x3, dev)
        nfeat = 12 * F
        if side is not None:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)                     # dfeat exists
            with torch.cuda.stream(side):
                dyn = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 8, 4, B, F, T, T, x3)
                d_n = TO.encoder_backward(plan["enc_n"], tape["tn"], dyn, grads, f"{prefix}.encoder_n", x3, need_input_grad=True)
                if tail is not None:
                    tail(d_n)
            dyx = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 0, 8, B, F, T, T, x3)
            TO.encoder_backward(plan["enc_x"], tape["tx"], dyx, grads, f"{prefix}.encoder_x", x3)
            if before_join is not None:
                before_join()
            # `dfeat` (allocated on the current stream, read on the side stream) stays referenced until here: the caching
            # allocator may hand its memory out again only after the join below has been enqueued
            cur.wait_stream(side)
            return d_n
        dyx = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 0, 8, B, F, T, T, x3)
        TO.encoder_backward(plan["enc_x"], tape["tx"], dyx, grads, f"{prefix}.encoder_x", x3)
        dyn = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 8, 4, B, F, T, T, x3)
        d_n = TO.encoder_backward(plan["enc_n"], tape["tn"], dyn, grads, f"{prefix}.encoder_n", x3, need_input_grad=True)
        if tail is not None:
            tail(d_n)
        return d_n

    def forward(self, x, n):
        raise RuntimeError("call ContextAggNet through JointModel (libsos_hip path)")


class _JointTrainFn(torch.autograd.Function):
    """Training-mode forward of JointModel with the hand-written HIP backward (both outputs)."""

    @staticmethod
    def forward(ctx, net, x, n, *params):
        (n_pred, out), tape = net._forward_train(x, n)
        ctx.net, ctx.tape = net, tape
        return n_pred, out

    @staticmethod
    def backward(ctx, g_npred, g_out):
        with E.deferred_reductions():         # weight gradients reduce on a side stream; joined when the scope closes
            grads = ctx.net._backward(ctx.tape, g_npred, g_out)
        ctx.tape = None
        if getattr(ctx.net, "grad_sink_factory", None) is not None:     # see detector/networks.py: the bucketer owns p.grad
            return (None, None, None) + (None,) * len(list(ctx.net.parameters()))
        return (None, None, None) + tuple(grads[name].reshape(p.shape) for name, p in ctx.net.named_parameters())


class JointModel(nn.Module):
    """M2/networks.py:208-217: n_pred = stage1(n, x); out = stage2(x, n_pred); returns both."""

    ANNOUNCES_STAGE2_BACKWARD = True     # _backward_scaled calls self.after_stage2_backward() (agent.train_concurrent's gate)
    ANNOUNCES_LSTM_FORWARD = True        # _forward_train calls self.before_lstm_forward() once both encoders are enqueued

    def __init__(self, config):
        super().__init__()
        self.stage1 = InpaintNet()
        self.stage2 = ContextAggNet(config.kernel_sizes, config.dilations)
        self._cache = E.PlanCache()
        self._tcache = E.PlanCache(record=True)

    def _build_plan(self):
        x3 = E.is_x3()
        return dict(x3=x3, s1=self.stage1.build_plan(x3), s2=self.stage2.build_plan(x3))

    def _build_train_plan(self):
        x3 = E.is_x3()
        return dict(x3=x3, s1=self.stage1.build_train_plan(x3), s2=self.stage2.build_train_plan(x3))

    # The branch [stage 1 -> encoder_n] (forward) / [encoder_n -> stage 1] (backward) runs on a side stream beside encoder_x
    # (HBM-bound BatchNorm passes and the low-efficiency U-Net launches of one branch under the MFMA-bound 96-channel
    # convolutions of the other).  Same kernels, same order per branch: bit-identical results.  Default since round 5 (the
    # driver's round-4 record: 552.0 vs ~530 utt/s, +4.2 %); SOS_BRANCH_STREAMS=0 is the one-stream schedule (A/B, and
    # bench.py's serial pre-pass that times the dominant kernel alone on the chip).
    BRANCH_STREAMS = __import__("os").environ.get("SOS_BRANCH_STREAMS", "1") != "0"

    _SIDE_STREAMS = {}      # (device, calling stream) -> its branch stream: one per calling stream for the life of the process, shared
                            # by every JointModel (HIP has 4 hardware queues: streams are not free, see agent._job_stream)

    def _side_stream(self, dev):
        if not self.BRANCH_STREAMS or torch.cuda.is_current_stream_capturing():
            return None
        cur = torch.cuda.current_stream(dev)
        key = (dev.index, cur.cuda_stream)
        ss = JointModel._SIDE_STREAMS
        if key not in ss:
            ss[key] = torch.cuda.Stream(device=dev)
        self.__dict__["_used_side_stream"] = True
        return ss[key]

    def _forward_train(self, x, n):
        plan = self._tcache.get(self, self._build_train_plan)
        x3 = plan["x3"]
        out, t2 = self.stage2.forward_train(plan["s2"], x, lambda: self.stage1.forward_train(plan["s1"], n, x, x3), x3,
                                            before_lstm=getattr(self, "before_lstm_forward", None), side=self._side_stream(x.device))
        n_pred, t1 = t2.pop("n"), t2.pop("n_extra")
        return (n_pred, out), dict(plan=plan, t1=t1, t2=t2, x3=x3, mode=get_precision())

    def _backward(self, tape, g_npred, g_out):
        g_npred = g_npred.contiguous().float() if g_npred is not None else None
        g_out = g_out.contiguous().float() if g_out is not None else None
        E.check_tape_weights(self, tape)
        # the pass runs in the mode its forward ran in; fp16 mode: one loss scale for both entering gradients
        with precision_scope(tape.get("mode")), E.backward_scale(g_npred, g_out, guard=E.guard_state(self)):
            return self._backward_scaled(tape, g_npred, g_out)

    def _backward_scaled(self, tape, g_npred, g_out):
        plan, x3 = tape["plan"], tape["x3"]
        factory = getattr(self, "grad_sink_factory", None)
        grads = factory() if factory is not None else {}
        dev = tape["t2"]["out"].device
        B, F, T = tape["t2"]["dims"]
        if g_out is None:
            g_out = torch.zeros((B, 2, F, T), dtype=torch.float32, device=dev)
        def stage1_tail(d_np):                                                          # d_np: Act [B,F,T,16]
            if g_npred is not None:
                direct = E.pack_input(g_npred, x3, mul=E.cur_gs().mul)                     # loss gradient on n_pred
                TO.reflect_fold(direct, F, T, 0, d_np, 0, 2, accumulate=True)              # pad 0: plain add
            self.stage1.backward(plan["s1"], tape["t1"], d_np, grads, x3)

        side = self._side_stream(dev)
        hook = getattr(self, "after_stage2_backward", None)
        if side is None:
            d_np = self.stage2.backward(plan["s2"], tape["t2"], g_out, grads, x3)
            if hook is not None:       # agent.train_concurrent: another model's step may start here (see there)
                hook()
            stage1_tail(d_np)
        else:
            # branch streams: stage 1's backward hangs off encoder_n's on the side stream; the gate of the other model's
            # backward is recorded behind encoder_x's backward on this stream, as in the one-stream schedule
            self.stage2.backward(plan["s2"], tape["t2"], g_out, grads, x3, side=side, tail=stage1_tail, before_join=hook)
        return grads

    @torch.no_grad()
    def begin_x(self, x, rag=None):
        """Eval only: start encoder_x(x) -- the part of forward(x, n) that does not need n -- on this model's branch stream and
        return a handle for forward(x, n, started=handle).  The pipeline calls it between the two passes of the `mixed` detector
        so that the parity pass over a handful of marked clips (a batch too small to fill the chip) runs under encoder_x's
        chip-filling convolutions.  Returns None when it cannot (training mode, inside a stream capture, branch streams off)."""
        if self.training or x.dim() != 4:
            return None
        side = self._side_stream(x.device)
        if side is None:
            return None
        plan = self._cache.get(self, self._build_plan)
        x = x.contiguous().float()
        return self.stage2.start_x(plan["s2"], x, plan["x3"], rag, side)

    def forward(self, x, n, rag=None, started=None):
        """rag (eval only): engine.Ragged with the clips' own frame counts of a variable-length batch; x, n are then
        (B,2,F,max T) and only the first rag.T[b] columns of clip b's outputs are valid -- each clip computed exactly as
        if it were run alone at its own length (M2/predict.py:405-412 runs one file at a time)."""
        L.require_cuda(x, n)
        if x.dim() != 4 or x.shape[1] != 2 or x.shape != n.shape:
            raise ValueError(f"expected two (B, 2, F, T) inputs, got {tuple(x.shape)} and {tuple(n.shape)}")
        if rag is not None and (self.training or len(rag.T) != x.shape[0] or max(rag.T) != x.shape[3]):
            raise ValueError("ragged batches are an inference feature; rag must describe this batch")
        if self.training:
            return _JointTrainFn.apply(self, x.contiguous().float(), n.contiguous().float(), *self.parameters())
        plan = self._cache.get(self, self._build_plan)
        x3 = plan["x3"]
        x = x.contiguous().float()
        n = n.contiguous().float()
        if started is not None and (started["x"].data_ptr() != x.data_ptr() or started["x3"] != x3 or
                                    started.get("mode", get_precision()) != get_precision()):
            # (the storage type matters too: a feature matrix written by the bf16 library must not be read by the fp16 one)
            raise ValueError("forward(started=): the handle belongs to another input or precision mode")
        n_pred = self.stage1.run(plan["s1"], n, x, x3, rag)
        out = self.stage2.run(plan["s2"], x, n_pred, x3, rag, started=started)
        return n_pred, out
