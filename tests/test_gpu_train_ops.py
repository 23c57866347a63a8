"""Training-mode HIP kernels (BatchNorm statistics/apply, weight gradient) vs torch-CPU fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import hashed, rel_err

pytestmark = pytest.mark.gpu


def _act_from_nchw(x, x3, cs=None):
    """f32 NCHW (cpu) -> engine.Act on the GPU (+ the values it actually holds, as f32 NCHW)."""
    from sos_amd import engine as E
    B, Cc, H, W = x.shape
    cs = E.pad_to(Cc, 16) if cs is None else cs
    a = E.Act(B, H, W, cs, x3, torch.device("cuda"), zero=True)
    st = E.act_dtype()                  # the 16-bit storage type of the current precision mode (bf16 / IEEE half)
    xh = x.to(st)
    nhwc = torch.zeros(B, H, W, cs, dtype=st)
    nhwc[..., :Cc] = xh.permute(0, 2, 3, 1)
    held = xh.float()
    if x3:
        lo = (x - xh.float()).to(st)
        nl = torch.zeros_like(nhwc)
        nl[..., :Cc] = lo.permute(0, 2, 3, 1)
        a.t.copy_(torch.cat([nhwc, nhwc, nl], dim=3).cuda())
        held = held + lo.float()
    else:
        a.t.copy_(nhwc.cuda())
    return a, held


def _act_to_nchw(a, Cc):
    t = a.t.float().cpu()
    v = t[..., :Cc]
    if a.x3:
        v = v + t[..., 2 * a.cs:2 * a.cs + Cc]
    return v.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(params=["bf16", "bf16x3", "fp16"])
def mode_x3(request):
    """Every storage mode of the kernels: the two builds of the library (bfloat16 / IEEE half -- the timed mode) and the
    three-pass split.  Yields x3 (bool)."""
    import sos_amd
    sos_amd.set_precision(request.param)
    try:
        yield request.param == "bf16x3"
    finally:
        sos_amd.set_precision("bf16")


def test_bn_train_stats_and_apply(mode_x3):
    from sos_amd import engine as E, _lib as L
    x3 = mode_x3
    B, Cc, H, W = 3, 48, 20, 13
    x = torch.from_numpy(hashed(31, (B, Cc, H, W), 2.0).astype(np.float32)) + 0.3
    raw, held = _act_from_nchw(x, x3)
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(1 + 0.2 * hashed(32, (Cc,)).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(0.1 * hashed(33, (Cc,)).astype(np.float32)))
    ref = torch.nn.BatchNorm2d(Cc)
    ref.load_state_dict(bn.state_dict())
    bn = bn.cuda().train()
    dst = E.Act(B, H, W, 48, x3, torch.device("cuda"), zero=True)
    saved = E.bn_train(raw, 0, Cc, bn, L.ACT_RELU, None, dst)
    ref.train()
    want = torch.relu(ref(held))
    tol = 2e-5 if x3 else 1e-2
    assert rel_err(_act_to_nchw(dst, Cc), want) < tol
    assert rel_err(bn.running_mean.cpu(), ref.running_mean) < 1e-5
    assert rel_err(bn.running_var.cpu(), ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    assert rel_err(saved["mean"].cpu(), held.mean(dim=(0, 2, 3))) < 1e-5
    assert rel_err(saved["invstd"].cpu(), torch.rsqrt(held.var(dim=(0, 2, 3), unbiased=False) + 1e-5)) < 1e-5


CASES = [
    # name, Cout(M), Cin(N), k, stride, dil, pad, pad_mode, H, W
    ("5x5 dil(2,1) zero", 48, 96, (5, 5), 1, (2, 1), (4, 2), "zeros", 20, 40),
    ("5x5 dil(8,8) zero", 96, 48, (5, 5), 1, (8, 8), (16, 16), "zeros", 33, 35),
    ("7x1 zero", 48, 48, (7, 1), 1, (1, 1), (3, 0), "zeros", 18, 21),
    ("3x3 s2 reflect", 64, 128, (3, 3), 2, (1, 1), (1, 1), "reflect", 17, 23),
    ("5x5 s2 reflect", 128, 64, (5, 5), 2, (1, 1), (2, 2), "reflect", 18, 41),
    ("1x1 linear", 100, 200, (1, 1), 1, (1, 1), (0, 0), "zeros", 1, 77),
    ("3x3 narrow", 2, 64, (3, 3), 1, (1, 1), (1, 1), "reflect", 12, 19),
    # 49 taps: more than one workgroup's 32 (tap, n-tile) pairs -> tap rows divided over the workgroups of one launch
    ("7x7 s2 16ch", 128, 16, (7, 7), 2, (1, 1), (3, 3), "zeros", 30, 44),
    ("7x7 s1 48x32", 48, 32, (7, 7), 1, (1, 1), (3, 3), "zeros", 17, 23),
    # GEMM path of the 1x1 gradients (M, N >= 128): ragged M / N tiles, pixel counts that are not multiples of the
    # 64-pixel stage, several splits
    ("1x1 gemm 160x256", 160, 256, (1, 1), 1, (1, 1), (0, 0), "zeros", 1, 77),
    ("1x1 gemm 288x416 long", 288, 416, (1, 1), 1, (1, 1), (0, 0), "zeros", 7, 331),
    # 16x16x32 kernel (M, N in (32, 48], 25 or 9 taps): dense, dilated with residue classes, reflect border
    ("16: 5x5 48x48", 48, 48, (5, 5), 1, (1, 1), (2, 2), "zeros", 37, 50),
    ("16: 5x5 dil(4,4) 48x40", 48, 40, (5, 5), 1, (4, 4), (8, 8), "zeros", 30, 41),
    ("16: 3x3 reflect 40x48", 40, 48, (3, 3), 1, (1, 1), (1, 1), "reflect", 21, 35),
    # round 3: k-steps without a pixel inside the image are skipped; shapes whose cheapest tile walks its pixels column-major
    # (few strided columns per residue class) and whose border tiles lose whole k-steps
    ("5x5 dil(16,16) 96x96", 96, 96, (5, 5), 1, (16, 16), (32, 32), "zeros", 64, 45),
    ("5x5 dil(16,1) 96x64", 96, 64, (5, 5), 1, (16, 1), (32, 2), "zeros", 40, 51),
    ("16: 5x5 dil(16,16) 48x48", 48, 48, (5, 5), 1, (16, 16), (32, 32), "zeros", 64, 45),
    ("16: 5x5 dil(32,32) 48x48", 48, 48, (5, 5), 1, (32, 32), (64, 64), "zeros", 70, 81),
    # round 5: the streaming kernel of the thin 1x1 gradients (one side <= 16 channels): pixel counts that are not multiples of
    # the 32-pixel stage, waves with an empty run, several workgroups
    ("thin 1x1 48x14", 48, 14, (1, 1), 1, (1, 1), (0, 0), "zeros", 37, 50),
    ("thin 1x1 96x14 long", 96, 14, (1, 1), 1, (1, 1), (0, 0), "zeros", 64, 201),
    ("thin 1x1 8x96", 8, 96, (1, 1), 1, (1, 1), (0, 0), "zeros", 33, 47),
    ("thin 1x1 4x48 short", 4, 48, (1, 1), 1, (1, 1), (0, 0), "zeros", 1, 20),
    ("thin 1x1 64x10", 64, 10, (1, 1), 1, (1, 1), (0, 0), "zeros", 19, 23),
    # ... and of the thin input under 5 vertical taps (the U-Net's folded first block): reflection and zero padding, dilation
    ("thin 5x1 64x10 reflect", 64, 10, (5, 1), 1, (1, 1), (2, 0), "reflect", 37, 50),
    ("thin 5x1 56x10 zero", 56, 10, (5, 1), 1, (1, 1), (2, 0), "zeros", 20, 33),
    ("thin 5x1 64x16 dil2", 64, 16, (5, 1), 1, (2, 1), (4, 0), "zeros", 23, 40),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_weight_grad(case, mode_x3):
    from sos_amd import engine as E, _lib as L
    x3 = mode_x3
    _, M, N, k, stride, dil, pad, pmode, H, W = case
    B = 2
    x = torch.from_numpy(hashed(41, (B, N, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, x3)
    xp = F.pad(xheld, (pad[1], pad[1], pad[0], pad[0]), mode="reflect") if pmode == "reflect" else xheld
    w = torch.zeros(M, N, k[0], k[1], requires_grad=True)
    y = F.conv2d(xp, w, None, stride, (0, 0) if pmode == "reflect" else pad, dil)
    g = torch.from_numpy(hashed(42, tuple(y.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, x3)
    y.backward(gheld)
    dw = torch.empty(M, N, k[0], k[1], dtype=torch.float32, device="cuda")
    E.wgrad(ga, 0, M, xa, 0, N, k[0], k[1], dw, stride=stride, dil=dil, pad=pad,
            pad_mode=L.PAD_REFLECT if pmode == "reflect" else L.PAD_ZERO)
    err = rel_err(dw.cpu(), w.grad)
    print(case[0], "x3" if x3 else "bf16", "wgrad rel err", err)
    assert err < (2e-4 if x3 else 2e-5 + 1e-3)


@pytest.mark.parametrize("tile", ["1,4,4,0", "1,4,4,1", "1,6,2,1", "1,2,6,0", "2,3,4,1", "4,4,2,1", "4,4,2,0"])
@pytest.mark.parametrize("ch", [96, 48])
def test_conv_weight_grad_forced_tiles(tile, ch, monkeypatch):
    """Every (residue classes, tile height, tile width, pixel order) of the weight-gradient kernels gives the same gradient:
    the tile and the order of its pixels along the contraction (row- or column-major, which decides the border k-steps that
    are skipped) are performance choices only.  SOS_WGRAD_TILE = "classes,log2 TH,log2 TW,order" forces one."""
    from sos_amd import engine as E
    monkeypatch.setenv("SOS_WGRAD_TILE", tile)
    B, H, W, k, dil = 2, 37, 45, (5, 5), (4, 4)
    pad = (8, 8)
    x = torch.from_numpy(hashed(51, (B, ch, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, False)
    w = torch.zeros(ch, ch, 5, 5, requires_grad=True)
    y = F.conv2d(xheld, w, None, 1, pad, dil)
    g = torch.from_numpy(hashed(52, tuple(y.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, False)
    y.backward(gheld)
    dw = torch.empty(ch, ch, 5, 5, dtype=torch.float32, device="cuda")
    E.wgrad(ga, 0, ch, xa, 0, ch, 5, 5, dw, dil=dil, pad=pad)
    err = rel_err(dw.cpu(), w.grad)
    print(tile, ch, "wgrad rel err", err)
    assert err < 2e-5 + 1e-3


def test_wgrad_tune_measured_plan_gives_the_same_gradient(tmp_path):
    """sos_wgrad_tune (ABI 7): timing the candidate plans of a shape and adopting the winner changes the launch, not the gradient
    -- the tuned launch matches torch autograd like the cost model's launch did; the table survives a save / load round trip and
    the second tune of the shape is a no-op (-1)."""
    import ctypes
    from sos_amd import engine as E, _lib as L
    B, M, N, H, W, k, dil = 2, 64, 96, 40, 37, (3, 3), (2, 2)
    pad = (2, 2)
    x = torch.from_numpy(hashed(61, (B, N, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, False)
    w = torch.zeros(M, N, 3, 3, requires_grad=True)
    y = F.conv2d(xheld, w, None, 1, pad, dil)
    g = torch.from_numpy(hashed(62, tuple(y.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, False)
    y.backward(gheld)
    dw0 = torch.empty(M, N, 3, 3, dtype=torch.float32, device="cuda")
    E.wgrad(ga, 0, M, xa, 0, N, 3, 3, dw0, dil=dil, pad=pad)
    e0 = rel_err(dw0.cpu(), w.grad)
    d = L.WgradDesc()
    d.g, d.B, d.Hg, d.Wg, d.g_cs, d.g_off = ga.t.data_ptr(), B, H, W, ga.nseg * ga.cs, 0
    d.x, d.Hx, d.Wx, d.x_cs, d.x_off = xa.t.data_ptr(), H, W, xa.nseg * xa.cs, 0
    d.M, d.N, d.kh, d.kw, d.stride, d.dil_h, d.dil_w = M, N, 3, 3, 1, 2, 2
    d.pad_top, d.pad_left, d.pad_mode, d.ksplit, d.scale = 2, 2, L.PAD_ZERO, 0, 1.0
    ws = torch.empty(L.lib().sos_wgrad_workspace_bytes(ctypes.byref(d)) // 4 + 1, dtype=torch.float32, device="cuda")
    scratch = torch.empty_like(dw0)
    d.partial, d.dw, d.accumulate = ws.data_ptr(), scratch.data_ptr(), 0
    best = ctypes.c_float(0.0)
    L.check(L.lib().sos_wgrad_tune(ctypes.byref(d), 2, ctypes.byref(best), L.stream_ptr()), "sos_wgrad_tune")
    assert best.value > 0.0
    L.check(L.lib().sos_wgrad_tune(ctypes.byref(d), 2, ctypes.byref(best), L.stream_ptr()), "sos_wgrad_tune")
    assert best.value == -1.0
    dw1 = torch.empty_like(dw0)
    E.wgrad(ga, 0, M, xa, 0, N, 3, 3, dw1, dil=dil, pad=pad)
    e1 = rel_err(dw1.cpu(), w.grad)
    print("wgrad vs autograd: model plan", e0, "measured plan", e1)
    assert e0 < 2e-5 + 1e-3 and e1 < 2e-5 + 1e-3
    path = str(tmp_path / "wg.txt").encode()
    assert L.lib().sos_wgrad_tune_save(path) == 0
    body = open(path).read().splitlines()
    assert body[0] == "sos_wgrad_tune 1 nkey 12" and any(ln.startswith(f"{H} {W} 3 3 1 2 2 {M} {N} ") for ln in body[1:])
    assert L.lib().sos_wgrad_tune_load(path) == len(body) - 1


@pytest.mark.parametrize("geom", [(20, 27, 2, 64), (40, 37, 16, 64), (9, 11, 4, 24), (5, 6, 3, 16), (4, 7, 3, 16), (33, 18, 1, 128)],
                         ids=lambda g: "%dx%d pad%d c%d" % g)
def test_reflect_fold_border_adds_the_mirrored_cells(geom):
    """sos_reflect_fold_border (the border kernel of round 4 enumerates only the pixels that have a mirrored cell; pad > min(H, W) - 2
    keeps the all-pixel walk): out += every border cell of the padded tensor that mirrors onto the pixel, interior cells untouched
    -- against the adjoint of torch's ReflectionPad2d.  Bands that meet (9 x 11, pad 4), pad = H - 2 and the fallback are covered."""
    from sos_amd import train_ops as TO
    H, W, pad, C = geom
    B = 3
    P = torch.from_numpy(hashed(71, (B, C, H + 2 * pad, W + 2 * pad)).astype(np.float32))
    O = torch.from_numpy(hashed(72, (B, C, H, W)).astype(np.float32))
    pa, pheld = _act_from_nchw(P, False)
    oa, oheld = _act_from_nchw(O, False)
    x = torch.zeros(B, C, H, W, requires_grad=True)
    F.pad(x, (pad, pad, pad, pad), mode="reflect").backward(pheld)
    want = oheld + x.grad - pheld[:, :, pad:pad + H, pad:pad + W]
    TO.reflect_fold_border(pa, H, W, pad, oa, 0, C)
    got = _act_to_nchw(oa, C)
    # one rounding to the 16-bit storage type of a sum of <= 9 stored values
    err = float((got - want).abs().max() / want.abs().max())
    print(geom, "fold border rel err", err)
    assert err < 6e-3
    inner = (slice(None), slice(None), slice(pad + 1, H - 1 - pad), slice(pad + 1, W - 1 - pad))
    if H - 2 - 2 * pad > 0 and W - 2 - 2 * pad > 0:
        assert torch.equal(got[inner], oheld[inner]), "an interior pixel was rewritten"


def test_pack_grad_rows_and_strided():
    """sos_pack_grad_f32: f32 gradients -> 16-bit rows.  Channel-contiguous sources (the LSTM gate gradients) take the row kernel
    of round 4 (8 channels per thread), anything else the element-wise kernel; both must equal a plain cast, and the sigmoid'
    form (the denoiser's mask head, channel stride = T) the cast of g y (1 - y)."""
    from sos_amd import engine as E, train_ops as TO, _lib as L
    B, T, C = 3, 37, 1600
    g = torch.from_numpy(hashed(81, (B, T, C)).astype(np.float32)).cuda()
    dst = E.Act(B, 1, T, C, False, torch.device("cuda"))
    dst.t.fill_(float("nan"))
    TO.pack_grad(g, None, L.ACT_NONE, B, T, C, T * C, C, 1, dst)
    got = dst.t.view(B, T, C).float().cpu()
    assert torch.equal(got, g.cpu().to(E.act_dtype()).float())
    # channel stride T (B, C, T source), with the sigmoid derivative
    C2 = 48
    g2 = torch.from_numpy(hashed(82, (B, C2, T)).astype(np.float32)).cuda()
    y2 = torch.sigmoid(torch.from_numpy(hashed(83, (B, C2, T)).astype(np.float32))).cuda()
    dst2 = E.Act(B, 1, T, C2, False, torch.device("cuda"))
    TO.pack_grad(g2, y2, L.ACT_SIGMOID, B, T, C2, C2 * T, 1, T, dst2)
    want2 = (g2 * y2 * (1 - y2)).permute(0, 2, 1).cpu()
    got2 = dst2.t.view(B, T, C2).float().cpu()
    assert float((got2 - want2).abs().max()) <= 2e-3 * float(want2.abs().max())


DOWN_CASES = [(64, 128, 5, 2, 1, 20, 27), (128, 128, 3, 1, 4, 16, 23), (64, 64, 5, 1, 1, 18, 21), (128, 64, 3, 2, 1, 17, 22),
              (64, 64, 3, 1, 16, 40, 37)]


@pytest.mark.parametrize("accumulate", [False, True], ids=["store", "accumulate"])
@pytest.mark.parametrize("case", DOWN_CASES, ids=[f"{c[0]}-{c[1]} k{c[2]} s{c[3]} d{c[4]}" for c in DOWN_CASES])
def test_down_block_input_gradient(case, accumulate, mode_x3):
    """DownConvBlock (ReflectionPad2d + strided / dilated conv + BN + PReLU, M2/networks.py:97-117) backward: the gradient of the
    block INPUT against torch autograd.  Its reflection-pad fold is fused into the data-gradient convolution (interior cells of the
    padded domain go straight into grad(src), stored or accumulated; only the border cells pass through the padded scratch tensor
    and sos_reflect_fold_border): stride 1 with dilation, the four phase convolutions of stride 2, a pad wider than a tile."""
    from sos_amd import engine as E, train_ops as TO
    from sos_amd.denoiser.networks import DownConvBlock
    x3 = mode_x3
    cin, cout, k, s, d, H, W = case
    torch.manual_seed(1)
    blk = DownConvBlock(cin, cout, k, s, dilation=d)
    ref = DownConvBlock(cin, cout, k, s, dilation=d)
    ref.load_state_dict(blk.state_dict())
    blk = blk.cuda().train()
    x = torch.from_numpy(hashed(3, (2, cin, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, x3)
    lp = TO.down_train_plan(blk, x3)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    dst = E.Act(2, Ho, Wo, cout, x3, torch.device("cuda"))
    t = TO.down_forward_train(lp, xa, 0, dst, 0, Ho, Wo, x3)
    xr = xheld.clone().requires_grad_(True)
    yr = ref.block(xr)
    g = torch.from_numpy(hashed(4, tuple(yr.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, x3)
    yr.backward(gheld)
    gb = TO.GradBufs(x3)
    gb.bufs[id(dst)] = ga
    gb.written[id(dst)] = [(0, cout)]
    expect = xr.grad
    if accumulate:                  # another consumer of the block input has already written its gradient
        prior = torch.from_numpy(hashed(5, (2, cin, H, W)).astype(np.float32))
        pa, pheld = _act_from_nchw(prior, x3)
        gb.bufs[id(xa)] = pa
        gb.written[id(xa)] = [(0, cin)]
        expect = expect + pheld
    grads = {}
    TO.down_backward(t, gb, grads, "b", x3)
    err = rel_err(_act_to_nchw(gb.of(xa), cin), expect)
    err_w = rel_err(grads["b.block.1.weight"], ref.block[1].weight.grad)
    print(case, "accumulate" if accumulate else "store", "x3" if x3 else "16-bit", "d_in rel err", err, "dw rel err", err_w)
    if x3:      # the three-pass mode is the logic check: a wrong fold is an O(1) error on the border pixels
        assert err < 3e-4 and err_w < 2e-4
        return
    # 16-bit modes: the bound is COMPUTED.  The same block in f32 with nothing but round trips through the storage type where
    # the kernels store (weights, raw conv output, block output; their gradients d_y, d_raw, d_in -- tests/storage_model.py)
    # deviates from f32 autograd by m / m_w: the kernels may add nothing beyond the format -- within 2x of the model's own
    # deviation (a 5 % scaling bug in the fold or the BatchNorm backward is 10-50x that)
    from oracle import nets as onet
    from storage_model import q, storage_model
    import sos_amd
    dt = torch.float16 if sos_amd.get_precision() == "fp16" else torch.bfloat16
    sdm = {"b.block.1.weight": ref.block[1].weight.detach().clone().requires_grad_(True),
           "b.block.2.weight": ref.block[2].weight.detach().clone(), "b.block.2.bias": ref.block[2].bias.detach().clone(),
           "b.block.3.weight": ref.block[3].weight.detach().clone()}
    with storage_model(dt):
        xm = xheld.clone().requires_grad_(True)
        onet.down_block(xm, sdm, "b", k, s, d, True).backward(gheld)
        m_in = xm.grad if not accumulate else q(xm.grad + pheld)
    m, m_w = rel_err(m_in, expect), rel_err(sdm["b.block.1.weight"].grad, ref.block[1].weight.grad)
    print("   storage model: d_in", m, "dw", m_w, " HIP / model", err / m, err_w / m_w)
    assert err < 2.0 * m + 1e-4 and err_w < 2.0 * m_w + 1e-4


def test_loss_scale_is_exact_under_power_of_two_rescaling():
    """fp16 mode: the backward pass carries a power-of-two loss scale chosen on the device from max|g| of the entering gradient
    (sos_amax_f32 + sos_loss_scale), multiplied in where f32 gradients become 16-bit (sos_pack_nchw_to_nhwc `mul`) and divided out
    where parameter gradients leave (sos_wgrad_desc.scale_dev, sos_bn_bwd out_scale).  The pass is linear, so rescaling the entering
    gradient by 2^+-6 must change NOTHING but the scale: the stored (scaled) activation gradients are the same bits, S moves by
    exactly 2^-+6, and every parameter gradient is exactly 2^+-6 times the unscaled run's (VERDICT r3 #3c)."""
    import sos_amd
    from sos_amd import engine as E, train_ops as TO
    from sos_amd.denoiser.networks import DownConvBlock
    sos_amd.set_precision("fp16")
    try:
        torch.manual_seed(4)
        cin, cout, k, s, d, H, W = 64, 64, 3, 1, 2, 21, 26
        blk = DownConvBlock(cin, cout, k, s, dilation=d).cuda().train()
        x = torch.from_numpy(hashed(13, (2, cin, H, W)).astype(np.float32))
        xa, _ = _act_from_nchw(x, False)
        lp = TO.down_train_plan(blk, False)
        dst = E.Act(2, H, W, cout, False, torch.device("cuda"))
        t = TO.down_forward_train(lp, xa, 0, dst, 0, H, W, False)
        g = torch.from_numpy(hashed(14, (2, cout, H, W)).astype(np.float32)) * 3e-3       # (small: unscaled it would sit in half's subnormals)
        res = {}
        for c in (1.0, 2.0 ** -6, 2.0 ** 6):
            gc = (g * c).cuda()
            with E.backward_scale(gc) as gs:
                ga = E.pack_input(gc, False, mul=gs.mul)
                gb = TO.GradBufs(False)
                gb.bufs[id(dst)] = ga
                gb.written[id(dst)] = [(0, cout)]
                grads = {}
                TO.down_backward(t, gb, grads, "b", False)
                res[c] = (float(gs.mul), ga.t.clone(), gb.of(xa).t.clone(), {n: v.clone() for n, v in grads.items()})
        S1, ga1, din1, gr1 = res[1.0]
        assert S1 >= 2.0 ** 14                      # max|g| = 3e-3 -> S = 2^16: the scale is doing something
        for c in (2.0 ** -6, 2.0 ** 6):
            S, ga, din, gr = res[c]
            assert S == S1 / c, (S, S1, c)
            assert torch.equal(ga, ga1) and torch.equal(din, din1)          # the scaled 16-bit gradients: the same bits
            for n in gr1:
                assert torch.equal(gr[n], gr1[n] * c), n                    # parameter gradients: exactly c x
        # and the unscaled run agrees with f32 autograd to the format's accuracy (the loss scale is what keeps 3e-3 * 2^-8-sized
        # gradient elements out of half's subnormal range)
        ref = DownConvBlock(cin, cout, k, s, dilation=d)
        ref.load_state_dict({kk: v.cpu() for kk, v in blk.state_dict().items()})
        xr = x.half().float().requires_grad_(True)
        ref.train()
        ref.block(xr).backward(g)
        assert rel_err(gr1["b.block.1.weight"], ref.block[1].weight.grad) < 2e-2
    finally:
        sos_amd.set_precision("bf16")


@pytest.mark.parametrize("reflect", [False, True], ids=["zero", "reflect"])
@pytest.mark.parametrize("ragged", [False, True], ids=["dense", "ragged"])
def test_pack_wtaps_matches_a_shifted_gather(mode_x3, reflect, ragged):
    """sos_pack_nchw_wtaps (horizontal taps of the 2-channel first layers on the channel axis): stored channel t*C + c of pixel
    (h, w) is x[c][h][w + t - pad] -- zero or mirrored outside the clip, at the clip's OWN end in a ragged batch -- rounded to the
    storage type exactly like the plain boundary pack (bit-identical values; the low parts in the three-pass mode)."""
    from sos_amd import engine as E, _lib as L
    x3 = mode_x3
    B, Cc, H, W, kw = 3, 2, 9, 21, 7 if not reflect else 5
    pad = (kw - 1) // 2
    x = torch.from_numpy(hashed(91, (B, Cc, H, W)).astype(np.float32))
    widths = [21, 13, 17] if ragged else [W] * B
    cw = torch.tensor(widths, dtype=torch.int32, device="cuda") if ragged else None
    a = E.pack_input(x.cuda(), x3, wtaps=(kw, pad, L.PAD_REFLECT if reflect else L.PAD_ZERO), clip_w=cw)
    assert a.cs == 16 and a.t.shape[-1] == (48 if x3 else 16)
    want = torch.zeros(B, H, W, 16)
    for b in range(B):
        Wc = widths[b]
        for t in range(kw):
            for w in range(Wc):
                ws = w + t - pad
                if reflect:
                    ws = -ws if ws < 0 else (2 * (Wc - 1) - ws if ws >= Wc else ws)
                if 0 <= ws < Wc:
                    want[b, :, w, t * Cc:(t + 1) * Cc] = x[b, :, :, ws].t()
    st = E.act_dtype()
    hi = want.to(st)
    got = a.t.cpu()
    assert torch.equal(got[..., :16], hi)
    if x3:
        assert torch.equal(got[..., 16:32], hi)
        assert torch.equal(got[..., 32:], (want - hi.float()).to(st))


def test_first_down_block_with_folded_taps(mode_x3):
    """DownConvBlock(2, 64, 5, 1) -- down1 / down3 of the U-Net (M2/networks.py:158,165) -- with its horizontal taps on the channel
    axis (train_ops.down_train_plan(first=True)): raw conv output, BatchNorm + PReLU output and the un-folded weight gradient
    against torch on the same (storage-rounded) input."""
    from sos_amd import engine as E, train_ops as TO
    from sos_amd.denoiser.networks import DownConvBlock
    x3 = mode_x3
    torch.manual_seed(2)
    blk = DownConvBlock(2, 64, 5, 1)
    ref = DownConvBlock(2, 64, 5, 1)
    ref.load_state_dict(blk.state_dict())
    blk = blk.cuda().train()
    B, H, W = 2, 19, 23
    x = torch.from_numpy(hashed(7, (B, 2, H, W)).astype(np.float32))
    lp = TO.down_train_plan(blk, x3, first=True)
    assert lp["kw"] == 1 and lp["cin"] == 10 and lp["wtaps"][0] == 5
    xa = E.pack_input(x.cuda(), x3, wtaps=lp["wtaps"])
    st = E.act_dtype()
    xheld = x.to(st).float() + ((x - x.to(st).float()).to(st).float() if x3 else 0)
    dst = E.Act(B, H, W, 64, x3, torch.device("cuda"))
    t = TO.down_forward_train(lp, xa, 0, dst, 0, H, W, x3)
    xr = xheld.clone().requires_grad_(True)
    yr = ref.block(xr)
    tol = 3e-5 if x3 else 2e-2
    assert rel_err(_act_to_nchw(dst, 64), yr.detach()) < tol
    g = torch.from_numpy(hashed(8, tuple(yr.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, x3)
    yr.backward(gheld)
    gb = TO.GradBufs(x3)
    gb.bufs[id(dst)] = ga
    gb.written[id(dst)] = [(0, 64)]
    grads = {}
    TO.down_backward(t, gb, grads, "b", x3, need_src_grad=False)
    assert grads["b.block.1.weight"].shape == ref.block[1].weight.shape
    assert rel_err(grads["b.block.1.weight"], ref.block[1].weight.grad) < (2e-4 if x3 else 1e-1)


def test_conv_transpose_weight_grad(mode_x3):
    """ConvTranspose2d(k3,s2,p1,output_padding=1): roles swap (G = layer input, X = output grad)."""
    from sos_amd import engine as E, _lib as L
    if mode_x3:
        pytest.skip("the three-pass split of this shape is covered by test_conv_weight_grad")
    B, Cin, Cout, H, W = 2, 64, 32, 9, 11
    x = torch.from_numpy(hashed(43, (B, Cin, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, False)
    w = torch.zeros(Cin, Cout, 3, 3, requires_grad=True)
    y = F.conv_transpose2d(xheld, w, None, 2, 1, 1)
    g = torch.from_numpy(hashed(44, tuple(y.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, False)
    y.backward(gheld)
    dw = torch.empty(Cin, Cout, 3, 3, dtype=torch.float32, device="cuda")
    E.wgrad(xa, 0, Cin, ga, 0, Cout, 3, 3, dw, stride=2, pad=(1, 1))
    assert rel_err(dw.cpu(), w.grad) < 1e-3


# ---------------------------------------------------------------------------- temporal taps (Conv3d) inside the kernels
def _frames_act(x5, x3):
    """f32 (B, C, T, H, W) -> Act of B*T frames [B*T, H, W, C] (+ the values it holds, same shape as x5)."""
    B, Cc, T, H, W = x5.shape
    a, held = _act_from_nchw(x5.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, H, W), x3, cs=Cc)
    return a, held.reshape(B, T, Cc, H, W).permute(0, 2, 1, 3, 4).contiguous()


@pytest.mark.parametrize("kt,stride", [(5, 1), (3, 2)])
def test_temporal_taps_conv_and_gradients_match_conv3d(kt, stride, mode_x3):
    """sos_conv_desc / sos_wgrad_desc temporal taps = nn.Conv3d(I, O, (kt,3,3), stride (1,s,s), padding ((kt-1)/2,1,1))
    (M1/networks.py:54-77) on clips of T frames: forward, weight gradient and data gradient against torch autograd, with
    clips short enough (T = 4) that most frames touch the temporal padding."""
    from sos_amd import engine as E, _lib as L, train_ops as TO
    x3 = mode_x3
    B, T, I, O, H, W = 2, 4, 128, 32, 9, 11
    x = torch.from_numpy(hashed(71, (B, I, T, H, W)).astype(np.float32))
    xa, xheld = _frames_act(x, x3)
    w = torch.from_numpy((0.05 * hashed(72, (O, I, kt, 3, 3))).astype(np.float32))
    xr = xheld.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, None, (1, stride, stride), ((kt - 1) // 2, 1, 1))
    Ho, Wo = y.shape[-2:]
    # ---- forward
    w2 = w.permute(0, 2, 1, 3, 4).reshape(O, kt * I, 3, 3).cuda()
    wp = E.pack_weight(w2, kt * I, x3)
    one, zero = TO.ones_zeros(wp.shape[1], torch.device("cuda"))
    dst = E.Act(B * T, Ho, Wo, 32, x3, torch.device("cuda"), zero=True)
    E.conv_to_act(xa, 0, I, wp, 3, 3, O, one, zero, L.ACT_NONE, dst, cout_store=32, stride=stride, pad=(1, 1), Ho=Ho, Wo=Wo,
                  temporal=(T, kt))
    got = _act_to_nchw(dst, O).reshape(B, T, O, Ho, Wo).permute(0, 2, 1, 3, 4)
    e = rel_err(got, y.detach())
    print("temporal conv", kt, stride, "x3" if x3 else "16-bit", e)
    assert e < (3e-5 if x3 else 1e-2)
    # ---- gradients
    g = torch.from_numpy(hashed(73, tuple(y.shape)).astype(np.float32))
    ga, gheld = _frames_act(g, x3)
    y.backward(gheld)
    dw2 = torch.empty(O, kt * I, 3, 3, dtype=torch.float32, device="cuda")
    E.wgrad(ga, 0, O, xa, 0, kt * I, 3, 3, dw2, stride=stride, pad=(1, 1), temporal=(T, kt, I))
    dw = dw2.reshape(O, kt, I, 3, 3).permute(0, 2, 1, 3, 4).cpu()
    e = rel_err(dw, wr.grad)
    print("temporal wgrad", e)
    assert e < (2e-4 if x3 else 2e-3)
    if stride == 1:         # data gradient: the same conv over dy with the temporal taps flipped (train_ops.video_train_plan)
        w2n = w.flip(2).permute(2, 0, 1, 3, 4).reshape(kt * O, I, 3, 3).cuda()
        wd = TO.dgrad_weight(w2n, x3)
        one, zero = TO.ones_zeros(wd.shape[1], torch.device("cuda"))
        dx = E.Act(B * T, H, W, I, x3, torch.device("cuda"), zero=True)
        E.conv_to_act(ga, 0, ga.cs, wd, 3, 3, I, one, zero, L.ACT_NONE, dx, cout_store=I, pad=(1, 1), Ho=H, Wo=W, temporal=(T, kt))
        gotx = _act_to_nchw(dx, I).reshape(B, T, I, H, W).permute(0, 2, 1, 3, 4)
        e = rel_err(gotx, xr.grad)
        print("temporal dgrad", e)
        assert e < (3e-5 if x3 else 1e-2)


@pytest.mark.parametrize("T,cin,cout", [(178, 3072, 1600), (60, 2048, 800), (178, 400, 600)])
def test_flattened_1x1_layers_are_bit_identical(T, cin, cout, monkeypatch):
    """1x1 layers over single-row images (LSTM input projections, FC head) run over the batch as ONE pixel row (engine.conv):
    a tiling choice only -- every output pixel's contraction order is unchanged, so the outputs are bit-identical to the
    per-clip launch (SOS_FLATTEN_1X1=0)."""
    from sos_amd import engine as E, _lib as L
    sos_amd = pytest.importorskip("sos_amd")
    sos_amd.set_precision("fp16")
    try:
        dev = torch.device("cuda")
        B = 5
        torch.manual_seed(3)
        w = E.pack_weight(lambda: torch.randn(cout, cin, 1, 1, device=dev) * 0.05, E.pad_to(cin, 16), False)
        src = E.Act(B, 1, T, E.pad_to(cin, 16), False, dev)
        src.t.normal_()
        scale = E.pad_vec(torch.ones(cout, device=dev), w.shape[1], 1.0)
        shift = E.pad_vec(torch.zeros(cout, device=dev), w.shape[1])
        outs = []
        for flat in (True, False):
            monkeypatch.setattr(E, "FLATTEN_1X1", flat)
            dst = E.Act(B, 1, T, E.pad_to(cout, 16), False, dev, zero=True)
            E.conv_to_act(src, 0, E.pad_to(cin, 16), w, 1, 1, cout, scale, shift, L.ACT_RELU, dst, cout_store=dst.cs, Ho=1, Wo=T)
            outs.append(dst.t.clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        assert float(outs[0].float().abs().max()) > 0
    finally:
        sos_amd.set_precision("bf16")


@pytest.mark.parametrize("dil", [(1, 1), (4, 1), (16, 16)], ids=["d1", "d4x1", "d16"])
def test_fused_input_batchnorm_equals_the_materialised_apply(dil):
    """sos_conv_desc.in_scale / in_shift (round 5): the consumer conv of a training-mode Conv2dBlock reads the producer's RAW conv
    output and applies the producer's BatchNorm + ReLU while it stages the patch.  The staged values are computed with
    sos_bn_act_apply's arithmetic and rounding, so the result must equal -- bit for bit -- the same conv over the tensor
    sos_bn_act_apply materialises, zero padding included (a padded pixel must stay 0, not relu(shift))."""
    import sos_amd
    from sos_amd import engine as E, _lib as L
    sos_amd.set_precision("fp16")
    try:
        dev = torch.device("cuda")
        B, H, W, Cc = 2, 40, 37, 96
        torch.manual_seed(5)
        raw = E.Act(B, H, W, Cc, False, dev)
        raw.t.normal_()
        scale = (torch.rand(Cc, device=dev) + 0.5).float()
        shift = (torch.randn(Cc, device=dev) * 0.3 + 0.2).float()        # mostly positive: relu(shift) != 0 would show at the borders
        y = E.Act(B, H, W, Cc, False, dev)
        E.bn_apply(E.view(raw, 0, Cc), scale, shift, L.ACT_RELU, None, y, 0, Cc)
        w = E.pack_weight(torch.randn(Cc, Cc, 5, 5, device=dev) * 0.05, Cc, False)
        pad = (2 * dil[0], 2 * dil[1])
        outs = []
        for src, in_bn in ((y, None), (raw, (scale, shift))):
            dst = E.Act(B, H, W, Cc, False, dev, zero=True)
            E.conv_to_act(src, 0, Cc, w, 5, 5, Cc, None, None, L.ACT_NONE, dst, cout_store=Cc, dil=dil, pad=pad, Ho=H, Wo=W, in_bn=in_bn)
            outs.append(dst.t.clone())
        assert bool(torch.isfinite(outs[0].float()).all()) and float(outs[0].float().abs().max()) > 0.1
        assert torch.equal(outs[0], outs[1])
    finally:
        sos_amd.set_precision("bf16")


@pytest.mark.parametrize("case", [("1x1 raw", 1, 96, 96, "zeros", 1, None), ("1x1 affine relu", 1, 96, 96, "zeros", 1, "relu"),
                                  ("1x1 cout 90", 1, 90, 96, "zeros", 1, "relu"), ("5x1 reflect prelu", 5, 64, 64, "reflect", 1, "prelu"),
                                  ("5x1 zero raw", 5, 64, 64, "zeros", 1, None), ("5x1 dil2 zero relu", 5, 60, 64, "zeros", 2, "relu")],
                         ids=lambda c: c[0])
def test_thin_input_streaming_conv_equals_the_tiled_kernel(case, monkeypatch):
    """conv_thin_kernel (round 5): the convs over the 16-channel packed module input (14 -> 96 1x1, 10 -> 64 5x1) and the 1x1 data
    gradients of the 8-channel heads as per-wave streams.  Same MFMA per tap, same epilogue arithmetic: the output must equal the
    tiled kernel's bit for bit (SOS_CONV_NO_THIN=1 selects it), borders, partial last stage and padded output channels included;
    and both match torch on the storage-rounded operands."""
    import sos_amd
    from sos_amd import engine as E, _lib as L
    _, kh, cout, cs, pmode, dil, act = case
    sos_amd.set_precision("fp16")
    try:
        dev = torch.device("cuda")
        B, H, W = 2, 41, 37                                     # 3 034 pixels: not a multiple of the 32-pixel stage
        torch.manual_seed(11)
        x = E.Act(B, H, W, 16, False, dev)
        x.t.normal_()
        wt = torch.randn(cout, 16, kh, 1, device=dev) * 0.2
        w = E.pack_weight(wt, 16, False)
        scale = shift = None
        actc, slope = L.ACT_NONE, None
        if act:
            scale = torch.zeros(E.pad_to(cout, 32), device=dev)
            shift = torch.zeros_like(scale)
            scale[:cout] = torch.rand(cout, device=dev) + 0.5
            shift[:cout] = torch.randn(cout, device=dev) * 0.3
            actc = L.ACT_RELU if act == "relu" else L.ACT_PRELU
            slope = torch.tensor([0.25], device=dev) if act == "prelu" else None
        pad = ((kh - 1) // 2 * dil, 0)
        outs = []
        for no_thin in (False, True):
            if no_thin:
                monkeypatch.setenv("SOS_CONV_NO_THIN", "1")
            else:
                monkeypatch.delenv("SOS_CONV_NO_THIN", raising=False)
            dst = E.Act(B, H, W, cs, False, dev, zero=True)
            E.conv_to_act(x, 0, 16, w, kh, 1, cout, scale, shift, actc, dst, cout_store=cs, dil=(dil, 1), pad=pad, Ho=H, Wo=W,
                          slope=slope, pad_mode=L.PAD_REFLECT if pmode == "reflect" else L.PAD_ZERO)
            outs.append(dst.t.clone())
        assert float(outs[0].float().abs().max()) > 0.1
        assert torch.equal(outs[0], outs[1])
        xin = x.t.float().permute(0, 3, 1, 2)
        xp = F.pad(xin, (0, 0, pad[0], pad[0]), mode="reflect") if pmode == "reflect" else xin
        ref = F.conv2d(xp, wt.half().float(), None, 1, (0, 0) if pmode == "reflect" else pad, (dil, 1))
        if act:
            ref = ref * scale[:cout].view(1, -1, 1, 1) + shift[:cout].view(1, -1, 1, 1)
            ref = torch.relu(ref) if act == "relu" else torch.where(ref >= 0, ref, 0.25 * ref)
        got = outs[0].float().permute(0, 3, 1, 2)
        assert rel_err(got[:, :cout].cpu(), ref.cpu()) < 2e-3
        if cs > cout:
            assert float(got[:, cout:].abs().max()) == 0.0
    finally:
        sos_amd.set_precision("bf16")
