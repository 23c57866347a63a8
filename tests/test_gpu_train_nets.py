"""Training-mode forward/backward of the HIP networks vs goldens produced by the reference
modules' autograd (tests/golden/make_goldens.py: losses, per-parameter gradient norms and heads)."""
import os

import numpy as np
import pytest
import torch

import sos_amd
from oracle import nets as onet
from util import rel_err, silent_gate, spec_input

pytestmark = pytest.mark.gpu

# whole-network gradient-norm bounds of the timed fp16 mode (observed worst 0.10 / median 7e-3 on the denoiser)
DET_TOL_FP16, JM_TOL_FP16 = 0.1, 0.15


# Bound on |d_slope - reference| as a fraction of the slope gradient's CONDITION SCALE (tests/golden/make_goldens.py:
# sum_{z<0} |dy*z| from the reference's own autograd, 130..470 here against gradients of 0.01..1, i.e. condition numbers
# of 2e2..2e4): d_slope = sum_{z<0} dy*z over ~1e6 signed terms cancels almost completely, so a relative perturbation
# eps of the terms moves it by ~eps * scale.  Observed: bf16x3 3e-4, fp16 1.7e-3, bf16 5e-3 of the scale.
SLOPE_TOL = {"bf16x3": 1e-3, "fp16": 5e-3, "bf16": 2e-2}

# ABSOLUTE ceilings of the 16-bit train-mode forward (VERDICT r4 #2b / ADVICE r4): the computed bound "2 x the storage model's own
# deviation" follows the model, so a regression that also moved tests/storage_model.py would pass; these do not follow anything.
# The model's deviations are themselves pinned on the CPU (tests/test_storage_model.py: fp16 9.5e-3 / 1.0e-3 / 7.3e-3, bf16
# 7.8e-2 / 7.1e-3 / 5.1e-2 for logits / n_pred / mask), the ceilings sit ~1.3x above twice those.
CEIL = {"fp16": dict(logits=2.5e-2, n_pred=3e-3, mask=2e-2), "bf16": dict(logits=2e-1, n_pred=2e-2, mask=1.3e-1)}


def _model_deviation(which, precision, x, n, nfr, g):
    """Deviation from the reference goldens of the storage model (tests/storage_model.py: the f32 oracle with round trips
    through the 16-bit storage type at the kernels' storage points) on the train-mode forward of the detector ('det':
    logits) or the denoiser ('jm': n_pred, mask)."""
    from storage_model import q, storage_dtype, storage_model
    with storage_model(storage_dtype(precision)), torch.no_grad():
        if which == "det":
            sd = onet.closed_form_state(onet.detector_spec(), seed=1)
            return rel_err(onet.detector_forward(sd, q(x), nfr, training=True), g["train_det_logits"])
        sd = onet.closed_form_state(onet.joint_spec(), seed=2)
        n_pred, mask = onet.joint_forward(sd, q(x), q(n), training=True)
        return rel_err(n_pred, g["train_n_pred"]), rel_err(mask, g["train_mask"])


def _check_grads(named_params, gradnorm, gradhead, tol, label, slope_scale=None):
    worst = 0.0
    bad = []
    errs = []
    for i, (name, p) in enumerate(named_params):
        assert p.grad is not None, name
        g = p.grad.detach().float().cpu().reshape(-1).numpy()
        gn = float(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
        e_norm = abs(gn - gradnorm[i]) / (gradnorm[i] + 1e-12)
        head = np.pad(g[:8], (0, max(0, 8 - len(g))))
        e_head = np.max(np.abs(head - gradhead[i])) / (np.max(np.abs(gradhead[i])) + 1e-3 * gradnorm[i] + 1e-12)
        if g.size == 1 and slope_scale is not None and np.isfinite(slope_scale[i]):
            # single shared PReLU slope: bounded against its condition scale (see SLOPE_TOL)
            e_s = abs(float(g[0]) - float(gradhead[i][0])) / float(slope_scale[i])
            print(f"  {label} {name:44s} d_slope {float(g[0]):+.4f} ref {float(gradhead[i][0]):+.4f} err/scale {e_s:.2e}")
            assert e_s < SLOPE_TOL[label], (name, float(g[0]), float(gradhead[i][0]), float(slope_scale[i]))
            continue
        errs.append(e_norm)
        worst = max(worst, e_norm, e_head / 10)
        if not (e_norm < tol and e_head < 10 * tol):
            bad.append((name, e_norm, e_head))
            print(f"  {label} {name:44s} |g| {gn:10.4e} ref {gradnorm[i]:10.4e} e_norm {e_norm:8.2e} e_head {e_head:8.2e}")
    assert not bad, bad
    return worst, float(np.median(errs))


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
def test_detector_train_step_matches_reference_autograd(golden, precision):
    from sos_amd.detector import networks as dnet
    g = golden("networks")
    sos_amd.set_precision(precision)
    try:
        det = dnet.get_network()
        det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1), strict=True)
        det = det.cuda().train()
        B, T, nfr = 2, 89, 30
        x = spec_input(100 + B, B, T).cuda()
        label = torch.from_numpy(g["train_label"]).cuda()
        logits = det(x, nfr)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, label)
        loss.backward()
        # Whole-network gradients are only conditionally stable: a forward difference of ~5e-5 flips a
        # few dozen ReLU decisions (|z| ~ 0) of the 364k outputs of a block, and BatchNorm's
        # backward sums (d_beta = sum dz) cancel heavily, so single elements move by ~1e-2 even
        # though every kernel matches torch to ~1e-5 in isolation (test_encoder_block_backward_exact,
        # tools/probe/archive/det_bwd_debug.py).  Norms stay within a few 1e-3 with the tuned tilings; other (equally valid)
        # conv tilings change the summation order and moved single tensors to 1.3e-2 (test_gpu_forced_tilings.py),
        # so the bound leaves room for whatever tiling the autotuner picks on a given box.
        tol = {"bf16x3": 2e-2, "fp16": DET_TOL_FP16, "bf16": 0.25}[precision]
        e_lo = rel_err(logits, g["train_det_logits"])
        print(precision, "train logits rel err", e_lo, "loss", float(loss), "ref", float(g["train_bce"]))
        if precision == "bf16x3":
            assert e_lo < 1e-3
        else:
            # 16-bit storage: the bound is COMPUTED -- the f32 oracle with nothing but round trips through the storage type at
            # the kernels' storage points (tests/storage_model.py) deviates from the reference by e_model; the kernels may
            # add nothing beyond the format: within 2x of that model's own deviation
            e_model = _model_deviation("det", precision, x.cpu(), None, nfr, g)
            print(precision, "  storage-model logits deviation", e_model, " HIP / model", e_lo / e_model)
            assert e_lo < 2.0 * e_model + 1e-4
            assert e_lo < CEIL[precision]["logits"]
        worst, med = _check_grads(list(det.named_parameters()), g["train_det_gradnorm"], g["train_det_gradhead"], tol, precision)
        print(precision, "worst grad err", worst, "median", med)
        assert med < {"bf16x3": 2e-3, "fp16": 2e-2, "bf16": 8e-2}[precision]
        # running statistics were updated like torch's
        rv = [v.detach().cpu().numpy().reshape(-1)[:4] for k, v in det.state_dict().items() if k.endswith("running_var")]
        want = g["train_det_running_var_head"]
        for a, b in zip(rv, want):
            assert rel_err(np.pad(a, (0, 4 - len(a))), b) < {"bf16x3": 1e-3, "fp16": 1e-2, "bf16": 5e-2}[precision]
    finally:
        sos_amd.set_precision("bf16")


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
def test_denoiser_train_step_matches_reference_autograd(golden, precision):
    from sos_amd.denoiser import networks as jnet
    from sos_amd.common import MyConfig
    from sos_amd import transform
    g = golden("networks")
    sos_amd.set_precision(precision)
    try:
        jm = jnet.get_network(MyConfig())
        jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2), strict=True)
        jm = jm.cuda().train()
        B, T = 2, 89
        x = spec_input(100 + B, B, T)
        n = silent_gate(x).cuda()
        clean = (spec_input(300, B, T) * 0.5)
        full_noise = (x - clean).cuda()
        x, clean = x.cuda(), clean.cuda()
        n_pred, out = jm(x, n)
        rec = transform.batch_fast_icRM_sigmoid(x, out)
        l1 = torch.nn.functional.mse_loss(n_pred, full_noise)
        l2 = torch.nn.functional.mse_loss(rec, clean)
        (l1 + l2).backward()
        x3 = precision == "bf16x3"
        e1, e2 = rel_err(n_pred, g["train_n_pred"]), rel_err(out, g["train_mask"])
        print(precision, "train n_pred/mask rel err", e1, e2, "losses", float(l1), float(l2), "ref", float(g["train_l1"]), float(g["train_l2"]))
        if precision == "bf16x3":
            assert max(e1, e2) < 1e-3
        else:       # computed bound, see the detector test
            m1, m2 = _model_deviation("jm", precision, x.cpu(), n.cpu(), None, g)
            print(precision, "  storage-model n_pred / mask deviation", m1, m2, " HIP / model", e1 / m1, e2 / m2)
            assert e1 < 2.0 * m1 + 1e-4 and e2 < 2.0 * m2 + 1e-4
            assert e1 < CEIL[precision]["n_pred"] and e2 < CEIL[precision]["mask"]
        ltol = {"bf16x3": 1e-3, "fp16": 5e-3, "bf16": 5e-2}[precision]
        assert abs(float(l1) / float(g["train_l1"]) - 1) < ltol
        assert abs(float(l2) / float(g["train_l2"]) - 1) < ltol
        worst, med = _check_grads(list(jm.named_parameters()), g["train_jm_gradnorm"], g["train_jm_gradhead"],
                                  {"bf16x3": 3e-2, "fp16": JM_TOL_FP16, "bf16": 0.4}[precision], precision,
                                  g["train_jm_slope_scale"])      # see the note on conditioning above
        print(precision, "worst grad err", worst, "median", med)
        assert med < {"bf16x3": 2e-3, "fp16": 1.5e-2, "bf16": 5e-2}[precision]
    finally:
        sos_amd.set_precision("bf16")


@pytest.mark.parametrize("first", [(5, 5), (1, 7)], ids=["first5x5", "first1x7"])
def test_encoder_block_backward_exact(first, monkeypatch):
    """Three Conv2d+BN(train)+ReLU blocks (dilated 5x5 or 1x7, 5x5, 1x1) forward + backward through the HIP
    kernels vs torch autograd on the same weights (bf16x3).  The first block reads 2 channels: it runs `plain` (the layer as
    stored: 16 channels per tap, 2 real) and with its horizontal taps on the channel axis (engine.wfold_spec: k x 1 forward conv
    and weight gradient over kw * 2 channels, un-folded weight gradient, data gradient in the layer's own geometry); the two
    must agree with each other to 5e-5 and with torch to 1e-4 (5x5 first block).  With the 1x7 first block on this 24-column
    input BOTH sit at 4e-4 (dw of block 0) / 1e-3 (d_in) from torch with the same digits -- a property of the data (one ReLU
    gate of block 0 within rounding of zero moves BatchNorm's backward sums), not of the fold: 3e-3 there."""
    import torch.nn.functional as F
    from sos_amd import engine as E, train_ops as TO, common_nets as CN
    from test_gpu_train_ops import _act_to_nchw
    from util import hashed
    sos_amd.set_precision("bf16x3")
    try:
        x3 = True
        B, H, W = 2, 32, 24
        kernels, dils = [first, (5, 5)], [(2, 1) if first == (5, 5) else (1, 1), (1, 1)]
        torch.manual_seed(0)
        ref = CN.make_encoder(kernels, dils, nf=48, outf=8)
        x = torch.from_numpy(hashed(5, (B, 2, H, W)).astype(np.float32))
        nfeat = 8 * H
        xr = x.clone().requires_grad_(True)
        h = xr
        for blk in ref:
            h = blk.block(h)
        fr = h.reshape(B, -1, W).permute(0, 2, 1)
        gd = torch.from_numpy(hashed(6, (B, W, nfeat)).astype(np.float32))
        fr.backward(gd)
        ghi = gd.to(torch.bfloat16)
        glo = (gd - ghi.float()).to(torch.bfloat16)
        dfeat = torch.cat([ghi, ghi, glo], dim=2).cuda().contiguous()
        wtol = 1e-4 if first == (5, 5) else 3e-3
        res = {}
        for wfold in (False, True):
            monkeypatch.setattr(E, "WFOLD", wfold)
            enc = CN.make_encoder(kernels, dils, nf=48, outf=8)
            enc.load_state_dict(ref.state_dict())
            enc = enc.cuda().train()
            plan = TO.encoder_train_plan(enc, x3)
            assert ("wtaps" in plan[0]) == wfold
            a = CN.pack_encoder_input(plan, x.cuda(), x3)
            feat = torch.empty((B, W, 3 * nfeat), dtype=torch.bfloat16, device="cuda")
            fspec = dict(t=feat, row=3 * nfeat, third=nfeat, c_off=0, H=H, W=W, Wo=W, gather=None, x3=x3)
            tape = TO.encoder_forward_train(plan, a, fspec, x3)
            got = feat.float().cpu()
            assert rel_err(got[..., :nfeat] + got[..., 2 * nfeat:], fr) < 1e-4
            dy = TO.feat_grad_to_nhwc(dfeat, 3 * nfeat, nfeat, 0, 8, B, H, W, W, x3)
            grads = {}
            din = TO.encoder_backward(plan, tape, dy, grads, "e", x3, need_input_grad=True)
            for i, blk in enumerate(ref):
                e_w = rel_err(grads[f"e.{i}.block.0.weight"], blk.block[0].weight.grad)
                print("wfold" if wfold else "plain", "block", i, "dw rel err", e_w)
                assert grads[f"e.{i}.block.0.weight"].shape == blk.block[0].weight.shape
                assert e_w < wtol
                assert rel_err(grads[f"e.{i}.block.1.weight"], blk.block[1].weight.grad) < wtol
                assert rel_err(grads[f"e.{i}.block.1.bias"], blk.block[1].bias.grad) < wtol
            e_in = rel_err(_act_to_nchw(din, 2), xr.grad)
            print("wfold" if wfold else "plain", "d_in rel err", e_in)
            assert e_in < wtol
            res[wfold] = ({k: v.detach().float().cpu() for k, v in grads.items()}, _act_to_nchw(din, 2))
        # the fold changes the summation order of the first block only: the two variants agree far below either's distance to torch
        for k in res[False][0]:
            assert rel_err(res[True][0][k], res[False][0][k]) < 5e-5, k
        assert rel_err(res[True][1], res[False][1]) < 5e-5
    finally:
        sos_amd.set_precision("bf16")
