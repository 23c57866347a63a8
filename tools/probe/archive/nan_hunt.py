import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import sos_amd
from sos_amd import engine as E, _lib as L, train_ops as TO, common_nets as CN
from util import hashed
sos_amd.set_precision("bf16x3"); x3 = True
# poison the allocator's free pool
p = torch.full((512 * 1024 * 1024,), float("nan"), dtype=torch.bfloat16, device="cuda"); del p
torch.manual_seed(0)
B, H, W = 2, 32, 24
enc = CN.make_encoder([(5, 5), (5, 5)], [(2, 1), (1, 1)], nf=48, outf=8).cuda().train()
x = torch.from_numpy(hashed(5, (B, 2, H, W)).astype(np.float32))
plan = TO.encoder_train_plan(enc, x3)
for i, lp in enumerate(plan): print("plan", i, "w nan", bool(torch.isnan(lp["w"].float()).any()), "wd nan", bool(torch.isnan(lp["wd"].float()).any()))
a = CN.pack_encoder_input(plan, x.cuda(), x3)       # (block 0 of a training plan may be folded: engine.wfold_spec)
print("pack nan", bool(torch.isnan(a.t.float()).any()))
dev = a.t.device
cur = a
nfeat = 8 * H
feat = torch.empty((B, W, 3 * nfeat), dtype=torch.bfloat16, device="cuda")
fspec = dict(t=feat, row=3 * nfeat, third=nfeat, c_off=0, H=H, W=W, Wo=W, gather=None, x3=x3)
for i, lp in enumerate(plan):
    cs = E.pad_to(lp["cout"], 16)
    one, zero = TO.ones_zeros(lp["w"].shape[1], dev)
    raw = E.Act(B, H, W, cs, x3, dev)
    E.conv_to_act(cur, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], one, zero, L.ACT_NONE, raw, cout_store=cs, dil=lp["dil"], pad=lp["pad"], Ho=H, Wo=W)
    torch.cuda.synchronize()
    n = torch.isnan(raw.t.float())
    print("block", i, "raw nan count", int(n.sum()), "of", n.numel(), "channels with nan:", torch.nonzero(n.any(dim=0).any(dim=0).any(dim=0)).flatten().tolist()[:20])
    last = i == len(plan) - 1
    y = None if last else E.Act(B, H, W, cs, x3, dev)
    saved = E.bn_train(raw, 0, lp["cout"], lp["bn"], L.ACT_RELU, None, y, 0, fspec if last else None)
    torch.cuda.synchronize()
    print("   saved nan:", {k: bool(torch.isnan(v).any()) for k, v in saved.items()}, "y nan", None if y is None else int(torch.isnan(y.t.float()).sum()))
    cur = y
print("feat nan", int(torch.isnan(feat.float()).sum()))
# ---- backward with poisoned pool
tape = TO.encoder_forward_train(plan, a, fspec, x3)
gd = torch.from_numpy(hashed(6, (B, W, nfeat)).astype(np.float32))
ghi = gd.to(torch.bfloat16); glo = (gd - ghi.float()).to(torch.bfloat16)
dfeat = torch.cat([ghi, ghi, glo], dim=2).cuda().contiguous()
p = torch.full((256 * 1024 * 1024,), float("nan"), dtype=torch.bfloat16, device="cuda"); del p
dy = TO.feat_grad_to_nhwc(dfeat, 3 * nfeat, nfeat, 0, 8, B, H, W, W, x3)
print("dy nan", int(torch.isnan(dy.t.float()).sum()))
for i in range(len(plan) - 1, -1, -1):
    lp, tp = plan[i], tape[i]
    raw = tp["raw"]
    d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev)
    dg, db, _ = TO.bn_bwd(dy, 0, raw, 0, lp["cout"], tp["saved"], lp["bn"].weight, L.ACT_RELU, None, d_raw)
    torch.cuda.synchronize()
    n = torch.isnan(d_raw.t.float())
    print(i, "d_raw nan", int(n.sum()), "chan:", torch.nonzero(n.any(0).any(0).any(0)).flatten().tolist()[:24], "dg nan", bool(torch.isnan(dg).any()), "db nan", bool(torch.isnan(db).any()))
    dw = torch.empty_like(lp["conv"].weight, dtype=torch.float32)
    E.wgrad(d_raw, 0, lp["cout"], tp["inp"], 0, lp["cin"], lp["kh"], lp["kw"], dw, dil=lp["dil"], pad=lp["pad"])
    torch.cuda.synchronize()
    print(i, "dw nan", int(torch.isnan(dw).sum()), "inp nan", int(torch.isnan(tp["inp"].t.float()).sum()))
    inp = tp["inp"]
    d_in = E.Act(inp.B, inp.H, inp.W, inp.cs, x3, dev)
    one, zero = TO.ones_zeros(lp["wd"].shape[1], dev)
    E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], lp["kh"], lp["kw"], lp["cin"], one, zero, L.ACT_NONE, d_in, cout_store=inp.cs, dil=lp["dil"],
                  pad=(lp["dil"][0] * (lp["kh"] - 1) - lp["pad"][0], lp["dil"][1] * (lp["kw"] - 1) - lp["pad"][1]), Ho=inp.H, Wo=inp.W)
    torch.cuda.synchronize()
    n = torch.isnan(d_in.t.float())
    print(i, "d_in nan", int(n.sum()), "chan:", torch.nonzero(n.any(0).any(0).any(0)).flatten().tolist()[:24])
    dy = d_in
