import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
import sos_amd
from sos_amd import engine as E, _lib as L, train_ops as TO, common_nets as CN
from util import hashed, rel_err
from test_gpu_train_ops import _act_from_nchw, _act_to_nchw

sos_amd.set_precision("bf16x3")
x3 = True
torch.manual_seed(0)
B, H, W = 2, 32, 24
ks = [(5, 5), (5, 5), (1, 1)]
dl = [(2, 1), (1, 1)]
enc = CN.make_encoder(ks[:2], dl, nf=48, outf=8)
# replace first conv in-channels 2 -> ok
ref = CN.make_encoder(ks[:2], dl, nf=48, outf=8)
ref.load_state_dict(enc.state_dict())
enc = enc.cuda().train()
x = torch.from_numpy(hashed(5, (B, 2, H, W)).astype(np.float32))
plan = TO.encoder_train_plan(enc, x3)
a = CN.pack_encoder_input(plan, x.cuda(), x3)       # (block 0 of a training plan may be folded: engine.wfold_spec)
nseg = 3
nfeat = 8 * H
feat = torch.empty((B, W, nseg * nfeat), dtype=torch.bfloat16, device="cuda")
fspec = dict(t=feat, row=nseg * nfeat, third=nfeat, c_off=0, H=H, W=W, Wo=W, gather=None, x3=x3)
tape = TO.encoder_forward_train(plan, a, fspec, x3)
# reference
xr = x.clone().requires_grad_(True)
h = xr
outs = []
for blk in ref:
    h = blk.block(h)
    outs.append(h)
fr = h.reshape(B, -1, W).permute(0, 2, 1)          # (B, W, 8*H)
got = feat.float().cpu()
got = got[..., :nfeat] + got[..., 2 * nfeat:]
print("fwd feat err", rel_err(got, fr))
gd = torch.from_numpy(hashed(6, (B, W, nfeat)).astype(np.float32))
fr.backward(gd)
ghi = gd.to(torch.bfloat16); glo = (gd - ghi.float()).to(torch.bfloat16)
dfeat = torch.cat([ghi, ghi, glo], dim=2).cuda().contiguous()
dy = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 0, 8, B, H, W, W, x3)
grads = {}
din = TO.encoder_backward(plan, tape, dy, grads, "e", x3, need_input_grad=True)
for i, blk in enumerate(ref):
    print(i, "dW", rel_err(grads[f"e.{i}.block.0.weight"], blk.block[0].weight.grad),
          "dgamma", rel_err(grads[f"e.{i}.block.1.weight"], blk.block[1].weight.grad),
          "dbeta", rel_err(grads[f"e.{i}.block.1.bias"], blk.block[1].bias.grad))
print("d_in", rel_err(_act_to_nchw(din, 2), xr.grad))
