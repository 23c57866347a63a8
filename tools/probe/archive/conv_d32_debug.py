#!/usr/bin/env python3
"""Round 6 debug: where does the 6e-3 error of dW / dbeta in tests/test_gpu_block_goldens.py::test_zero_padded_dilation_32_block come
from -- a ReLU gate within rounding of zero (legitimate: one flipped element moves dbeta by one dy) or a kernel?  Compares the HIP
block with the oracle on the same box, element by element."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sos_amd  # noqa: E402
from oracle import nets as onet  # noqa: E402
from sos_amd import _lib as L, common_nets as CN, engine as E, train_ops as TO  # noqa: E402
from test_gpu_train_ops import _act_from_nchw, _act_to_nchw  # noqa: E402
from util import hashed  # noqa: E402

sos_amd.set_precision("bf16x3")
x3, C = True, 48
spec = [("b.block.0.weight", (48, 48, 5, 5), "conv")] + onet._bn_spec("b.block.1", 48)
sd = onet.closed_form_state(spec, seed=11)
x = torch.from_numpy(hashed(801, (2, 48, 80, 70)).astype(np.float32))
g = torch.from_numpy(hashed(900, (2, 48, 80, 70)).astype(np.float32))
# oracle in float64
sdo = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
xo = x.double().requires_grad_(True)
w = sdo["b.block.0.weight"]
raw_o = torch.nn.functional.conv2d(xo, w, None, 1, (64, 64), (32, 32))
raw_o.retain_grad()
bn_o = onet.batch_norm(raw_o, sdo, "b.block.1", True, {})
yo = torch.relu(bn_o)
yo.backward(g.double())
blk = CN.Conv2dBlock(48, 48, (5, 5), (32, 32))
blk.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
blk = blk.cuda().train()
dev = torch.device("cuda")
xa, _ = _act_from_nchw(x, x3)
lp = TO.encoder_train_plan(torch.nn.Sequential(blk), x3)[0]
one, zero = TO.ones_zeros(lp["w"].shape[1], dev)
raw = E.Act(2, 80, 70, 48, x3, dev)
st = E.conv_to_act(xa, 0, lp["cin_store"], lp["w"], 5, 5, C, one, zero, L.ACT_NONE, raw, cout_store=48, dil=(32, 32), pad=(64, 64), Ho=80, Wo=70, stats_c=C)
y = E.Act(2, 80, 70, 48, x3, dev)
saved = E.bn_train(raw, 0, C, lp["bn"], L.ACT_RELU, None, y, 0, None, stats=st)
ga, _ = _act_from_nchw(g, x3)
d_raw = E.Act(2, 80, 70, 48, x3, dev)
dgamma, dbeta, _ = TO.bn_bwd(ga, 0, raw, 0, C, saved, lp["bn"].weight, L.ACT_RELU, None, d_raw)
dw = torch.empty((C, C, 5, 5), dtype=torch.float32, device=dev)
E.wgrad(d_raw, 0, C, xa, 0, C, 5, 5, dw, dil=(32, 32), pad=(64, 64))
torch.cuda.synchronize()
rawn, yn, drn = _act_to_nchw(raw, C).double(), _act_to_nchw(y, C).double(), _act_to_nchw(d_raw, C).double()
print("raw rel err", float((rawn - raw_o.detach()).abs().max() / raw_o.detach().abs().max()))
gate_h, gate_o = yn > 0, bn_o.detach() > 0
flips = (gate_h != gate_o).nonzero()
print("gate flips", flips.shape[0], [(tuple(int(v) for v in f), float(bn_o.detach()[tuple(f)])) for f in flips[:10]])
e_draw = (drn - raw_o.grad).abs()
print("d_raw max err", float(e_draw.max()), "of", float(raw_o.grad.abs().max()), "at", np.unravel_index(int(e_draw.argmax()), e_draw.shape))
db_o = sdo["b.block.1.bias"].grad
e_db = (dbeta.double().cpu() - db_o).abs()
print("dbeta err per channel (top)", sorted([(float(e), c) for c, e in enumerate(e_db)], reverse=True)[:5], "max |dbeta|", float(db_o.abs().max()))
e_dw = (dw.double().cpu() - w.grad).abs()
per_co = e_dw.amax(dim=(1, 2, 3))
print("dw err per cout (top)", sorted([(float(e), c) for c, e in enumerate(per_co)], reverse=True)[:5], "max |dw|", float(w.grad.abs().max()))
# dw from the HIP d_raw by torch: isolates the wgrad kernel
dw_t = torch.nn.grad.conv2d_weight(x.double(), w.shape, drn, 1, (64, 64), (32, 32))
print("wgrad kernel vs torch on the SAME d_raw: rel err", float((dw.double().cpu() - dw_t).abs().max() / dw_t.abs().max()))
