import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import sos_amd
from sos_amd import engine as E, _lib as L, train_ops as TO
from sos_amd.detector import networks as dnet
from oracle import nets as onet
from util import spec_input, rel_err
from test_gpu_train_ops import _act_to_nchw

sos_amd.set_precision("bf16x3")
g = np.load(os.path.join(R, "tests/golden/networks.npz"))
sd = onet.closed_form_state(onet.detector_spec(), seed=1)
det = dnet.get_network(); det.load_state_dict(sd); det = det.cuda().train()
B, T, nfr = 2, 89, 30
x = spec_input(100 + B, B, T)
label = torch.from_numpy(g["train_label"])
# oracle with autograd on CPU, keeping intermediates
sdr = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
import torch.nn.functional as F
h = x
inter = []
for i, d in enumerate(onet.DET_DILATIONS + [(1, 1)]):
    w = sdr[f"encoder_audio.{i}.block.0.weight"]
    pad = ((w.shape[2] - 1) // 2 * d[0], (w.shape[3] - 1) // 2 * d[1])
    raw = F.conv2d(h, w, None, 1, pad, d); raw.retain_grad()
    y = torch.relu(onet.batch_norm(raw, sdr, f"encoder_audio.{i}.block.1", True)); y.retain_grad()
    inter.append((raw, y)); h = y
f = h.reshape(B, -1, T)
f = onet.nearest_resize_last(f, nfr)
m = onet.lstm_bidir(f.permute(2, 0, 1), sdr, "lstm").permute(1, 0, 2)
m = torch.relu(onet.linear(m, sdr, "fc1.0")); lo = onet.linear(m, sdr, "fc1.2").squeeze(2)
loss = F.binary_cross_entropy_with_logits(lo, label); loss.backward()
# ours
out, tape = det._forward_train(x.cuda(), nfr)
gl = (torch.sigmoid(out) - label.cuda()) / label.numel()
plan = tape["plan"]; dev = out.device; x3 = True
grads = {}
dz2 = E.Act(B, 1, nfr, 16, x3, dev, zero=True)
TO.pack_grad(gl.contiguous(), None, L.ACT_NONE, B, nfr, 1, nfr, 1, 1, dz2)
d_m = TO.linear_backward(plan["fc2"], tape["m"], dz2, grads, "fc1.2", x3, dev)
dz0 = E.Act(B, 1, nfr, tape["m"].cs, x3, dev, zero=True)
TO.act_bwd_from_y(d_m, tape["m"], L.ACT_RELU, dz0, 100)
dh = TO.linear_backward(plan["fc0"], tape["h"], dz0, grads, "fc1.0", x3, dev)
dfeat = TO.lstm_backward(plan["lstm"], tape["lstm"], dh, grads, "lstm", B, nfr, x3, dev)
lo_, hi_ = TO.gather_ranges(tape["gather"].cpu().numpy(), T)
dy = TO.feat_grad_to_nhwc(dfeat, 3 * 2048, 2048, 0, 8, B, 256, T, nfr, x3, torch.from_numpy(lo_).to(dev), torch.from_numpy(hi_).to(dev))
print("dy11 vs ref y11.grad", rel_err(_act_to_nchw(dy, 8), inter[11][1].grad))
for i in range(11, 7, -1):
    lp, tp = plan["enc"][i], tape["enc"][i]
    raw = tp["raw"]
    print(i, "raw fwd err", rel_err(_act_to_nchw(raw, lp["cout"]), inter[i][0].detach()))
    d_raw = E.Act(raw.B, raw.H, raw.W, raw.cs, x3, dev)
    dg, db, _ = TO.bn_bwd(dy, 0, raw, 0, lp["cout"], tp["saved"], lp["bn"].weight, L.ACT_RELU, None, d_raw)
    print(i, "dbeta", rel_err(db, sdr[f"encoder_audio.{i}.block.1.bias"].grad), "dgamma", rel_err(dg, sdr[f"encoder_audio.{i}.block.1.weight"].grad),
          "d_raw", rel_err(_act_to_nchw(d_raw, lp["cout"]), inter[i][0].grad))
    # same reduction in torch fp32 on the GPU from OUR tensors
    rawf = _act_to_nchw(raw, lp["cout"]); dyf = _act_to_nchw(dy, lp["cout"])
    z = rawf * tp["saved"]["scale"].cpu()[None, :, None, None] + tp["saved"]["shift"].cpu()[None, :, None, None]
    dz = dyf * (z > 0)
    print(i, "   torch S1 from our tensors vs ours", rel_err(dz.sum(dim=(0, 2, 3)), db), " vs ref", rel_err(dz.sum(dim=(0, 2, 3)), sdr[f"encoder_audio.{i}.block.1.bias"].grad))
    inp = tp["inp"]
    d_in = E.Act(inp.B, inp.H, inp.W, inp.cs, x3, dev)
    one, zero = TO.ones_zeros(lp["wd"].shape[1], dev)
    E.conv_to_act(d_raw, 0, d_raw.cs, lp["wd"], lp["kh"], lp["kw"], lp["cin"], one, zero, L.ACT_NONE, d_in, cout_store=inp.cs, dil=lp["dil"],
                  pad=(lp["dil"][0] * (lp["kh"] - 1) - lp["pad"][0], lp["dil"][1] * (lp["kw"] - 1) - lp["pad"][1]), Ho=inp.H, Wo=inp.W)
    print(i, "d_in vs ref", rel_err(_act_to_nchw(d_in, lp["cin"]), inter[i - 1][1].grad))
    dy = d_in
