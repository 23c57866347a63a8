"""Per-parameter gradient errors of one training step vs the reference-autograd goldens, per precision mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sos_amd
from oracle import nets as onet
from util import spec_input, silent_gate
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
from sos_amd.common import MyConfig
from sos_amd import transform
g = np.load(os.path.join(ROOT, "tests/golden/networks.npz"))
for prec in sys.argv[1:] or ["fp16"]:
    sos_amd.set_precision(prec)
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2), strict=True)
    jm = jm.cuda().train()
    B, T = 2, 89
    x = spec_input(100 + B, B, T); n = silent_gate(x).cuda(); clean = spec_input(300, B, T) * 0.5
    full_noise = (x - clean).cuda(); x, clean = x.cuda(), clean.cuda()
    n_pred, out = jm(x, n); rec = transform.batch_fast_icRM_sigmoid(x, out)
    (torch.nn.functional.mse_loss(n_pred, full_noise) + torch.nn.functional.mse_loss(rec, clean)).backward()
    rows = []
    for i, (name, p) in enumerate(jm.named_parameters()):
        gg = p.grad.detach().float().cpu().reshape(-1).numpy()
        gn = float(np.sqrt(np.sum(gg.astype(np.float64) ** 2)))
        rows.append((abs(gn - g["train_jm_gradnorm"][i]) / (g["train_jm_gradnorm"][i] + 1e-12), name, gn, float(g["train_jm_gradnorm"][i]), gg.size,
                     float(gg[0]), float(g["train_jm_gradhead"][i][0])))
    rows.sort(reverse=True)
    print("====", prec)
    for r in rows[:25]:
        print("%.3e %-46s |g| %.4e ref %.4e n=%d g0 %.4e ref0 %.4e" % r)
    e = np.array([r[0] for r in rows]); print("median", np.median(e), "p90", np.percentile(e, 90), "max", e.max())
