import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import sos_amd
from oracle import nets as onet
from util import spec_input
from sos_amd.detector import networks as dnet
g = np.load(os.path.join(R, "tests/golden/networks.npz"))
sos_amd.set_precision("bf16")
det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1)); det = det.cuda().train()
x = spec_input(102, 2, 89).cuda(); label = torch.from_numpy(g["train_label"]).cuda()
lo = det(x, 30); loss = torch.nn.functional.binary_cross_entropy_with_logits(lo, label); loss.backward()
torch.cuda.synchronize()
print("bf16 step done; any nan grads:", any(bool(torch.isnan(p.grad).any()) for p in det.parameters()))
del det, lo, loss
from sos_amd import engine as E, _lib as L, train_ops as TO, common_nets as CN
from util import hashed
sos_amd.set_precision('bf16x3'); x3 = True
exec(open(os.path.join(R, "tools/probe/archive/nan_hunt.py")).read().split("# poison the allocator's free pool")[1].split("\n", 2)[2].replace('p = torch.full((256 * 1024 * 1024,), float("nan"), dtype=torch.bfloat16, device="cuda"); del p', ''))
