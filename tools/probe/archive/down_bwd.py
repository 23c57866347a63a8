import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.nn as nn
import sos_amd
from sos_amd import engine as E, _lib as L, train_ops as TO
from sos_amd.denoiser.networks import DownConvBlock, UpConvBlock
from util import hashed, rel_err
from test_gpu_train_ops import _act_from_nchw, _act_to_nchw
sos_amd.set_precision("bf16x3"); x3 = True
torch.manual_seed(1)
for (cin, cout, k, s, d, H, W) in [(64, 128, 5, 2, 1, 20, 27), (128, 128, 3, 1, 4, 16, 23), (64, 64, 5, 1, 1, 18, 21)]:
    blk = DownConvBlock(cin, cout, k, s, dilation=d)
    ref = DownConvBlock(cin, cout, k, s, dilation=d); ref.load_state_dict(blk.state_dict())
    blk = blk.cuda().train()
    x = torch.from_numpy(hashed(3, (2, cin, H, W)).astype(np.float32))
    xa, xheld = _act_from_nchw(x, x3)
    lp = TO.down_train_plan(blk, x3)
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    dst = E.Act(2, Ho, Wo, cout, x3, torch.device("cuda"))
    t = TO.down_forward_train(lp, xa, 0, dst, 0, Ho, Wo, x3)
    xr = xheld.clone().requires_grad_(True)
    yr = ref.block(xr)
    print("fwd", rel_err(_act_to_nchw(dst, cout), yr))
    g = torch.from_numpy(hashed(4, tuple(yr.shape)).astype(np.float32))
    ga, gheld = _act_from_nchw(g, x3)
    yr.backward(gheld)
    gb = TO.GradBufs(x3)
    gb.bufs[id(dst)] = ga
    grads = {}
    TO.down_backward(t, gb, grads, "b", x3)
    print((cin, cout, k, s, d), "dW", rel_err(grads["b.block.1.weight"], ref.block[1].weight.grad), "dg", rel_err(grads["b.block.2.weight"], ref.block[2].weight.grad),
          "db", rel_err(grads["b.block.2.bias"], ref.block[2].bias.grad), "dslope", float(grads["b.block.3.weight"]), float(ref.block[3].weight.grad),
          "d_in", rel_err(_act_to_nchw(gb.of(xa), cin), xr.grad))
