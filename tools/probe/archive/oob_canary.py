import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import pipeline, transform
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
sos_amd.set_precision("fp16")
torch.manual_seed(0)
det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
base = torch.from_numpy(synth_batch(500, 6)["mixed"]).cuda()
x = base[1:4, :14000].contiguous()
S = transform.stft_batch(x)
torch.cuda.synchronize()
import random
random.seed(0)
for name, fn in (("det", lambda: det(s=S, v_num_frames=30)), ("jm", lambda: jm(S, S))):
    for trial in range(3):
        # interleave canaries with freed holes so that the net's activations land between canaries
        canaries, holes = [], []
        for k in range(200):
            n = random.choice([64 << 10, 256 << 10, 1 << 20, 3 << 20, 365 << 10])
            t = torch.full((n // 4,), 7.25, dtype=torch.float32, device="cuda")
            (canaries if k % 2 else holes).append(t)
        del holes
        fn()
        torch.cuda.synchronize()
        bad = [(i, int((c != 7.25).sum()), c.numel()) for i, c in enumerate(canaries) if not bool((c == 7.25).all())]
        print(name, "trial", trial, "corrupted canaries:", bad[:5], flush=True)
        for i, cnt, n in bad[:2]:
            c = canaries[i]; idx = (c != 7.25).nonzero().flatten()
            print("   first/last bad index", int(idx[0]), int(idx[-1]), "of", n, "values", c[idx[:4]].tolist())
        del canaries
