"""How long does the host need to ENQUEUE one training step (both models, two streams) compared with the GPU time of the
step?  If the two are close, the step is launch bound in places."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sos_amd
from sos_amd import agent, tools, transform
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet

sos_amd.set_precision("fp16")
torch.manual_seed(0)
B, N = 64, 28000
det, jm = dnet.get_network().cuda().train(), jnet.get_network(MyConfig()).cuda().train()
raw = synth_batch(0, 8)
tile = lambda a: torch.from_numpy(np.tile(a, (8, 1))[:B]).cuda().contiguous()
mixed, clean, full_noise, bits = tile(raw["mixed"]), tile(raw["clean"]), tile(raw["full_noise"]), tile(raw["bits"])
mask, noise_sig = tools.bits_to_mask_batch(bits, 14000 / 30.0, N, mixed)
S = transform.stft_batch(torch.cat([mixed, clean * (1 - mask), noise_sig, full_noise]))
bj = {"mixed": S[:B].contiguous(), "clean": S[B:2 * B].contiguous(), "noise": S[2 * B:3 * B].contiguous(), "full_noise": S[3 * B:].contiguous()}
bd = {"audio": bj["mixed"], "label": bits.float()}
ad, aj = agent.DetectorAgent(det, lr=1e-3), agent.DenoiserAgent(jm, lr=1e-3)
for _ in range(3):
    agent.train_concurrent([(aj, bj), (ad, bd)])
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_concurrent([(aj, bj), (ad, bd)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
print("host enqueue ms per step:", [round(v, 1) for v in enq])
print("step ms (enqueue + drain):", [round(v, 1) for v in tot])
