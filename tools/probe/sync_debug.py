"""Does the training step contain host<->device synchronisation points?  torch.cuda.set_sync_debug_mode('warn') + timing of the
forward / backward / optimizer phases of each agent on its own."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sos_amd
from sos_amd import agent, tools, transform
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
if os.environ.get("SOS_FORCE_BUCKETS") == "1":          # the data-parallel gradient path in a world of one
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
    ad.train_func(bd); aj.train_func(bj)
torch.cuda.synchronize()
for name, ag, b in (("detector", ad, bd), ("denoiser", aj, bj)):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ag.train_func(b)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name}: host {1e3*(t1-t0):.1f} ms, until drained {1e3*(t2-t0):.1f} ms")
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    aj.train_func(bj); ad.train_func(bd)
    torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("default")
seen = {}
for x in w:
    k = str(x.message)[:100] + " @ " + f"{x.filename.split('/')[-1]}:{x.lineno}"
    seen[k] = seen.get(k, 0) + 1
print("sync warnings:", len(w))
for k, v in seen.items():
    print(v, k)
