#!/usr/bin/env python3
"""Per-signature time of the conv / wgrad launches of one training (or inference) step (B=64)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sos_amd
from sos_amd import agent, engine, tools, transform
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--top", type=int, default=40)
a = ap.parse_args(); B = a.batch
sos_amd.set_precision("bf16"); torch.manual_seed(0)
det = dnet.get_network().cuda(); jm = jnet.get_network(MyConfig()).cuda()
raw = synth_batch(0, min(B, 8)); rep = (B + 7) // 8
tile = lambda x: torch.from_numpy(np.tile(x, (rep, 1))[:B]).cuda().contiguous()
mixed, clean, full_noise, bits = tile(raw["mixed"]), tile(raw["clean"]), tile(raw["full_noise"]), tile(raw["bits"])
mask, noise_sig = tools.bits_to_mask_batch(bits, 14000 / 30.0, 28000, mixed)
S = transform.stft_batch(torch.cat([mixed, clean * (1 - mask), noise_sig, full_noise]))
bj = {"mixed": S[:B].contiguous(), "clean": S[B:2 * B].contiguous(), "noise": S[2 * B:3 * B].contiguous(), "full_noise": S[3 * B:].contiguous()}
bd = {"audio": bj["mixed"], "label": bits.float()}
ad, aj = agent.DetectorAgent(det.train(), lr=1e-3), agent.DenoiserAgent(jm.train(), lr=1e-3)
def step(): ad.train_func(bd); aj.train_func(bj)
for _ in range(2): step()
torch.cuda.synchronize()
engine.PROFILER = engine.LaunchProfiler()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); step(); e.record(); torch.cuda.synchronize()
summ = engine.PROFILER.summary(); engine.PROFILER = None
tot = sum(v["total_ms"] for v in summ.values())
print(f"step (profiled) {s.elapsed_time(e):.1f} ms; conv+wgrad launches {tot:.1f} ms")
for kind in ("conv", "wgrad"):
    print(f"  {kind}: {sum(v['total_ms'] for k, v in summ.items() if k[0] == kind):.1f} ms")
print(f"{'signature':70s} {'n':>3s} {'tot ms':>8s} {'avg ms':>8s} {'TF/s':>7s}")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:a.top]:
    print(f"{str(k):70s} {v['launches']:3d} {v['total_ms']:8.2f} {v['avg_ms']:8.3f} {v['flops'] / v['avg_ms'] / 1e9:7.0f}")
