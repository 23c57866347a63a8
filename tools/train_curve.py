"""Loss curves of the denoiser (JointModel) trained from the same initial weights on the same synthetic batches in the three
precision modes -- bf16x3 (~fp32 accuracy: the parity mode), fp16 (the timed mode), bf16 -- plus the SI-SDR gain of the trained
chain on held-out clips.  Evidence that the 16-bit modes follow the same optimisation as the parity mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sos_amd
from sos_amd import agent
from sos_amd.common import MyConfig
from sos_amd.dataset import make_batch
from sos_amd.denoiser import networks as jnet

STEPS, B = int(os.environ.get("STEPS", 200)), 8
curves = {}
for precision in ("bf16x3", "fp16", "bf16"):
    sos_amd.set_precision(precision)
    torch.manual_seed(0)
    ag = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
    c = []
    for it in range(STEPS):
        _, ls = ag.train_func(make_batch("denoiser", 7000 + B * it, B))
        c.append((float(ls["stage1"].detach()), float(ls["stage2"].detach())))
    curves[precision] = np.array(c)
    # held-out loss in eval-free form: one more batch, forward only
    del ag
sos_amd.set_precision("bf16")
print(f"# denoiser, B = {B}, Adam 1e-3, {STEPS} steps, same init (manual_seed 0) and batches; columns: stage1 MSE | stage2 MSE")
print("# step   " + "   ".join(f"{p:>17s}" for p in curves))
for it in list(range(0, STEPS, 10)) + [STEPS - 1]:
    print(f"{it:5d}   " + "   ".join(f"{curves[p][it, 0]:8.4f} {curves[p][it, 1]:8.4f}" for p in curves))
ref = curves["bf16x3"].sum(1)
for p in ("fp16", "bf16"):
    tot = curves[p].sum(1)
    w = 10
    sm = lambda a: np.convolve(a, np.ones(w) / w, mode="valid")
    rel = np.abs(sm(tot) - sm(ref)) / sm(ref)
    print(f"# {p} vs bf16x3: |total loss difference| / loss, 10-step means: first 20 steps max {np.abs(tot[:20] - ref[:20]).max() / ref[:20].mean():.4f}, "
          f"whole run max {rel.max():.4f}, mean {rel.mean():.4f}; mean loss of the last 20 steps {tot[-20:].mean():.4f} (bf16x3 {ref[-20:].mean():.4f})")
