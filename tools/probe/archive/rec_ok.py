import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sos_amd
from sos_amd import agent, tools, transform, engine as E
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
sos_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp16")
torch.manual_seed(0)
B, N = 8, 28000
det, jm = dnet.get_network().cuda().train(), jnet.get_network(MyConfig()).cuda().train()
raw = synth_batch(0, 8)
t = lambda a: torch.from_numpy(a[:B]).cuda().contiguous()
mixed, clean, full_noise, bits = t(raw["mixed"]), t(raw["clean"]), t(raw["full_noise"]), t(raw["bits"])
mask, noise_sig = tools.bits_to_mask_batch(bits, 14000 / 30.0, N, mixed)
S = transform.stft_batch(torch.cat([mixed, clean * (1 - mask), noise_sig, full_noise]))
bj = {"mixed": S[:B].contiguous(), "clean": S[B:2 * B].contiguous(), "noise": S[2 * B:3 * B].contiguous(), "full_noise": S[3 * B:].contiguous()}
bd = {"audio": bj["mixed"], "label": bits.float()}
ad, aj = agent.DetectorAgent(det, lr=1e-3), agent.DenoiserAgent(jm, lr=1e-3)
built = []
orig = E.PackRecorder.__init__
def spy(self, module, build):
    orig(self, module, build); built.append((type(module).__name__, self.ok, len(self.outs)))
E.PackRecorder.__init__ = spy
for i in range(3):
    ad.train_func(bd); aj.train_func(bj)
    torch.cuda.synchronize()
    print("step", i, "recorders built so far:", built)
