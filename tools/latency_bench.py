"""Request latency of the inference chain (STFT -> detector -> mask -> denoiser -> ISTFT), eager launches vs the
hipGraph replay of pipeline.GraphedDenoiser (BASELINE configs[3]).  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sos_amd  # noqa: E402,F401
from sos_amd import pipeline  # noqa: E402
from sos_amd.common import MyConfig  # noqa: E402
from sos_amd.denoiser import networks as jnet  # noqa: E402
from sos_amd.detector import networks as dnet  # noqa: E402

torch.manual_seed(0)
det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
g = pipeline.GraphedDenoiser(det, jm)


def wall(fn, iters):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


for b, secs in ((1, 1), (1, 2), (1, 10), (4, 2), (16, 2), (64, 2)):
    x = torch.randn(b, int(14000 * secs), device="cuda") * 0.1
    for _ in range(3):
        pipeline.denoise(det, jm, x)
    e = wall(lambda: pipeline.denoise(det, jm, x), 20)
    r = wall(lambda: g(x, clone=False), 20)
    print(f"B={b:3d} x {secs:2d} s   eager {e:7.2f} ms   graph {r:7.2f} ms   ({e / r:4.2f}x, {b * secs / r * 1e3:7.0f} x real time)")
