"""Audio-visual variant (BASELINE configs[4] shape: 2 s clips = 60 frames of 224x224 + the 2x256x178 spectrogram),
inference throughput on one MI355X.  Algorithmic work: 1471 GFLOP (video branch) + 49 GFLOP (audio) per clip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sos_amd  # noqa: E402,F401
from sos_amd.detector import networks as dnet  # noqa: E402

torch.manual_seed(0)
net = dnet.get_network(video=True).cuda().eval()
for B in (1, 4, 16):
    s = torch.randn(B, 2, 256, 178, device="cuda")
    v = torch.rand(B, 3, 60, 224, 224, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            net(s, v=v)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            net(s, v=v)
        b.record()
        torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"B={B:3d}  {ms:8.2f} ms  {B / ms * 1e3:7.1f} clips/s  {B * 1520 / ms:7.1f} TFLOP/s end to end "
          f"({torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak)")

# ---- training step of the variant (forward with batch statistics, backward, Adam) -- BASELINE configs[4] is 32 clips per GPU
from sos_amd import agent  # noqa: E402

for B in (8, 32):
    torch.cuda.reset_peak_memory_stats()
    ag = agent.DetectorAgent(dnet.get_network(video=True), lr=1e-3)
    batch = {"audio": torch.randn(B, 2, 256, 178, device="cuda"), "frames": torch.rand(B, 3, 60, 224, 224, device="cuda"),
             "label": (torch.rand(B, 60, device="cuda") > 0.3).float()}
    for _ in range(2):
        ag.train_func(batch)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        ag.train_func(batch)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"train B={B:3d}  {ms:8.1f} ms/step  {B / ms * 1e3:7.1f} clips/s  {B * 3 * 1520 / ms:7.1f} TFLOP/s end to end "
          f"({torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak)")
    del ag, batch
