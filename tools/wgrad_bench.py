#!/usr/bin/env python3
"""Micro-benchmark of sos_conv2d_wgrad (B=64): TFLOP/s per shape."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import _lib as L, engine as E
sos_amd.set_precision(os.environ.get('SOS_PRECISION', 'bf16'))
SHAPES = [("ctx96 d1x1", 256, 178, 96, 96, (5, 5), (1, 1), 1), ("ctx96 d8x1", 256, 178, 96, 96, (5, 5), (8, 1), 1),
          ("ctx96 d32x32", 256, 178, 96, 96, (5, 5), (32, 32), 1), ("ctx48 d1x1", 256, 178, 48, 48, (5, 5), (1, 1), 1),
          ("inp 256 3x3", 64, 45, 256, 256, (3, 3), (1, 1), 1), ("inp 128 5x5", 128, 89, 128, 128, (5, 5), (1, 1), 1),
          ("inp 64->128 s2", 256, 178, 64, 128, (5, 5), (1, 1), 2), ("lstm proj 3072->1600", 1, 178, 3072, 1600, (1, 1), (1, 1), 1),
          # round 3: the dilated layers (border k-steps skipped), the 7x1 layers and the thin first / last layers
          ("ctx96 d2x2", 256, 178, 96, 96, (5, 5), (2, 2), 1), ("ctx96 d4x4", 256, 178, 96, 96, (5, 5), (4, 4), 1),
          ("ctx96 d8x8", 256, 178, 96, 96, (5, 5), (8, 8), 1), ("ctx96 d16x16", 256, 178, 96, 96, (5, 5), (16, 16), 1),
          ("ctx96 d16x1", 256, 178, 96, 96, (5, 5), (16, 1), 1), ("ctx96 d32x1", 256, 178, 96, 96, (5, 5), (32, 1), 1),
          ("ctx48 d16x16", 256, 178, 48, 48, (5, 5), (16, 16), 1), ("ctx48 d32x32", 256, 178, 48, 48, (5, 5), (32, 32), 1),
          ("ctx96 7x1", 256, 178, 96, 96, (7, 1), (1, 1), 1), ("ctx48 7x1", 256, 178, 48, 48, (7, 1), (1, 1), 1),
          ("thin 2->96 1x7", 256, 178, 2, 96, (1, 7), (1, 1), 1), ("thin 2->48 1x7", 256, 178, 2, 48, (1, 7), (1, 1), 1),
          ("thin 2->64 5x5", 256, 178, 2, 64, (5, 5), (1, 1), 1), ("thin 64->2 5x5", 256, 178, 64, 2, (5, 5), (1, 1), 1),
          ("thin 96->8 1x1", 256, 178, 96, 8, (1, 1), (1, 1), 1), ("thin 48->4 1x1", 256, 178, 48, 4, (1, 1), (1, 1), 1),
          ("inp 256 3x3 d16", 64, 45, 256, 256, (3, 3), (16, 16), 1), ("inp 256 3x3 d8", 64, 45, 256, 256, (3, 3), (8, 8), 1),
          ("inp 64->128 3x3", 256, 178, 64, 128, (3, 3), (1, 1), 1), ("inp 128->256 3x3", 128, 89, 128, 256, (3, 3), (1, 1), 1),
          # round 4: the thin first layers with their horizontal taps on the channel axis (engine.wfold_spec): compare the ms with the "thin" rows
          ("fold 14->96 1x1", 256, 178, 14, 96, (1, 1), (1, 1), 1), ("fold 14->48 1x1", 256, 178, 14, 48, (1, 1), (1, 1), 1),
          ("fold 10->64 5x1", 256, 178, 10, 64, (5, 1), (1, 1), 1)]
ap = argparse.ArgumentParser(); ap.add_argument("--only", default=""); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--warm", type=float, default=0.3)
a = ap.parse_args()
dev = torch.device("cuda"); B = 64
for name, H, W, cin, cout, k, dil, st in SHAPES:
    if a.only and a.only not in name: continue
    zero = os.environ.get('SOS_BENCH_ZERO') == '1'       # all-zero operands: same instruction stream, no toggling (power A/B)
    x = E.Act(B, H, W, E.pad_to(cin, 16), False, dev); x.t.zero_() if zero else x.t.normal_()
    Ho, Wo = (H + st - 1) // st, (W + st - 1) // st
    g = E.Act(B, Ho, Wo, E.pad_to(cout, 16), False, dev); g.t.zero_() if zero else g.t.normal_()
    dw = torch.empty(cout, cin, k[0], k[1], device=dev)
    pad = ((k[0] - 1) // 2 * dil[0], (k[1] - 1) // 2 * dil[1])
    run = lambda: E.wgrad(g, 0, cout, x, 0, cin, k[0], k[1], dw, stride=st, dil=dil, pad=pad)
    run(); torch.cuda.synchronize(); t_end = time.time() + a.warm
    while time.time() < t_end:
        for _ in range(5): run()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    fl = 2.0 * B * Ho * Wo * cout * cin * k[0] * k[1]
    print(f"{name:24s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  ({100 * fl / ms / 1e9 / 2500:.1f}% of 2.5 PF)", flush=True)
