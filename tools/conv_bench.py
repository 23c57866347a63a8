#!/usr/bin/env python3
"""Micro-benchmark of sos_conv2d_fwd on the model's layer shapes (B=64): TFLOP/s per shape."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sos_amd import _lib as L  # noqa: E402
from sos_amd import engine as E  # noqa: E402

SHAPES = [
    # name, H, W, cin, cout, k, dil, stride, pad_mode
    ("ctx96 d1x1", 256, 178, 96, 96, (5, 5), (1, 1), 1, 0),
    ("ctx96 d2x1", 256, 178, 96, 96, (5, 5), (2, 1), 1, 0),
    ("ctx96 d8x1", 256, 178, 96, 96, (5, 5), (8, 1), 1, 0),
    ("ctx96 d32x1", 256, 178, 96, 96, (5, 5), (32, 1), 1, 0),
    ("ctx96 d4x4", 256, 178, 96, 96, (5, 5), (4, 4), 1, 0),
    ("ctx96 d16x16", 256, 178, 96, 96, (5, 5), (16, 16), 1, 0),
    ("ctx96 d32x32", 256, 178, 96, 96, (5, 5), (32, 32), 1, 0),
    ("ctx96 7x1", 256, 178, 96, 96, (7, 1), (1, 1), 1, 0),
    ("ctx48 d1x1", 256, 178, 48, 48, (5, 5), (1, 1), 1, 0),
    ("ctx48 d32x32", 256, 178, 48, 48, (5, 5), (32, 32), 1, 0),
    ("inp 128->128 5x5", 128, 89, 128, 128, (5, 5), (1, 1), 1, 1),
    ("inp 64->128 5x5 s2", 256, 178, 64, 128, (5, 5), (1, 1), 2, 1),
    ("inp 256->256 3x3 d1", 64, 45, 256, 256, (3, 3), (1, 1), 1, 1),
    ("inp 256->256 3x3 d16", 64, 45, 256, 256, (3, 3), (16, 16), 1, 1),
    ("inp 256->128 3x3", 128, 89, 256, 128, (3, 3), (1, 1), 1, 1),
    ("inp 128->64 3x3", 256, 178, 128, 64, (3, 3), (1, 1), 1, 1),
    # data gradients of the U-Net's reflect-padded blocks: full correlation of d_raw (fwd cout channels) with the flipped
    # weights onto the PADDED domain (train_ops._reflect_dgrad): zero padding (k-1)*dil, output (H + 2p) x (W + 2p), no epilogue
    ("dgrad up2.0 64->128 3x3", 256, 178, 64, 128, (3, 3), (1, 1), 1, "dgrad"),
    ("dgrad up1.0 128->256 3x3", 128, 89, 128, 256, (3, 3), (1, 1), 1, "dgrad"),
    ("dgrad down2.1 128->128 5x5", 128, 89, 128, 128, (5, 5), (1, 1), 1, "dgrad"),
    ("dgrad mid 256->256 3x3 d2", 64, 45, 256, 256, (3, 3), (2, 2), 1, "dgrad"),
    ("dgrad mid 256->256 3x3 d8", 64, 45, 256, 256, (3, 3), (8, 8), 1, "dgrad"),
    ("dgrad mid 256->256 3x3 d16", 64, 45, 256, 256, (3, 3), (16, 16), 1, "dgrad"),
    # round 4: the 2-channel first layers, as stored (16 channels per tap, 2 real) and with their horizontal taps folded into the
    # channel axis by the boundary pack (engine.wfold_spec: 14 / 10 of 16 channels real, 1 / kw of the taps); TFLOP/s of the
    # STORED contraction -- compare the milliseconds
    ("thin 2->96 1x7 stored", 256, 178, 16, 96, (1, 7), (1, 1), 1, 0),
    ("thin 2->96 folded 1x1", 256, 178, 16, 96, (1, 1), (1, 1), 1, 0),
    ("thin 2->48 1x7 stored", 256, 178, 16, 48, (1, 7), (1, 1), 1, 0),
    ("thin 2->48 folded 1x1", 256, 178, 16, 48, (1, 1), (1, 1), 1, 0),
    ("thin 2->64 5x5 stored", 256, 178, 16, 64, (5, 5), (1, 1), 1, 1),
    ("thin 2->64 folded 5x1", 256, 178, 16, 64, (5, 1), (1, 1), 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--warm", type=float, default=0.3)
    a = ap.parse_args()
    B = a.batch
    dev = torch.device("cuda")
    for name, H, W, cin, cout, k, dil, stride, pm in SHAPES:
        if a.only and a.only not in name:
            continue
        src = E.Act(B, H, W, cin, False, dev)
        zmode = os.environ.get('SOS_BENCH_ZERO', '')      # '1': all-zero operands, 'a': zero activations, 'w': zero weights (power experiments)
        if zmode in ('1', 'a'):
            src.t.zero_()
        else:
            src.t.normal_()
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        dgrad = pm == "dgrad"
        if dgrad:
            pm = 0
            Ho, Wo = H + (k[0] - 1) * dil[0], W + (k[1] - 1) * dil[1]
        dst = E.Act(B, Ho, Wo, E.pad_to(cout, 16), False, dev)
        w = E.pack_weight(torch.randn(cout, cin, k[0], k[1], device=dev) * (0.0 if zmode in ('1', 'w') else 0.05), cin, False)
        scale = torch.ones(w.shape[1], device=dev)
        shift = torch.zeros(w.shape[1], device=dev)
        pad = ((k[0] - 1) // 2 * dil[0], (k[1] - 1) // 2 * dil[1])
        if dgrad:
            pad = ((k[0] - 1) * dil[0], (k[1] - 1) * dil[1])

        # SOS_BENCH_EPI: 'eval' (default: folded BatchNorm + ReLU epilogue), 'raw' (the training forward's / the data
        # gradient's store of the bare accumulators), 'stats' (raw + the fused BatchNorm partial sums of the training forward)
        epi = os.environ.get('SOS_BENCH_EPI', 'eval')
        raw = dgrad or epi in ('raw', 'stats')

        # SOS_BENCH_INBN=1: the producer's BatchNorm + ReLU fused into the patch staging (sos_conv_desc.in_scale, round 5)
        in_bn = None
        if os.environ.get('SOS_BENCH_INBN') == '1':
            in_bn = (torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1)

        def run():
            E.conv_to_act(src, 0, cin, w, k[0], k[1], cout, None if raw else scale, None if raw else shift,
                          L.ACT_NONE if raw else L.ACT_RELU, dst, cout_store=dst.cs,
                          stride=stride, dil=dil, pad=pad, pad_mode=pm, Ho=Ho, Wo=Wo, stats_c=cout if epi == 'stats' and not dgrad else 0,
                          in_bn=in_bn)
        import time
        t_end = time.time() + a.warm
        run()
        torch.cuda.synchronize()
        while time.time() < t_end:      # let the clocks ramp: the box idles in a low-power state
            for _ in range(10):
                run()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        fl = 2.0 * B * Ho * Wo * cout * cin * k[0] * k[1]
        print(f"{name:24s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  ({100 * fl / ms / 1e9 / 2500:.1f}% of 2.5 PF)", flush=True)


if __name__ == "__main__":
    main()
