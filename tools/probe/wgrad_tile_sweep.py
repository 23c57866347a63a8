#!/usr/bin/env python3
"""Times sos_conv2d_wgrad on one layer shape under every forced pixel tile (SOS_WGRAD_TILE = "classes,log2 TH,log2 TW,order"):
calibration data for the tile cost model in wgrad.hip.   python tools/probe/wgrad_tile_sweep.py "ctx48 d32x32" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import engine as E
sos_amd.set_precision(os.environ.get('SOS_PRECISION', 'bf16'))
SH = {"ctx96 d1x1": (96, (5, 5), (1, 1)), "ctx96 d4x4": (96, (5, 5), (4, 4)), "ctx96 d16x16": (96, (5, 5), (16, 16)),
      "ctx96 d32x32": (96, (5, 5), (32, 32)), "ctx96 d8x8": (96, (5, 5), (8, 8)), "ctx96 d2x2": (96, (5, 5), (2, 2)), "ctx48 d1x1": (48, (5, 5), (1, 1)), "ctx48 d4x4": (48, (5, 5), (4, 4)),
      "ctx48 d8x8": (48, (5, 5), (8, 8)), "ctx48 d16x16": (48, (5, 5), (16, 16)), "ctx48 d32x32": (48, (5, 5), (32, 32)),
      "ctx48 d32x1": (48, (5, 5), (32, 1)), "ctx96 7x1": (96, (7, 1), (1, 1)), "ctx48 7x1": (48, (7, 1), (1, 1)),
      # U-Net: (cin, cout, k, dil, stride, H, W)
      "inp 256 3x3": (256, 256, (3, 3), (1, 1), 1, 64, 45), "inp 256 3x3 d16": (256, 256, (3, 3), (16, 16), 1, 64, 45),
      "inp 256 3x3 d4": (256, 256, (3, 3), (4, 4), 1, 64, 45), "inp 128 5x5": (128, 128, (5, 5), (1, 1), 1, 128, 89),
      "inp 64->128 s2": (64, 128, (5, 5), (1, 1), 2, 256, 178), "inp 64->128 3x3": (64, 128, (3, 3), (1, 1), 1, 256, 178),
      "inp 128->256 3x3": (128, 256, (3, 3), (1, 1), 1, 128, 89), "inp 128->256 s2": (128, 256, (3, 3), (1, 1), 2, 128, 89)}
dev = torch.device("cuda"); B = 64
for name in sys.argv[1:]:
    sh = SH[name]
    if len(sh) == 3:
        cin = cout = sh[0]; k, dil = sh[1], sh[2]; st, H, W = 1, 256, 178
    else:
        cin, cout, k, dil, st, H, W = sh
    Ho, Wo = (H + st - 1) // st, (W + st - 1) // st
    x = E.Act(B, H, W, cin, False, dev); x.t.normal_()
    g = E.Act(B, Ho, Wo, cout, False, dev); g.t.normal_()
    dw = torch.empty(cout, cin, k[0], k[1], device=dev)
    pad = ((k[0] - 1) // 2 * dil[0], (k[1] - 1) // 2 * dil[1])
    run = lambda: E.wgrad(g, 0, cout, x, 0, cin, k[0], k[1], dw, stride=st, dil=dil, pad=pad)
    def timed():
        run(); torch.cuda.synchronize()
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): run()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 5
    os.environ.pop("SOS_WGRAD_TILE", None)
    first = timed()
    res = []
    for lnc in range(0, 7):
        nc = 1 << lnc
        if nc > 1 and (st > 1 or nc > dil[1] or dil[1] % nc): break
        for lth in range(0, 9 - lnc):
            ltw = 8 - lnc - lth
            if ltw < 2: continue
            for ko in (0, 1):
                os.environ["SOS_WGRAD_TILE"] = f"{nc},{lth},{ltw},{ko}"
                try:
                    run(); torch.cuda.synchronize()
                except RuntimeError:
                    continue
                for _ in range(3): run()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): run()
                e.record(); torch.cuda.synchronize()
                res.append((s.elapsed_time(e) / 5, nc, lth, ltw, ko))
    os.environ.pop("SOS_WGRAD_TILE", None)
    last = timed()
    bms, bnc, blth, bltw, bko = min(res)
    os.environ["SOS_WGRAD_TILE"] = f"{bnc},{blth},{bltw},{bko}"
    again = timed()
    os.environ.pop("SOS_WGRAD_TILE", None)
    print(f"== {name}: model pick {last:.3f} ms (before the sweep {first:.3f}; best forced tile re-timed {again:.3f})")
    for ms, nc, lth, ltw, ko in sorted(res):
        print(f"   {ms:.3f} ms  NC={nc} TH={1 << lth} TW={1 << ltw} order={ko}")
    sys.stdout.flush()
