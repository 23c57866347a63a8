#!/usr/bin/env python3
"""Times sos_conv2d_wgrad on the U-Net / few-tap layer shapes under every forced channel tile of the workgroup
(SOS_WGRAD_MT m-tiles x SOS_WGRAD_NTB n-tiles of 32; the pixel tile is the cost model's pick for that channel tile):
which (MT, NTB) the host should choose where the (tap, n-tile) pairs do not divide over the 8 waves or the operands of the
default tile leave no room for the second LDS buffer.   python tools/probe/wgrad_cfg_sweep.py ["inp 256 3x3" ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import engine as E
sos_amd.set_precision(os.environ.get('SOS_PRECISION', 'bf16'))
# name: (N = cin, M = cout, k, dil, stride, H, W) of the layer input
SH = {"inp 256 3x3": (256, 256, (3, 3), (1, 1), 1, 64, 45), "inp 256 3x3 d2": (256, 256, (3, 3), (2, 2), 1, 64, 45),
      "inp 256 3x3 d4": (256, 256, (3, 3), (4, 4), 1, 64, 45), "inp 256 3x3 d8": (256, 256, (3, 3), (8, 8), 1, 64, 45),
      "inp 256 3x3 d16": (256, 256, (3, 3), (16, 16), 1, 64, 45),
      "inp 128 5x5": (128, 128, (5, 5), (1, 1), 1, 128, 89), "inp 64->128 s2": (64, 128, (5, 5), (1, 1), 2, 256, 178),
      "inp 64->128 3x3": (64, 128, (3, 3), (1, 1), 1, 256, 178), "inp 128->256 3x3": (128, 256, (3, 3), (1, 1), 1, 128, 89),
      "inp 256->256 s2": (256, 256, (3, 3), (1, 1), 2, 128, 89), "up 128<-64 s2": (64, 128, (3, 3), (1, 1), 2, 256, 178),
      "up 256<-128 s2": (128, 256, (3, 3), (1, 1), 2, 128, 89),
      "ctx96 7x1": (96, 96, (7, 1), (1, 1), 1, 256, 178), "ctx48 7x1": (48, 48, (7, 1), (1, 1), 1, 256, 178),
      "ctx96 d1x1": (96, 96, (5, 5), (1, 1), 1, 256, 178)}
dev = torch.device("cuda"); B = 64
names = sys.argv[1:] or [n for n in SH if n != "ctx96 d1x1"]
for name in names:
    cin, cout, k, dil, st, H, W = SH[name]
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
        for _ in range(8): run()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 8
    for v in ("SOS_WGRAD_MT", "SOS_WGRAD_NTB"): os.environ.pop(v, None)
    fl = 2.0 * B * Ho * Wo * cout * cin * k[0] * k[1]
    base = timed()
    ref = dw.clone()
    print(f"== {name}: default {base:.3f} ms = {fl / base / 1e9 / 2500:.3f} of peak", flush=True)
    taps = k[0] * k[1]
    for mt in (1, 2, 3):
        if mt > (cout + 31) // 32: continue
        for ntb in (1, 2, 3, 4):
            if ntb * taps > 32 or ntb > (cin + 31) // 32: continue
            os.environ["SOS_WGRAD_MT"] = str(mt); os.environ["SOS_WGRAD_NTB"] = str(ntb)
            try:
                ms = timed()
            except RuntimeError as ex:
                print(f"   MT={mt} NTB={ntb}: {str(ex)[:80]}"); continue
            err = float((dw - ref).abs().max() / ref.abs().max())
            print(f"   MT={mt} NTB={ntb}: {ms:.3f} ms = {fl / ms / 1e9 / 2500:.3f}   (max rel diff vs default {err:.1e})", flush=True)
    for v in ("SOS_WGRAD_MT", "SOS_WGRAD_NTB"): os.environ.pop(v, None)
