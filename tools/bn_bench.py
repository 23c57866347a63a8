#!/usr/bin/env python3
"""Micro-benchmark of the BatchNorm elementwise / reduction kernels on a 96-channel B=64 activation."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sos_amd import _lib as L, engine as E
dev = torch.device("cuda"); B, C = 64, int(os.environ.get("BN_C", "96"))
H, W = int(os.environ.get("BN_H", "256")), int(os.environ.get("BN_W", "178"))      # BN_C / BN_H / BN_W: the U-Net's tensors
x = E.Act(B, H, W, C, False, dev); x.t.normal_()
dy = E.Act(B, H, W, C, False, dev); dy.t.normal_()
dx = E.Act(B, H, W, C, False, dev)
y = E.Act(B, H, W, C, False, dev)
npix = B * H * W
nblk = L.lib().sos_bn_stats_blocks(npix)
partial = torch.empty(nblk * 3 * C, device=dev); coef = torch.empty(4 * C, device=dev)
scale = torch.ones(C, device=dev); shift = torch.zeros(C, device=dev); mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev)
gamma = torch.ones(C, device=dev); dgamma = torch.empty(C, device=dev); dbeta = torch.empty(C, device=dev)
vx, vdy, vdx, vy = E.view(x, 0, C), E.view(dy, 0, C), E.view(dx, 0, C), E.view(y, 0, C)
def bwd(): L.check(L.lib().sos_bn_bwd(ctypes.byref(vdy), ctypes.byref(vx), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                      L.ACT_RELU, None, L.ptr(partial), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), None, ctypes.byref(vdx), None, L.stream_ptr()), "bwd")
def stats(): L.check(L.lib().sos_bn_stats(ctypes.byref(vx), L.ptr(partial), L.stream_ptr()), "stats")
def apply(): L.check(L.lib().sos_bn_act_apply(ctypes.byref(vx), L.ptr(scale), L.ptr(shift), L.ACT_RELU, None, ctypes.byref(vy), 0, 0, 0, None, L.stream_ptr()), "apply")
byt = npix * C * 2
for fn, nm, nb in ((bwd, "bn_bwd (reduce 2R + apply 2R1W)", 5 * byt), (stats, "bn_stats (1R)", byt), (apply, "bn_apply (1R1W)", 2 * byt)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{nm:34s} {ms * 1e3:8.1f} us  {nb / ms / 1e9:6.2f} TB/s", flush=True)
