"""Do an MFMA-bound kernel (wgrad / conv) and an HBM-bound kernel (bn_bwd) overlap when issued on two HIP streams?"""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sos_amd import _lib as L, engine as E
dev = torch.device("cuda"); B, H, W, C = 64, 256, 178, 96
x = E.Act(B, H, W, C, False, dev); x.t.normal_()
g = E.Act(B, H, W, C, False, dev); g.t.normal_()
dw = torch.empty(C, C, 5, 5, device=dev)
dy = E.Act(B, H, W, C, False, dev); dy.t.normal_(); dx = E.Act(B, H, W, C, False, dev)
npix = B * H * W; nblk = L.lib().sos_bn_stats_blocks(npix)
partial = torch.empty(nblk * 3 * C, device=dev); coef = torch.empty(4 * C, device=dev)
one = torch.ones(C, device=dev); zero = torch.zeros(C, device=dev); dgam = torch.empty(C, device=dev); dbet = torch.empty(C, device=dev)
vx, vdy, vdx = E.view(x, 0, C), E.view(dy, 0, C), E.view(dx, 0, C)
w = E.pack_weight(torch.randn(C, C, 5, 5, device=dev) * 0.05, C, False)
dst = E.Act(B, H, W, C, False, dev)
def wgrad(): E.wgrad(g, 0, C, x, 0, C, 5, 5, dw, pad=(2, 2))
def conv(): E.conv_to_act(x, 0, C, w, 5, 5, C, None, None, L.ACT_NONE, dst, cout_store=dst.cs, pad=(2, 2))
def bn(): L.check(L.lib().sos_bn_bwd(ctypes.byref(vdy), ctypes.byref(vx), L.ptr(one), L.ptr(zero), L.ptr(zero), L.ptr(one), L.ptr(one), L.ACT_RELU, None,
                                     L.ptr(partial), L.ptr(coef), L.ptr(dgam), L.ptr(dbet), None, ctypes.byref(vdx), L.stream_ptr()), "bn")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, heavy in (("wgrad", wgrad), ("conv", conv)):
    heavy(); bn(); torch.cuda.synchronize()
    th, tb = timeit(heavy), timeit(bn)
    def both_serial(): heavy(); bn(); bn()
    def both_par():
        with torch.cuda.stream(s1): heavy()
        with torch.cuda.stream(s2): bn(); bn()
    ts, tp = timeit(both_serial), timeit(both_par)
    print(f"{name}: alone {th:.3f} ms, bn_bwd alone {tb:.3f} ms, serial (1 + 2 bn) {ts:.3f} ms, two streams {tp:.3f} ms")
