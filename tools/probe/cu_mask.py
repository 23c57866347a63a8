#!/usr/bin/env python3
"""CU-partitioned streams (VERDICT r4 #1b): does a bandwidth-bound BatchNorm pass on a FEW compute units run beside an MFMA-bound
convolution on the REST without either paying for it?  hipExtStreamCreateWithCUMask streams (bit i of the mask -> XCD i % 8,
then shader engine, then CU: the first n bits are n / 8 CUs of every XCD), wrapped as torch.cuda.ExternalStream.
  1. conv 96->96 5x5 (B = 64) alone on 256 / 224 / 192 / 160 / 128 CUs
  2. bn_bwd (2R + 2R1W passes over the same tensor) alone on 32 / 64 / 96 / 128 / 256 CUs
  3. N convs and M bn_bwd passes back to back on one stream, on two unmasked streams, and on complementary masks"""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sos_amd  # noqa: E402
from sos_amd import _lib as L, engine as E  # noqa: E402

sos_amd.set_precision("fp16")
dev = torch.device("cuda")
hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*"))[0])


def masked(lo, n):
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, lo + n):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


B, H, W, Cc = 64, 256, 178, 96
src = E.Act(B, H, W, Cc, False, dev); src.t.normal_()
dst = E.Act(B, H, W, Cc, False, dev)
w = E.pack_weight(torch.randn(Cc, Cc, 5, 5, device=dev) * 0.05, Cc, False)
x = E.Act(B, H, W, Cc, False, dev); x.t.normal_()
dy = E.Act(B, H, W, Cc, False, dev); dy.t.normal_()
dx = E.Act(B, H, W, Cc, False, dev)
npix = B * H * W
nblk = L.lib().sos_bn_stats_blocks(npix)
partial = torch.empty(nblk * 3 * Cc, device=dev); coef = torch.empty(4 * Cc, device=dev)
one = torch.ones(Cc, device=dev); zero = torch.zeros(Cc, device=dev)
dgamma = torch.empty(Cc, device=dev); dbeta = torch.empty(Cc, device=dev)
vx, vdy, vdx = E.view(x, 0, Cc), E.view(dy, 0, Cc), E.view(dx, 0, Cc)


def conv():
    E.conv_to_act(src, 0, Cc, w, 5, 5, Cc, None, None, L.ACT_NONE, dst, cout_store=dst.cs, dil=(1, 1), pad=(2, 2), Ho=H, Wo=W)


def bn():
    L.check(L.lib().sos_bn_bwd(ctypes.byref(vdy), ctypes.byref(vx), L.ptr(one), L.ptr(zero), L.ptr(zero), L.ptr(one), L.ptr(one),
                               L.ACT_RELU, None, L.ptr(partial), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta), None, ctypes.byref(vdx),
                               None, L.stream_ptr()), "bwd")


def timed(jobs, iters=20, warm=10):
    """jobs: [(stream, fn, count per iteration)]; wall time per iteration (ms) with every stream joined."""
    main = torch.cuda.current_stream()

    def once():
        for st, fn, cnt in jobs:
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for _ in range(cnt):
                    fn()
        for st, _, _ in jobs:
            main.wait_stream(st)
    for _ in range(warm):
        once()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        once()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


fl = 2.0 * B * H * W * Cc * Cc * 25
byt = 5 * npix * Cc * 2
full = torch.cuda.Stream()
print("1. conv 96->96 5x5 alone")
print(f"   unmasked        {timed([(full, conv, 4)]) / 4:7.3f} ms")
for n in (256, 224, 192, 160, 128):
    t = timed([(masked(0, n), conv, 4)]) / 4
    print(f"   {n:3d} CUs         {t:7.3f} ms  {fl / t / 1e9:7.0f} TFLOP/s")
print("2. bn_bwd (5 tensor passes) alone")
print(f"   unmasked        {timed([(full, bn, 4)]) / 4:7.3f} ms")
for n in (32, 64, 96, 128, 256):
    t = timed([(masked(0, n), bn, 4)]) / 4
    print(f"   {n:3d} CUs         {t:7.3f} ms  {byt / t / 1e9:6.2f} TB/s")
print("3. 8 convs + 8 bn_bwd per iteration")
t_serial = timed([(full, lambda: (conv(), bn()), 8)])
print(f"   one stream, alternating        {t_serial:7.3f} ms")
s2 = torch.cuda.Stream()
print(f"   two unmasked streams           {timed([(full, conv, 8), (s2, bn, 8)]):7.3f} ms")
for nb in (32, 64, 96):
    a, b = masked(nb, 256 - nb), masked(0, nb)
    print(f"   conv on {256 - nb:3d} CUs, bn on {nb:3d} CUs {timed([(a, conv, 8), (b, bn, 8)]):7.3f} ms")
    print(f"   conv unmasked,  bn on {nb:3d} CUs  {timed([(full, conv, 8), (b, bn, 8)]):7.3f} ms")
print("4. 8 convs + 16 bn_bwd per iteration (the step's ratio is ~ 60 ms MFMA : 33 ms passes)")
print(f"   one stream                     {timed([(full, lambda: (conv(), bn(), bn()), 8)]):7.3f} ms")
print(f"   two unmasked streams           {timed([(full, conv, 8), (s2, bn, 16)]):7.3f} ms")
for nb in (64, 96, 128):
    a, b = masked(nb, 256 - nb), masked(0, nb)
    print(f"   conv on {256 - nb:3d} CUs, bn on {nb:3d} CUs {timed([(a, conv, 8), (b, bn, 16)]):7.3f} ms")
