#!/usr/bin/env python3
"""Micro-benchmark of the recurrent LSTM kernels (B=64): denoiser (T=178, H=200) and detector (T=60, H=100)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sos_amd import _lib as L, engine as E
dev = torch.device("cuda"); B = 64
for name, T, H in (("denoiser", 178, 200), ("detector", 60, 100)):
    m = torch.nn.LSTM(64, H, bidirectional=True, batch_first=True).to(dev)
    pk = E.lstm_pack(m, False)
    xproj = torch.randn(B, T, 8 * H, device=dev)
    h = E.Act(B, 1, T, E.pad_to(2 * H, 16), False, dev, zero=True)
    gates = torch.zeros(B, T, 2, 4 * H, device=dev); cs = torch.zeros(B, T, 2, H, device=dev)
    dgates = torch.empty(B, T, 2, 4 * H, device=dev)
    dh = E.Act(B, 1, T, E.pad_to(2 * H, 16), False, dev, zero=True); dh.t.normal_()
    def fwd(): E.lstm(xproj, pk, B, T, H, h, gates, cs)
    def bwd(): L.check(L.lib().sos_lstm_bidir_bwd(L.ptr(dh.t), dh.nseg * dh.cs, dh.dtype_code, dh.cs, L.ptr(gates), L.ptr(cs),
                                                   L.ptr(pk["bh"]), L.ptr(pk["bl"]), B, T, H, L.ptr(dgates), L.stream_ptr()), "bwd")
    for fn, nm in ((fwd, "fwd"), (bwd, "bwd")):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"{name} {nm}: {ms:.3f} ms  ({1e3 * ms / T:.2f} us/step)", flush=True)
