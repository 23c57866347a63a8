#!/usr/bin/env python3
"""STFT / ISTFT / mask-apply / bits->mask kernels at B = 64 clips of 2 s against the HBM roofline (SURVEY.md 8-d bytes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sos_amd import tools, transform  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


def main():
    for B in (64, 256):          # BASELINE configs[1] and configs[3]'s batch
        run(B)


def run(B):
    N = 28000
    print(f"--- B = {B}")
    wave = torch.randn(B, N, device="cuda") * 0.1
    S = transform.stft_batch(wave)
    crm = torch.rand_like(S) * 0.8 + 0.1
    bits = (torch.rand(B, 60, device="cuda") > 0.3).to(torch.uint8)
    rows = [("stft   (476 544 B/clip)", lambda: transform.stft_batch(wave), 476544),
            ("istft  (476 408 B/clip)", lambda: transform.istft_batch(S), 476408),
            ("crm apply (1 093 632 B/clip)", lambda: transform.batch_fast_icRM_sigmoid(S, crm), 1093632),
            ("bits->mask + noise (336 060 B/clip)", lambda: tools.bits_to_mask_batch(bits, 14000 / 30.0, N, wave), 3 * N * 4 + 60)]
    for name, fn, bytes_per_clip in rows:
        us = timed(fn)
        gbs = B * bytes_per_clip / us / 1e3
        print(f"{name:38s} {us:8.1f} us   {gbs:8.0f} GB/s algorithmic   {gbs / 8000:6.3f} of the 8 TB/s HBM roofline")


if __name__ == "__main__":
    main()
