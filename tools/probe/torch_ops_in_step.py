#!/usr/bin/env python3
"""Which torch device ops (not our HIP launches) does one training step enqueue, from where?  torch.profiler over one steady-state
step of the bench workload, grouped by op name and by the Python source line that issued them."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = bench.Workload("train", "fp16", 64, 0)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows = []
for e in ev:
    if e.device_time_total <= 0 and e.count < 5:
        continue
    if not e.key.startswith("aten::"):
        continue
    st = [s for s in e.stack if "sos" in s or "listening" in s or "bench" in s]
    rows.append((e.count, e.device_time_total, e.key, str(e.input_shapes)[:60], (st[0] if st else (e.stack[0] if e.stack else ""))[-110:]))
rows.sort(key=lambda r: -r[0])
print("count  device_us  op  shapes  source")
for r in rows[:70]:
    print("%5d %9.0f  %-28s %-60s %s" % r)
tot = collections.Counter()
for e in prof.key_averages():
    if e.key.startswith("aten::"):
        tot[e.key] += e.count
print("aten ops per step:", sum(tot.values()), tot.most_common(25))
