#!/usr/bin/env python3
"""Per-stream timeline of ONE steady-state training step from a rocprofv3 kernel trace (csv): for every HIP stream its first / last
kernel, busy time and every gap longer than 0.4 ms with the kernels on either side -- where a stream waits for another one.
usage: stream_timeline.py kernel_trace.csv"""
import collections
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))))
adam = [s for s, e, n, st, q in rows if "adam_multi" in n]
t0, t1 = adam[-5], adam[-3]          # from the first Adam of the step before the last two to the same point one step later
print(f"step window {(t1 - t0) / 1e6:.2f} ms")
per = collections.defaultdict(list)
for s, e, n, st, q in rows:
    if t0 <= s < t1:
        per[(st, q)].append((s, e, n))
for key, ks in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, n in kv[1])):
    busy = sum(e - s for s, e, n in ks)
    print(f"stream {key}: {len(ks)} kernels, first {(ks[0][0] - t0) / 1e6:.2f} ms, last end {(max(e for s, e, n in ks) - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms")
    if len(ks) < 20:
        continue
    prev_e, prev_n = ks[0][1], ks[0][2]
    for s, e, n in ks[1:]:
        if s - prev_e > 400000:
            print(f"     gap {(s - prev_e) / 1e6:6.2f} ms at {(prev_e - t0) / 1e6:7.2f} ms   after {prev_n[:44]:44s} before {n[:44]}")
        if e > prev_e:
            prev_e, prev_n = e, n
