#!/bin/bash
# How much of a steady-state training step is the GPU idle (no kernel of either stream running)?  rocprofv3 kernel trace -> union of
# the kernel intervals of the timed steps.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/idle; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o k -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/idle/t/**/*kernel_trace.csv", recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
# the timed region: the last 6 steps = from the 3rd-from... find by splitting on adam_multi_kernel launches of the denoiser (2 per step)
adam = [s for s, e, n in iv if "adam_multi" in n]
t0 = adam[-13] if len(adam) >= 13 else iv[0][0]      # start after the adam of the step before the last 6
t1 = max(e for s, e, n in iv)
busy, cur_s, cur_e = 0, None, None
for s, e, n in iv:
    if e <= t0: continue
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = t1 - t0
ksum = sum(min(e, t1) - max(s, t0) for s, e, n in iv if e > t0)
import collections
small = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in iv:
    if e_ > t0 and e_ - s_ < 20000:
        small[n_[:80]][0] += 1; small[n_[:80]][1] += e_ - s_
ns = sum(v[0] for v in small.values()); ts = sum(v[1] for v in small.values())
print(f"kernels shorter than 20 us in the window: {ns/6:.0f} per step, {ts/6e6:.2f} ms per step")
for k, v in sorted(small.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {v[1]/6e6:7.3f} ms/step {v[0]/6:7.1f} calls/step  {k}")
print(f"window {tot/1e6:.1f} ms, GPU busy (union of kernels) {busy/1e6:.1f} ms = {100*busy/tot:.1f} %, idle {100*(tot-busy)/tot:.1f} %, sum of kernel times {ksum/1e6:.1f} ms")
PY
rm -rf $O/t
