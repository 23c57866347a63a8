#!/bin/bash
# Where is the chip under-filled during a steady-state training step?  rocprofv3 kernel trace of bench.py; every instant of the last
# 6 steps is weighted by the fraction of the chip its running kernels can occupy (workgroups x waves per workgroup against 256 CUs x
# 8 waves; a kernel with >= 2048 waves counts as full).  Prints the time per step spent below 50 % / 25 % fill and the kernels
# that run during that time.   gpurun -- bash tools/probe/archive/low_occupancy.sh [extra bench flags]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/lowocc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o k -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $O/bench.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lowocc/t/**/*kernel_trace.csv", recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    waves = grid // 64
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], min(1.0, waves / 2048.0)))
iv.sort()
adam = [s for s, e, n, w in iv if "adam_multi" in n]
t0 = adam[-13] if len(adam) >= 13 else iv[0][0]
t1 = max(e for s, e, n, w in iv)
ev = []
for i, (s, e, n, w) in enumerate(iv):
    if e <= t0: continue
    ev.append((max(s, t0), 1, i)); ev.append((e, 0, i))
ev.sort()
active = set(); last = t0
low50 = low25 = idle = 0
who = collections.defaultdict(float)
for t, kind, i in ev:
    if t > last:
        fill = sum(iv[j][3] for j in active)
        dt = t - last
        if not active: idle += dt
        elif fill < 0.25: low25 += dt
        elif fill < 0.5: low50 += dt
        if active and fill < 0.5:
            for j in active: who[iv[j][2][:70]] += dt
        last = t
    if kind: active.add(i)
    else: active.discard(i)
tot = t1 - t0
print(f"window {tot/6e6:.1f} ms/step: idle {idle/6e6:.2f}, fill < 25 % {low25/6e6:.2f}, fill 25-50 % {low50/6e6:.2f} ms/step")
for k, v in sorted(who.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v/6e6:7.3f} ms/step  {k}")
PY
