#!/bin/bash
# Same-box A/B of two builds of the library on the training bench: gpurun -- bash tools/probe/ab_bench.sh [rounds] [extra bench flags]
# A = ab/base/libsos_hip*.so (a build of an earlier commit, see DESIGN.md "how A/B numbers are taken"), B = the in-tree build;
# the runs alternate A B A B ... so that box-to-box and warm-up differences cancel.
R=${1:-3}; shift
O=gpurun_out/ab; mkdir -p $O; : > $O/ab.txt
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then export SOS_HIP_LIB=$PWD/ab/base/libsos_hip.so SOS_HIP_LIB_F16=$PWD/ab/base/libsos_hip_f16.so; else unset SOS_HIP_LIB SOS_HIP_LIB_F16; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$v', round(d['value'], 1), 'utt/s', round(d['ms_per_step'], 2), 'ms', 'roofline', round(d['roofline']['frac'], 3))" >> $O/ab.txt
  done
done
cat $O/ab.txt
python - <<'PY'
import re
a = [float(l.split()[1]) for l in open("gpurun_out/ab/ab.txt") if l.startswith("A")]
b = [float(l.split()[1]) for l in open("gpurun_out/ab/ab.txt") if l.startswith("B")]
print(f"A mean {sum(a)/len(a):.1f}  B mean {sum(b)/len(b):.1f}  B/A {sum(b)/len(b)/(sum(a)/len(a)):.4f}")
PY
