#!/bin/bash
# Same-box A/B of two bench.py flag sets (alternating runs):  gpurun -- bash tools/probe/ab_flags.sh 3 "" "--force-buckets"
R=${1:-3}; A="$2"; B="$3"
O=gpurun_out/ab; mkdir -p $O; : > $O/ab_flags.txt
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then F="$A"; else F="$B"; fi
    python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-secondary $F 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$v', round(d['value'], 1), 'utt/s', round(d['ms_per_step'], 2), 'ms', '[$F]')" >> $O/ab_flags.txt
  done
done
cat $O/ab_flags.txt
python - <<'PY'
a = [float(l.split()[1]) for l in open("gpurun_out/ab/ab_flags.txt") if l.startswith("A")]
b = [float(l.split()[1]) for l in open("gpurun_out/ab/ab_flags.txt") if l.startswith("B")]
print(f"A mean {sum(a)/len(a):.1f}  B mean {sum(b)/len(b):.1f}  B/A {sum(b)/len(b)/(sum(a)/len(a)):.4f}")
PY
