#!/bin/bash
# Same-box A/B of two environment settings on the training bench (alternating runs):
#   gpurun -- bash tools/probe/ab_env.sh 3 "SOS_STREAM_OVERLAP=gated" "SOS_STREAM_OVERLAP=split" [extra bench flags]
R=${1:-3}; A="$2"; B="$3"; shift 3
O=gpurun_out/ab; mkdir -p $O; : > $O/ab_env.txt
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$v', round(d['value'], 1), 'utt/s', round(d['ms_per_step'], 2), 'ms', 'roofline', round(d['roofline']['frac'], 3), '$E')" >> $O/ab_env.txt
  done
done
cat $O/ab_env.txt
python - <<'PY'
a = [float(l.split()[1]) for l in open("gpurun_out/ab/ab_env.txt") if l.startswith("A")]
b = [float(l.split()[1]) for l in open("gpurun_out/ab/ab_env.txt") if l.startswith("B")]
print(f"A mean {sum(a)/len(a):.1f}  B mean {sum(b)/len(b):.1f}  B/A {sum(b)/len(b)/(sum(a)/len(a)):.4f}")
PY
