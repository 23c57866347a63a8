#!/bin/bash
# Power / clock samples (rocm-smi) while the 96->96 5x5 conv runs on random data and on all-zero data.
cd $GRAFT_REPO_ROOT
for Z in "" 1 w a; do
  echo "== conv96 d1x1, SOS_BENCH_ZERO=$Z  (empty: random data, 1: all zero, w: zero weights, a: zero activations)"
  ( SOS_BENCH_ZERO=$Z python tools/conv_bench.py --only "ctx96 d1x1" --iters 2000 --warm 1.0 > /tmp/pe.log 2>&1 ) &
  PID=$!
  sleep 2.2
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/^.*GPU\[0\]//' | tr '\n' ' '; echo
    sleep 0.3
  done
  wait $PID
  tail -1 /tmp/pe.log
done
