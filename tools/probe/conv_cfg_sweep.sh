#!/bin/bash
# Times the first N candidates of the cost-ordered tiling list of one conv_bench shape (SOS_CONV_FORCE_CFG = k):
#   gpurun -- bash tools/probe/conv_cfg_sweep.sh "ctx96 d32x32" 16
NAME="$1"; N=${2:-16}
for k in $(seq 0 $((N-1))); do
  SOS_CONV_TUNE=0 SOS_CONV_TUNE_TABLE=0 SOS_CONV_FORCE_CFG=$k SOS_CONV_LIST=1 python tools/conv_bench.py --only "$NAME" --iters 5 --warm 0.1 2>&1 | grep -v amdgpu | awk -v k=$k '/cfg/ && !s {c=$0; s=1} / ms / {print k": "$0" | "c}'
done
