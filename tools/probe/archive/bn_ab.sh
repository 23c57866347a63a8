#!/bin/bash
# BatchNorm kernel bandwidths of several library builds on the model's tensor shapes: gpurun -- bash tools/probe/archive/bn_ab.sh "" ab/v2 ab/v3
for cfg in "96 256 178" "48 256 178" "64 256 178" "128 128 89" "256 64 45"; do
  set -- $cfg; C=$1; H=$2; W=$3
  for lib in "" ab/v2 ab/v3; do
    if [ -n "$lib" ]; then export SOS_HIP_LIB=$PWD/$lib/libsos_hip.so SOS_HIP_LIB_F16=$PWD/$lib/libsos_hip_f16.so; else unset SOS_HIP_LIB SOS_HIP_LIB_F16; fi
    echo "C=$C ${H}x$W lib=${lib:-tree}: $(BN_C=$C BN_H=$H BN_W=$W python tools/bn_bench.py 2>&1 | grep bn_bwd | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')"
  done
done
