#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "conv_weight_grad" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for sw in "SOS_NOP=1" "SOS_WGRAD_NO_THIN=1"; do
  echo "== fp16 $sw" >> $O/wgrad.txt
  for only in "fold 10->64"; do
    env SOS_PRECISION=fp16 $sw timeout 300 python tools/wgrad_bench.py --only "$only" 2>&1 | grep -v amdgpu >> $O/wgrad.txt
  done
done
cat $O/wgrad.txt
