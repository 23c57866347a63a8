#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "conv_weight_grad" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for P in fp16; do
for sw in "" "SOS_WGRAD_NO_THIN=1 SOS_WGRAD_NO16_7=1"; do
  echo "== $P $sw" >> $O/wgrad.txt
  for only in "thin 96->8" "thin 48->4" "fold 14->96" "fold 14->48" "7x1"; do
    env SOS_PRECISION=$P $sw timeout 300 python tools/wgrad_bench.py --only "$only" >> $O/wgrad.txt 2>&1
  done
done
for occ in 1 2 3; do echo "== occ $occ" >> $O/wgrad.txt; for only in "thin 96->8" "fold 14->48"; do env SOS_PRECISION=$P SOS_WGT_OCC=$occ timeout 300 python tools/wgrad_bench.py --only "$only" >> $O/wgrad.txt 2>&1; done; done
done
cat $O/wgrad.txt
