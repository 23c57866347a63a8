#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j50; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "thin_input_streaming or first_down_block" > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-250
for sw in "SOS_NOP=1" "SOS_CONV_NO_THIN=1" "SOS_CONV_THIN_WGS=256" "SOS_CONV_THIN_WGS=768"; do
  echo "== $sw"; for only in "thin 2->96 folded" "thin 2->64 folded"; do env SOS_PRECISION=fp16 $sw python tools/conv_bench.py --only "$only" 2>&1 | grep -v amdgpu; done
done | tee $O/conv_thin.txt
echo "== raw epilogue"; for only in "thin 2->96 folded" "thin 2->64 folded"; do env SOS_PRECISION=fp16 SOS_BENCH_EPI=raw python tools/conv_bench.py --only "$only" 2>&1 | grep -v amdgpu; done | tee -a $O/conv_thin.txt
