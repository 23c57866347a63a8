#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "fused_input" > $O/pytest_inbn.log 2>&1; tail -15 $O/pytest_inbn.log
for e in raw stats; do for i in 0 1 0 1; do echo "== EPI=$e INBN=$i" >> $O/inbn.txt; SOS_BENCH_EPI=$e SOS_BENCH_INBN=$i python tools/conv_bench.py --only ctx96 --iters 20 2>&1 | grep -v amdgpu >> $O/inbn.txt; done; done
cat $O/inbn.txt
python tools/bn_bench.py 2>&1 | grep -v amdgpu
