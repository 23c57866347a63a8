#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_agent.py tests/test_gpu_determinism.py tests/test_audiovisual.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/bn_bench.py 2>&1 | grep -v amdgpu > $O/bn.txt; BN_C=48 python tools/bn_bench.py 2>&1 | grep -v amdgpu >> $O/bn.txt; cat $O/bn.txt
bash tools/probe/ab_bench.sh 3 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
