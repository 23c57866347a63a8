#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_env.sh 3 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" > $O/ab_branch.txt 2>&1; tail -7 $O/ab_branch.txt
SOS_BENCH_TILE8=1 bash tools/probe/ab_env.sh 2 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" > $O/ab_branch_tile8.txt 2>&1; tail -5 $O/ab_branch_tile8.txt
