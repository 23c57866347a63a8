#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_determinism.py tests/test_gpu_coresidency.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_env.sh 3 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" --mode infer --precision fp16 > $O/ab_infer_fp16.txt 2>&1; tail -1 $O/ab_infer_fp16.txt
bash tools/probe/ab_env.sh 2 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" --mode infer > $O/ab_infer_mixed.txt 2>&1; tail -1 $O/ab_infer_mixed.txt
bash tools/probe/ab_env.sh 2 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" --mode infer-ragged --steps 3 > $O/ab_ragged.txt 2>&1; tail -1 $O/ab_ragged.txt
