#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_determinism.py tests/test_gpu_handoff.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_env.sh 3 "SOS_MIXED_OVERLAP_X=0" "SOS_MIXED_OVERLAP_X=1" --mode infer --train-detector 60 > $O/ab_trained.txt 2>&1; tail -1 $O/ab_trained.txt
bash tools/probe/ab_env.sh 2 "SOS_MIXED_OVERLAP_X=0" "SOS_MIXED_OVERLAP_X=1" --mode infer > $O/ab_mixed.txt 2>&1; tail -1 $O/ab_mixed.txt
bash tools/probe/ab_env.sh 2 "SOS_MIXED_OVERLAP_X=0" "SOS_MIXED_OVERLAP_X=1" --mode infer-ragged --steps 3 > $O/ab_ragged.txt 2>&1; tail -1 $O/ab_ragged.txt
