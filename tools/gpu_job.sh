#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j28; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_agent.py -m gpu -q -x -k "bucketer" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
