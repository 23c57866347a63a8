#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j46; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_forced_tilings.py -m gpu -q -x -k alternative > $O/pytest.log 2>&1; tail -3 $O/pytest.log
