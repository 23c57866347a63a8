#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r05 > gpurun_out/refresh_stdout.txt 2>&1; tail -30 gpurun_out/refresh_stdout.txt
