#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 400 tools/probe/bin/occ_probe > $O/occ_probe2.txt 2>&1
cat $O/occ_probe2.txt
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_pipeline.txt
