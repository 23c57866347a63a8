#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j56; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/refresh_profiles.sh r05 > $O/refresh.log 2>&1; tail -4 $O/refresh.log | cut -c1-200
