#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
( time timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee $O/pytest_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke_final.txt
( time timeout 1500 python bench.py > $O/bench_final.json 2> $O/bench_final.err ) 2>&1 | tail -4; cut -c1-300 $O/bench_final.json
