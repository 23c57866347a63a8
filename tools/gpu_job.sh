#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j01; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -rP > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/probe/cu_mask.py > $O/cu_mask.txt 2>&1; tail -40 $O/cu_mask.txt
for k in "" 3 7; do
  if [ -z "$k" ]; then timeout 300 python tools/probe/track_bound.py >> $O/track.txt 2>&1
  else SOS_CONV_TUNE=0 SOS_CONV_FORCE_CFG=$k timeout 300 python tools/probe/track_bound.py >> $O/track.txt 2>&1; fi
done
grep FORCE_CFG $O/track.txt | cut -c1-120
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-3000
