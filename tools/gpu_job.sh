#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j25; mkdir -p $O
run() { env $1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary $2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(round(d['value'], 1), 'utt/s', '[$1] [$2]')"; }
for i in 1 2 3; do
run "SOS_DUMMY=0" "" >> $O/q.txt
run "SOS_BW_PRIO=1" "" >> $O/q.txt
run "SOS_SIDE_PRIO=1" "" >> $O/q.txt
run "SOS_DET_PRIO=1" "" >> $O/q.txt
run "SOS_SIDE_PRIO=1 SOS_DET_PRIO=1" "" >> $O/q.txt
done
cat $O/q.txt
