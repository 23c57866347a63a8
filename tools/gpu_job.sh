#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j58; mkdir -p $O
for i in 1 2 3; do for sw in "SOS_BN_STREAM=256" "SOS_BN_STREAM=512" "SOS_BN_STREAM=1024"; do
  env $sw timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > $O/b.json
  python -c "
import json; d=json.loads(open('$O/b.json').read()); print('$sw', round(d['value'],1), d['ms_per_step'])" | tee -a $O/ab.txt
done; done
