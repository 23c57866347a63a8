#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j44; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
SOS_BN_STREAM=512 timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py -m gpu -q -x > $O/pytest_stream.log 2>&1; tail -2 $O/pytest_stream.log
for i in 1 2 3; do
  for sw in "SOS_FEAT_NO_TILE=1" "SOS_NOP=1"; do
    env $sw timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > $O/b.json
    python -c "
import json; d=json.loads(open('$O/b.json').read()); print('$sw', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/ab.txt
  done
done
rocprofv3 --kernel-trace --stats -d $O/p2 -o t --output-format csv -- python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/p2/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pack_wtaps' in r['Name'] or 'feat_to' in r['Name'] or 'copy_crop' in r['Name'] or 'gather_pack' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'])
PY
rm -rf $O/p2
