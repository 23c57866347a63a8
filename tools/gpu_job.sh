#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j48; mkdir -p $O
for only in "thin 2->96 folded" "thin 2->64 folded"; do SOS_PRECISION=fp16 python tools/conv_bench.py --only "$only" 2>&1 | grep -v amdgpu; done | tee $O/conv_thin.txt
for i in 1 2 3; do
  for sw in "SOS_CONV_NO_THIN=1" "SOS_NOP=1"; do
    env $sw timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > $O/b.json
    python -c "
import json; d=json.loads(open('$O/b.json').read()); print('train $sw', round(d['value'],1), d['ms_per_step'])" | tee -a $O/ab.txt
    env $sw timeout 600 python bench.py --mode infer --precision fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > $O/b.json
    python -c "
import json; d=json.loads(open('$O/b.json').read()); print('infer $sw', round(d['value'],1), d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
timeout 2000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
