#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
python -c "
import json; d=json.loads(open('$O/bench.json').read()); s=d['secondary']
print({k:v for k,v in s.items() if k!='note' and 'roofline' not in k})
print(d['roofline'])
"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
