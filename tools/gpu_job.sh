#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j32; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k "pipelined or two_pass" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read()); s=d['secondary']
print(round(d['value'],1), d['roofline']['frac'])
print({k:v for k,v in s.items() if k!='note' and 'roofline' not in k and 'runs' not in k})"
for m in "infer --precision fp16" "infer" "infer-ragged --steps 5 --warmup 2"; do python bench.py --mode $m --no-cpu-baseline --no-secondary 2>/dev/null > $O/bench_$(echo $m | tr ' -' '__').json; cut -c1-200 $O/bench_$(echo $m | tr ' -' '__').json; done
