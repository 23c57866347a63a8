#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j04; mkdir -p $O
SOS_CONV16_MODE=3 timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_train_nets.py tests/test_gpu_pipeline.py -m gpu -q -x -k "not train_step" > $O/pytest_m3.log 2>&1; tail -3 $O/pytest_m3.log
for m in "" 1 0 3; do
  echo "== SOS_CONV16_MODE=$m" >> $O/conv48.txt
  if [ -z "$m" ]; then python tools/conv_bench.py --only ctx48 --iters 20 >> $O/conv48.txt 2>&1; SOS_BENCH_EPI=stats python tools/conv_bench.py --only "ctx48 d1x1" --iters 20 >> $O/conv48.txt 2>&1
  else SOS_CONV16_MODE=$m python tools/conv_bench.py --only ctx48 --iters 20 >> $O/conv48.txt 2>&1; SOS_CONV16_MODE=$m SOS_BENCH_EPI=stats python tools/conv_bench.py --only "ctx48 d1x1" --iters 20 >> $O/conv48.txt 2>&1; fi
done
cat $O/conv48.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV16_MODE=3" > $O/ab_m3_train.txt 2>&1; tail -1 $O/ab_m3_train.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV16_MODE=3" --mode infer --precision fp16 > $O/ab_m3_infer.txt 2>&1; tail -1 $O/ab_m3_infer.txt
