#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
{
for i in 1 2; do
  echo "== A: slab DMA re-reads a piece for the pad lanes (lib before the change)"; SOS_HIP_LIB=$PWD/tools/probe/bin/libsos_prepad.so timeout 300 python tools/conv_bench.py --only "ctx96 d" --iters 20 2>/dev/null
  echo "== B: pad lanes fetch nothing"; timeout 300 python tools/conv_bench.py --only "ctx96 d" --iters 20 2>/dev/null
done
echo "== B, forced three-per-CU"; SOS_CONV_FORCE_W3=1 timeout 300 python tools/conv_bench.py --only "ctx96" --iters 20 2>/dev/null
} > $O/padskip_ab.txt 2>&1
cat $O/padskip_ab.txt
timeout 2400 python -m pytest tests/test_gpu_forced_tilings.py -x -q -m gpu -k "384 or three" 2>&1 | tail -5 | tee $O/pytest_forced_r6.txt
timeout 1500 python tools/make_tune_table.py --retune-pt3 > $O/retune_w3.log 2>&1; grep "^tune" $O/retune_w3.log | cut -c1-200
cp gpurun_out/tune_table_gfx950.txt $O/tune_r6.txt; cp gpurun_out/tune_table_gfx950.txt $O/tune_r6.txt.f16
: > $O/ab_table.txt
for i in 1 2 3; do
  for v in A B C; do
    case $v in
      A) E="SOS_HIP_LIB_F16=$PWD/tools/probe/bin/libsos_prepad_f16.so";;
      B) E="SOS_X=1";;
      C) E="SOS_CONV_TUNE_CACHE=$PWD/$O/tune_r6.txt";;
    esac
    env $E timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$v', round(d['value'], 1), 'utt/s', round(d['ms_per_step'], 2), 'ms', 'roofline', round(d['roofline']['frac'], 3))" >> $O/ab_table.txt
  done
done
cat $O/ab_table.txt
