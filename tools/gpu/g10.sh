cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g10; mkdir -p $O
python tools/wgrad_bench.py --only "7x1" > $O/wgrad_ntb.txt 2>&1
python tools/wgrad_bench.py --only "inp 256" >> $O/wgrad_ntb.txt 2>&1
timeout 1500 python tools/make_tune_table.py --retune-wide > $O/retune_wide.log 2>&1
cp gpurun_out/tune_table_gfx950.txt $O/tune_table_wide.txt
cp gpurun_out/tune_table_gfx950.txt gpurun_out/tune_table_gfx950.txt.f16
bash tools/probe/ab_env.sh 3 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tune_table_gfx950.txt" > $O/ab_wide.txt 2>&1
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tune_table_gfx950.txt" --mode infer --precision fp16 > $O/ab_wide_infer.txt 2>&1
cat $O/wgrad_ntb.txt; tail -5 $O/retune_wide.log; tail -8 $O/ab_wide.txt; tail -3 $O/ab_wide_infer.txt
