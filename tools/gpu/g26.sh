cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g26; mkdir -p $O
timeout 1500 python tools/make_tune_table.py --retune-wide > $O/retune_wide.log 2>&1
cp gpurun_out/tune_table_gfx950.txt $O/tune_table_wide.txt
cp gpurun_out/tune_table_gfx950.txt gpurun_out/tune_table_gfx950.txt.f16
timeout 1500 python tools/make_tune_table.py --retune-wgrad > $O/retune_wgrad.log 2>&1
cp gpurun_out/wgrad_table_gfx950.txt $O/wgrad_table.txt
bash tools/probe/ab_env.sh 3 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tune_table_gfx950.txt SOS_WGRAD_TUNE_CACHE=$PWD/gpurun_out/wgrad_table_gfx950.txt" > $O/ab.txt 2>&1; tail -7 $O/ab.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tune_table_gfx950.txt" --mode infer --precision fp16 > $O/ab_infer.txt 2>&1; tail -1 $O/ab_infer.txt
