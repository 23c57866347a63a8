cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g12; mkdir -p $O
timeout 900 python -m pytest tests/test_abi_loads.py tests/test_gpu_train_ops.py -m gpu -x -q > $O/pytest_pre.log 2>&1; tail -2 $O/pytest_pre.log
timeout 2400 python tools/make_tune_table.py --retune-wide-all > $O/retune_wide_all.log 2>&1
tail -3 $O/retune_wide_all.log
cp gpurun_out/tune_table_gfx950.txt $O/tune_table_wide_all.txt
cp gpurun_out/tune_table_gfx950.txt gpurun_out/tune_table_gfx950.txt.f16
T="SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tune_table_gfx950.txt"
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "$T" > $O/ab_train.txt 2>&1; tail -1 $O/ab_train.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "$T" --mode infer > $O/ab_infer_mixed.txt 2>&1; tail -1 $O/ab_infer_mixed.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "$T" --mode infer-ragged > $O/ab_ragged.txt 2>&1; tail -1 $O/ab_ragged.txt
bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "$T" --precision mixed > $O/ab_train_mixed.txt 2>&1; tail -1 $O/ab_train_mixed.txt
