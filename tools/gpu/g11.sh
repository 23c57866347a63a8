cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g11; mkdir -p $O
timeout 900 python -m pytest tests/test_abi_loads.py tests/test_gpu_train_ops.py -m gpu -x -q > $O/pytest_pre.log 2>&1; tail -2 $O/pytest_pre.log
timeout 1800 python tools/make_tune_table.py --retune-wgrad > $O/retune_wgrad.log 2>&1
cp gpurun_out/wgrad_table_gfx950.txt $O/wgrad_table.txt
bash tools/probe/ab_env.sh 3 "SOS_DUMMY=0" "SOS_WGRAD_TUNE_CACHE=$PWD/gpurun_out/wgrad_table_gfx950.txt" > $O/ab_wgrad.txt 2>&1
grep "wgrad tune" $O/retune_wgrad.log | cut -c1-330; tail -3 $O/retune_wgrad.log; tail -8 $O/ab_wgrad.txt
