cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g3; mkdir -p $O
python tools/conv_bench.py --only ctx48 > $O/conv48_default.txt 2>&1
SOS_CONV16_FORCE512=1 python tools/conv_bench.py --only ctx48 > $O/conv48_force512.txt 2>&1
timeout 1500 python tools/make_tune_table.py --retune-16row > $O/retune.log 2>&1
cp gpurun_out/tune_table_gfx950.txt $O/tune_table_new.txt; cp gpurun_out/tune_table_gfx950.txt gpurun_out/tune_table_gfx950.txt.f16
SOS_CONV_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_table_gfx950.txt python tools/conv_bench.py --only ctx48 > $O/conv48_retuned.txt 2>&1
SOS_CONV_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_table_gfx950.txt python tools/conv_bench.py --only thin > $O/conv_thin_retuned.txt 2>&1
bash tools/probe/ab_env.sh 3 "SOS_X=0" "SOS_CONV_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_table_gfx950.txt" > $O/ab_table.txt 2>&1
bash tools/probe/ab_env.sh 2 "SOS_X=0" "SOS_CONV_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/tune_table_gfx950.txt" --mode infer > $O/ab_table_infer.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1
