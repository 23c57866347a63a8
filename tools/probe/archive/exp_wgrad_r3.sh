O=gpurun_out/e1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_ops.py -x -q -k "weight_grad" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/wgrad_bench.py > $O/wgrad_new.txt 2>&1
SOS_WGRAD_DBG=32 python tools/wgrad_bench.py > $O/wgrad_dbg32.txt 2>&1
for occ in 2 3; do SOS_WGRAD_OCC=$occ python tools/wgrad_bench.py --only thin > $O/wgrad_occ$occ.txt 2>&1; SOS_WGRAD_OCC=$occ python tools/wgrad_bench.py --only 7x1 >> $O/wgrad_occ$occ.txt 2>&1; done
cat $O/wgrad_new.txt
