cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g5; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_train_nets.py tests/test_gpu_forced_tilings.py tests/test_gpu_train_ops.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
rocprofv3 --kernel-trace --stats -d $O/prof_bn -o t --output-format csv -- python tools/bn_bench.py > $O/bn_bench.txt 2>&1
for f in $(find $O/prof_bn -name "*kernel_stats.csv"); do cp $f $O/bn_kernel_stats.csv; done
BN_C=48 rocprofv3 --kernel-trace --stats -d $O/prof_bn48 -o t --output-format csv -- python tools/bn_bench.py > $O/bn_bench48.txt 2>&1
for f in $(find $O/prof_bn48 -name "*kernel_stats.csv"); do cp $f $O/bn48_kernel_stats.csv; done
rm -rf $O/prof_bn $O/prof_bn48
python tools/wgrad_bench.py --only fold > $O/wgrad_fold.txt 2>&1
