cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g6; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_coresidency.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
rocprofv3 --kernel-trace --stats -d $O/prof_bn -o t --output-format csv -- python tools/bn_bench.py > $O/bn_bench.txt 2>&1
for f in $(find $O/prof_bn -name "*kernel_stats.csv"); do cp $f $O/bn_kernel_stats.csv; done
rm -rf $O/prof_bn
bash tools/probe/ab_bench.sh 3 > $O/ab_bn.txt 2>&1
SOS_BRANCH_STREAMS=1 bash tools/probe/ab_env.sh 2 "SOS_STREAM_OVERLAP=split" "SOS_STREAM_OVERLAP=full" > $O/ab_branch_full.txt 2>&1
