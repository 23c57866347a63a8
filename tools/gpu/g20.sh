cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g20; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_gpu_forced_tilings.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/probe/ab_bench.sh 3 --mode infer --precision fp16 > $O/ab_infer16.txt 2>&1; tail -1 $O/ab_infer16.txt
bash tools/probe/ab_bench.sh 3 --mode infer > $O/ab_infer_mixed.txt 2>&1; tail -1 $O/ab_infer_mixed.txt
bash tools/probe/ab_bench.sh 2 > $O/ab_train.txt 2>&1; tail -1 $O/ab_train.txt
bash tools/probe/ab_bench.sh 2 --mode infer-ragged > $O/ab_ragged.txt 2>&1; tail -1 $O/ab_ragged.txt
