cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g35; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_nets.py tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_forced_tilings.py tests/test_audiovisual.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_bench.sh 3 > $O/ab.txt 2>&1; tail -1 $O/ab.txt
bash tools/probe/ab_bench.sh 3 --mode infer > $O/ab_infer_mixed.txt 2>&1; tail -1 $O/ab_infer_mixed.txt
bash tools/probe/ab_bench.sh 2 --precision bf16x3 --steps 6 > $O/ab_x3.txt 2>&1; tail -1 $O/ab_x3.txt
