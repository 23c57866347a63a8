cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g34; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_nets.py tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_forced_tilings.py tests/test_audiovisual.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_bench.sh 3 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
bash tools/probe/ab_bench.sh 3 --mode infer --precision fp16 > $O/ab_infer.txt 2>&1; tail -1 $O/ab_infer.txt
