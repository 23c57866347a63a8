cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g16; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_audiovisual.py tests/test_gpu_agent.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_bench.sh 3 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
SOS_HIP_LIB=$PWD/ab/base/libsos_hip.so python tools/wgrad_bench.py --only "ctx" > $O/wg_A.txt 2>&1
python tools/wgrad_bench.py --only "ctx" > $O/wg_B.txt 2>&1
paste $O/wg_A.txt $O/wg_B.txt | cut -c1-40,80-120
