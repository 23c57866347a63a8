cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g13; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py tests/test_gpu_nets.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
bash tools/probe/ab_bench.sh 3 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
SOS_HIP_LIB=$PWD/ab/base/libsos_hip.so python tools/conv_bench.py --only "inp" > $O/conv_A.txt 2>&1
python tools/conv_bench.py --only "inp" > $O/conv_B.txt 2>&1
paste $O/conv_A.txt $O/conv_B.txt | cut -c1-40,80-140
