cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g25; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_agent.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_env.sh 3 "SOS_FUSED_STATS_UNET=0" "SOS_FUSED_STATS_UNET=1" > $O/ab.txt 2>&1; tail -7 $O/ab.txt
