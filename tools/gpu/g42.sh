cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g42; mkdir -p $O
timeout 2400 python -m pytest tests/test_abi_loads.py tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_agent.py tests/test_gpu_determinism.py tests/test_gpu_ddp_agents.py tests/test_audiovisual.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/probe/ab_env.sh 3 "SOS_WGRAD_DEFER=0" "SOS_WGRAD_DEFER=1" > $O/ab.txt 2>&1; tail -7 $O/ab.txt
