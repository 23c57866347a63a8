cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g9; mkdir -p $O
SOS_WGRAD_VERBOSE=1 python tools/probe/wgrad_cfg_sweep.py > $O/wgrad_cfg_sweep.txt 2>&1
grep -v "^sos_conv2d_wgrad" $O/wgrad_cfg_sweep.txt | tail -150
