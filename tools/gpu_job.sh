#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 600 python tools/probe/conv_d32_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_d32_debug.txt
