#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 900 python tools/probe/torch_ops_in_step.py 2>&1 | grep -v amdgpu.ids | tail -110 > $O/torch_ops.txt; cat $O/torch_ops.txt | cut -c1-330
