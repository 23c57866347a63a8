#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j21; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/slab_probe tools/probe/slab_probe.hip && timeout 120 /tmp/slab_probe > $O/slab_probe.txt 2>&1; echo "rc=$?" >> $O/slab_probe.txt; cat $O/slab_probe.txt
