#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call; replaces round 4's tools/gpu/g*.sh):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/j38; mkdir -p $O
rm -f $O/launch_train.log
SOS_LAUNCH_LOG=$O/launch_train.log rocprofv3 --kernel-trace --stats -d $O/prof_train -o t -- python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_train.log 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_train -name "*.db" | head -1) $O/train_kernels.md $O/launch_train.log > /dev/null 2>&1
find $O -name "*.db" -delete
grep -n "signature" $O/train_kernels.md
