#!/bin/bash
# rocprofv3 kernel stats of the audio-visual variant (inference B=16 and one training configuration) -> gpurun_out/av
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/av; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python tools/av_bench.py > $O/av.log 2>&1
python profiles/summarize_rocpd.py $(find $O/prof -name "*.db" | head -1) $O/av_kernels.md > /dev/null 2>&1
find $O -name "*.db" -delete
cat $O/av.log | tail -6
head -45 $O/av_kernels.md | cut -c1-180
