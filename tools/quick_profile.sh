#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of the training and inference bench (fp16, shipped tilings) -> gpurun_out/$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-qprof}; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_train -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_train.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_infer -o t -- python bench.py --mode infer --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_infer.log 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_train -name "*.db" | head -1) $O/train_kernels.md > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_infer -name "*.db" | head -1) $O/infer_kernels.md > /dev/null 2>&1
find $O -name "*.db" -delete
head -30 $O/train_kernels.md | cut -c1-200
grep -E "stft|istft" $O/infer_kernels.md | cut -c1-200
