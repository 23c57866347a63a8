cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g21; mkdir -p $O
bash tools/probe/ab_bench.sh 3 --mode infer --precision fp16 > $O/ab_infer16.txt 2>&1; tail -7 $O/ab_infer16.txt
bash tools/probe/ab_bench.sh 2 --mode infer > $O/ab_infer_mixed.txt 2>&1; tail -1 $O/ab_infer_mixed.txt
