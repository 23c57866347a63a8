cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g19; mkdir -p $O
python tools/wgrad_bench.py --only "ctx" > $O/wg_A.txt 2>&1
SOS_WGRAD_BATCH_DMA=1 python tools/wgrad_bench.py --only "ctx" > $O/wg_B.txt 2>&1
paste $O/wg_A.txt $O/wg_B.txt | cut -c1-40,80-120
python tools/wgrad_bench.py --only "inp" > $O/wg_A2.txt 2>&1
SOS_WGRAD_BATCH_DMA=1 python tools/wgrad_bench.py --only "inp" > $O/wg_B2.txt 2>&1
paste $O/wg_A2.txt $O/wg_B2.txt | cut -c1-40,80-120
bash tools/probe/ab_env.sh 3 "SOS_WGRAD_BATCH_DMA=0" "SOS_WGRAD_BATCH_DMA=1" > $O/ab.txt 2>&1; tail -7 $O/ab.txt
