cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g1; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/conv_bench.py --only thin > $O/conv_thin.txt 2>&1
python tools/wgrad_bench.py --only thin > $O/wgrad_thin.txt 2>&1
python tools/wgrad_bench.py --only fold >> $O/wgrad_thin.txt 2>&1
bash tools/probe/ab_env.sh 3 "SOS_WFOLD=0" "SOS_WFOLD=1" > $O/ab_wfold.txt 2>&1
bash tools/probe/ab_env.sh 2 "SOS_WFOLD=0" "SOS_WFOLD=1" --mode infer --precision fp16 > $O/ab_wfold_infer.txt 2>&1
bash tools/probe/ab_env.sh 3 "SOS_BRANCH_STREAMS=0" "SOS_BRANCH_STREAMS=1" > $O/ab_branch.txt 2>&1
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python tools/step_breakdown.py > $O/breakdown.txt 2>&1
