cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g40; mkdir -p $O
for V in c48 c96 rest; do
  bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tvar/$V.txt" > $O/train_$V.txt 2>&1; echo "train $V: $(tail -1 $O/train_$V.txt)"
  bash tools/probe/ab_env.sh 2 "SOS_DUMMY=0" "SOS_CONV_TUNE_CACHE=$PWD/gpurun_out/tvar/$V.txt" --mode infer --precision fp16 > $O/infer_$V.txt 2>&1; echo "infer $V: $(tail -1 $O/infer_$V.txt)"
done
