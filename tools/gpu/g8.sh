cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_stdout.txt 2>&1
python tools/step_breakdown.py --top 300 > gpurun_out/refresh/breakdown_full.txt 2>&1
tail -20 gpurun_out/refresh_stdout.txt
