cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g23; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_nets.py tests/test_gpu_agent.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/step_breakdown.py --top 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"])):
    n=r["Name"][:70]
    if float(r["TotalDurationNs"])/tot<0.0015: break
    print(f'{n:70s} {int(r["Calls"]):5d} {float(r["TotalDurationNs"])/1e6:8.2f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
rm -rf $O/prof
