#!/bin/bash
# What does the data-parallel gradient path cost in a world of one (bench.py --force-buckets)?  Kernel trace: RCCL kernels, bucket copies.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/bucket; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python bench.py --force-buckets --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/bucket/t/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select substr(name,1,90), count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels where name like '%ccl%' or name like '%foreach%' or name like '%multi_tensor%' or name like '%copy%' or name like '%Copy%' or name like '%cat%' group by name order by 3 desc").fetchall()
for r in rows[:15]:
    print(f"{r[2]/6:8.3f} ms/step {r[1]/6:7.1f} calls/step {r[3]:9.1f} us avg  {r[0]}")
PY
rm -rf $O/t
