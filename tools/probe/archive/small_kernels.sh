#!/bin/bash
# Per-kernel-NAME totals of a training profile (all launch geometries of a name merged): where do the ~2400 launches of a
# step spend their time, and how much of it is kernels shorter than 20 us?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/small; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/small/t/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tot = c.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
n = c.execute("select count(*) from kernels").fetchone()[0]
small = c.execute("select count(*), sum(end-start)/1e6 from kernels where end-start < 20000").fetchone()
print(f"total {tot:.1f} ms over {n} dispatches (4 steps); shorter than 20 us: {small[0]} dispatches, {small[1]:.1f} ms")
rows = c.execute("select substr(name,1,70), count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
for r in rows[:70]:
    print(f"{r[2]/4:8.3f} ms/step {r[1]//4:5d} calls/step {r[3]:9.1f} us avg  {r[0]}")
PY
rm -rf $O/t
