#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table.
usage: summarize_rocpd.py results.db [out.md]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
tot = c.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
rows = c.execute("select name, grid_x/workgroup_x, grid_y, lds_size, vgpr_count, accum_vgpr_count, count(*), sum(end-start)/1e6, "
                 "avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name, grid_x, grid_y "
                 "order by 8 desc").fetchall()
import re
fam = {}
for r in rows:
    n = r[0]
    m = re.match(r"(?:void )?(?:at::native::)?([A-Za-z_0-9]+)", n)
    key = m.group(1) if m else n[:30]
    key = {"vectorized_elementwise_kernel": "torch elementwise", "__amd_rocclr_copyBuffer": "copyBuffer (torch copies)",
           "__amd_rocclr_fillBufferAligned": "fillBuffer (memset)"}.get(key, key)
    if key.startswith("bn_"):
        key = "BatchNorm: " + key
    f = fam.setdefault(key, [0, 0.0])
    f[0] += r[6]
    f[1] += r[7]
lines = [f"total kernel time {tot:.2f} ms over {sum(r[6] for r in rows)} dispatches", "",
         "| kernel family | calls | total ms | % |", "|---|---|---|---|"] + \
        [f"| {k} | {v[0]} | {v[1]:.2f} | {100 * v[1] / tot:.1f} |" for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]] + ["",
         "| kernel | blocks.x | grid.y | LDS B | vgpr | agpr | calls | total ms | % | avg us | min us |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows[:60]:
    lines.append(f"| {r[0][:90]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]:.2f} | {100*r[7]/tot:.1f} | {r[8]:.1f} | {r[9]:.1f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
