#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table.
usage: summarize_rocpd.py results.db [out.md [launch_log]]"""
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
# ---- one row per (kernel, layer signature): join with the engine's launch log (SOS_LAUNCH_LOG, one line per conv / wgrad launch
# in enqueue order): the k-th dispatch of the conv family (conv_mfma_kernel / conv16_kernel / conv_thin_kernel) is the k-th "conv" line, the k-th
# dispatch of the weight-gradient family (wgrad_kernel / wgrad16_kernel / wgrad_gemm_kernel / wgrad_thin*_kernel; not the reduce) the k-th "wgrad" line
if len(sys.argv) > 3:
    log = [ln.rstrip("\n").split("|") for ln in open(sys.argv[3]) if "|" in ln]
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    order = "dispatch_id" if "dispatch_id" in cols else "start"
    for kind, pat in (("conv", r"conv_mfma_kernel|conv16_kernel|conv_thin_kernel"), ("wgrad", r"wgrad_kernel|wgrad16_kernel|wgrad_gemm_kernel|wgrad_thin_kernel|wgrad_thin_taps_kernel")):
        disp = [r for r in c.execute(f"select name, end-start from kernels order by {order}").fetchall() if re.search(pat, r[0])]
        ents = [e for e in log if e[0] == kind]
        lines += ["", f"### {kind} launches by layer signature ({len(disp)} dispatches, {len(ents)} logged launches)"]
        if len(disp) != len(ents):
            lines.append("(counts differ: the log and the trace are not from the same process -- no per-signature rows)")
            continue
        agg = {}
        for (name, ns), (_, sig, flops) in zip(disp, ents):
            m = re.search(r"(conv_mfma_kernel|conv16_kernel|conv_thin_kernel|wgrad16_kernel|wgrad_gemm_kernel|wgrad_thin_taps_kernel|wgrad_thin_kernel|wgrad_kernel)(<[^>]*>)?", name)
            a = agg.setdefault((m.group(0) if m else name[:40], sig), [0, 0.0, 1e30, float(flops)])
            a[0] += 1
            a[1] += ns
            a[2] = min(a[2], ns)
        hdr = "kh,kw,dil_h,dil_w,stride,cin,cout,B,Ho,Wo" if kind == "conv" else "kh,kw,dil_h,dil_w,stride,M,N,B,Hg,Wg"
        lines += [f"| kernel | signature ({hdr}) | calls | total ms | avg us | min us | TFLOP/s (avg) | of 2.5 PF |", "|---|---|---|---|---|---|---|---|"]
        for (kn, sig), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
            tf = a[3] / (a[1] / a[0] * 1e-9) / 1e12
            lines.append(f"| {kn} | {sig} | {a[0]} | {a[1] / 1e6:.2f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {tf:.0f} | {tf / 2500:.3f} |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
