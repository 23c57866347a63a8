#!/bin/bash
# Steady-state kernel time per training step by kernel family: two rocprofv3 kernel traces of bench.py (K1 and K2 timed steps,
# same warm-up), per-family (calls, ms) differences divided by K2 - K1 -- the first step's one-time work (plan building, weight
# packing recorders, allocator warm-up) cancels.   gpurun -- bash tools/probe/steady_families.sh [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/steady; rm -rf $O; mkdir -p $O
for K in 2 10; do
  rocprofv3 --kernel-trace -d $O/p$K -o t -- python bench.py --steps $K --warmup 1 --no-cpu-baseline --no-secondary "$@" > $O/p$K.log 2>&1
done
python - <<'PY'
import glob, re, sqlite3
def fam(db):
    c = sqlite3.connect(db)
    out = {}
    for name, n, ms in c.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name"):
        m = re.match(r"(?:void )?(?:at::native::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", name)
        k = m.group(1) if m else name[:30]
        if k in ("vectorized_elementwise_kernel", "elementwise_kernel_manual_unroll", "elementwise_kernel", "index_elementwise_kernel",
                 "reduce_kernel", "at", "CatArrayBatchedCopy", "CatArrayBatchedCopy_contig"):
            f = re.search(r"(\w+Functor|\w+_kernel_cuda|copy|Copy|fill|Fill|index|cat|Cat)", name[20:])
            k = "torch: " + (f.group(1) if f else k)
        a = out.setdefault(k, [0, 0.0]); a[0] += n; a[1] += ms
    return out
a = fam(glob.glob("gpurun_out/steady/p2/**/*.db", recursive=True)[0])
b = fam(glob.glob("gpurun_out/steady/p10/**/*.db", recursive=True)[0])
rows = []
for k in b:
    n0, t0 = a.get(k, [0, 0.0])
    rows.append((k, (b[k][0] - n0) / 8.0, (b[k][1] - t0) / 8.0))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
lines = [f"steady-state kernel time per training step (sum over both streams; kernels that co-run are each charged the shared time): {tot:.1f} ms, "
         f"{sum(r[1] for r in rows):.0f} launches", "", "| kernel family | launches / step | ms / step | % |", "|---|---|---|---|"]
lines += [f"| {k} | {n:.1f} | {t:.2f} | {100 * t / tot:.1f} |" for k, n, t in rows if n > 0.05][:45]
open("gpurun_out/steady/families.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $O -name "*.db" -delete
