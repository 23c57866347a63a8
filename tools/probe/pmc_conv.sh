#!/bin/bash
# PMC passes over the 96->96 5x5 conv kernel; prints per-dispatch averages of each counter
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (the shipped tiling table is used: no SOS_CONV_TUNE_CACHE)
CMD=${PMC_CMD:-'python tools/conv_bench.py --only "ctx96 d1x1" --iters 3 --warm 0.05'}
KREGEX=${PMC_KERNEL:-conv_mfma}
eval "$CMD" > /dev/null 2>&1

i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "$KREGEX" -d gpurun_out/pmc_$i -o p --output-format csv -- bash -c "$CMD" > gpurun_out/pmc_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:42s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
