#!/bin/bash
# SQ counters of one micro-benchmark: pmc_kernel.sh <tag> <kernel regex> <--only selector> [bench script, default tools/conv_bench.py]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$1; RX=$2; SEL=$3; BENCH=${4:-tools/conv_bench.py}
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "$RX" -d gpurun_out/pmc_${T}_$i -o p --output-format csv -- python $BENCH --only "$SEL" --iters 3 --warm 0.05 > gpurun_out/pmc_${T}_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_${T}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$T", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf gpurun_out/pmc_${T}_*/
