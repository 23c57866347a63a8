#!/bin/bash
# LDS bank-conflict counters of the 96->96 conv kernel for the library in SOS_HIP_LIB (default: in-tree)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-new}
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex conv_mfma -d gpurun_out/pmc_lds_$T -o p --output-format csv -- python tools/conv_bench.py --only "ctx96 d1x1" --iters 3 --warm 0.05 > gpurun_out/pmc_lds_$T.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_lds_$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$T", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
