#!/bin/bash
# Dynamic instruction counts of the conv kernel's phases: the ABLATE build (ab/ablate/libsos_hip.so, `make ABLATE=1`) run with
# SOS_CONV_DBG = 0 (everything), 1 (no patch staging), 2 (no tap loop), 4 (no epilogue), 8 (no slab DMA), 15 (none of them)
# under one rocprofv3 --pmc pass each; per-wave averages per mask.   pmc_phases.sh <conv_bench --only selector> [epilogue: eval|raw|stats]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SEL="$1"; EPI=${2:-eval}
export SOS_HIP_LIB=$PWD/ab/ablate/libsos_hip.so SOS_BENCH_EPI=$EPI
for DBG in 0 1 2 4 8 15; do
  for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    D=gpurun_out/pmcph_$DBG_$(echo $SET | cut -c4-12)
    SOS_CONV_DBG=$DBG rocprofv3 --pmc $SET --kernel-include-regex "conv" -d gpurun_out/pmcph/$DBG/$(echo $SET | cut -c4-14) -o p --output-format csv -- python tools/conv_bench.py --only "$SEL" --iters 3 --warm 0.05 > gpurun_out/pmcph.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
for dbg in (0, 1, 2, 4, 8, 15):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcph/{dbg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in acc.items()}
    w = a.get("SQ_WAVES", 1.0)
    print(f"$SEL [$EPI] dbg={dbg:2d} per wave:", {k[3:]: round(v / w, 1) for k, v in sorted(a.items()) if k != "SQ_WAVES"}, "waves", round(w))
PY
rm -rf gpurun_out/pmcph
