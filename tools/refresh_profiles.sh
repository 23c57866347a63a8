#!/bin/bash
# Run on the GPU box (gpurun -- tools/refresh_profiles.sh [round tag]): regenerates the measured artefacts that profiles/
# keeps (written to gpurun_out/refresh/, copy from there):
#   bench JSON lines (train = BASELINE configs[1] in fp16 / bf16 / bf16x3, infer, infer-ragged = configs[3]),
#   rocprofv3 --kernel-trace --stats summaries of the train / infer / ragged benches,
#   the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the dominant conv kernel,
#   the micro-benchmarks.  Every process loads the shipped tiling table: nothing is tuned here.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; mkdir -p $O
python tools/conv_bench.py > $O/conv_bench.txt 2>&1
python tools/wgrad_bench.py > $O/wgrad_bench.txt 2>&1
python tools/lstm_bench.py > $O/lstm_bench.txt 2>&1
python tools/bn_bench.py > $O/bn_bench.txt 2>&1
python tools/frontend_bench.py > $O/frontend_bench.txt 2>&1
rm -f $O/launch_train.log
# (--serial: every kernel alone on the chip, like the serial pre-pass bench.py takes roofline.avg_ms in -- the per-signature averages of
# this trace are the ones to compare with the bench line; the concurrent schedule is profiled by steady_families.sh below)
SOS_LAUNCH_LOG=$O/launch_train.log rocprofv3 --kernel-trace --stats -d $O/prof_train -o t -- python bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_train.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_infer -o t -- python bench.py --mode infer --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_infer.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_ragged -o t -- python bench.py --mode infer-ragged --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_ragged.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-include-regex conv_mfma -d $O/pmc_$n -o p --output-format csv -- python tools/conv_bench.py --only "ctx96 d1x1" --iters 3 --warm 0.05 > $O/pmc_$n.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/refresh/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
B, H, W, C = 64, 256, 178, 96
out = {"kernel": "conv_mfma_kernel 96->96 5x5 B=64 (tools/conv_bench.py --only 'ctx96 d1x1')",
       "FETCH_SIZE_KB": avg.get("FETCH_SIZE"), "WRITE_SIZE_KB": avg.get("WRITE_SIZE"),
       "hbm_bytes_per_launch": (2 * avg.get("FETCH_SIZE", 0) + avg.get("WRITE_SIZE", 0)) * 1024,
       "algorithmic_bytes_per_launch": 2 * B * H * W * C * 2,
       "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md HBM section)",
       "l2_hit": avg.get("TCC_HIT_sum"), "l2_miss": avg.get("TCC_MISS_sum")}
json.dump(out, open("gpurun_out/refresh/pmc_conv96.json", "w"), indent=1)
print(json.dumps(out))
PY
cp $O/pmc_conv96.json profiles/${R}_pmc_conv96.json     # bench.py reports this record as roofline.traffic
python bench.py --steps 10 --warmup 3 > $O/bench_train_fp16.json 2> $O/bench_train_fp16.err
python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
python bench.py --precision bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_train_bf16x3.json 2> $O/bench_train_bf16x3.err
python bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_infer_mixed.json 2> $O/bench_infer_mixed.err
python bench.py --mode infer --precision fp16 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_infer_fp16.json 2> $O/bench_infer_fp16.err
python bench.py --mode infer-ragged --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_ragged_mixed.json 2> $O/bench_ragged_mixed.err
bash tools/probe/steady_families.sh > $O/steady_families.txt 2>&1
bash tools/probe/archive/low_occupancy.sh > $O/low_occupancy.txt 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_train -name "*.db" | head -1) $O/train_kernels.md $O/launch_train.log > /dev/null 2>&1
for m in infer ragged; do python profiles/summarize_rocpd.py $(find $O/prof_$m -name "*.db" | head -1) $O/${m}_kernels.md > /dev/null 2>&1; done
find $O -name "*.db" -delete            # the summaries stay; gpurun copies at most 64 MiB back
for f in $O/bench_*.json; do echo $f; cut -c1-420 $f; done
