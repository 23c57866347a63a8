O=gpurun_out/g17; mkdir -p $O
export SOS_BENCH_EPI=stats
bash tools/probe/pmc_kernel.sh c96 conv "ctx96 d1x1" > $O/pmc.txt 2>&1
bash tools/probe/pmc_kernel.sh c48 conv "ctx48 d1x1" >> $O/pmc.txt 2>&1
bash tools/probe/pmc_kernel.sh u256 conv "inp 256->256 3x3 d1" >> $O/pmc.txt 2>&1
bash tools/probe/pmc_kernel.sh c967 conv "ctx96 7x1" >> $O/pmc.txt 2>&1
grep "^c96\|^c48\|^u256\|^c967" $O/pmc.txt
