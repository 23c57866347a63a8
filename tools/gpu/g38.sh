O=gpurun_out/g38; mkdir -p $O
bash tools/probe/pmc_phases.sh "ctx48 d1x1" stats > $O/phases.txt 2>&1
bash tools/probe/pmc_phases.sh "ctx48 d1x1" raw >> $O/phases.txt 2>&1
bash tools/probe/pmc_phases.sh "ctx96 d1x1" stats >> $O/phases.txt 2>&1
bash tools/probe/pmc_phases.sh "ctx96 d1x1" raw >> $O/phases.txt 2>&1
grep "per wave" $O/phases.txt | cut -c1-230
