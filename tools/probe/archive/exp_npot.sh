O=gpurun_out/n1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forced_tilings.py -x -q -k "forced_candidate" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== table"; python tools/conv_bench.py 2>&1 | grep -v amdgpu > $O/table.txt
echo "== tuned, npot off"; SOS_CONV_NPOT=0 SOS_CONV_TUNE=1 SOS_CONV_TUNE_TABLE=0 python tools/conv_bench.py 2>&1 | grep -v amdgpu > $O/tuned_pow2.txt
echo "== tuned, npot on"; SOS_CONV_TUNE=1 SOS_CONV_TUNE_TABLE=0 python tools/conv_bench.py 2>&1 | grep -v amdgpu > $O/tuned_npot.txt
paste $O/table.txt $O/tuned_pow2.txt $O/tuned_npot.txt | awk '{printf "%-28s table %s  pow2 %s  npot %s\n", $1" "$2" "$3, $4, $13, $22}' 
