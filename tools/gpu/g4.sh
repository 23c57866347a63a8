cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/g4; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
bash tools/probe/pmc_kernel.sh c48_pt4 conv16_kernel "ctx48 d1x1" > $O/pmc_c48_pt4.txt 2>&1
SOS_CONV16_FORCE512=1 bash tools/probe/pmc_kernel.sh c48_pt8 conv16_kernel "ctx48 d1x1" > $O/pmc_c48_pt8.txt 2>&1
for occ in 1 2 3 4; do echo "== SOS_WGRAD_OCC=$occ"; SOS_WGRAD_OCC=$occ python tools/wgrad_bench.py --only fold 2>&1 | grep -v amdgpu; SOS_WGRAD_OCC=$occ python tools/wgrad_bench.py --only thin 2>&1 | grep -v amdgpu; done > $O/wgrad_occ.txt 2>&1
