#!/bin/bash
# The ONE metered-GPU batch script (rewritten per call):  gpurun -- bash tools/gpu_job.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "doubled_batch and denoiser and bf16 and not x3" -s 2>&1 | grep -v "rel diff" | tail -40 | tee $O/pytest_doubled.txt
SOS_CONV_NO_W3=1 timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "doubled_batch and denoiser and bf16 and not x3" 2>&1 | tail -3 | tee -a $O/pytest_doubled.txt
timeout 1500 python -m pytest tests/test_gpu_block_goldens.py -q -m gpu -s 2>&1 | tail -40 | tee $O/pytest_blocks.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "band_follows" -s 2>&1 | tail -12 | tee $O/pytest_band.txt
