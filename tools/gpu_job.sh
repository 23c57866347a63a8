#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
{ echo "== streaming kernel (SOS_STFT_STREAM=1)"; SOS_STFT_STREAM=1 python tools/frontend_bench.py 2>/dev/null | grep -E "B =|stft"; echo "== register-resident matrix"; python tools/frontend_bench.py 2>/dev/null | grep -E "B =|stft"; 
  echo "== again"; SOS_STFT_STREAM=1 python tools/frontend_bench.py 2>/dev/null | grep -E "B =|stft "; python tools/frontend_bench.py 2>/dev/null | grep -E "B =|stft "; } | tee $O/stft_resident.txt
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import torch, os, sys
sys.path.insert(0, os.getcwd())
from sos_amd import transform
w = torch.randn(64, 28000, device="cuda") * 0.1
a = transform.stft_batch(w).clone()
os.environ["X"]="1"
import subprocess
torch.save(a.cpu(), "/tmp/a.pt")
PY
SOS_STFT_STREAM=1 python - <<'PY'
import torch, os, sys
sys.path.insert(0, os.getcwd())
from sos_amd import transform
torch.manual_seed(0)
PY
python - <<'PY'
import torch, os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = "import torch,sys,os; sys.path.insert(0, os.getcwd()); from sos_amd import transform; torch.manual_seed(0); w=torch.randn(64,28000,device='cuda')*0.1; torch.save(transform.stft_batch(w).cpu(), sys.argv[1])"
subprocess.run([sys.executable, "-c", code, "/tmp/res.pt"], check=True)
subprocess.run([sys.executable, "-c", code, "/tmp/str.pt"], check=True, env=dict(os.environ, SOS_STFT_STREAM="1"))
a, b = torch.load("/tmp/res.pt"), torch.load("/tmp/str.pt")
print("resident == streaming bit for bit:", torch.equal(a, b), float((a - b).abs().max()))
PY
