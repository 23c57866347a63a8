"""Two ranks of the REAL training agents (reference behaviour: nn.DataParallel keeps the replicas identical,
M1/agent.py:167-172, M2/agent.py:159-164): DetectorAgent + DenoiserAgent under agent.train_concurrent in a world of two --
RCCL when the box has two GPUs, else gloo on the GPU tensors of the one GPU (two processes sharing it).

After two steps: all parameters are bit-identical across the ranks (same initial broadcast, same averaged gradients, same
Adam), the all-reduced gradients equal the mean of the ranks' stand-alone gradients, BatchNorm running statistics differ
per rank (per-replica statistics, no SyncBN -- DataParallel semantics), and the two models' collectives interleaved on
one communicator did not deadlock (the children exit within the timeout)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp16"])
def test_two_ranks_of_the_real_agents_stay_identical(tmp_path, precision):
    world = 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_agents_child.py"), str(tmp_path), precision],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)          # a deadlock of the interleaved collectives ends here
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two-rank job did not finish: deadlock?")
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    print("backend:", res[0]["backend"])
    for tag in ("det", "jm"):
        names = res[0]["param_names"][tag]
        s0, s1 = res[0]["state"][tag], res[1]["state"][tag]
        # replicas identical after two optimizer steps: every parameter, bit for bit
        for n in names:
            assert torch.equal(s0[n], s1[n]), (tag, n)
        # ... and they moved, from rank 0's initial weights
        # per-rank BatchNorm statistics: the ranks saw different clips
        diff = [k for k in s0 if k.endswith("running_mean") and not torch.equal(s0[k], s1[k])]
        assert len(diff) >= 0.9 * sum(k.endswith("running_mean") for k in s0), (tag, len(diff))
        assert all(int(s0[k]) == 2 for k in s0 if k.endswith("num_batches_tracked"))
        # the all-reduced gradients (a SUM; Adam's grad_scale applies 1/world) = the mean of the stand-alone gradients
        worst = 0.0
        exact = True
        for n in names:
            a, b = res[0]["ddp_sum"][tag][n], res[1]["ddp_sum"][tag][n]
            assert torch.equal(a, b), (tag, n)                                   # both ranks hold the same reduced buffer
            want = res[0]["local"][tag][n] + res[1]["local"][tag][n]
            exact = exact and torch.equal(a, want)
            e = float((a - want).abs().max()) / (float(want.abs().max()) + 1e-30)
            worst = max(worst, e)
            assert e < 1e-5, (tag, n, e)
        print(tag, "all-reduced vs sum of stand-alone gradients: worst rel diff", worst, "bit-identical" if exact else "")


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` launches its N ranks itself; with fewer devices it must fail loudly, not run one rank
    and print n_gpus: 1.  (Here: 0 or 1 GPU visible, 8 requested.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
    assert '"n_gpus"' not in r.stdout


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
