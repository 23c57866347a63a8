"""Determinism and thread safety of the boundary (SURVEY.md 8-b threading row; the reference wraps its networks in
nn.DataParallel -- one host thread per replica calling forward concurrently, M1/predict.py:58-63, M2/predict.py:392-397).

* Two FRESH processes produce bit-identical outputs, gradients and running statistics: every process loads the shipped
  tiling table (or falls back to the deterministic cost-model pick), nothing is chosen by a timing race, and no kernel
  uses floating-point atomics.
* Two host threads running inference concurrently on their own HIP streams through the SAME modules (shared packed
  weights, shared tiling table in the .so) reproduce the serial results bit for bit."""
import os
import subprocess
import sys
import threading

import pytest
import torch

import sos_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(precision, B, T):
    env = {k: v for k, v in os.environ.items() if not k.startswith("SOS_CONV_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_det_child.py"), precision, str(B), str(T)], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return dict(ln.split() for ln in r.stdout.splitlines() if len(ln.split()) == 2)


@pytest.mark.parametrize("precision,B,T", [("fp16", 2, 89), ("bf16", 1, 178)])
def test_two_fresh_processes_are_bit_identical(precision, B, T):
    a, b = _child(precision, B, T), _child(precision, B, T)
    assert len(a) == 8 and a.keys() == b.keys()
    assert a == b, {k: (a[k][:12], b[k][:12]) for k in a if a[k] != b[k]}


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_two_host_threads_match_serial(precision):
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision(precision)
    try:
        torch.manual_seed(0)
        det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
        base = torch.from_numpy(synth_batch(500, 6)["mixed"]).cuda()
        # different shapes per thread: both threads miss / hit the tiling table and the per-kernel attribute setup at
        # the same time
        inputs = [base[:4].contiguous(), base[1:4, :14000].contiguous()]
        serial = [pipeline.denoise(det, jm, x) for x in inputs]
        torch.cuda.synchronize()
        results, errors = [None, None], []
        start = threading.Barrier(2)

        def work(i):
            try:
                st = torch.cuda.Stream()
                start.wait()
                with torch.cuda.stream(st):
                    outs = [pipeline.denoise(det, jm, inputs[i]) for _ in range(6)]
                st.synchronize()
                results[i] = outs
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        assert not errors, errors
        for i in range(2):
            assert results[i] is not None
            for y in results[i]:
                assert torch.equal(y, serial[i])
    finally:
        sos_amd.set_precision("bf16")


def test_fresh_threads_first_use_races():
    """First use of every kernel variant from two threads at once (nothing warmed up in this process for these shapes):
    the per-device attribute setup and the tiling table are touched concurrently."""
    from sos_amd import engine as E
    from sos_amd import _lib as L
    outs, errors = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                src = E.Act(3, 40, 36, 48, False, torch.device("cuda"))
                src.t.copy_(torch.linspace(-1, 1, src.t.numel(), device="cuda").reshape(src.t.shape).to(src.t.dtype))
                w = E.pack_weight(torch.linspace(-0.1, 0.1, 48 * 48 * 25).reshape(48, 48, 5, 5).cuda(), 48, False)
                dst = E.Act(3, 40, 36, 48, False, torch.device("cuda"))
                start.wait()
                for _ in range(4):
                    E.conv_to_act(src, 0, 48, w, 5, 5, 48, None, None, L.ACT_NONE, dst, cout_store=48, dil=(2, 2), pad=(4, 4),
                                  Ho=40, Wo=36)
            st.synchronize()
            outs[i] = dst.t.float().cpu()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errors, errors
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0


def test_steady_state_steps_do_not_synchronise_with_the_host():
    """A training step and the inference pipelines enqueue their launches without waiting for the device: under
    torch.cuda.set_sync_debug_mode('error') every synchronising torch call (a pageable host->device copy, .item(), ...)
    raises.  (One such copy per step used to hold the host until the stage-2 backward had drained; per-call tables held
    the ragged pipeline 24 times per batch.)  Warm-up runs first: plans, tables and workspaces are built there."""
    from sos_amd import agent, pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch, synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("fp16")
    try:
        torch.manual_seed(0)
        ad = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
        aj = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
        bd, bj = make_batch("detector", 100, 4), make_batch("denoiser", 200, 4)
        det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
        base = torch.from_numpy(synth_batch(300, 4)["mixed"]).cuda()
        clips = [base[0, :14000].contiguous(), base[1], base[2, :20000].contiguous(), base[3]]
        for _ in range(2):
            ad.train_func(bd); aj.train_func(bj)
            pipeline.denoise(det, jm, base)
            pipeline.denoise_ragged(det, jm, clips)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            ad.train_func(bd); aj.train_func(bj)
            agent.train_concurrent([(aj, bj), (ad, bd)])
            pipeline.denoise(det, jm, base)
            pipeline.denoise_ragged(det, jm, clips)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    finally:
        sos_amd.set_precision("bf16")
