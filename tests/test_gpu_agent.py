"""Trainer semantics on the GPU: fused losses, Adam step vs torch.optim.Adam, checkpoint format."""
import os

import numpy as np
import pytest
import torch

from oracle import nets as onet
from util import hashed, rel_err, silent_gate, spec_input

pytestmark = pytest.mark.gpu


def test_fused_losses_match_torch():
    from sos_amd import agent
    a = torch.from_numpy(hashed(61, (3, 2, 32, 20)).astype(np.float32)).cuda().requires_grad_(True)
    b = torch.from_numpy(hashed(62, (3, 2, 32, 20)).astype(np.float32)).cuda()
    l = agent.mse_loss(a, b)
    (3.0 * l).backward()
    ar = a.detach().cpu().requires_grad_(True)
    lr = torch.nn.functional.mse_loss(ar, b.cpu())
    (3.0 * lr).backward()
    assert abs(float(l) - float(lr)) < 1e-6 and rel_err(a.grad, ar.grad) < 1e-6
    x = torch.from_numpy(hashed(63, (4, 60), 3.0).astype(np.float32)).cuda().requires_grad_(True)
    y = (torch.from_numpy(hashed(64, (4, 60))) > 0).float().cuda()
    l = agent.bce_with_logits_loss(x, y)
    l.backward()
    xr = x.detach().cpu().requires_grad_(True)
    lr = torch.nn.functional.binary_cross_entropy_with_logits(xr, y.cpu())
    lr.backward()
    assert abs(float(l) - float(lr)) < 1e-6 and rel_err(x.grad, xr.grad) < 1e-6


def test_fused_adam_matches_torch_adam():
    from sos_amd.agent import FusedAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(17, 5)), torch.nn.Parameter(torch.randn(300))]
    pr = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ps = [torch.nn.Parameter(p.detach().cuda()) for p in ps]
    o1, o2 = FusedAdam(ps, lr=1e-3), torch.optim.Adam(pr, lr=1e-3)
    for it in range(5):
        for p, q in zip(ps, pr):
            g = torch.randn(q.shape)
            p.grad, q.grad = g.cuda(), g.clone()
        v0 = ps[0]._version
        o1.step()
        o2.step()
        assert ps[0]._version > v0                    # packed-weight caches see the update
    for p, q in zip(ps, pr):
        assert rel_err(p, q) < 1e-6
    sd = o1.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["lr"] == 1e-3


def test_agents_train_step_and_checkpoint(tmp_path):
    import sos_amd
    from sos_amd import agent
    from sos_amd.common import MyConfig
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    B, T = 2, 89
    x = spec_input(100 + B, B, T)
    batch2 = {"mixed": x, "noise": silent_gate(x), "clean": spec_input(300, B, T) * 0.5,
              "full_noise": x - spec_input(300, B, T) * 0.5}
    det = dnet.get_network()
    det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    ag1 = agent.DetectorAgent(det, lr=1e-3, model_dir=str(tmp_path))
    label = (torch.from_numpy(hashed(301, (B, 60))) > 0).float()
    losses = [float(ag1.train_func({"label": label, "audio": x})[1]["bce"]) for _ in range(4)]
    print("detector bce over 4 steps", losses)
    assert losses[-1] < losses[0]                       # same batch: Adam must reduce the loss
    jm = jnet.get_network(MyConfig())
    jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    ag2 = agent.DenoiserAgent(jm, lr=1e-3, model_dir=str(tmp_path))
    tot = []
    for _ in range(3):
        _, ls = ag2.train_func(batch2)
        tot.append(float(ls["stage1"]) + float(ls["stage2"]))
    print("denoiser loss over 3 steps", tot)
    assert tot[-1] < tot[0]
    # eval after training uses the UPDATED weights and running statistics
    (n_pred, out), ls = ag2.val_func(batch2)
    assert torch.isfinite(out).all() and torch.isfinite(n_pred).all()
    ag2.clock.tick()
    path = ag2.save_ckpt()
    ck = torch.load(path, map_location="cpu")
    # the reference's four keys (M1/agent.py:62-78) + the fp16 mode's overflow-guard state (round 4: a resumed run keeps its
    # loss-scale back-off; a reference-side loader indexes the four keys by name and never sees the fifth)
    assert set(ck.keys()) == {"clock", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "overflow_guard"}
    assert list(ck["model_state_dict"].keys()) == [k for k, _, _ in onet.joint_spec()]
    assert ck["clock"] == {"epoch": 1, "minibatch": 1, "step": 1}
    jm2 = jnet.get_network(MyConfig())
    ag3 = agent.DenoiserAgent(jm2, lr=1e-3, model_dir=str(tmp_path))
    ag3.load_ckpt(1)
    (n2, o2), _ = ag3.val_func(batch2)
    assert torch.equal(o2, out) and torch.equal(n2, n_pred)
    with pytest.raises(ValueError):
        ag3.load_ckpt(99)


@pytest.mark.parametrize("branch", [False, True], ids=["two-streams", "branch-streams"])
def test_train_concurrent_matches_serial(branch, monkeypatch):
    """The two models trained on two HIP streams (agent.train_concurrent) end up bit-identical to back-to-back
    training: the kernels are deterministic and the agents share no mutable device state.  branch-streams: the denoiser
    additionally runs its [stage 1 -> encoder_n] branch on a side stream beside encoder_x (JointModel.BRANCH_STREAMS), forward
    and backward -- same kernels in the same order per branch, so still bit-identical to the serial schedule."""
    import sos_amd
    from sos_amd import agent
    from sos_amd.common import MyConfig
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("bf16")
    B, T = 2, 89
    x = spec_input(100 + B, B, T).cuda()
    clean = (spec_input(300, B, T) * 0.5).cuda()
    bj = {"mixed": x, "noise": silent_gate(x.cpu()).cuda(), "clean": clean, "full_noise": x - clean}
    bd = {"label": (torch.from_numpy(hashed(301, (B, 60))) > 0).float().cuda(), "audio": x}

    def make():
        det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
        jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
        return agent.DetectorAgent(det, lr=1e-3), agent.DenoiserAgent(jm, lr=1e-3)

    d1, j1 = make()
    d2, j2 = make()
    for _ in range(3):
        monkeypatch.setattr(jnet.JointModel, "BRANCH_STREAMS", False)
        d1.train_func(bd); j1.train_func(bj)
        monkeypatch.setattr(jnet.JointModel, "BRANCH_STREAMS", branch)
        agent.train_concurrent([(j2, bj), (d2, bd)])
    torch.cuda.synchronize()
    assert bool(j2.net.__dict__.get("_used_side_stream")) == branch
    for a, b in ((d1, d2), (j1, j2)):
        for (k, p), (_, q) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
            assert torch.equal(p, q), k


def test_deferred_weight_gradient_reductions_are_bit_identical(monkeypatch):
    """engine.wgrad(defer=True) inside engine.deferred_reductions() (SOS_WGRAD_DEFER=1, ABI 8: sos_conv2d_wgrad_partial on the
    model's stream, sos_conv2d_wgrad_reduce on a side stream behind an event, a ring of workspaces, a join before the gradients
    reach autograd): three training steps of both models under train_concurrent end bit-identical to the same steps with every
    reduce on the launching stream -- the same kernels in the same order, only the stream of the small one differs."""
    import sos_amd
    from sos_amd import agent, engine
    from sos_amd.common import MyConfig
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("fp16")
    try:
        B, T = 2, 89
        x = spec_input(100 + B, B, T).cuda()
        clean = (spec_input(300, B, T) * 0.5).cuda()
        bj = {"mixed": x, "noise": silent_gate(x.cpu()).cuda(), "clean": clean, "full_noise": x - clean}
        bd = {"label": (torch.from_numpy(hashed(301, (B, 60))) > 0).float().cuda(), "audio": x}

        def make():
            det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
            jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
            return agent.DetectorAgent(det, lr=1e-3), agent.DenoiserAgent(jm, lr=1e-3)

        d1, j1 = make()
        d2, j2 = make()
        for _ in range(3):
            monkeypatch.setattr(engine, "WGRAD_DEFER", False)
            agent.train_concurrent([(j1, bj), (d1, bd)])
            monkeypatch.setattr(engine, "WGRAD_DEFER", True)
            agent.train_concurrent([(j2, bj), (d2, bd)])
        torch.cuda.synchronize()
        # (the state itself is per host thread and backward runs on the autograd engine's thread: process-wide counters)
        assert engine.WG_STATS["deferred"] > 0, "no reduce was deferred"
        assert engine.WG_STATS["joined"] > 0, "a deferred reduce was never joined"
        for a, b in ((d1, d2), (j1, j2)):
            for (k, p), (_, q) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
                assert torch.equal(p, q), k
    finally:
        sos_amd.set_precision("bf16")


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["SOS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SOS_ROOT"], "tests"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import sos_amd
from sos_amd import agent
from sos_amd.common import MyConfig
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
from oracle import nets as onet
from util import hashed, silent_gate, spec_input
B, T = 2, 89
x = spec_input(100 + B, B, T).cuda()
clean = (spec_input(300, B, T) * 0.5).cuda()
bj = {"mixed": x, "noise": silent_gate(x.cpu()).cuda(), "clean": clean, "full_noise": x - clean}
bd = {"label": (torch.from_numpy(hashed(301, (B, 60))) > 0).float().cuda(), "audio": x}
def make():
    det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    return agent.DetectorAgent(det, lr=1e-3), agent.DenoiserAgent(jm, lr=1e-3)
os.environ["SOS_FORCE_BUCKETS"] = "1"
d1, j1 = make()
assert d1.bucketer is not None and d1.bucketer.collective
os.environ["SOS_FORCE_BUCKETS"] = "0"
d2, j2 = make()
assert d2.bucketer is None
for _ in range(2):
    agent.train_concurrent([(j1, bj), (d1, bd)])
    agent.train_concurrent([(j2, bj), (d2, bd)])
torch.cuda.synchronize()
n = 0
for a, b in ((d1, d2), (j1, j2)):
    for (k, p), (_, q) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert torch.equal(p, q), k
        n += 1
dist.barrier(); dist.destroy_process_group()
print("RCCL_PATH_OK", n)
"""


@pytest.mark.parametrize("branch", ["0", "1"], ids=["two-streams", "branch-streams"])
def test_bucketed_all_reduce_path_on_rccl_world_of_one(branch):
    """The multi-GPU gradient path (GradBucketer: flat buckets, async all_reduce on the RCCL stream issued from the
    two model streams -- one communicator per model --, wait, Adam on the bucket views) with backend 'nccl' in a world of one,
    where the all-reduce is the identity: parameters after two steps are bit-identical to the single-process path.
    branch-streams: gradients are produced on the denoiser's side stream as well (the bucketer waits for the producing stream
    before its batched copy)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOS_ROOT=root, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0", SOS_BRANCH_STREAMS=branch)
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_detector_learns_the_synthetic_task(precision):
    """End-to-end sanity beyond one-step parity: 40 Adam steps on fresh synthetic batches (speech bursts gated by the
    frame labels + coloured noise) lower the BCE loss of the silent-interval detector and lift its frame accuracy
    above the majority-class rate."""
    import sos_amd
    from sos_amd import agent
    from sos_amd.dataset import make_batch
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision(precision)
    torch.manual_seed(0)
    ag = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
    losses = []
    for it in range(40):
        out, ls = ag.train_func(make_batch("detector", 5000 + 16 * it, 16))
        losses.append(float(ls["bce"]))
    test = make_batch("detector", 90000, 32)
    out, _ = ag.val_func(test)
    acc = float(((out >= 0) == (test["label"] > 0.5)).float().mean())
    base = float(max(test["label"].mean(), 1 - test["label"].mean()))
    print("detector losses", [round(x, 3) for x in losses[::5]], "val accuracy", acc, "majority", base)
    sos_amd.set_precision("bf16")
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]) and acc > base + 0.02


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_denoiser_learns_the_synthetic_task(precision):
    """60 Adam steps on fresh synthetic batches lower both MSE terms of the two-stage denoiser, and the masked output of
    a held-out batch ends up closer to the clean spectrogram than the noisy input is."""
    import sos_amd
    from sos_amd import agent, transform
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch
    from sos_amd.denoiser import networks as jnet
    sos_amd.set_precision(precision)
    torch.manual_seed(0)
    ag = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
    l1, l2 = [], []
    for it in range(60):
        _, ls = ag.train_func(make_batch("denoiser", 7000 + 8 * it, 8))
        l1.append(float(ls["stage1"].detach())); l2.append(float(ls["stage2"].detach()))
    test = make_batch("denoiser", 95000, 16)
    (n_pred, crm), _ = ag.val_func(test)
    rec = transform.batch_fast_icRM_sigmoid(test["mixed"], crm)
    err_out = float(((rec - test["clean"]) ** 2).mean())
    err_in = float(((test["mixed"] - test["clean"]) ** 2).mean())
    print("denoiser stage1", [round(x, 4) for x in l1[::10]], "stage2", [round(x, 4) for x in l2[::10]], "val MSE out/in", err_out, err_in)
    sos_amd.set_precision("bf16")
    assert np.mean(l1[-5:]) < 0.8 * np.mean(l1[:5]) and np.mean(l2[-5:]) < 0.8 * np.mean(l2[:5])
    assert err_out < err_in


def test_fp16_training_tracks_the_parity_mode():
    """The timed mode's training (half storage, loss-scaled backward) against the bf16x3 parity mode (~fp32 accuracy) from
    the same initial weights on the same batches: the loss curves of the first 25 denoiser steps agree step by step (the
    trajectories are those of the same optimisation, not merely both decreasing)."""
    import sos_amd
    from sos_amd import agent
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch
    from sos_amd.denoiser import networks as jnet
    curves = {}
    try:
        for precision in ("bf16x3", "fp16"):
            sos_amd.set_precision(precision)
            torch.manual_seed(0)
            ag = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
            c = []
            for it in range(25):
                _, ls = ag.train_func(make_batch("denoiser", 7000 + 8 * it, 8))
                c.append(float(ls["stage1"].detach()) + float(ls["stage2"].detach()))
            curves[precision] = np.array(c)
            del ag
    finally:
        sos_amd.set_precision("bf16")
    rel = np.abs(curves["fp16"] - curves["bf16x3"]) / curves["bf16x3"]
    print("loss curves bf16x3", np.round(curves["bf16x3"][::4], 4), "fp16", np.round(curves["fp16"][::4], 4), "max rel diff", rel.max())
    # bounds = 1.5x the worst value over the shipped tilings and two forced ones (tools/probe/track_bound.py under SOS_CONV_FORCE_CFG
    # unset / 3 / 7 on one MI355X, round 5: first five steps 1.70e-2 / 1.04e-2 / 1.46e-2, all 25 steps 5.6e-2 / 6.2e-2 / 3.4e-2):
    # the quantity moves with the summation order of the tilings, so the bound must not sit on one tiling's value (VERDICT r4 #2c)
    assert rel[:5].max() < 2.6e-2 and rel.max() < 0.1 and curves["fp16"][-5:].mean() < 0.7 * curves["fp16"][:3].mean()


@pytest.mark.parametrize("precision", ["fp16", "bf16x3"])
def test_gather_refresh_of_packed_weights_equals_torch_repacking(precision):
    """After an optimizer step the packed 16-bit weights of a training plan are refreshed by ONE sos_gather_pack_multi
    launch that replays relocation maps recorded once (engine.PackRecorder) instead of ~1000 torch kernels.  The maps
    are verified when they are recorded; here the end-to-end guarantee: four training steps of both networks give
    bit-identical losses with the refresh on and off, and the recorder did take the gather path."""
    import sos_amd
    from sos_amd import agent, engine
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision(precision)
    res = {}
    try:
        for gather in (True, False):
            engine.PackRecorder.ENABLED = gather
            torch.manual_seed(0)
            aj = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
            ad = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
            ls = []
            for it in range(4):
                _, l = aj.train_func(make_batch("denoiser", 100 + 4 * it, 4))
                _, l2 = ad.train_func(make_batch("detector", 100 + 4 * it, 4))
                ls.append((float(l["stage1"].detach()), float(l["stage2"].detach()), float(l2["bce"].detach())))
            res[gather] = ls
            assert bool(aj.net._tcache.rec.ok) == gather and bool(ad.net._tcache.rec.ok) == gather
    finally:
        engine.PackRecorder.ENABLED = True
        sos_amd.set_precision("bf16")
    assert res[True] == res[False], res


def test_overflow_guard_skips_the_step_on_the_device():
    """fp16 training (bench.py's timed mode): a parameter gradient that is Inf / NaN must not reach Adam's moments or the
    weights.  The guard (sos_grad_guard, decided on the device, no host sync) turns the optimizer launch of THAT step
    into a no-op for every parameter group, halves the loss-scale target, and the next finite step trains again."""
    import sos_amd
    from sos_amd import _lib as L, agent, engine
    from sos_amd.dataset import make_batch
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("fp16")
    try:
        torch.manual_seed(0)
        ag = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
        batch = make_batch("detector", 8100, 2)
        ag.train_func(batch)                                                 # a normal step: the moments exist
        guard = ag.optimizer.guard
        assert guard is engine.guard_state(ag.net) and guard.cpu().tolist()[:4] == [0.0, 1.0, 1.0, 0.0]
        snap = {n: p.detach().clone() for n, p in ag.net.named_parameters()}
        mom = {n: (ag.optimizer.state[p]["exp_avg"].clone(), ag.optimizer.state[p]["exp_avg_sq"].clone()) for n, p in ag.net.named_parameters()}
        # an overflowing backward: the loss itself is finite, ONE weight-gradient element is Inf (what a 16-bit store
        # beyond 65504 inside the pass produces)
        ag.net.train()
        _, losses = ag.forward(batch)
        ag.optimizer.zero_grad(set_to_none=True)
        sum(losses.values()).backward()
        name, p_bad = list(ag.net.named_parameters())[7]
        p_bad.grad.view(-1)[3] = float("inf")
        ag.optimizer.step()
        torch.cuda.synchronize()
        st = guard.cpu().tolist()
        assert st[0] == 1.0 and st[1] == 0.5 and st[3] == 1.0, st           # found, target backed off, one step skipped
        for n, p in ag.net.named_parameters():
            assert torch.equal(p.detach(), snap[n]), n
            assert torch.equal(ag.optimizer.state[p]["exp_avg"], mom[n][0]) and torch.equal(ag.optimizer.state[p]["exp_avg_sq"], mom[n][1]), n
        # NaN in the INPUT: the loss and every gradient are NaN -> skipped as well, the weights stay finite
        bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        bad["audio"][0, 0, 3, 5] = float("nan")
        ag.train_func(bad)
        torch.cuda.synchronize()
        st = guard.cpu().tolist()
        assert st[0] == 1.0 and st[1] == 0.25 and st[3] == 2.0, st
        assert all(torch.equal(p.detach(), snap[n]) for n, p in ag.net.named_parameters())
        # the next finite step is applied (with the backed-off loss scale) and learns
        _, l0 = ag.train_func(batch)
        torch.cuda.synchronize()
        st = guard.cpu().tolist()
        assert st[0] == 0.0 and st[1] == 0.25 and st[3] == 2.0, st
        assert any(not torch.equal(p.detach(), snap[n]) for n, p in ag.net.named_parameters())
        assert all(bool(torch.isfinite(p).all()) for p in ag.net.parameters())
        # the back-off recovers: SOS_GUARD_GROWTH finite steps double it again (driven through the C ABI on a small buffer)
        g = torch.ones(1000, device="cuda")
        for _ in range(200):
            L.check(L.lib().sos_grad_guard(L.ptr(g), g.numel(), L.ptr(guard), 1, L.stream_ptr()), "sos_grad_guard")
        assert guard.cpu().tolist()[1] == 0.5
        # several buffers, one decision: the flag of the first buffer survives until the finalising call
        g2 = g.clone()
        g2[17] = float("nan")
        L.check(L.lib().sos_grad_guard(L.ptr(g2), g2.numel(), L.ptr(guard), 0, L.stream_ptr()), "sos_grad_guard")
        L.check(L.lib().sos_grad_guard(L.ptr(g), g.numel(), L.ptr(guard), 1, L.stream_ptr()), "sos_grad_guard")
        assert guard.cpu().tolist()[0] == 1.0
    finally:
        sos_amd.set_precision("bf16")


def test_skipped_steps_do_not_advance_adams_bias_correction(tmp_path):
    """torch.cuda.amp.GradScaler does not advance the optimizer's step on a skipped update.  FusedAdam's host counter
    counts attempts; the guard's skipped-step count lives on the device and the kernel takes its bias corrections at
    attempts - skipped: after [finite, Inf, finite] the parameters equal torch.optim.Adam's after TWO steps on the finite
    gradients.  save_ckpt stores the applied count and the guard's back-off state; load_ckpt restores both."""
    from sos_amd import agent, engine
    torch.manual_seed(3)
    net = torch.nn.Linear(7, 5).cuda()
    ref = torch.nn.Linear(7, 5)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    opt = agent.FusedAdam(net.parameters(), lr=1e-2)
    opt.guard = engine.guard_state(net)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    gs = [{n: torch.randn_like(p) for n, p in ref.named_parameters()} for _ in range(2)]

    def step(g, poison=False):
        for n, p in net.named_parameters():
            p.grad = g[n].cuda().clone()
            if poison:
                p.grad.view(-1)[0] = float("inf")
        opt.step()

    step(gs[0])
    step(gs[1], poison=True)            # skipped on the device
    step(gs[1])
    for g in gs:
        for n, p in ref.named_parameters():
            p.grad = g[n].clone()
        ropt.step()
    torch.cuda.synchronize()
    assert opt.guard.cpu().tolist()[3] == 1.0
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert float((p.detach().cpu() - q.detach()).abs().max()) < 2e-6, n       # (with attempts = 3: 1.2e-4 off)
    # checkpoint: applied steps + the guard's back-off; a fresh agent resumes with both
    class _A(agent.BaseAgent):
        pass
    ag = _A(torch.nn.Linear(7, 5), lr=1e-2, model_dir=str(tmp_path))
    ag.net.load_state_dict(net.state_dict())
    ag.optimizer.load_state_dict(opt.state_dict())
    engine.guard_state(ag.net).copy_(opt.guard)
    ag.save_ckpt("latest")
    ck = torch.load(str(tmp_path / "latest.pth"))
    assert all(float(v["step"]) == 2.0 for v in ck["optimizer_state_dict"]["state"].values())
    assert ck["overflow_guard"].tolist()[:4] == [0.0, 0.5, 1.0, 0.0]        # {found, back-off, finite steps since, skipped}
    ag2 = _A(torch.nn.Linear(7, 5), lr=1e-2, model_dir=str(tmp_path))
    ag2.load_ckpt("latest")
    assert engine.guard_state(ag2.net).cpu().tolist()[:4] == [0.0, 0.5, 1.0, 0.0]
    assert all(int(st["step"]) == 2 for st in ag2.optimizer.state.values())


def test_backward_after_an_optimizer_step_is_refused():
    """A training tape is valid for the weights its forward ran with (the packed weights are refreshed in place): forward,
    optimizer step, backward of the OLD forward raises instead of silently mixing old activations with new weights."""
    from sos_amd import agent
    from sos_amd.dataset import make_batch
    from sos_amd.detector import networks as dnet
    torch.manual_seed(0)
    ag = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
    batch = make_batch("detector", 8200, 2)
    ag.net.train()
    _, stale = ag.forward(batch)
    ag.train_func(batch)                       # forward + backward + Adam: the weights move on
    with pytest.raises(RuntimeError, match="parameters changed"):
        sum(stale.values()).backward()


def test_bucketer_keeps_every_bucket_on_its_producing_stream():
    """GradBucketer, SOS_DDP_COMM=inline (round 5): gradients produced on two streams (the denoiser's branch streams) never share a
    bucket -- a bucket is closed when the producing stream changes, and its gather copy is enqueued on the stream that produced it,
    so no stream ever waits for another one before finalize().  After finalize() every parameter's .grad is the bucket view holding
    exactly the gradient that was handed in, whatever the stream it came from."""
    from sos_amd import agent
    assert agent.GradBucketer.COMM_MODE == "inline"
    dev = torch.device("cuda")
    torch.manual_seed(3)
    params = [(f"p{i}", torch.nn.Parameter(torch.zeros(1000 + 37 * i, device=dev))) for i in range(12)]
    b = agent.GradBucketer(params, bucket_bytes=16 * 1024)          # ~4 gradients per bucket
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    grads, order = {}, []
    cur = torch.cuda.current_stream()
    for i, (name, p) in enumerate(params):
        st = s1 if (i // 3) % 2 == 0 else s2                          # the producing stream changes every three gradients
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            g = torch.randn(p.numel(), device=dev) * (i + 1)
            # keep the producing stream busy behind the gradient: a copy enqueued on ANOTHER stream without a wait would race
            torch.cuda._sleep(200000)
            grads[name] = g.clone()
            b.ready(name, g)
            order.append(st)
    # buckets never mix streams: with the stream changing every 3 gradients and ~4 fitting a bucket, there are >= 4 buckets
    assert len(b.buckets) >= 4
    b.finalize()
    torch.cuda.synchronize()
    for name, p in params:
        assert p.grad is not None and torch.equal(p.grad.reshape(-1), grads[name]), name
