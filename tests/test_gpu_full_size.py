"""BASELINE configs[1] sizes (B = 64 clips of 2 s -> 64 x 2 x 256 x 178) where the oracle is too slow to run:
size-independent properties of the path instead (the same kernels, tilings and split-K factors the bench uses).

* inference: eval-mode networks treat clips independently -> permuting the batch permutes the output, bit for bit
  (tiles never span clips, so the summation order of a clip does not depend on its position);
  the mixed spectrogram masked by the net and inverted is a linear function of the mask (checked on the front-end)
* training: a batch made of the same 32 clips twice has the BatchNorm statistics, losses and (mean-reduced)
  gradients of the 32-clip batch.  Only the summation order of the f32 batch statistics differs (1e-7), but the
  arithmetic is discontinuous: a rounding to bf16 turns that into a 4e-3 step for a few elements, after three
  layers the activations differ at the rounding-noise floor, and every element whose pre-activation changes sign
  flips its ReLU gate in the backward pass -- a fraction eps of the elements, i.e. a sqrt(eps) relative change
  of the gradient per layer.  Observed, growing from the last layer to the first and varying with the tuned tilings: bf16x3 2e-6 .. 3e-2
  (losses 1e-7), plain bf16 6e-4 .. 0.19 for the detector and up to 0.66 for the first U-Net layer of the
  (random-init, 36 blocks deep) denoiser (losses 1e-5 .. 4e-5).  The gradient bound (0.1) is therefore asserted in
  the bf16x3 parity mode, where it still separates this from a wrong batch normalisation or a dropped half
  batch (0.5 .. 1.0); plain bf16 asserts the losses, finiteness and determinism
* two identical steps are bit-identical (no atomics anywhere in the path)."""
import numpy as np
import pytest
import torch

import sos_amd

pytestmark = pytest.mark.gpu
B = 64


def _nets():
    from sos_amd.common import MyConfig
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    torch.manual_seed(0)
    return dnet.get_network().cuda(), jnet.get_network(MyConfig()).cuda()


@pytest.mark.parametrize("precision", ["bf16", "fp16", "mixed"])
def test_inference_is_permutation_equivariant_at_batch_64(precision):
    """BASELINE configs[1]'s batch through the inference chain in every 1x-cost mode: 'fp16' is the storage type bench.py
    times, 'mixed' the pipeline's mode (detector in bf16x3: both libraries in one chain)."""
    from sos_amd import pipeline
    from sos_amd.dataset import synth_batch
    det, jm = _nets()
    det, jm = det.eval(), jm.eval()
    base = synth_batch(500, 8)["mixed"]
    mixed = torch.from_numpy(np.tile(base, (8, 1))).cuda() * torch.linspace(0.5, 1.5, B, device="cuda")[:, None]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    sos_amd.set_precision(precision)
    try:
        r = pipeline.denoise(det, jm, mixed, return_all=True)
        rp = pipeline.denoise(det, jm, mixed[perm].contiguous(), return_all=True)
    finally:
        sos_amd.set_precision("bf16")
    assert r["out"].shape == (B, 158 * 177)
    for k in ("logits", "bits", "mask", "n_pred", "crm", "out"):
        assert torch.equal(r[k][perm], rp[k]), k
    assert torch.isfinite(r["out"]).all() and float(r["out"].abs().max()) > 0
    # the batch really is 64 different problems
    assert len({float(v) for v in r["out"].abs().sum(1)}) == B


def _train_batches(n):
    from sos_amd.dataset import make_batch
    return make_batch("denoiser", 700, n), make_batch("detector", 700, n)


def _grads(agent_cls, net, batch):
    ag = agent_cls(net.train(), lr=1e-3)
    _, losses = ag.forward(batch)
    ag.optimizer.zero_grad(set_to_none=True)
    sum(losses.values()).backward()
    return {k: float(v) for k, v in losses.items()}, {n: p.grad.detach().clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16"])
@pytest.mark.parametrize("which", ["detector", "denoiser"])
def test_training_step_on_a_doubled_batch_equals_the_half_batch(which, precision):
    from sos_amd import agent
    sos_amd.set_precision(precision)
    try:
        _doubled_batch(which, precision, agent)
    finally:
        sos_amd.set_precision("bf16")


def _doubled_batch(which, precision, agent):
    # fp16 (the timed mode: loss-scaled half gradients; the doubled batch halves the entering gradient, the power-of-two
    # loss scale doubles, nothing else changes) is bounded like bf16, 4x tighter on the losses
    # (bf16: 4e-4 since round 6 -- the B = 64 shapes of the 96-channel layers run the three-per-CU tiling with two k-steps per chunk,
    # the B = 32 shapes their own table entries: other summation orders of the fused BatchNorm statistics, 1.8e-4 observed on the
    # denoiser's stage-2 loss where 2e-4 was asserted)
    l_tol, g_tol = {"bf16x3": (2e-6, 0.1), "bf16": (4e-4, None), "fp16": (5e-5, None)}[precision]
    bj, bd = _train_batches(B // 2)
    half = bd if which == "detector" else bj
    half = {k: v for k, v in half.items() if torch.is_tensor(v)}
    full = {k: torch.cat([v, v]).contiguous() for k, v in half.items()}
    cls = agent.DetectorAgent if which == "detector" else agent.DenoiserAgent
    nets = []
    for _ in range(3):
        det, jm = _nets()
        nets.append(det if which == "detector" else jm)
    sd0 = {k: v.clone() for k, v in nets[0].state_dict().items()}
    for n in nets[1:]:
        n.load_state_dict(sd0)
    l_half, g_half = _grads(cls, nets[0], half)
    l_full, g_full = _grads(cls, nets[1], full)
    l_again, g_again = _grads(cls, nets[2], full)
    assert next(iter(full.values())).shape[0] == B
    for k in l_half:
        print(which, precision, k, l_full[k], l_half[k])
        assert abs(l_full[k] - l_half[k]) <= l_tol * abs(l_half[k]) + 1e-7, (k, l_full[k], l_half[k])
        assert l_full[k] == l_again[k]
    worst = 0.0
    for n in g_half:
        assert torch.equal(g_full[n], g_again[n]), n                        # deterministic
        ref = float(g_half[n].float().norm())
        err = float((g_full[n].float() - g_half[n].float()).norm()) / (ref + 1e-20)
        if g_half[n].numel() == 1:
            continue        # single shared PReLU slope: a cancelling sum (see tests/test_gpu_train_nets.py)
        worst = max(worst, err)
        print(f"   {n:48s} |g| {ref:10.3e} rel diff {err:9.2e}")
    for n in g_half:
        if g_half[n].numel() == 1:
            continue
        err = float((g_full[n].float() - g_half[n].float()).norm()) / (float(g_half[n].float().norm()) + 1e-20)
        assert torch.isfinite(g_full[n]).all(), n
        if g_tol is not None:
            assert err < g_tol, (n, err)
    print(which, precision, "worst relative gradient difference doubled vs half batch:", worst)
    # BatchNorm running statistics see the same batch moments (unbiased variance differs by n/(n-1): 2e-7 here)
    for (k, a), (_, b) in zip(nets[0].state_dict().items(), nets[1].state_dict().items()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max()) + 1e-6, k
