#!/usr/bin/env python3
"""Generate tests/golden/blocks.npz: per-block intermediates of ONE block of each
kind and the parameters after ONE optimizer step, from the REFERENCE's own live classes imported from /root/reference
(M2/networks.py: ConvBlock, DownConvBlock, UpConvBlock; torch's nn.LSTM / nn.Linear as M1/networks.py:91-98 wires them;
torch.optim.Adam as M1/agent.py:48 does).  Runs only in the build container; stores arrays of numbers only (strided samples of
the big tensors, everything of the small ones), never reference source.  It also pins oracle/nets.py's block functions against
the same reference outputs (hard asserts).

Weights are closed-form (oracle.nets.closed_form_state), inputs hashed (tests/util.py: hashed), so the GPU tests rebuild both
without this file's help and compare against the stored reference OUTPUTS.

    python tests/golden/make_goldens_blocks.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_goldens as mg            # noqa: E402  (the stub / loader helpers)
from oracle import nets as onet      # noqa: E402
from util import hashed, rel_err, silent_gate, spec_input    # noqa: E402

STRIDE = 7        # every 7th element of a large tensor is stored (flattened order)


def samples(t):
    return t.detach().reshape(-1)[::STRIDE].numpy().copy()


def main():
    torch.set_num_threads(8)
    mg.install_stubs()
    ref2 = mg.load(os.path.join(mg.M2, "networks.py"), "ref_networks2")
    ref1 = mg.load(os.path.join(mg.M1, "networks.py"), "ref_networks1")
    out = {}
    dy_idx = {}

    def run(tag, ref_blk, sd, x, key_map, oracle_fn):
        idx = 900 + len(dy_idx)
        dy_idx[tag] = idx
        ref_blk.load_state_dict({key_map(k): v for k, v in sd.items()}, strict=True)
        ref_blk.train()
        xr = x.clone().requires_grad_(True)
        y = ref_blk(xr)
        g = torch.from_numpy(hashed(idx, tuple(y.shape)).astype(np.float32))
        y.backward(g)
        st = {}
        sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        yo = oracle_fn(xo, sdo, st)
        yo.backward(g)
        assert rel_err(yo, y) < 2e-5, (tag, rel_err(yo, y))
        assert rel_err(xo.grad, xr.grad) < 1e-4, (tag, rel_err(xo.grad, xr.grad))
        out[f"{tag}_y"] = samples(y)
        out[f"{tag}_y_absmax"] = np.float64(float(y.detach().abs().max()))
        out[f"{tag}_dx"] = samples(xr.grad)
        out[f"{tag}_dx_absmax"] = np.float64(float(xr.grad.abs().max()))
        out[f"{tag}_dy_idx"] = np.int64(idx)
        for k, p in ref_blk.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad.detach().numpy().copy()
        for k, v in ref_blk.state_dict().items():
            if "running" in k:
                out[f"{tag}_{k}"] = v.numpy().copy()
                cand = [kk for kk in st if kk.endswith(k)]
                assert cand and rel_err(st[cand[0]], v) < 1e-4, (tag, k)
        print(tag, "y absmax", float(y.detach().abs().max()), "dx absmax", float(xr.grad.abs().max()), "stored y samples", out[f"{tag}_y"].size)

    # ---- (1) zero-padded dilated Conv2d + BN + ReLU at dilation (32, 32): M2/networks.py:28-51 (== M1/networks.py:28-51)
    spec = [("b.block.0.weight", (48, 48, 5, 5), "conv")] + onet._bn_spec("b.block.1", 48)
    sd = onet.closed_form_state(spec, seed=11)
    x = torch.from_numpy(hashed(801, (2, 48, 80, 70)).astype(np.float32))
    run("conv_d32", ref2.ConvBlock(48, 48, (5, 5), (32, 32)), sd, x, lambda k: k[2:],
        lambda xo, s, st: onet.conv_block(xo, s, "b", (32, 32), True, st))
    # the ReLU gate of this block: an element whose pre-activation lies within rounding of zero may legitimately sit on the other
    # side of the gate in another implementation, which moves dbeta / dgamma / dW of ITS output channel by one dy-sized term (the
    # PReLU blocks only change slope there).  Stored: the smallest |pre-activation| per output channel; the test skips the channels
    # whose margin is below its own forward accuracy (tools/probe/archive/conv_d32_debug.py: one such element here, 3.1e-6, channel 2).
    with torch.no_grad():
        pre = onet.batch_norm(torch.nn.functional.conv2d(x, sd["b.block.0.weight"], None, 1, (64, 64), (32, 32)), sd, "b.block.1", True, {})
    out["conv_d32_gate_margin"] = pre.abs().amin(dim=(0, 2, 3)).numpy().copy()
    print("conv_d32 gate margins below 1e-4:", [(c, float(v)) for c, v in enumerate(out["conv_d32_gate_margin"]) if v < 1e-4])
    # ---- (2) reflection-padded block at dilation 16 (the U-Net's middle): M2/networks.py:97-117
    sd = onet.closed_form_state(onet._down_spec("b", 64, 64, 3), seed=12)
    x = torch.from_numpy(hashed(802, (2, 64, 40, 37)).astype(np.float32))
    run("down_d16", ref2.DownConvBlock(64, 64, 3, 1, dilation=16), sd, x, lambda k: k[2:],
        lambda xo, s, st: onet.down_block(xo, s, "b", 3, 1, 16, True, st))
    # ---- (3) stride-2 block
    sd = onet.closed_form_state(onet._down_spec("b", 64, 128, 5), seed=13)
    x = torch.from_numpy(hashed(803, (2, 64, 20, 27)).astype(np.float32))
    run("down_s2", ref2.DownConvBlock(64, 128, 5, 2), sd, x, lambda k: k[2:],
        lambda xo, s, st: onet.down_block(xo, s, "b", 5, 2, 1, True, st))
    # ---- (4) ConvTranspose2d(k3, s2, p1, output_padding=1 -- the reference passes `dilation` there, M2/networks.py:130) + BN + PReLU
    sd = onet.closed_form_state(onet._up_spec("b", 128, 64, 3), seed=14)
    x = torch.from_numpy(hashed(804, (2, 128, 10, 13)).astype(np.float32))
    run("up", ref2.UpConvBlock(128, 64, 3, 2), sd, x, lambda k: k[2:], lambda xo, s, st: onet.up_block(xo, s, "b", True, st))

    # ---- (5) BiLSTM(2048, 100) + FC head, wired as M1/networks.py:91-98,143-153 (seq-first, flatten_parameters is a no-op on CPU)
    T, B = 30, 2
    spec = onet._lstm_spec("lstm", 2048, 100) + [("fc1.0.weight", (100, 200), "lin"), ("fc1.0.bias", (100,), "bias"),
                                                  ("fc1.2.weight", (1, 100), "lin"), ("fc1.2.bias", (1,), "bias")]
    sd = onet.closed_form_state(spec, seed=15)
    lstm = torch.nn.LSTM(input_size=2048, hidden_size=100, bidirectional=True)
    fc1 = torch.nn.Sequential(torch.nn.Linear(200, 100), torch.nn.ReLU(True), torch.nn.Linear(100, 1))
    lstm.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("lstm.")}, strict=True)
    fc1.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("fc1.")}, strict=True)
    feat = torch.from_numpy(hashed(805, (T, B, 2048)).astype(np.float32) * 0.5)
    fr = feat.clone().requires_grad_(True)
    h, _ = lstm(fr)
    logits = fc1(h).squeeze(-1).permute(1, 0)                 # (B, T): M1/networks.py:151-153
    g = torch.from_numpy(hashed(905, (B, T)).astype(np.float32))
    logits.backward(g)
    sdo = {k: v.clone() for k, v in sd.items()}
    ho = onet.lstm_bidir(feat, sdo, "lstm")
    lo = onet.linear(torch.relu(onet.linear(ho, sdo, "fc1.0")), sdo, "fc1.2").squeeze(-1).permute(1, 0)
    assert rel_err(ho, h) < 2e-5 and rel_err(lo, logits) < 2e-5
    out["lstm_h"] = h.detach().numpy().copy()
    out["lstm_logits"] = logits.detach().numpy().copy()
    out["lstm_dfeat"] = samples(fr.grad)
    out["lstm_dfeat_absmax"] = np.float64(float(fr.grad.abs().max()))
    for k, p in list(lstm.named_parameters()) + [("fc1." + k, p) for k, p in fc1.named_parameters()]:
        gflat = p.grad.detach().reshape(-1).numpy()
        out[f"lstm_gradnorm_{k}"] = np.float64(np.sqrt(np.sum(gflat.astype(np.float64) ** 2)))
        out[f"lstm_gradsamples_{k}"] = gflat[::101].copy()
    print("lstm: h absmax", float(h.abs().max()), "logits", float(logits.min()), float(logits.max()))

    # ---- (6) ONE optimizer step of the reference trainers on the B = 2, T = 89 training batch of networks.npz
    # (M1/agent.py:48,106-111: Adam(lr 1e-3), zero_grad -> backward -> step; M2/agent.py:101-106 the same on loss1 + loss2)
    ref_tf = mg.load(os.path.join(mg.M2, "transform.py"), "ref_transform")
    det = ref1.get_network()
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    det.load_state_dict(sd1, strict=True)

    class Cfg:
        kernel_sizes = onet.CTX_KERNELS
        dilations = onet.CTX_DILATIONS
    jm = ref2.get_network(Cfg())
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    jm.load_state_dict(sd2, strict=True)
    Bt, Tt, nfr = 2, 89, 30
    x = spec_input(100 + Bt, Bt, Tt)
    n = silent_gate(x)
    clean = spec_input(300, Bt, Tt) * 0.5
    full_noise = x - clean
    label = torch.from_numpy((onet._hash_uniform(301, Bt * nfr).reshape(Bt, nfr) > 0).astype(np.float32))
    for name, net, loss_fn in (("det", det, lambda: torch.nn.BCEWithLogitsLoss()(det(x, nfr), label)),
                               ("jm", jm, None)):
        net.train()
        opt = torch.optim.Adam(net.parameters(), 1e-3)
        opt.zero_grad()
        if name == "det":
            loss = loss_fn()
        else:
            n_pred, mask = jm(x, n)
            rec = ref_tf.batch_fast_icRM_sigmoid(x, mask)
            loss = torch.nn.MSELoss()(n_pred, full_noise) + torch.nn.MSELoss()(rec, clean)
        before = {k: p.detach().clone() for k, p in net.named_parameters()}
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        opt.step()
        # Adam's first update is -lr * g / (|g| + eps): lr * sign(g) wherever |g| >> 1e-8.  Stored per parameter: strided samples of
        # the UPDATE (p_after - p_before) and of the gradient that produced it (the test skips elements whose reference gradient
        # is too small for its sign to be decided at the test's precision), and the sum of the updates
        upd_s, g_s, sums, names = [], [], [], []
        for k, p in net.named_parameters():
            d = (p.detach() - before[k]).reshape(-1)
            upd_s.append(d[::53].numpy().copy())
            g_s.append(grads[k].reshape(-1)[::53].numpy().copy())
            sums.append(float(d.double().sum()))
            names.append(k)
        out[f"adam_{name}_update_samples"] = np.concatenate(upd_s)
        out[f"adam_{name}_grad_samples"] = np.concatenate(g_s)
        out[f"adam_{name}_sample_counts"] = np.array([len(u) for u in upd_s], dtype=np.int64)
        out[f"adam_{name}_update_sum"] = np.array(sums)
        out[f"adam_{name}_loss"] = np.float64(float(loss))
        print("adam", name, "loss", float(loss), "params", len(names), "samples", out[f"adam_{name}_update_samples"].size)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("wrote blocks.npz", os.path.getsize(os.path.join(HERE, "blocks.npz")), "bytes")


if __name__ == "__main__":
    main()
