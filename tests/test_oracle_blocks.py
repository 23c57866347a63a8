"""tests/golden/blocks.npz (per-block outputs of the REFERENCE's live classes, tests/golden/make_goldens_blocks.py) against the
oracle's block functions on the CPU: the pin of oracle/nets.py's blocks can be re-checked without the reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import nets as onet       # noqa: E402
from util import hashed               # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blocks.npz"))


def _check(tag, y, x, st, sd, names):
    g = torch.from_numpy(hashed(int(G[f"{tag}_dy_idx"]), tuple(y.shape)).astype(np.float32))
    y.backward(g)
    e = lambda a, b, s=None: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) if s is None else s))     # noqa: E731
    assert e(y.detach().reshape(-1)[::7].numpy(), G[f"{tag}_y"], float(G[f"{tag}_y_absmax"])) < 2e-5
    assert e(x.grad.reshape(-1)[::7].numpy(), G[f"{tag}_dx"], float(G[f"{tag}_dx_absmax"])) < 1e-4
    for ours, theirs in names.items():
        assert e(sd[ours].grad.numpy(), G[f"{tag}_grad_{theirs}"]) < 2e-4, (tag, ours)
    for k, v in st.items():
        if "running" in k:
            assert e(v.numpy(), G[f"{tag}_{k[2:]}"]) < 1e-4, (tag, k)


def _sd(spec, seed):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
            for k, v in onet.closed_form_state(spec, seed=seed).items()}


def test_oracle_blocks_match_the_reference_classes():
    torch.set_num_threads(8)
    sd = _sd([("b.block.0.weight", (48, 48, 5, 5), "conv")] + onet._bn_spec("b.block.1", 48), 11)
    x = torch.from_numpy(hashed(801, (2, 48, 80, 70)).astype(np.float32)).requires_grad_(True)
    st = {}
    _check("conv_d32", onet.conv_block(x, sd, "b", (32, 32), True, st), x, st, sd,
           {"b.block.0.weight": "block.0.weight", "b.block.1.weight": "block.1.weight", "b.block.1.bias": "block.1.bias"})
    for tag, spec, seed, idx, shape, fn in (
            ("down_d16", onet._down_spec("b", 64, 64, 3), 12, 802, (2, 64, 40, 37), lambda x, sd, st: onet.down_block(x, sd, "b", 3, 1, 16, True, st)),
            ("down_s2", onet._down_spec("b", 64, 128, 5), 13, 803, (2, 64, 20, 27), lambda x, sd, st: onet.down_block(x, sd, "b", 5, 2, 1, True, st))):
        sd = _sd(spec, seed)
        x = torch.from_numpy(hashed(idx, shape).astype(np.float32)).requires_grad_(True)
        st = {}
        _check(tag, fn(x, sd, st), x, st, sd, {"b.block.1.weight": "block.1.weight", "b.block.2.weight": "block.2.weight",
                                               "b.block.2.bias": "block.2.bias", "b.block.3.weight": "block.3.weight"})
    sd = _sd(onet._up_spec("b", 128, 64, 3), 14)
    x = torch.from_numpy(hashed(804, (2, 128, 10, 13)).astype(np.float32)).requires_grad_(True)
    st = {}
    _check("up", onet.up_block(x, sd, "b", True, st), x, st, sd, {"b.block.0.weight": "block.0.weight", "b.block.1.weight": "block.1.weight",
                                                                  "b.block.1.bias": "block.1.bias", "b.block.2.weight": "block.2.weight"})


def test_oracle_bilstm_head_matches_torch_lstm():
    spec = onet._lstm_spec("lstm", 2048, 100) + [("fc1.0.weight", (100, 200), "lin"), ("fc1.0.bias", (100,), "bias"),
                                                  ("fc1.2.weight", (1, 100), "lin"), ("fc1.2.bias", (1,), "bias")]
    sd = onet.closed_form_state(spec, seed=15)
    feat = torch.from_numpy(hashed(805, (30, 2, 2048)).astype(np.float32) * 0.5)
    h = onet.lstm_bidir(feat, sd, "lstm")
    lo = onet.linear(torch.relu(onet.linear(h, sd, "fc1.0")), sd, "fc1.2").squeeze(-1).permute(1, 0)
    assert np.max(np.abs(h.numpy() - G["lstm_h"])) < 2e-5 * np.max(np.abs(G["lstm_h"]))
    assert np.max(np.abs(lo.numpy() - G["lstm_logits"])) < 2e-5 * np.max(np.abs(G["lstm_logits"]))
