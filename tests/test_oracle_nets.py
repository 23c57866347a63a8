"""Oracle networks vs golden vectors produced by the imported reference modules."""
import numpy as np
import torch

from oracle import nets as onet
from util import rel_err, silent_gate, spec_input, hashed


def test_state_dict_inventory():
    s1 = onet.detector_spec()
    s2 = onet.joint_spec()
    assert len(s1) == 84 and len(s2) == 322
    n1 = sum(int(np.prod(s)) for k, s, kind in s1 if kind not in ("bn_rm", "bn_rv", "nbt"))
    n2 = sum(int(np.prod(s)) for k, s, kind in s2 if kind not in ("bn_rm", "bn_rv", "nbt"))
    assert n1 == 2276857 and n2 == 16389372


def test_eval_forward_matches_reference_goldens(golden):
    g = golden("networks")
    torch.set_num_threads(8)
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    for tag, B, T, nfr in (("a", 1, 178, 60), ("b", 2, 89, 30)):
        x = spec_input(100 + B, B, T)
        n = silent_gate(x)
        with torch.no_grad():
            lo = onet.detector_forward(sd1, x, nfr)
            n_pred, out = onet.joint_forward(sd2, x, n)
        assert rel_err(lo, g[f"det_logits_{tag}"]) < 1e-4
        assert rel_err(n_pred, g[f"n_pred_{tag}"]) < 1e-4
        assert rel_err(out, g[f"mask_{tag}"]) < 1e-4


def test_train_forward_losses_match_reference_goldens(golden):
    g = golden("networks")
    torch.set_num_threads(8)
    B, T, nfr = 2, 89, 30
    x = spec_input(100 + B, B, T)
    n = silent_gate(x)
    clean = spec_input(300, B, T) * 0.5
    full_noise = x - clean
    label = torch.from_numpy(g["train_label"])
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    with torch.no_grad():
        lo, l = onet.detector_loss(sd1, {"audio": x, "label": label}, True, {})
        (n_pred, out), ls = onet.denoiser_losses(
            sd2, {"mixed": x, "noise": n, "clean": clean, "full_noise": full_noise}, True, {})
    assert rel_err(lo, g["train_det_logits"]) < 2e-4
    assert abs(float(l["bce"]) - float(g["train_bce"])) < 1e-5
    assert rel_err(n_pred, g["train_n_pred"]) < 2e-4
    assert rel_err(out, g["train_mask"]) < 2e-4
    assert abs(float(ls["stage1"]) / float(g["train_l1"]) - 1) < 1e-4
    assert abs(float(ls["stage2"]) / float(g["train_l2"]) - 1) < 1e-4
