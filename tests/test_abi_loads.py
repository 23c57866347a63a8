"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/sos_hip.h declares; the module mirrors expose the reference's state_dict layout."""
import os
import re

import pytest
import torch

from oracle import nets as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sos_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sos_hip.h")).read()
    declared = set(re.findall(r"\b(sos_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("sos_stream_t")
    assert declared, "header parse failed"
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name), f"{name} declared in sos_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"sos_last_error"} == declared
    assert h.sos_abi_version() == 1


def test_state_dict_keys_match_reference_layout():
    from sos_amd.detector import networks as dnet
    from sos_amd.denoiser import networks as jnet
    from sos_amd.common import MyConfig
    det = dnet.get_network()
    spec = onet.detector_spec()
    sd = det.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in spec)
    jm = jnet.get_network(MyConfig())
    spec = onet.joint_spec()
    sd = jm.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec]
    assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in spec)
    # strict load of a reference-shaped checkpoint works
    jm.load_state_dict(onet.closed_form_state(spec, seed=2), strict=True)


def test_no_cpu_fallback():
    from sos_amd.detector import networks as dnet
    from sos_amd import transform
    det = dnet.get_network().eval()
    with pytest.raises(RuntimeError):
        det(torch.zeros(1, 2, 256, 64))
    with pytest.raises(RuntimeError):
        transform.stft_batch(torch.zeros(1, 28000))


def test_lstm_gate_permutation_is_consistent():
    """engine.lstm_gate_perm: torch's (dir, gate, unit) row order -> the kernels' gate-interleaved (dir, unit, gate)
    order (include/sos_hip.h: channel dir*4H + 4*j + q), and its inverse."""
    import torch
    from sos_amd import engine as E
    H = 12
    perm, inv = E.lstm_gate_perm(H, torch.device("cpu"))
    assert sorted(perm.tolist()) == list(range(8 * H))
    assert torch.equal(perm[inv], torch.arange(8 * H)) and torch.equal(inv[perm], torch.arange(8 * H))
    for d in range(2):
        for j in range(H):
            for q in range(4):
                assert int(perm[d * 4 * H + 4 * j + q]) == d * 4 * H + q * H + j
