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
