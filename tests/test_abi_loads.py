"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/sos_hip.h declares; the module mirrors expose the reference's state_dict layout."""
import os
import re

import pytest
import torch

from oracle import nets as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("precision,storage", [("bf16", b"bf16"), ("bf16x3", b"bf16"), ("fp16", b"fp16")])
def test_library_exports_every_declared_symbol(precision, storage):
    """Both builds of the C ABI (libsos_hip.so: bfloat16 storage, libsos_hip_f16.so: IEEE half) export every symbol the
    header declares, and the ctypes table binds exactly those."""
    import sos_amd
    from sos_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sos_hip.h")).read()
    declared = set(re.findall(r"\b(sos_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("sos_stream_t")
    assert declared, "header parse failed"
    sos_amd.set_precision(precision)
    try:
        h = _lib.lib()
    finally:
        sos_amd.set_precision("bf16")
    for name in declared:
        assert hasattr(h, name), f"{name} declared in sos_hip.h but not exported"
    assert set(_lib.SIGNATURES) | {"sos_last_error"} == declared
    assert h.sos_abi_version() == _lib.EXPECTED_ABI == 10
    assert h.sos_storage_dtype() == storage


def test_tune_table_load_rejects_foreign_and_illegal_entries(tmp_path):
    """sos_conv2d_tune_load (host only, no GPU): a table of another format / ABI is an error, an entry whose tiling the
    build would not offer for that shape is dropped, a legal one is accepted; the shipped table loads completely."""
    from sos_amd import _lib, engine
    h = _lib.lib()
    bad = tmp_path / "old.txt"
    bad.write_text("64 256 178 178 96 1 96 5 5 1 1 1 256 178 0 1 0 0 96 1 4 4 6\n")      # round-1 format: no header
    assert h.sos_conv2d_tune_load(str(bad).encode()) < 0
    shape = "64 256 178 178 96 1 96 5 5 1 1 1 256 178 0 1 0 0 96"
    mixed = tmp_path / "mixed.txt"
    mixed.write_text("sos_conv_tune 2 abi 3 nkey 19\n"
                     f"{shape} 1 4 4 6\n"            # legal: 16x16 pixels, 6 k-steps per chunk
                     f"{shape} 1 4 4 7\n"            # 7 does not divide cin/16
                     f"{shape} 4 2 4 6\n"            # 4 residue classes need dil_w >= 4
                     f"{shape} 1 8 8 6\n")           # 2^16 pixels per workgroup
    assert h.sos_conv2d_tune_load(str(mixed).encode()) == 1
    assert h.sos_conv2d_tune_load(str(tmp_path / "missing.txt").encode()) == 0
    if os.path.exists(engine.SHIPPED_TUNE_TABLE):
        lines = [ln for ln in open(engine.SHIPPED_TUNE_TABLE).read().splitlines()[1:] if ln.strip()]
        assert h.sos_conv2d_tune_load(engine.SHIPPED_TUNE_TABLE.encode()) == len(lines) > 0


def test_wgrad_table_load_rejects_foreign_and_illegal_entries(tmp_path):
    """sos_wgrad_tune_load (host only, no GPU; ABI 7): a table of another format is an error, a plan the build would not offer
    for that shape is dropped, a legal one is accepted; the shipped table of measured weight-gradient plans loads completely."""
    from sos_amd import _lib, engine
    h = _lib.lib()
    bad = tmp_path / "foreign.txt"
    bad.write_text("sos_wgrad_tune 9 nkey 12\n")
    assert h.sos_wgrad_tune_load(str(bad).encode()) < 0
    shape = "256 178 5 5 1 4 4 96 96 0 0 0"          # Hg Wg kh kw stride dil_h dil_w M N 16x16x32-kernel temporal flat
    mixed = tmp_path / "mixed.txt"
    mixed.write_text("sos_wgrad_tune 1 nkey 12\n"
                     f"{shape} 3 1 0 4 1 1\n"        # legal: 3 m-tiles x 1 n-tile, 16 x 16 pixels, column-major, one workgroup per CU
                     f"{shape} 3 2 0 4 1 1\n"        # 2 n-tiles x 25 taps > 32 (tap, n-tile) pairs
                     f"{shape} 3 1 3 4 1 1\n"        # 8 residue classes need dil_w % 8 == 0
                     f"{shape} 4 1 0 4 1 1\n"        # 4 m-tiles
                     f"{shape} 3 1 0 4 1 2\n"        # two workgroups per CU: 3 m-tiles + the 5x5 patch do not fit half the LDS
                     "256 178 5 5 1 1 1 48 48 0 0 0 2 1 0 4 0 1\n")   # 48 x 48 x 25 taps is the 16x16x32 kernel's shape: flag mismatch
    assert h.sos_wgrad_tune_load(str(mixed).encode()) == 1
    assert h.sos_wgrad_tune_load(str(tmp_path / "missing.txt").encode()) == 0
    lines = [ln for ln in open(engine.SHIPPED_WGRAD_TABLE).read().splitlines()[1:] if ln.strip()]
    assert h.sos_wgrad_tune_load(engine.SHIPPED_WGRAD_TABLE.encode()) == len(lines)


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


def test_bench_picks_the_dominant_signature_by_work_not_by_warm_up_time():
    """bench.py's roofline leg: the dominant launch signature is the one carrying the most algorithmic FLOPs (per launch x
    launches), whatever one-time host work inflated the HIP-event brackets of other signatures during warm-up (a run once
    reported a 15 TFLOP/s first-layer gradient as dominant: roofline.frac 0.005)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    summ = {("conv", 5, 5, 1, 1, 1, 96, 96, 64, 256, 178): dict(flops=1.344e12, launches=12, total_ms=14.0, avg_ms=1.17),
            ("conv", 5, 5, 2, 2, 1, 96, 96, 64, 256, 178): dict(flops=1.344e12, launches=6, total_ms=7.2, avg_ms=1.2),
            ("wgrad", 1, 7, 1, 1, 1, 48, 2, 64, 256, 178): dict(flops=3.9e9, launches=6, total_ms=900.0, avg_ms=150.0)}
    assert bench._dominant(summ) == ("conv", 5, 5, 1, 1, 1, 96, 96, 64, 256, 178)
