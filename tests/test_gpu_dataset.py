"""The asynchronous data layer (round 6): worker processes + producer thread + side stream give the batches of
the synchronous path bit for bit, and the consumer never waits for the host."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import sos_amd  # noqa: E402

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
        elif k != "_raw":
            assert a[k] == b[k], k


@pytest.mark.parametrize("model", ["detector", "denoiser"])
def test_async_synthetic_loader_equals_the_synchronous_one(model):
    """get_dataloader(num_workers=2, prefetch=2) against num_workers=0 (the caller's thread and stream): the same batch dicts
    (M1/dataset.py:348-352 / M2/dataset.py:311-320), every tensor bit for bit -- the draws are a function of the clip index, the
    device half runs the same kernels on a side stream."""
    from sos_amd.dataset import get_dataloader, make_batch
    sync = list(get_dataloader("training", batch_size=4, num_workers=0, model=model, n_batches=5))
    ld = get_dataloader("training", batch_size=4, num_workers=2, model=model, n_batches=5, prefetch=2)
    got = []
    for b in ld:
        got.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})     # consumed on the current stream
    ld.close()
    assert len(got) == len(sync) == 5
    for a, b in zip(sync, got):
        _same(a, b)
    _same(sync[1], make_batch(model, 4, 4))
    if model == "denoiser":
        assert got[0]["mixed"].shape == (4, 2, 256, 178) and got[0]["_bits"].dtype == torch.uint8
        assert got[0]["bitstream"][0] == "".join(str(int(v)) for v in got[0]["_bits"][0].tolist())
    torch.cuda.synchronize()


def test_the_consumer_of_the_async_loader_never_synchronises():
    """Drawing batches from the asynchronous loader and training on them under torch.cuda.set_sync_debug_mode('error'): the
    consumer thread only makes its stream wait for an event; the producer uploads from pinned memory (a pageable host -> device
    copy, a .cpu() or an .item() anywhere on the way would raise -- the round-5 loaders made four device -> host copies per
    batch)."""
    from sos_amd import agent
    from sos_amd.common import MyConfig
    from sos_amd.dataset import get_dataloader
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("fp16")
    ld = None
    try:
        torch.manual_seed(0)
        ad = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
        aj = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
        ld = get_dataloader("training", batch_size=4, num_workers=2, model="denoiser", n_batches=7, prefetch=2)
        it = iter(ld)
        for _ in range(3):                       # warm-up: plans, tables, workspaces, the workers, the pinned ring
            b = next(it)
            aj.train_func(b)
            ad.train_func({"audio": b["mixed"], "label": b["_bits"].float()})
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            losses = []
            for _ in range(4):
                b = next(it)
                agent.train_concurrent([(aj, b), (ad, {"audio": b["mixed"], "label": b["_bits"].float()})])
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        assert all(torch.isfinite(p).all() for p in aj.net.parameters())
    finally:
        if ld is not None:
            ld.close()
        sos_amd.set_precision("bf16")


def test_async_loader_hands_over_producer_errors():
    """An exception in the producer thread (here: an impossible batch) reaches the consumer instead of hanging it."""
    from sos_amd import dataset as D
    ld = D._SyntheticLoader("nonsense-model-that-has-no-cRM", D.PHASE_TRAINING, 2, 2, "cuda", num_workers=1)
    bad = D.batch_from_raw
    try:
        def boom(*a, **k):
            raise RuntimeError("producer failed")
        D.batch_from_raw = boom
        with pytest.raises(RuntimeError, match="producer failed"):
            list(ld)
    finally:
        D.batch_from_raw = bad
        ld.close()
