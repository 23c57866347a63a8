"""Host half of the data layer (no GPU): the synthetic draws are a pure function of the clip index, whoever computes them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_host_draws_do_not_depend_on_the_worker_count():
    """dataset._HostPool: `num_workers` spawned processes (numpy / scipy only -- sos_amd._synth_worker imports no torch) split a
    batch into chunks; the stacked draws equal the in-process ones bit for bit, in clip order."""
    from sos_amd import dataset as D
    want = D.host_draws(40, 7, n_samples=4000)
    pool = D._HostPool(3)
    try:
        got = pool.submit(40, 7, 4000, D.DATA_REQUIRED_SR, D.FPS, None)()
        again = pool.submit(41, 5, 4000, D.DATA_REQUIRED_SR, D.FPS, None)()
    finally:
        pool.close()
    for k in ("speech", "noise", "bits"):
        assert np.array_equal(want[k], got[k]), k
        assert np.array_equal(want[k][1:6], again[k]), k
    assert want["snr"] == got["snr"] and want["snr"][1:6] == again["snr"]
    assert want["speech"].dtype == np.float32 and want["bits"].dtype == np.uint8 and want["bits"].shape == (7, round(4000 / 14000 * 30))


def test_synth_worker_module_does_not_import_torch():
    """The worker processes of the loader import sos_amd._synth_worker only: it must stay free of torch (import time, and no HIP
    initialisation in a worker)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import sos_amd._synth_worker as w; "
            "assert 'torch' not in sys.modules, 'torch imported'; print(w.synth_chunk((0, 2, 1000, 14000, 30.0, None))[0].shape)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "(2, 1000)" in r.stdout, r.stdout + r.stderr


def test_rank_shards_of_the_synthetic_loader_are_disjoint_and_equally_long():
    from sos_amd import dataset as D
    seen = []
    for r in range(4):
        ld = D._SyntheticLoader("detector", D.PHASE_TRAINING, 8, 5, "cuda", rank=r, world_size=4, num_workers=2)
        st = ld.starts()
        assert len(st) == 5
        seen += [s + k for s in st for k in range(8)]
        ld.close()
    assert len(set(seen)) == len(seen) == 4 * 5 * 8
