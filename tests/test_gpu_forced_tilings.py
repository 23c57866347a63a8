"""The conv autotuner picks a tiling (and between the 32-row and the 16-row MFMA kernel) per shape by timing, so
which code paths the other GPU tests exercise depends on the device's timings.  These tests pin the choice instead:
a child process with autotuning off (SOS_CONV_TUNE=0) runs the network parity tests with the k-th candidate of the
cost-ordered list forced (SOS_CONV_FORCE_CFG=k; k = 0 is the 16-row kernel for every eligible shape, larger k walk
through single/double weight-slab buffers, other pixel tiles and channel-chunk sizes)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k", [0, 3, 7])
def test_network_parity_with_forced_candidate(k):
    env = dict(os.environ, SOS_CONV_TUNE="0", SOS_CONV_FORCE_CFG=str(k))
    # forward parity of both networks + the exact (1e-4) encoder-block backward; the whole-network gradient tests are
    # left to the tuned run: their tolerances sit at the ReLU-gating noise floor, which moves with the summation order
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_nets.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py"), "-k", "not train_step"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_network_parity_with_forced_16_row_slab_mode(mode):
    """The 16-row conv kernel's three weight-slab schedules (two buffers, one buffer refilled behind a barrier, ring of three
    with the DMA two windows ahead) are performance choices: the network parity tests pass with each forced for every
    16-row launch (SOS_CONV16_MODE; a mode whose buffers do not fit falls back to the tiling's own)."""
    env = dict(os.environ, SOS_CONV16_MODE=str(mode))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_nets.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py"), "-k", "not train_step"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_network_parity_with_forced_512_pixel_16_row_workgroups():
    """Round 4: the 16-row conv kernel with 512-pixel workgroups (a wave owns 128 pixels x all output channels; conv16_kernel<.., PT = 8>)
    is a tiling candidate the table may or may not pick per shape: the network parity tests (forward of both networks, the
    fp16 / bf16 / three-pass modes, the exact encoder-block backward) pass with it forced for every eligible launch
    (SOS_CONV16_FORCE512=1), single- and double-buffered slab."""
    env = dict(os.environ, SOS_CONV16_FORCE512="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_nets.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py"), "-k", "not train_step"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_network_parity_with_forced_384_slot_tiles():
    """Round 6: the 32-row conv kernel with 384-slot tiles (a wave owns 96 pixels x 96 output channels; conv_mfma_kernel<3, ks,
    false, false, 3>) is a tiling candidate of the 96-channel stride-1 layers: the network parity tests (forward of both networks in
    every precision mode, the exact encoder-block backward, the fused training statistics whose scratch now lies over the staged
    tile) pass with the cheapest such candidate forced for every eligible launch (SOS_CONV_FORCE_PT3=1)."""
    # (the fused-input-BatchNorm test compares two launches bit for bit: forcing a tile on one of them is not its subject)
    env = dict(os.environ, SOS_CONV_FORCE_PT3="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_nets.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py"), os.path.join(ROOT, "tests", "test_gpu_train_ops.py"),
                        "-k", "not fused_input_batchnorm"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_network_parity_with_forced_three_workgroups_per_cu():
    """Round 6: conv_mfma_kernel<3, ks, false, false, 2, true> -- the 96-channel-wide double-slab kernel compiled for THREE
    workgroups per CU (<= 168 registers) with nothing but the staged tile in the epilogue's LDS (pixel offsets in the rows' pad
    bytes, the fused statistics' partial sums over the walked tile).  Forced for every eligible launch (SOS_CONV_FORCE_W3=1): forward
    parity of both networks in every precision mode, the training forward's fused statistics, the block backward tests."""
    env = dict(os.environ, SOS_CONV_FORCE_W3="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_nets.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py"), os.path.join(ROOT, "tests", "test_gpu_train_ops.py"),
                        "-k", "not fused_input_batchnorm"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("env_kv", [("SOS_BN_STREAM", "0"), ("SOS_WGRAD_NO_THIN", "1")], ids=["no-bn-stream", "no-thin-wgrad"])
def test_training_parity_with_the_alternative_round5_kernels(env_kv):
    """Round 5 left performance choices behind switches: the per-wave streaming BatchNorm backward reduce of the full-resolution ReLU
    blocks and the per-wave streaming thin weight gradients (both default; the switches select the kernels they replaced).  The BatchNorm / block backward tests and the per-parameter gradient tests of both networks pass on either side
    of each switch.  (The tiled feature-matrix gradient's predecessor still serves the pooled case, the thin-input conv has its
    own bit-for-bit test against the tiled kernel.)"""
    env = dict(os.environ, **{env_kv[0]: env_kv[1]})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_train_ops.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train_nets.py")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
