"""HIP front-end kernels (through the C ABI) vs the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from util import rel_err, hashed

pytestmark = pytest.mark.gpu


def test_stft_golden(golden):
    from sos_amd import transform as T
    g = golden("frontend")
    for i in range(3):
        S = T.fast_stft(g[f"wave{i}"])
        assert S.shape == g[f"stft{i}"].shape and S.dtype == np.float64
        assert rel_err(S, g[f"stft{i}"]) < 1e-5          # north_star: spectrograms within 1e-3 rel


def test_stft_batch_matches_oracle_full_size():
    from sos_amd import transform as T
    B, N = 64, 28000
    w = (hashed(7, (B, N)) * 0.3).astype(np.float32)
    S = T.stft_batch(torch.from_numpy(w).cuda())
    assert tuple(S.shape) == (B, 2, 256, 178)
    for b in (0, 17, 63):
        ref = ofe.fast_stft(w[b]).transpose(2, 0, 1)
        assert rel_err(S[b].cpu().numpy(), ref) < 1e-5


def test_istft_golden_and_roundtrip(golden):
    from sos_amd import transform as T
    g = golden("frontend")
    for i in range(3):
        y = T.fast_istft(g[f"stft{i}"])
        assert y.shape == g[f"istft{i}"].shape           # hop*(T-1): 27 966 for a 28 000-sample clip
        assert rel_err(y, g[f"istft{i}"]) < 1e-5
    y = T.fast_istft(g["rand_spec"])
    assert rel_err(y, g["rand_spec_istft"]) < 1e-5
    # size-independent property at BASELINE batch: istft(stft(x)) == x on the common length
    B, N = 64, 28000
    w = torch.from_numpy((hashed(8, (B, N)) * 0.3).astype(np.float32)).cuda()
    r = T.istft_batch(T.stft_batch(w))
    assert tuple(r.shape) == (B, 158 * 177)
    assert float((r - w[:, :r.shape[1]]).abs().max()) < 2e-5


def test_stft_linearity():
    from sos_amd import transform as T
    a = torch.from_numpy(hashed(9, (2, 9000)).astype(np.float32)).cuda()
    b = torch.from_numpy(hashed(10, (2, 9000)).astype(np.float32)).cuda()
    lhs = T.stft_batch(a + 2 * b)
    rhs = T.stft_batch(a) + 2 * T.stft_batch(b)
    assert float((lhs - rhs).abs().max()) < 1e-3 * float(rhs.abs().max())


def test_mask_ops_golden(golden):
    from sos_amd import transform as T
    g = golden("maskops")
    Y = torch.from_numpy(g["Y"]).cuda()
    crm = torch.from_numpy(g["crm"]).cuda()
    rec = T.batch_fast_icRM_sigmoid(Y, crm)
    assert rel_err(rec.cpu().numpy(), g["rec"]) < 1e-5
    Y1 = g["Y"][0].transpose(1, 2, 0)
    c1 = g["crm"][0].transpose(1, 2, 0)
    assert rel_err(T.fast_icRM_sigmoid(Y1, c1), g["rec1"]) < 1e-5
    assert rel_err(T.fast_cRM_sigmoid(g["S1"], Y1.astype(np.float64)), g["tgt"]) < 1e-5


def test_mask_apply_backward_matches_autograd():
    from sos_amd import transform as T
    from oracle import nets as onet
    Y = torch.from_numpy(hashed(11, (2, 2, 16, 12), 2.0).astype(np.float32))
    crm = torch.from_numpy((0.5 + 0.45 * hashed(12, (2, 2, 16, 12))).astype(np.float32))
    gout = torch.from_numpy(hashed(13, (2, 2, 16, 12)).astype(np.float32))
    c_ref = crm.clone().requires_grad_(True)
    onet.mask_apply(Y, c_ref).backward(gout)
    c_gpu = crm.cuda().requires_grad_(True)
    T.batch_fast_icRM_sigmoid(Y.cuda(), c_gpu).backward(gout.cuda())
    assert rel_err(c_gpu.grad.cpu().numpy(), c_ref.grad.numpy()) < 1e-5


def test_bits_to_mask_bit_exact(golden):
    from sos_amd import tools
    g = golden("bitmask")
    for i in range(20):
        n = int(g[f"n{i}"])
        bits = "".join(str(int(b)) for b in g[f"bits{i}"])
        m = tools.convert_bitstreammask_to_audiomask(np.zeros(n, np.float32), float(g[f"ratio{i}"]), bits)
        want = np.unpackbits(g[f"mask{i}"])[:n]
        assert m.dtype == np.float32 and np.array_equal(m.astype(np.uint8), want), i
    with pytest.raises(RuntimeError):
        tools.convert_bitstreammask_to_audiomask(np.zeros(100, np.float32), 466.6, "012")


def test_bits_to_mask_batch_and_masked_signal():
    from sos_amd import tools
    B, nfr, n = 64, 60, 28000
    bits = (hashed(14, (B, nfr)) > -0.4).astype(np.uint8)
    sig = hashed(15, (B, n)).astype(np.float32)
    mask, masked = tools.bits_to_mask_batch(torch.from_numpy(bits).cuda(), 14000 / 30.0, n, torch.from_numpy(sig).cuda())
    for b in (0, 31, 63):
        want = ofe.convert_bitstreammask_to_audiomask(sig[b], 14000 / 30.0, list(bits[b]))
        assert np.array_equal(mask[b].cpu().numpy(), want)
        assert np.array_equal(masked[b].cpu().numpy(), sig[b] * want)


def test_threshold_bits():
    from sos_amd import tools
    lg = torch.tensor([[-3.0, -1e-3, 0.0, 1e-3, 2.0]], device="cuda")
    bits, conf = tools.threshold_bits(lg)
    assert bits.cpu().tolist() == [[0, 0, 1, 1, 1]]
    assert rel_err(conf.cpu().numpy(), torch.sigmoid(lg.cpu()).numpy()) < 1e-6


def test_bits_to_mask_randomised_bit_exact():
    """Integer work is bit-exact: 40 seeded random cases over frame counts, sample rates / frame rates (non-integer
    samples per frame) and signal lengths that do not match the bit stream, incl. isolated single bits and tails."""
    from sos_amd import tools
    rng = np.random.default_rng(2024)
    # the last two are shorter than the kernel's 16-sample fast-path bound: they take the general path
    ratios = [14000 / 30.0, 16000 / 25.0, 44100 / 29.97, 8000 / 24.0, 14000 / 60.0, 22050 / 30.0, 12.5, 7.3]
    for case in range(40):
        nfr = int(rng.integers(1, 700))
        ratio = ratios[case % len(ratios)]
        p_flip = [0.02, 0.2, 0.5][case % 3]
        bits = np.zeros(nfr, dtype=np.uint8)
        cur = int(rng.integers(0, 2))
        for i in range(nfr):                       # runs with occasional isolated frames
            if rng.random() < p_flip:
                cur ^= 1
            bits[i] = cur
        n = max(8, int(nfr * ratio) + int(rng.integers(-300, 300)))
        want = ofe.convert_bitstreammask_to_audiomask(np.zeros(n, np.float32), ratio, list(bits))
        got = tools.bits_to_mask_batch(torch.from_numpy(bits[None]).cuda(), ratio, n)[0].cpu().numpy()
        assert np.array_equal(got, want), (case, nfr, ratio, n)


def test_add_signals_golden_and_edge_cases(golden):
    """a16 (M2/tools.py:217-276): the PRODUCT functions (tools.add_signals = numpy API, tools.add_signals_batch = device
    API, both sos_add_signals_f32) against vectors produced by the imported reference, plus its edge cases: a list of
    noises, a silent signal, a silent noise, norm=None."""
    from sos_amd import tools
    g = golden("addsignals")
    for snr in (-10, 0, 7):
        m, c, n = tools.add_signals(g["sig"], [g["noi"]], snr, 0.5)
        assert isinstance(n, list) and m.dtype == np.float32
        # the reference sums float32 energies pairwise in float32, the kernel in f64: 1e-7-level differences of the gains
        assert np.allclose(m, g[f"mixed_{snr}"], atol=3e-7) and np.allclose(c, g[f"clean_{snr}"], atol=3e-7)
        assert np.allclose(n[0], g[f"noise_{snr}"], atol=3e-7)
        assert abs(np.max(np.abs(m)) - 0.5) < 1e-6
        assert np.allclose(m, c + n[0], atol=1e-6)
    # batched, different SNR per clip, against the oracle restatement
    sig = np.stack([g["sig"], g["noi"][::-1] * 0.3, np.zeros_like(g["sig"])])
    noi = np.stack([g["noi"], g["sig"], g["noi"]])
    snrs = [3.0, -7.0, 0.0]
    M, C, N = tools.add_signals_batch(torch.from_numpy(sig).cuda(), torch.from_numpy(noi).cuda(), snrs)
    for i in range(3):
        m, c, n = ofe.add_signals(sig[i], noi[i], snrs[i], 0.5)
        assert np.allclose(M[i].cpu().numpy(), m, atol=3e-7) and np.allclose(C[i].cpu().numpy(), c, atol=3e-7)
        assert np.allclose(N[i].cpu().numpy(), n, atol=3e-7)
    # two noises, no normalisation; silent noise left alone
    m, c, n = tools.add_signals(g["sig"], [g["noi"], np.zeros_like(g["noi"])], 5, None)
    m2, c2, n2 = ofe.add_signals(g["sig"], g["noi"], 5, None)
    assert np.allclose(m, m2, atol=3e-6) and np.array_equal(c, g["sig"]) and not n[1].any()


def test_power_law_front_end():
    """`power=True` of fast_stft / fast_istft (M1/transform.py:178-202, unused by the reference's callers): sign(x)|x|^0.3
    before the STFT, ^(1/0.3) after the ISTFT; product functions vs the oracle restatement."""
    from sos_amd import transform
    x = (hashed(77, (14000,)) * 0.4).astype(np.float32)
    x[::97] = 0.0
    assert rel_err(transform.power_law(x), ofe.power_law(x)) < 2e-6
    assert transform.power_law(x).dtype == np.float64 and not transform.power_law(x)[::97].any()
    S = transform.fast_stft(x, power=True)
    assert rel_err(S, ofe.fast_stft(ofe.power_law(x).astype(np.float32))) < 1e-5
    y = transform.fast_istft(S, power=True)
    want = ofe.power_law(ofe.fast_istft(ofe.fast_stft(ofe.power_law(x).astype(np.float32))), 1.0 / 0.3)
    assert y.shape == want.shape and rel_err(y, want) < 1e-4
    assert rel_err(y, x[:len(y)]) < 1e-3                       # companding then expanding is the identity
