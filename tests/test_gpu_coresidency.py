"""Victim / aggressor harness for the round-1 wrong-result event: the VALU `stft_kernel` of round 1
returned wrong imaginary parts in 20-90 % of its launches whenever workgroups of the first detector layer
(`conv16_kernel<3,1,true>`: cin = 16, cout = 48, 1x7 taps) were co-resident on its CU.  That victim was rewritten (the
MFMA STFT) but the aggressor still ships, so every LDS-using kernel family of the library is run here BESIDE it:

    stream A (its own host thread): the aggressor in a loop, its own output re-checked as it goes
    stream B (another host thread): the victim N times, every output bit-compared with its solo run

A single mismatch fails the test and names the victim.  Sizes are small enough that victim workgroups really share CUs
with aggressor workgroups (a chip-filling victim would simply queue behind it)."""
import ctypes
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NIT = 300


def _aggressors(dev):
    """name -> callable launching it once + a checker of its own output."""
    from sos_amd import _lib as L, engine as E, transform
    from sos_amd.dataset import synth_batch
    from sos_amd.detector import networks as dnet
    det = dnet.get_network().cuda().eval()
    plan = det._cache.get(det, det._build_plan)
    wave = torch.from_numpy(synth_batch(500, 4)["mixed"]).cuda()
    S = transform.stft_batch(wave[1:4, :14000].contiguous())
    B, _, F, T = S.shape
    a0 = E.pack_input(S, False)
    out = {}
    for tag, li in (("conv16<3,1,true> (detector layer 0, the round-1 aggressor)", 0), ("conv16<3,3,*> (detector layer 2)", 2)):
        lp = plan["enc"][li]
        src = a0 if li == 0 else E.Act(B, F, T, 48, False, dev)
        if li:
            src.t.normal_()
        dst = E.Act(B, F, T, 48, False, dev)

        def run(lp=lp, src=src, dst=dst):
            E.conv_to_act(src, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU,
                          dst, cout_store=48, dil=lp["dil"], pad=lp["pad"], Ho=F, Wo=T)
        run()
        torch.cuda.synchronize()
        ref = dst.t.clone()
        out[tag] = (run, (lambda dst=dst, ref=ref: torch.equal(dst.t, ref)))
    return out


def _victims(dev):
    """name -> zero-argument callable returning a tensor (or tuple of tensors) that must be bit-reproducible."""
    from sos_amd import _lib as L, agent, engine as E, metrics, tools, train_ops as TO, transform
    from sos_amd.dataset import synth_batch
    raw = synth_batch(700, 4)
    wave = torch.from_numpy(raw["mixed"]).cuda()
    clean = torch.from_numpy(raw["clean"]).cuda()
    bits = torch.from_numpy(raw["bits"]).cuda()
    S = transform.stft_batch(wave)
    Bc, Hc, Wc, C = 3, 64, 45, 96
    x = E.Act(Bc, Hc, Wc, C, False, dev); x.t.normal_()
    dy = E.Act(Bc, Hc, Wc, C, False, dev); dy.t.normal_()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    gamma = torch.rand(C, device=dev) + 0.5
    saved = dict(scale=torch.rand(C, device=dev) + 0.5, shift=torch.randn(C, device=dev), mean=torch.randn(C, device=dev) * 0.1,
                 invstd=torch.rand(C, device=dev) + 0.5)

    def bn_stats():
        xv = E.view(x, 0, C)
        nblk = L.lib().sos_bn_stats_blocks(xv.npix)
        partial = torch.empty((2, C, nblk), dtype=torch.float32, device=dev)
        L.check(L.lib().sos_bn_stats(ctypes.byref(xv), L.ptr(partial), L.stream_ptr()), "sos_bn_stats")
        return partial

    def bn_apply():
        y = E.Act(Bc, Hc, Wc, C, False, dev)
        E.bn_apply(E.view(x, 0, C), saved["scale"], saved["shift"], L.ACT_RELU, None, y, 0, C)
        return y.t

    def bn_bwd():
        dx = E.Act(Bc, Hc, Wc, C, False, dev)
        dg, db, _ = TO.bn_bwd(dy, 0, x, 0, C, saved, gamma, L.ACT_RELU, None, dx)
        return dx.t, dg, db

    Bl, T, H = 16, 60, 100
    lstm_mod = torch.nn.LSTM(64, H, bidirectional=True, batch_first=True).to(dev)
    pk = E.lstm_pack(lstm_mod, False)
    xproj = torch.randn(Bl, T, 8 * H, device=dev)
    dh = E.Act(Bl, 1, T, E.pad_to(2 * H, 16), False, dev, zero=True); dh.t.normal_()
    gates0 = torch.zeros(Bl, T, 2, 4 * H, device=dev); cs0 = torch.zeros(Bl, T, 2, H, device=dev)
    h0 = E.Act(Bl, 1, T, E.pad_to(2 * H, 16), False, dev, zero=True)
    E.lstm(xproj, pk, Bl, T, H, h0, gates0, cs0)

    def lstm_fwd():
        h = E.Act(Bl, 1, T, E.pad_to(2 * H, 16), False, dev, zero=True)
        gates = torch.zeros(Bl, T, 2, 4 * H, device=dev); cs = torch.zeros(Bl, T, 2, H, device=dev)
        E.lstm(xproj, pk, Bl, T, H, h, gates, cs)
        return h.t, gates, cs

    def lstm_bwd():
        dgates = torch.empty(Bl, T, 2, 4 * H, device=dev)
        L.check(L.lib().sos_lstm_bidir_bwd(L.ptr(dh.t), dh.nseg * dh.cs, dh.dtype_code, dh.cs, L.ptr(gates0), L.ptr(cs0),
                                           L.ptr(pk["bh"]), L.ptr(pk["bl"]), Bl, T, H, L.ptr(dgates), L.stream_ptr()), "lstm bwd")
        return dgates

    a = torch.randn(4, 2, 256, 89, device=dev)
    b = torch.randn(4, 2, 256, 89, device=dev)
    lab = (torch.rand(4, 60, device=dev) > 0.5).float()
    logit = torch.randn(4, 60, device=dev)
    crm = torch.rand(4, 2, 256, 178, device=dev) * 0.9 + 0.05

    def mse():
        ar = a.clone().requires_grad_(True)
        l = agent.mse_loss(ar, b)
        l.backward()
        return l.detach().reshape(1), ar.grad

    def bce():
        xr = logit.clone().requires_grad_(True)
        l = agent.bce_with_logits_loss(xr, lab)
        l.backward()
        return l.detach().reshape(1), xr.grad

    # a 3x3 conv / its weight gradient on a small 64-channel map (conv_mfma_kernel / wgrad_kernel)
    Cc = 64
    xa = E.Act(2, 32, 45, Cc, False, dev); xa.t.normal_()
    ga = E.Act(2, 32, 45, Cc, False, dev); ga.t.normal_()
    w = E.pack_weight(torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05, Cc, False)

    def conv3():
        dst = E.Act(2, 32, 45, Cc, False, dev)
        E.conv_to_act(xa, 0, Cc, w, 3, 3, Cc, None, None, L.ACT_NONE, dst, cout_store=Cc, pad=(1, 1), Ho=32, Wo=45)
        return dst.t

    def wgrad3():
        dw = torch.empty(Cc, Cc, 3, 3, dtype=torch.float32, device=dev)
        E.wgrad(ga, 0, Cc, xa, 0, Cc, 3, 3, dw, pad=(1, 1))
        return dw

    ref16, deg16 = clean[0].contiguous(), wave[0].contiguous()
    return {
        "stft_mfma": lambda: transform.stft_batch(wave),
        "istft_mfma": lambda: transform.istft_batch(S),
        "bn_stats": bn_stats, "bn_act_apply": bn_apply, "bn_bwd (reduce + finalize + apply)": bn_bwd,
        "lstm_fwd": lstm_fwd, "lstm_bwd": lstm_bwd,
        "mse_kernel": mse, "bce_kernel": bce,
        "crm_apply": lambda: transform.batch_fast_icRM_sigmoid(S, crm),
        "bits_to_mask": lambda: tools.bits_to_mask_batch(bits, 14000 / 30.0, wave.shape[1], wave),
        "conv_mfma 3x3": conv3, "wgrad 3x3": wgrad3,
        "metric llr (LDS autocorrelation)": lambda: torch.from_numpy(np.ascontiguousarray(metrics.llr(ref16, deg16, 14000))),
        "metric wss (LDS spectra)": lambda: torch.from_numpy(np.ascontiguousarray(metrics.wss(ref16, deg16, 14000))).reshape(-1),
    }


def _flat(o):
    """Outputs as raw bit patterns (NaN == NaN: the comparison is about bits, not values)."""
    def bits(t):
        t = t.detach().clone().contiguous()
        return t.view({2: torch.int16, 4: torch.int32, 8: torch.int64}.get(t.element_size(), t.dtype)) if t.is_floating_point() else t
    return tuple(bits(t) for t in (o if isinstance(o, tuple) else (o,)))


@pytest.mark.parametrize("precision", ["fp16"])
def test_every_lds_kernel_family_is_bit_stable_beside_the_conv16_aggressor(precision):
    import sos_amd
    dev = torch.device("cuda")
    sos_amd.set_precision(precision)
    try:
        aggr = _aggressors(dev)
        victims = _victims(dev)
        refs = {}
        for name, vf in victims.items():
            r1, r2 = _flat(vf()), _flat(vf())
            torch.cuda.synchronize()
            assert all(torch.equal(p, q) for p, q in zip(r1, r2)), f"{name} is not reproducible even alone"
            refs[name] = r1
        failures = []
        for aname, (arun, acheck) in aggr.items():
            for name, vf in victims.items():
                bad = [0, 0]
                stop = threading.Event()
                bar = threading.Barrier(2)
                errs = []

                def victim_thread():
                    try:
                        sos_amd.set_precision(precision)
                        st = torch.cuda.Stream()
                        bar.wait()
                        with torch.cuda.stream(st):
                            for lo in range(0, NIT, 50):          # outputs are kept in rounds of 50 (memory)
                                outs = [_flat(vf()) for _ in range(50)]
                                st.synchronize()
                                bad[0] += sum(0 if all(torch.equal(p, q) for p, q in zip(o, refs[name])) else 1 for o in outs)
                    except Exception as e:      # noqa: BLE001
                        errs.append(e)
                    finally:
                        stop.set()

                def aggressor_thread():
                    try:
                        sos_amd.set_precision(precision)
                        st = torch.cuda.Stream()
                        bar.wait()
                        with torch.cuda.stream(st):
                            while not stop.is_set():
                                for _ in range(10):
                                    arun()
                                st.synchronize()
                                bad[1] += 0 if acheck() else 1
                    except Exception as e:      # noqa: BLE001
                        errs.append(e)
                        stop.set()

                ths = [threading.Thread(target=victim_thread), threading.Thread(target=aggressor_thread)]
                [t.start() for t in ths]
                [t.join() for t in ths]
                assert not errs, errs
                print(f"aggressor {aname[:28]:28s} victim {name:36s} corrupted {bad[0]:3d} / {NIT}, aggressor wrong {bad[1]}", flush=True)
                if bad[0] or bad[1]:
                    failures.append((aname, name, bad[0], bad[1]))
        assert not failures, failures
    finally:
        sos_amd.set_precision("bf16")
