#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE modules from /root/reference.

Runs only in the build container (the reference never travels).  It does two things:
  1. pins oracle/ against the imported reference (hard asserts below), and
  2. writes input/output vectors so the pin can be re-checked anywhere
     (tests/test_oracle_*.py) without the reference.

Nothing of the reference's source text is stored; only arrays of numbers.
Usage:  python tests/golden/make_goldens.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
M1 = os.path.join(REF, "model_1_silent_interval_detection/audioonly_model")
M2 = os.path.join(REF, "model_2_audio_denoising/audio_denoising_model")
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import frontend as ofe  # noqa: E402
from oracle import nets as onet     # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    """Empty stand-ins for third-party packages the reference imports at module top
    but which the functions we call never touch."""
    _stub("librosa")
    _stub("librosa.display")
    sys.modules["librosa"].display = sys.modules["librosa.display"]
    _stub("backports")
    _stub("backports.tempfile", TemporaryDirectory=None)
    tv = _stub("torchvision")
    _noop = lambda *a, **k: None
    tr = _stub("torchvision.transforms", Compose=_noop, Resize=_noop, ToTensor=_noop, RandomRotation=_noop,
               RandomHorizontalFlip=_noop, Normalize=_noop, CenterCrop=_noop, RandomCrop=_noop, ColorJitter=_noop)
    tv.transforms = tr
    _stub("imageio")
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches", "cv2", "PIL", "PIL.Image",
                 "tqdm", "joblib", "sklearn", "sklearn.metrics", "pandas"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _stub(name, tqdm=lambda x, **k: x, Image=None, Parallel=None, delayed=None)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    sys.path.insert(0, os.path.dirname(path))
    try:
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
        sys.path.pop(0)
    return mod


def hashed(idx, shape, scale=1.0):
    return (onet._hash_uniform(idx, int(np.prod(shape))) * scale).reshape(shape)


def spec_input(idx, B, T, F=256):
    """Closed-form spectrogram-like input (exact across machines)."""
    u = hashed(idx, (B, 2, F, T))
    env = 2.0 / (1.0 + np.arange(F, dtype=np.float64) / 16.0)
    return torch.from_numpy((u * env[None, None, :, None]).astype(np.float32))


def silent_gate(x):
    T = x.shape[-1]
    g = ((np.arange(T) // 10) % 3 == 0).astype(np.float32)
    return x * torch.from_numpy(g)[None, None, None, :]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def main():
    torch.set_num_threads(8)
    install_stubs()
    ref_tf = load(os.path.join(M2, "transform.py"), "ref_transform")
    ref_tools2 = load(os.path.join(M2, "tools.py"), "ref_tools2")
    ref_tools1 = load(os.path.join(M1, "tools.py"), "ref_tools1")
    ref_net1 = load(os.path.join(M1, "networks.py"), "ref_networks1")
    ref_net2 = load(os.path.join(M2, "networks.py"), "ref_networks2")

    # ------------------------------------------------------------------ front-end
    fe = {}
    for i, n in enumerate((14000, 28000, 28123)):
        y = (hashed(50 + i, (n,)) * 0.3 * (0.5 + 0.5 * np.sin(np.arange(n) / 900.0))).astype(np.float32)
        S = ofe.fast_stft(y)
        # cross-check against torch.stft (independent implementation of the same definition)
        St = torch.stft(torch.from_numpy(y), 510, 158, 400, window=torch.hann_window(400, periodic=True),
                        center=True, pad_mode="reflect", return_complex=True).numpy()
        e = np.max(np.abs(S[:, :, 0] + 1j * S[:, :, 1] - St))
        assert e < 5e-6, e
        yi = ofe.fast_istft(S)
        yt = torch.istft(torch.from_numpy(St), 510, 158, 400, window=torch.hann_window(400, periodic=True),
                         center=True).numpy()
        assert len(yi) == 158 * (S.shape[1] - 1)
        m = min(len(yi), len(yt))
        assert np.max(np.abs(yi[:m] - yt[:m])) < 5e-6
        fe[f"wave{i}"] = y
        fe[f"stft{i}"] = S.astype(np.float32)
        fe[f"istft{i}"] = yi
    Sr = hashed(60, (256, 40, 2)).astype(np.float32)
    fe["rand_spec"] = Sr
    fe["rand_spec_istft"] = ofe.fast_istft(Sr)
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), **fe)
    print("frontend ok")

    # ------------------------------------------------------------------- mask ops
    mk = {}
    Y = hashed(70, (3, 2, 32, 20), 3.0).astype(np.float32)
    crm = (0.5 + 0.5 * hashed(71, (3, 2, 32, 20))).astype(np.float32)
    crm.reshape(-1)[:6] = [1e-6, 1 - 1e-6, 0.5, 1e-3, 0.999, 0.25]
    crm = np.clip(crm, 1e-7, 1 - 1e-7)
    rec = ref_tf.batch_fast_icRM_sigmoid(torch.from_numpy(Y), torch.from_numpy(crm)).numpy()
    assert rel_err(ofe.batch_fast_icRM_sigmoid(Y, crm), rec) < 1e-5
    assert rel_err(onet.mask_apply(torch.from_numpy(Y), torch.from_numpy(crm)).numpy(), rec) < 1e-6
    Y1 = Y[0].transpose(1, 2, 0).astype(np.float64)
    c1 = crm[0].transpose(1, 2, 0).astype(np.float64)
    rec1 = ref_tf.fast_icRM_sigmoid(Y1, c1)
    assert np.array_equal(ofe.fast_icRM_sigmoid(Y1, c1), rec1)
    S1 = hashed(72, Y1.shape, 2.0)
    tgt = ref_tf.fast_cRM_sigmoid(S1, Y1)
    assert np.array_equal(ofe.fast_cRM_sigmoid(S1, Y1), tgt)
    mk.update(Y=Y, crm=crm, rec=rec, rec1=rec1, S1=S1, tgt=tgt)
    np.savez_compressed(os.path.join(OUT, "maskops.npz"), **mk)
    print("maskops ok")

    # ---------------------------------------------------------------- bits -> mask
    bm = {}
    rng_bits = []
    for i in range(20):
        nfr = 60 if i < 12 else 645
        u = onet._hash_uniform(80 + i, nfr)
        if i % 4 == 0:      # isolated single bits
            bits = (u > 0.9).astype(int)
        elif i % 4 == 1:
            bits = (u > -0.9).astype(int)
        else:               # runs
            bits = (np.convolve(u, np.ones(7) / 7, mode="same") > 0).astype(int)
        rng_bits.append(bits)
    for i, bits in enumerate(rng_bits):
        sr, fps = (14000, 30.0) if i % 3 else (14000, 29.97)
        n = int(len(bits) / fps * sr) + (0 if i % 2 else 17)
        refsig = np.zeros(n, dtype=np.float32)
        s = "".join(str(b) for b in bits)
        m2 = ref_tools2.convert_bitstreammask_to_audiomask(refsig, float(sr) / fps, s)
        m1 = ref_tools1.convert_bitstreammask_to_audiomask(refsig, float(sr) / fps, list(bits))
        mo = ofe.convert_bitstreammask_to_audiomask(refsig, float(sr) / fps, s)
        assert np.array_equal(m2, m1) and np.array_equal(m2, mo), i
        bm[f"bits{i}"] = bits.astype(np.uint8)
        bm[f"ratio{i}"] = np.float64(float(sr) / fps)
        bm[f"n{i}"] = np.int64(n)
        bm[f"mask{i}"] = np.packbits(m2.astype(np.uint8))
    np.savez_compressed(os.path.join(OUT, "bitmask.npz"), **bm)
    print("bitmask ok")

    # -------------------------------------------------------------- nearest index
    ni = {}
    for (a, b) in [(178, 60), (90, 89), (89, 30), (887, 300), (1903, 645), (128, 128), (46, 45), (256, 256), (180, 178)]:
        t = torch.arange(a, dtype=torch.float32)[None, None, :]
        idx = torch.nn.functional.interpolate(t, size=b).numpy()[0, 0].astype(np.int64)
        assert np.array_equal(idx, ofe.nearest_index(a, b)), (a, b)
        ni[f"{a}_{b}"] = idx
    np.savez_compressed(os.path.join(OUT, "nearest.npz"), **ni)

    # ---------------------------------------------------------------- add_signals
    ad = {}
    sig = hashed(90, (5000,), 0.2).astype(np.float32)
    noi = hashed(91, (5000,), 0.7).astype(np.float32)
    for snr in (-10, 0, 7):
        a, b, c = ref_tools2.add_signals(sig, [noi], snr, norm=0.5)
        a2, b2, c2 = ofe.add_signals(sig, noi, snr, norm=0.5)
        assert np.allclose(a, a2, rtol=0, atol=1e-7) and np.allclose(c[0], c2, rtol=0, atol=1e-7)
        ad[f"mixed_{snr}"] = a
        ad[f"clean_{snr}"] = b
        ad[f"noise_{snr}"] = c[0]
    ad["sig"] = sig
    ad["noi"] = noi
    np.savez_compressed(os.path.join(OUT, "addsignals.npz"), **ad)
    print("addsignals ok")

    # ------------------------------------------------------------------- networks
    nets = {}
    det = ref_net1.get_network()
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    assert list(det.state_dict().keys()) == list(sd1.keys())
    for k, v in det.state_dict().items():
        assert tuple(v.shape) == tuple(sd1[k].shape), k
    det.load_state_dict(sd1, strict=True)

    class Cfg:
        kernel_sizes = onet.CTX_KERNELS
        dilations = onet.CTX_DILATIONS
    jm = ref_net2.get_network(Cfg())
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    assert list(jm.state_dict().keys()) == list(sd2.keys())
    for k, v in jm.state_dict().items():
        assert tuple(v.shape) == tuple(sd2[k].shape), k
    jm.load_state_dict(sd2, strict=True)

    for tag, B, T, nfr in (("a", 1, 178, 60), ("b", 2, 89, 30)):
        x = spec_input(100 + B, B, T)
        n = silent_gate(x)
        det.eval()
        jm.eval()
        with torch.no_grad():
            lo_ref = det(s=x, v_num_frames=nfr)
            lo_or = onet.detector_forward(sd1, x, nfr)
            assert rel_err(lo_or, lo_ref) < 2e-5, rel_err(lo_or, lo_ref)
            np_ref, out_ref = jm(x, n)
            np_or, out_or = onet.joint_forward(sd2, x, n)
            assert rel_err(np_or, np_ref) < 2e-5 and rel_err(out_or, out_ref) < 2e-5
        print(tag, "eval: logits range", float(lo_ref.min()), float(lo_ref.max()),
              "n_pred absmax", float(np_ref.abs().max()), "mask range", float(out_ref.min()), float(out_ref.max()),
              "mask std", float(out_ref.std()))
        nets[f"det_logits_{tag}"] = lo_ref.numpy()
        nets[f"n_pred_{tag}"] = np_ref.numpy()
        nets[f"mask_{tag}"] = out_ref.numpy()

    # train-mode forward + losses + gradients (B=2, T=89)
    B, T, nfr = 2, 89, 30
    x = spec_input(100 + B, B, T)
    n = silent_gate(x)
    clean = spec_input(300, B, T) * 0.5
    full_noise = x - clean
    label = torch.from_numpy((onet._hash_uniform(301, B * nfr).reshape(B, nfr) > 0).astype(np.float32))
    det.train()
    jm.train()
    det.zero_grad()
    jm.zero_grad()
    lo = det(x, nfr)
    bce = torch.nn.BCEWithLogitsLoss()(lo, label)
    bce.backward()
    st1 = {}
    lo_o, l_o = onet.detector_loss({k: v.clone() for k, v in sd1.items()}, {"audio": x, "label": label}, True, st1)
    assert rel_err(lo_o.detach(), lo.detach()) < 5e-5
    assert abs(float(l_o["bce"]) - float(bce)) < 1e-5
    # condition scale of every PReLU slope gradient: d_slope = sum_{z<0} dy*z is a heavily cancelling sum of ~1e6
    # signed terms; sum_{z<0} |dy*z| is the magnitude a relative perturbation of the terms is amplified by (the tests
    # bound |d_slope - reference| by a fraction of it)
    slope_scale, hooks = {}, []
    for mname, mod in jm.named_modules():
        if isinstance(mod, torch.nn.PReLU):
            def fwd_hook(m, inp, outp, mname=mname):
                z = inp[0].detach()
                outp.register_hook(lambda dy, z=z, mname=mname: slope_scale.__setitem__(
                    mname + ".weight", float((dy.double() * z.double()).abs()[z < 0].sum())))
            hooks.append(mod.register_forward_hook(fwd_hook))
    n_pred, out = jm(x, n)
    rec = ref_tf.batch_fast_icRM_sigmoid(x, out)
    l1 = torch.nn.MSELoss()(n_pred, full_noise)
    l2 = torch.nn.MSELoss()(rec, clean)
    (l1 + l2).backward()
    for h in hooks:
        h.remove()
    nets["train_jm_slope_scale"] = np.array([slope_scale.get(k, np.nan) for k, _ in jm.named_parameters()])
    print("PReLU slope gradients vs their condition scale:",
          [(k, round(float(p.grad), 4), round(slope_scale[k], 3)) for k, p in jm.named_parameters() if k in slope_scale])
    st2 = {}
    (np_o, out_o), ls = onet.denoiser_losses({k: v.clone() for k, v in sd2.items()},
                                             {"mixed": x, "noise": n, "clean": clean, "full_noise": full_noise}, True, st2)
    assert rel_err(np_o.detach(), n_pred.detach()) < 5e-5 and rel_err(out_o.detach(), out.detach()) < 5e-5
    assert abs(float(ls["stage1"]) - float(l1)) < 1e-4 * abs(float(l1))
    assert abs(float(ls["stage2"]) - float(l2)) < 1e-4 * abs(float(l2))
    for k, v in st1.items():
        assert rel_err(v, det.state_dict()[k]) < 1e-4, k
    for k, v in st2.items():
        assert rel_err(v, jm.state_dict()[k]) < 1e-4, k
    nets["train_det_logits"] = lo.detach().numpy()
    nets["train_bce"] = np.float64(float(bce))
    nets["train_n_pred"] = n_pred.detach().numpy()
    nets["train_mask"] = out.detach().numpy()
    nets["train_l1"] = np.float64(float(l1))
    nets["train_l2"] = np.float64(float(l2))
    nets["train_label"] = label.numpy()
    for name, mod in (("det", det), ("jm", jm)):
        gn, gh = [], []
        for k, p in mod.named_parameters():
            g = p.grad.detach().reshape(-1).numpy()
            gn.append(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
            gh.append(np.pad(g[:8], (0, max(0, 8 - len(g)))))
        nets[f"train_{name}_gradnorm"] = np.array(gn)
        nets[f"train_{name}_gradhead"] = np.stack(gh)
        rs = [v.numpy().reshape(-1)[:4] for k, v in mod.state_dict().items() if k.endswith("running_var")]
        nets[f"train_{name}_running_var_head"] = np.stack([np.pad(r, (0, 4 - len(r))) for r in rs])
    np.savez_compressed(os.path.join(OUT, "networks.npz"), **nets)
    print("networks ok; train losses", float(bce), float(l1), float(l2))


if __name__ == "__main__":
    main()
