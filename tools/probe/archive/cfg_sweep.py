"""Correctness of EVERY candidate conv tiling (SOS_CONV_FORCE_CFG) vs torch, several shapes."""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import numpy as np, torch, torch.nn.functional as F
    from sos_amd import engine as E, _lib as L
    from util import hashed, rel_err
    from test_gpu_train_ops import _act_from_nchw, _act_to_nchw
    E.AUTOTUNE = False
    bad = 0
    shapes = [(2, 48, 48, 32, 24, (5, 5), (2, 1), 1, "zeros"), (2, 16, 48, 32, 24, (5, 5), (2, 1), 1, "zeros"),
              (2, 48, 16, 32, 24, (5, 5), (2, 1), 1, "zeros"), (1, 96, 96, 70, 45, (5, 5), (8, 8), 1, "zeros"),
              (2, 64, 128, 33, 29, (5, 5), (1, 1), 2, "reflect"), (1, 256, 256, 16, 23, (3, 3), (4, 4), 1, "reflect"),
              (2, 48, 8, 256, 30, (1, 1), (1, 1), 1, "zeros")]
    for x3 in (False, True):
        for (B, ci, co, H, W, k, dil, st, pm) in shapes:
            x = torch.from_numpy(hashed(1, (B, ci, H, W)).astype(np.float32))
            w = torch.from_numpy(hashed(2, (co, ci, k[0], k[1]), 0.1).astype(np.float32))
            xa, xheld = _act_from_nchw(x, x3)
            pad = ((k[0] - 1) // 2 * dil[0], (k[1] - 1) // 2 * dil[1])
            xp = F.pad(xheld, (pad[1], pad[1], pad[0], pad[0]), mode="reflect") if pm == "reflect" else xheld
            wq = w.to(torch.bfloat16).float() + ((w - w.to(torch.bfloat16).float()).to(torch.bfloat16).float() if x3 else 0)
            want = F.conv2d(xp, wq, None, st, (0, 0) if pm == "reflect" else pad, dil)
            Ho, Wo = want.shape[2], want.shape[3]
            wp = E.pack_weight(w.cuda(), E.pad_to(ci, 16), x3)
            one = torch.ones(wp.shape[1], device="cuda"); zero = torch.zeros(wp.shape[1], device="cuda")
            dst = E.Act(B, Ho, Wo, E.pad_to(co, 16), x3, torch.device("cuda"))
            dst.t.fill_(float("nan"))
            E.conv_to_act(xa, 0, E.pad_to(ci, 16), wp, k[0], k[1], co, one, zero, L.ACT_NONE, dst, cout_store=dst.cs, stride=st, dil=dil, pad=pad,
                          pad_mode=L.PAD_REFLECT if pm == "reflect" else L.PAD_ZERO, Ho=Ho, Wo=Wo)
            got = _act_to_nchw(dst, co)
            e = rel_err(got, want) if not torch.isnan(got).any() else float("nan")
            ok = e < (2e-5 if x3 else 8e-3)
            if not ok:
                bad += 1
                print("  BAD cfg", os.environ.get("SOS_CONV_FORCE_CFG"), "x3" if x3 else "bf16", (B, ci, co, H, W, k, dil, st, pm), "err", e)
    sys.exit(1 if bad else 0)
else:
    nbad = 0
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        env = dict(os.environ, SOS_CONV_FORCE_CFG=str(k))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        out = (r.stdout + r.stderr).strip().splitlines()
        msgs = [l for l in out if "BAD" in l or "Error" in l or "error" in l]
        print("cfg", k, "rc", r.returncode, *msgs[:6], sep="\n   " if msgs else " ")
        nbad += r.returncode != 0
    print("configs with failures:", nbad)
