"""Child process of tests/test_gpu_determinism.py: one eval forward and one training step of both networks, digests of
every output and gradient on stdout (one `name sha256` line each)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sos_amd  # noqa: E402
from oracle import nets as onet  # noqa: E402
from sos_amd import transform  # noqa: E402
from sos_amd.common import MyConfig  # noqa: E402
from sos_amd.denoiser import networks as jnet  # noqa: E402
from sos_amd.detector import networks as dnet  # noqa: E402
from util import silent_gate, spec_input  # noqa: E402


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().float().cpu().numpy()).tobytes()).hexdigest()


def main():
    sos_amd.set_precision(sys.argv[1])
    B, T, nfr = int(sys.argv[2]), int(sys.argv[3]), 30
    det = dnet.get_network()
    det.load_state_dict(onet.closed_form_state(onet.detector_spec(), seed=1))
    jm = jnet.get_network(MyConfig())
    jm.load_state_dict(onet.closed_form_state(onet.joint_spec(), seed=2))
    det, jm = det.cuda(), jm.cuda()
    x = spec_input(100 + B, B, T).cuda()
    n = silent_gate(spec_input(100 + B, B, T)).cuda()
    with torch.no_grad():
        lo = det.eval()(x, nfr)
        n_pred, out = jm.eval()(x, n)
    print("eval_logits", digest(lo))
    print("eval_n_pred", digest(n_pred))
    print("eval_mask", digest(out))
    label = (spec_input(7, B, nfr, 1)[:, 0, 0] > 0).float().cuda()
    torch.nn.functional.binary_cross_entropy_with_logits(det.train()(x, nfr), label).backward()
    n_pred, out = jm.train()(x, n)
    rec = transform.batch_fast_icRM_sigmoid(x, out)
    (torch.nn.functional.mse_loss(n_pred, x * 0.5) + torch.nn.functional.mse_loss(rec, x * 0.25)).backward()
    print("train_n_pred", digest(n_pred))
    for tag, m in (("det", det), ("jm", jm)):
        h = hashlib.sha256()
        for k, p in m.named_parameters():
            h.update(np.ascontiguousarray(p.grad.detach().float().cpu().numpy()).tobytes())
        print(f"grads_{tag}", h.hexdigest())
        h = hashlib.sha256()
        for k, b in m.named_buffers():
            h.update(np.ascontiguousarray(b.detach().float().cpu().numpy()).tobytes())
        print(f"buffers_{tag}", h.hexdigest())


if __name__ == "__main__":
    main()
