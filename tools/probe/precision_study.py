"""CPU numerics study (no GPU): which 16-bit roundings of the eval forward dominate the error of the detector logits /
the denoiser mask / the waveform.  Emulates IEEE-half storage (q = x.half().float()) at selectable points of the ORACLE
network (test infrastructure; this tool is not part of the product path) and prints the relative max error per group.

  python tools/probe/precision_study.py det      # detector logits
  python tools/probe/precision_study.py jm       # denoiser n_pred / mask / reconstructed spectrogram
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import frontend as ofe   # noqa: E402
from oracle import nets as onet      # noqa: E402

DT = torch.float16 if os.environ.get("STUDY_DT", "fp16") == "fp16" else torch.bfloat16


def q(x):
    return x.to(DT).float()


class Flags:
    conv_w = False       # conv weights rounded
    conv_act = False     # block outputs rounded
    conv_in = False      # network input rounded
    lstm_x = False       # LSTM input features + W_ih rounded (projection GEMM operands)
    lstm_h = False       # W_hh and h rounded (recurrent product operands, stored h)
    fc = False           # FC operands rounded
    only_prefix = None   # restrict conv rounding to blocks whose prefix starts with this


FL = Flags()


def _match(prefix):
    return FL.only_prefix is None or prefix.startswith(FL.only_prefix)


_conv_block, _down_block, _up_block, _lstm, _linear = onet.conv_block, onet.down_block, onet.up_block, onet.lstm_bidir, onet.linear


def _sdq(sd, keys):
    out = dict(sd)
    for k in keys:
        out[k] = q(sd[k])
    return out


def conv_block(x, sd, prefix, dilation, training, stats_out=None):
    if FL.conv_w and _match(prefix):
        sd = _sdq(sd, [prefix + ".block.0.weight"])
    y = _conv_block(x, sd, prefix, dilation, training, stats_out)
    return q(y) if FL.conv_act and _match(prefix) else y


def down_block(x, sd, prefix, k, stride, dilation, training, stats_out=None, bn=True, act=True):
    if FL.conv_w and _match(prefix):
        sd = _sdq(sd, [prefix + ".block.1.weight"])
    y = _down_block(x, sd, prefix, k, stride, dilation, training, stats_out, bn, act)
    return q(y) if FL.conv_act and _match(prefix) and bn else y      # the last block writes f32


def up_block(x, sd, prefix, training, stats_out=None):
    if FL.conv_w and _match(prefix):
        sd = _sdq(sd, [prefix + ".block.0.weight"])
    y = _up_block(x, sd, prefix, training, stats_out)
    return q(y) if FL.conv_act and _match(prefix) else y


def lstm_bidir(x, sd, prefix):
    T, B, _ = x.shape
    outs = []
    for sfx in ("", "_reverse"):
        wih, whh = sd[f"{prefix}.weight_ih_l0{sfx}"], sd[f"{prefix}.weight_hh_l0{sfx}"]
        bias = sd[f"{prefix}.bias_ih_l0{sfx}"] + sd[f"{prefix}.bias_hh_l0{sfx}"]
        H = whh.shape[1]
        xp = (q(x) @ q(wih).t() if FL.lstm_x else x @ wih.t()) + bias
        if FL.lstm_h:
            whh = q(whh)
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        hs = [None] * T
        order = range(T) if sfx == "" else range(T - 1, -1, -1)
        for t in order:
            g = xp[t] + h @ whh.t()
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            if FL.lstm_h:
                h = q(h)
            hs[t] = h
        outs.append(torch.stack(hs, 0))
    return torch.cat(outs, dim=2)


def linear(x, sd, prefix):
    if FL.fc:
        return q(x) @ q(sd[prefix + ".weight"]).t() + sd[prefix + ".bias"]
    return _linear(x, sd, prefix)


onet.conv_block, onet.down_block, onet.up_block, onet.lstm_bidir, onet.linear = conv_block, down_block, up_block, lstm_bidir, linear


def setf(**kw):
    for k in ("conv_w", "conv_act", "conv_in", "lstm_x", "lstm_h", "fc"):
        setattr(FL, k, False)
    FL.only_prefix = None
    for k, v in kw.items():
        setattr(FL, k, v)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "det"
    torch.set_num_threads(8)
    from sos_amd.dataset import _synth_raw
    nclip = int(os.environ.get("STUDY_CLIPS", "2"))
    raw = dict(mixed=[], bits=[])
    for i in range(nclip):                      # host-only version of dataset.synth_batch (that one mixes on the GPU)
        sp, nz, bits, snr = _synth_raw(i, 28000, 14000, 30.0, None)
        m = ofe.convert_bitstreammask_to_audiomask(sp, 14000 / 30.0, list(bits))
        raw["mixed"].append(ofe.add_signals(sp * (1 - m), nz, snr, 0.5)[0].astype(np.float32))
        raw["bits"].append(bits)
    S = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in raw["mixed"]]).astype(np.float32))
    ALL = dict(conv_w=True, conv_act=True, lstm_x=True, lstm_h=True, fc=True)
    if which == "det":
        sd = onet.closed_form_state(onet.detector_spec(), seed=1)
        with torch.no_grad():
            setf()
            ref = onet.detector_forward(sd, S, 60)
            print("logits: max %.3f  min|logit| %.4f" % (ref.abs().max(), ref.abs().min()))
            for name, fl in [("all", ALL), ("conv_w", dict(conv_w=True)), ("conv_act", dict(conv_act=True)),
                             ("conv (w+act)", dict(conv_w=True, conv_act=True)),
                             ("lstm_x", dict(lstm_x=True)), ("lstm_h", dict(lstm_h=True)), ("fc", dict(fc=True)),
                             ("head (lstm+fc)", dict(lstm_x=True, lstm_h=True, fc=True)),
                             ("all but lstm_x", dict(ALL, lstm_x=False)),
                             ("all but head", dict(conv_w=True, conv_act=True))]:
                setf(**fl)
                out = onet.detector_forward(sd, q(S) if fl.get("conv_act") else S, 60)
                print("  %-18s logits rel err %.2e   abs %.2e" % (name, rel(out, ref), float((out - ref).abs().max())))
    else:
        sd = onet.closed_form_state(onet.joint_spec(), seed=2)
        bits = raw["bits"]
        Sn = torch.from_numpy(np.stack([ofe.fast_stft(w * ofe.convert_bitstreammask_to_audiomask(w, 14000 / 30.0, list(b))).transpose(2, 0, 1)
                                        for w, b in zip(raw["mixed"], bits)]).astype(np.float32))
        with torch.no_grad():
            setf()
            n_ref, m_ref = onet.joint_forward(sd, S, Sn)
            rec_ref = onet.mask_apply(S, m_ref)
            print("mask range %.4f..%.4f" % (m_ref.min(), m_ref.max()))
            for name, fl in [("all", ALL), ("conv_w", dict(conv_w=True)), ("conv_act", dict(conv_act=True)),
                             ("stage1 convs", dict(conv_w=True, conv_act=True, only_prefix="stage1")),
                             ("stage2 convs", dict(conv_w=True, conv_act=True, only_prefix="stage2")),
                             ("lstm_x", dict(lstm_x=True)), ("lstm_h", dict(lstm_h=True)), ("fc", dict(fc=True)),
                             ("head (lstm+fc)", dict(lstm_x=True, lstm_h=True, fc=True)),
                             ("all but head", dict(conv_w=True, conv_act=True)),
                             ("all but lstm_x", dict(ALL, lstm_x=False))]:
                setf(**fl)
                qi = fl.get("conv_act")
                n_p, m = onet.joint_forward(sd, q(S) if qi else S, q(Sn) if qi else Sn)
                rec = onet.mask_apply(S, m)
                print("  %-18s n_pred %.2e  mask %.2e  rec %.2e" % (name, rel(n_p, n_ref), rel(m, m_ref), rel(rec, rec_ref)))




def per_layer():
    """Weight + activation rounding of ONE block at a time (stage 2 and stage 1): where does the mask error come from?"""
    torch.set_num_threads(8)
    from sos_amd.dataset import _synth_raw
    raw = dict(mixed=[], bits=[])
    for i in range(2):
        sp, nz, bits, snr = _synth_raw(i, 28000, 14000, 30.0, None)
        m = ofe.convert_bitstreammask_to_audiomask(sp, 14000 / 30.0, list(bits))
        raw["mixed"].append(ofe.add_signals(sp * (1 - m), nz, snr, 0.5)[0].astype(np.float32))
        raw["bits"].append(bits)
    S = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in raw["mixed"]]).astype(np.float32))
    Sn = torch.from_numpy(np.stack([ofe.fast_stft(w * ofe.convert_bitstreammask_to_audiomask(w, 14000 / 30.0, list(b))).transpose(2, 0, 1)
                                    for w, b in zip(raw["mixed"], raw["bits"])]).astype(np.float32))
    sd = onet.closed_form_state(onet.joint_spec(), seed=2)
    with torch.no_grad():
        setf()
        n_ref, m_ref = onet.joint_forward(sd, S, Sn)
        rec_ref = onet.mask_apply(S, m_ref)
        names = [f"stage2.encoder_x.{i}" for i in range(15)] + [f"stage2.encoder_n.{i}" for i in range(15)] + \
                ["stage1.down", "stage1.mid", "stage1.up"]
        for nm in names:
            for what in ("conv_w", "conv_act"):
                setf(**{what: True, "only_prefix": nm})
                n_p, m = onet.joint_forward(sd, S, Sn)
                rec = onet.mask_apply(S, m)
                print("  %-22s %-8s n_pred %.2e  mask %.2e  rec %.2e" % (nm, what, rel(n_p, n_ref), rel(m, m_ref), rel(rec, rec_ref)), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "layers":
    per_layer()

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "layers"):
    main()
