import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import transform, engine as E, _lib as L
from sos_amd.dataset import synth_batch
sos_amd.set_precision("fp16")
NIT = 1500
base = torch.from_numpy(synth_batch(500, 4)["mixed"]).cuda()
x0 = base.contiguous()
ref = transform.stft_batch(x0)
dev = torch.device("cuda")
def mk_conv(cin, cout, k, dil):
    src = E.Act(3, 256, 89, cin, False, dev); src.t.normal_()
    w = E.pack_weight(torch.randn(cout, cin, k, k, device="cuda") * 0.05, cin, False)
    dst = E.Act(3, 256, 89, E.pad_to(cout, 16), False, dev)
    return lambda: E.conv_to_act(src, 0, cin, w, k, k, cout, None, None, L.ACT_NONE, dst, cout_store=dst.cs, dil=(dil, dil),
                                 pad=(dil * (k - 1) // 2,) * 2, Ho=256, Wo=89)
from sos_amd import common_nets as CN
from sos_amd.detector import networks as dnet
det = dnet.get_network().cuda().eval()
plan = det._cache.get(det, det._build_plan)
S = transform.stft_batch(base[1:4, :14000].contiguous())
B, _, F, T = S.shape; n = 30
def enc_only():
    a_ = E.pack_input(S, False)
    feat = torch.empty((B, n, 8 * F), dtype=E.act_dtype(), device=dev)
    CN.run_encoder(plan["enc"], a_, feat, 8 * F, 8 * F, 0, False, w_gather=CN.nearest_index(T, n, dev), T_out=n)
    return feat
feat0 = enc_only()
def lstm_only():
    return CN.run_lstm(plan["lstm"], (feat0, B, 1, n, 8 * F, 1), B, n, False, dev)
h0 = lstm_only()
def fc_only():
    f0, f2 = plan["fc0"], plan["fc2"]
    m = E.Act(B, 1, n, E.pad_to(f0["cout"], 16), False, dev)
    E.conv_to_act(h0, 0, f0["cin_store"], f0["w"], 1, 1, f0["cout"], f0["scale"], f0["shift"], L.ACT_RELU, m, cout_store=m.cs, Ho=1, Wo=n)
    out = torch.empty((B, n), dtype=torch.float32, device=dev)
    E.conv(m, 0, f2["cin_store"], f2["w"], 1, 1, 1, f2["scale"], f2["shift"], L.ACT_NONE, out=out, out_dtype=L.DT_F32, sb=n, sh=0, sw=1, sc=1, Ho=1, Wo=n)
a0 = E.pack_input(S, False)
acts = [E.Act(B, F, T, 48, False, dev) for _ in range(2)]
for a_ in acts: a_.t.normal_()
def layer(i):
    lp = plan["enc"][i]
    src = a0 if i == 0 else acts[0]
    def f():
        E.conv_to_act(src, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU,
                      acts[1], cout_store=48, dil=lp["dil"], pad=lp["pad"], Ho=F, Wo=T)
    return f
def featconv():
    lp = plan["enc"][-1]
    feat = torch.empty((B, n, 8 * F), dtype=E.act_dtype(), device=dev)
    E.conv(acts[0], 0, lp["cin_store"], lp["w"], 1, 1, lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU, out=feat, out_dtype=L.DT_BF16,
           sb=n * 8 * F, sh=1, sw=8 * F, sc=F, c_off=0, third=8 * F, Ho=F, Wo=n, w_gather=CN.nearest_index(T, n, dev))
import ctypes
def layer0_dbg(bits):
    def f():
        os.environ["SOS_CONV_DBG"] = str(bits)
        layer(0)()
    return f
jobs = {f"layer0 dbg={b}": layer0_dbg(b) for b in (0, 1, 8, 4, 2, 1 | 8, 1 | 8 | 4, 16)}
for f in jobs.values():
    f()
torch.cuda.synchronize()
for name, fn in jobs.items():
    bad = [0, 0]
    bar = threading.Barrier(2)
    stop = [False]
    def t0():
        st = torch.cuda.Stream(); bar.wait()
        with torch.cuda.stream(st):
            outs = [transform.stft_batch(x0) for _ in range(NIT)]
        st.synchronize(); stop[0] = True
        bad[0] = sum(0 if torch.equal(o, ref) else 1 for o in outs)
    def t1():
        st = torch.cuda.Stream(); bar.wait()
        with torch.cuda.stream(st):
            while not stop[0]:
                for _ in range(10):
                    fn()
                st.synchronize()
    ths = [threading.Thread(target=t0), threading.Thread(target=t1)]
    [t.start() for t in ths]; [t.join() for t in ths]
    print(f"{name:24s} corrupted STFT outputs: {bad[0]} / {NIT}", flush=True)
