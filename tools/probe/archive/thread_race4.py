import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import transform, engine as E, _lib as L, common_nets as CN
from sos_amd.dataset import synth_batch
from sos_amd.detector import networks as dnet
sos_amd.set_precision("fp16")
NIT = 1000
dev = torch.device("cuda")
base = torch.from_numpy(synth_batch(500, 4)["mixed"]).cuda()
x0 = base.contiguous()
det = dnet.get_network().cuda().eval()
plan = det._cache.get(det, det._build_plan)
S = transform.stft_batch(base[1:4, :14000].contiguous())
B, _, F, T = S.shape
a0 = E.pack_input(S, False)
dst = E.Act(B, F, T, 48, False, dev)
lp = plan["enc"][0]
def aggressor():
    E.conv_to_act(a0, 0, lp["cin_store"], lp["w"], lp["kh"], lp["kw"], lp["cout"], lp["scale"], lp["shift"], L.ACT_RELU,
                  dst, cout_store=48, dil=lp["dil"], pad=lp["pad"], Ho=F, Wo=T)
aggressor(); torch.cuda.synchronize()
conv_ref = dst.t.clone()
big = torch.randn(1 << 20, device="cuda")
victims = {
    "stft": lambda: transform.stft_batch(x0),
    "clone 4MB": lambda: big.clone(),
    "sin 4MB": lambda: torch.sin(big),
    "mul-add 4MB": lambda: big * 1.5 + 0.25,
    "istft": (lambda Sx=transform.stft_batch(x0): transform.istft_batch(Sx)),
    "pack_input": lambda: E.pack_input(S, False).t,
}
for name, vf in victims.items():
    ref = vf().clone(); torch.cuda.synchronize()
    bad = [0, 0]; stop = [False]; bar = threading.Barrier(2)
    def t0():
        st = torch.cuda.Stream(); bar.wait()
        with torch.cuda.stream(st):
            outs = [vf() for _ in range(NIT)]
        st.synchronize(); stop[0] = True
        bad[0] = sum(0 if torch.equal(o, ref) else 1 for o in outs)
    def t1():
        st = torch.cuda.Stream(); bar.wait()
        nbad = 0
        with torch.cuda.stream(st):
            while not stop[0]:
                for _ in range(10):
                    aggressor()
                st.synchronize()
                nbad += 0 if torch.equal(dst.t, conv_ref) else 1
        bad[1] = nbad
    ths = [threading.Thread(target=t0), threading.Thread(target=t1)]
    [t.start() for t in ths]; [t.join() for t in ths]
    print(f"victim {name:14s} corrupted: {bad[0]} / {NIT}   aggressor's own output wrong in {bad[1]} checks", flush=True)
