import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import pipeline, transform, tools
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
sos_amd.set_precision("fp16")
torch.manual_seed(0)
det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
base = torch.from_numpy(synth_batch(500, 6)["mixed"]).cuda()
inputs = [base[:4].contiguous(), base[1:4, :14000].contiguous()]
ref = [transform.stft_batch(x) for x in inputs]
full = [pipeline.denoise(det, jm, x) for x in inputs]
torch.cuda.synchronize()

def stage(name, x, i):
    if name == "stft":
        return transform.stft_batch(x)
    if name == "full":
        pipeline.denoise(det, jm, x)
        return transform.stft_batch(x)
    if name == "det":
        S = transform.stft_batch(x)
        det(s=S, v_num_frames=pipeline.n_video_frames(x.shape[1]))
        return S
    if name == "jm":
        S = transform.stft_batch(x)
        jm(S, S)
        return S

for modes in (("stft", "det"), ("stft", "jm"), ("stft", "det"), ("stft", "jm")):
    bad = [0, 0]
    bar = threading.Barrier(2)
    def work(i):
        st = torch.cuda.Stream()
        bar.wait()
        with torch.cuda.stream(st):
            outs = [stage(modes[i], inputs[i], i) for _ in range(30)]
        st.synchronize()
        bad[i] = sum(0 if torch.equal(o, ref[i]) else 1 for o in outs)
        for j, o in enumerate(outs):
            if not torch.equal(o, ref[i]):
                idx = (o != ref[i]).nonzero()
                print("  thread", i, "iter", j, "n_bad", len(idx), "b", idx[:, 0].unique().tolist(), "c", idx[:, 1].unique().tolist(),
                      "f range", int(idx[:, 2].min()), int(idx[:, 2].max()), "t", idx[:, 3].unique().tolist()[:40],
                      "sample got/ref", o[tuple(idx[0])].item(), ref[i][tuple(idx[0])].item(), "zeros?", int((o[tuple(idx.T)] == 0).sum()))
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ths]; [t.join() for t in ths]
    print(modes, "mismatching STFT outputs per thread:", bad, flush=True)
