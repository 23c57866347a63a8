import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sos_amd
from sos_amd import pipeline
from sos_amd.common import MyConfig
from sos_amd.dataset import synth_batch
from sos_amd.denoiser import networks as jnet
from sos_amd.detector import networks as dnet
nser = int(os.environ.get("NSER", "2"))
for prec in sys.argv[1:]:
  sos_amd.set_precision(prec)
  print("=====", prec)
  if True:
    torch.manual_seed(0)
    det, jm = dnet.get_network().cuda().eval(), jnet.get_network(MyConfig()).cuda().eval()
    base = torch.from_numpy(synth_batch(500, 6)["mixed"]).cuda()
    inputs = [base[:4].contiguous(), base[1:4, :14000].contiguous()]
    serial = [pipeline.denoise(det, jm, x, return_all=True) for x in inputs]
    serial2 = [pipeline.denoise(det, jm, x, return_all=True) for x in inputs] if nser > 1 else serial
    keys = ["S_mixed", "logits", "bits", "mask", "S_noise", "n_pred", "crm", "S_out", "out"]
    for i in range(2):
        print("serial repeat", i, {k: bool(torch.equal(serial[i][k], serial2[i][k])) for k in keys})
    torch.cuda.synchronize()
    res = [None, None]
    bar = threading.Barrier(2)
    def work(i):
        st = torch.cuda.Stream()
        bar.wait()
        with torch.cuda.stream(st):
            outs = [pipeline.denoise(det, jm, inputs[i], return_all=True) for _ in range(6)]
        st.synchronize()
        res[i] = outs
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for i in range(2):
        for j, r in enumerate(res[i]):
            d = {k: float((r[k].float() - serial[i][k].float()).abs().max()) for k in keys}
            print("thread", i, "iter", j, {k: v for k, v in d.items() if v != 0})
