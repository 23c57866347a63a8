import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, torch.distributed as dist
force = len(sys.argv) > 1 and sys.argv[1] == "buckets"
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577"); os.environ["SOS_FORCE_BUCKETS"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from sos_amd import agent
wl = bench.Workload("train", "fp16", 64, 0)
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(key, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for ag, nm in zip(wl.agents, ("det", "jm")):
    wrap(ag, "forward", nm + ".forward")
    wrap(ag.optimizer, "step", nm + ".adam")
    if ag.bucketer is not None:
        wrap(ag.bucketer, "finalize", nm + ".finalize")
    orig = ag.update_network
    def upd(loss_dict, ag=ag, nm=nm, orig=orig):
        t0 = time.perf_counter(); orig(loss_dict); T.setdefault(nm + ".update_network", []).append(time.perf_counter() - t0)
    ag.update_network = upd
for _ in range(4): wl.step()
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(8): wl.step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("mode", "buckets" if force else "plain", "host enqueue per step %.1f ms, wall per step %.1f ms" % (1e3 * t_enq / 8, 1e3 * t_all / 8))
for k, v in sorted(T.items()):
    print("   %-22s %.2f ms per call" % (k, 1e3 * sum(v) / len(v)))
