import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
wl = bench.Workload("train", "fp16", 64, 0)
for _ in range(5): wl.step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); wl.step(); ts.append(time.perf_counter() - t0)
t1 = time.perf_counter(); torch.cuda.synchronize(); tail = time.perf_counter() - t1
print("host enqueue ms per step:", [round(1e3 * t, 1) for t in ts], "drain after the last enqueue ms:", round(1e3 * tail, 1))
