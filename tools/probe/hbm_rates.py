import torch, time
n = 64 * 256 * 178 * 96
a = torch.randn(n, device="cuda", dtype=torch.bfloat16); b = torch.randn_like(a); c = torch.empty_like(a)
def t(fn, bytes_, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{name:28s} {ms*1e3:8.1f} us  {bytes_/ms/1e9:6.2f} TB/s")
B = n * 2
t(lambda: c.copy_(a), 2 * B, "copy (1R+1W)")
t(lambda: torch.add(a, b, out=c), 3 * B, "add (2R+1W)")
t(lambda: a.sum(), B, "sum (1R)")
t(lambda: c.zero_(), B, "fill (1W)")
t(lambda: torch.mul(a, b).sum(), 2*B + 0, "mul+sum (2R+W+R)")
