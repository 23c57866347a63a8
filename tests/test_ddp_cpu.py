"""world_size-2 gloo test of the data-parallel gradient path (GradBucketer / GradSink) on CPU tensors:
the buckets are filled in backward order, all-reduced asynchronously and averaged via grad_scale."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sos_amd.agent import GradBucketer, GradSink
    torch.manual_seed(0)
    params = [("a.weight", torch.nn.Parameter(torch.zeros(7, 5))), ("a.bias", torch.nn.Parameter(torch.zeros(7))),
              ("b.weight", torch.nn.Parameter(torch.zeros(300, 40))), ("c.slope", torch.nn.Parameter(torch.zeros(1)))]
    bk = GradBucketer(params, bucket_bytes=4096)          # small buckets -> several all-reduces
    sink = GradSink(bk)
    # backward produces grads in reverse order; values depend on the rank
    for name, p in reversed(params):
        sink[name] = torch.full(p.shape, float(rank + 1)) * (1 + len(name))
    bk.finalize()
    ok = True
    for name, p in params:
        want = (1 + 2) * (1 + len(name)) * torch.ones(p.shape)          # SUM over ranks 1 and 2
        ok = ok and p.grad is not None and torch.equal(p.grad, want) and p.grad.shape == p.shape
    q.put((rank, ok))
    dist.destroy_process_group()


def test_grad_bucketer_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
