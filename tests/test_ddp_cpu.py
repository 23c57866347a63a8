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
    from sos_amd.agent import GradBucketer, GradSink, broadcast_module_state
    from sos_amd.dataset import get_dataloader
    # rank 0's initial weights and buffers win (DataParallel replicates module 0)
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    net[1].running_mean.fill_(float(rank + 1))
    broadcast_module_state(net)
    torch.manual_seed(100)
    want_net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    same = all(torch.equal(a, b) for a, b in zip(net.parameters(), want_net.parameters())) and \
        torch.equal(net[1].running_mean, torch.ones(3))
    # the loader shards by torch.distributed's rank / world size by default: disjoint clips, same number of steps
    ld = get_dataloader("training", batch_size=4, n_batches=3, model="detector")
    starts = ld.starts()
    torch.manual_seed(0)
    params = [("a.weight", torch.nn.Parameter(torch.zeros(7, 5))), ("a.bias", torch.nn.Parameter(torch.zeros(7))),
              ("b.weight", torch.nn.Parameter(torch.zeros(300, 40))), ("c.slope", torch.nn.Parameter(torch.zeros(1)))]
    os.environ["SOS_DDP_PROFILE"] = "1"                   # the instrumentation bench.py --gpus N reports (comm_stats)
    bk = GradBucketer(params, bucket_bytes=4096)          # small buckets -> several all-reduces
    sink = GradSink(bk)
    # backward produces grads in reverse order; values depend on the rank
    for name, p in reversed(params):
        sink[name] = torch.full(p.shape, float(rank + 1)) * (1 + len(name))
    bk.finalize()
    ok = same
    for name, p in params:
        want = (1 + 2) * (1 + len(name)) * torch.ones(p.shape)          # SUM over ranks 1 and 2
        ok = ok and p.grad is not None and torch.equal(p.grad, want) and p.grad.shape == p.shape
        # the optimizer's grad_scale = 1/world turns the sum into the data-parallel AVERAGE (the kernel multiplies it in)
        ok = ok and torch.allclose(p.grad * (1.0 / bk.world), 1.5 * (1 + len(name)) * torch.ones(p.shape))
    st = bk.comm_stats()
    # three buckets (1 | 12000 | 7 + 35 floats), every gradient byte once, the waits timed on the host
    ok = ok and st["world"] == 2 and st["steps"] == 1 and st["collective"] and st["buckets_per_step"] == 3
    ok = ok and st["bytes_per_step"] == 4 * (1 + 12000 + 7 + 35) and st["comm_ms_per_step"] >= 0.0 and st["mode"] in ("inline", "own", "shared")
    bk.comm_reset()
    ok = ok and bk.comm_stats()["buckets_per_step"] == 0
    q.put((rank, ok, starts))
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
    res = sorted(res)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)]
    s0, s1 = res[0][2], res[1][2]
    clips = lambda starts: {c for st in starts for c in range(st, st + 4)}       # noqa: E731
    assert len(s0) == len(s1) == 3 and not (clips(s0) & clips(s1)) and clips(s0) | clips(s1) == set(range(24))


def test_shard_indices_cover_every_item_once_per_epoch():
    from sos_amd.dataset import shard_indices
    for n, world in ((10, 1), (10, 3), (7, 8), (64, 8), (0, 2)):
        shards = [shard_indices(n, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1                      # same number of steps on every rank
        seen = [i for s in shards for i in s]
        assert set(seen) == set(range(n))                              # nothing dropped (the tail wraps around)
        assert len(seen) - n < world or n == 0
