"""Trainer mirroring the reference `agent.py` (M1/agent.py:25-206, M2/agent.py:20-190) on the HIP
kernels, one process per GPU.

Kept from the reference: the loss definitions (BCEWithLogits for the detector, MSE(n_pred,
full_noise) + MSE(rec, clean) summed for the denoiser), Adam(lr) + StepLR(step_size), the
`{clock, model_state_dict, optimizer_state_dict, scheduler_state_dict}` checkpoint dict with
un-prefixed state-dict keys, TrainClock's `{epoch, minibatch, step}`.

Replaced: nn.DataParallel (single process, replicate/scatter/gather every step, gradients reduced
to GPU 0) by data parallelism across processes -- each rank owns its batch shard and its own
BatchNorm statistics (DataParallel semantics, no SyncBN) and the gradients are averaged with
bucketed RCCL all-reduces launched while the hand-written backward pass is still running.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import transform
from .engine import bump_version

PHASE_TRAINING, PHASE_TESTING = "training", "testing"


class TrainClock(object):
    """M1/utils.py:8-34."""

    def __init__(self):
        self.epoch, self.minibatch, self.step = 1, 0, 0

    def tick(self):
        self.minibatch += 1
        self.step += 1

    def tock(self):
        self.epoch += 1
        self.minibatch = 0

    def make_checkpoint(self):
        return {"epoch": self.epoch, "minibatch": self.minibatch, "step": self.step}

    def restore_checkpoint(self, clock_dict):
        self.epoch, self.minibatch, self.step = clock_dict["epoch"], clock_dict["minibatch"], clock_dict["step"]


# ------------------------------------------------------------------------------------ fused losses
def _scale_by(grad, g):
    """grad * g for the 0-dim upstream gradient `g` of a scalar loss (sos_scale_f32 on a copy: a second backward through
    a retained graph must see the unscaled gradient again)."""
    out = grad.clone()
    g = g.detach().reshape(1).to(device=grad.device, dtype=torch.float32)
    L.check(L.lib().sos_scale_f32(L.ptr(out), out.numel(), L.ptr(g), L.stream_ptr()), "sos_scale_f32")
    return out


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L.require_cuda(a, b)
        a, b = a.contiguous().float(), b.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a)
        partial = torch.empty(1024, dtype=torch.float32, device=a.device)
        L.check(L.lib().sos_mse_loss(L.ptr(a), L.ptr(b), a.numel(), 1.0, L.ptr(loss), L.ptr(grad), L.ptr(partial),
                                     L.stream_ptr()), "sos_mse_loss")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scale_by(grad, g), None


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        L.require_cuda(x, y)
        x, y = x.contiguous().float(), y.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        partial = torch.empty(1024, dtype=torch.float32, device=x.device)
        L.check(L.lib().sos_bce_logits_loss(L.ptr(x), L.ptr(y), x.numel(), 1.0, L.ptr(loss), L.ptr(grad), L.ptr(partial),
                                            L.stream_ptr()), "sos_bce_logits_loss")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _scale_by(grad, g), None


def mse_loss(a, b):
    """nn.MSELoss() (mean), M2/agent.py:174."""
    return _MSE.apply(a, b)


def bce_with_logits_loss(x, y):
    """nn.BCEWithLogitsLoss() (mean), M1/agent.py:187."""
    return _BCE.apply(x, y)


# ------------------------------------------------------------------------------------- optimizer
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad=False) semantics and state_dict layout.  One step = the gradients gathered into one flat
    buffer (a batched copy) + ONE sos_adam_multi_step launch per parameter group over a device table of (param, grad,
    exp_avg, exp_avg_sq) pointers built once; `grad_scale` folds the 1/world_size of the data-parallel average in."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self.guard = None          # engine.guard_state(net): device-side overflow guard, the step is skipped when a gradient is Inf / NaN
        self._tables = {}
        # Skipped steps are counted ONCE for the whole model (guard[3]) while `step` is per parameter.  A parameter whose state is
        # created in a later step() call than the first (it joined the group late, or had no gradient before) must not be
        # charged the steps skipped before it existed: its baseline -- guard[3] at the moment its state is created, a
        # device scalar, no host sync -- is kept here and subtracted in its (per-tensor) launches and in save_ckpt.  Parameters
        # born in the first call have baseline 0 by construction: the guard is created (or restored from the same checkpoint)
        # together with the optimizer state (BaseAgent.__init__ / load_ckpt).
        self._calls = 0
        self._skip_base = {}       # late-born parameter -> device f32 [1]

    def _table(self, gi, ps):
        key = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        ent = self._tables.get(gi)
        if ent is None or ent["key"] != key:
            dev = ps[0].device
            sizes = [p.numel() for p in ps]
            flat_g = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            rows, chunks, off = [], [], 0
            for ti, p in enumerate(ps):
                st = self.state[p]
                rows.append([p.data_ptr(), flat_g.data_ptr() + 4 * off, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), sizes[ti]])
                chunks += [[ti, c] for c in range((sizes[ti] + L.ADAM_CHUNK - 1) // L.ADAM_CHUNK)]
                off += sizes[ti]
            ent = self._tables[gi] = dict(key=key, flat_g=flat_g, n=len(ps), nchunks=len(chunks),
                                          views=list(torch.split(flat_g, sizes)),
                                          tab=torch.tensor(rows, dtype=torch.int64, device=dev),
                                          chunks=torch.tensor(chunks, dtype=torch.int32, device=dev))
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        work = []
        self._calls += 1
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            L.require_cuda(*ps)
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    if self.guard is not None and self._calls > 1:
                        self._skip_base[p] = self.guard[3:4].clone()
                st["step"] += 1
            ent = self._table(gi, ps)
            torch._foreach_copy_(ent["views"], [p.grad.reshape(-1) for p in ps])        # batched gather of the gradients
            work.append((group, ps, ent))
        skip = None
        if self.guard is not None and work:
            # one pass over the flat gradients (65 MB for the denoiser: ~15 us): any Inf / NaN -> found = 1, the Adam
            # kernels return at once, exp_avg / exp_avg_sq / the weights stay as they are, and the loss-scale target backs
            # off.  Decided on the device: no host synchronisation.  The host-side step counters count ATTEMPTS; the guard's
            # [3] counts the skipped ones on the device and the Adam kernels take their bias corrections at attempts - skipped,
            # i.e. at the number of updates actually applied (GradScaler semantics).  save_ckpt folds the two into the applied
            # count and stores the guard's back-off state next to the optimizer's.
            for k, (_, _, ent) in enumerate(work):
                L.check(L.lib().sos_grad_guard(L.ptr(ent["flat_g"]), ent["flat_g"].numel(), L.ptr(self.guard),
                                               1 if k == len(work) - 1 else 0, L.stream_ptr()), "sos_grad_guard")
            skip = self.guard
        for group, ps, ent in work:
            b1, b2 = group["betas"]
            steps = {int(self.state[p]["step"]) for p in ps}
            if len(steps) == 1 and not any(p in self._skip_base for p in ps):
                L.check(L.lib().sos_adam_multi_step(L.ptr(ent["tab"]), ent["n"], L.ptr(ent["chunks"]), ent["nchunks"],
                                                    float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                    float(group["weight_decay"]), steps.pop(), float(self.grad_scale),
                                                    L.ptr(skip), L.stream_ptr()), "sos_adam_multi_step")
            else:       # parameters that joined the group at different times: per-tensor launches
                for p, g in zip(ps, ent["views"]):
                    st = self.state[p]
                    sk = skip
                    if skip is not None and p in self._skip_base:
                        sk = skip.clone()                       # this parameter's view of the guard: steps skipped since it was born
                        sk[3:4].sub_(self._skip_base[p])
                    L.check(L.lib().sos_adam_step(L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(),
                                                  float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                  float(group["weight_decay"]), int(st["step"]), float(self.grad_scale),
                                                  L.ptr(sk), L.stream_ptr()), "sos_adam_step")
            for p in ps:
                bump_version(p)     # raw-pointer write: invalidate the packed-weight caches keyed on _version
        return None


# ----------------------------------------------------------------------------- gradient all-reduce
class GradBucketer:
    """Flat ~`bucket_bytes` gradient buckets in the order the backward pass produces them; a bucket's
    all-reduce (RCCL over xGMI with backend 'nccl'; gloo in CPU tests) starts as soon as its last
    gradient has been written, overlapping the rest of the backward.  xGMI is point-to-point, so a
    ring all-reduce is bound by one link: a few large buckets (25 MB) beat many small ones."""

    def __init__(self, named_params, bucket_bytes=25 * 1024 * 1024, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # SOS_FORCE_BUCKETS=1: issue the collectives even in a world of one (exercises the RCCL path on a 1-GPU box)
        self.collective = dist.is_initialized() and (self.world > 1 or os.environ.get("SOS_FORCE_BUCKETS") == "1")
        self.shapes = {n: p.shape for n, p in named_params}
        self.params = dict(named_params)
        self.bucket_bytes = bucket_bytes
        self._comm = None
        self._flush_stream = None
        self._inline_streams = set()
        # SOS_DDP_PROFILE=1 (bench.py --gpus N / --force-buckets): every bucket's collective is bracketed with HIP events on the
        # stream it is enqueued on (host timers around the waits for CPU tensors) -- comm_stats() reports buckets, bytes and
        # communication milliseconds per step, so that a scaling run can explain itself
        self.profile = os.environ.get("SOS_DDP_PROFILE") == "1"
        self._prof, self._prof_steps, self._prof_host_s = [], 0, 0.0
        self.reset()

    def reset(self):
        self.buckets, self.cur, self.cur_fill, self.handles, self.views = [], None, 0, [], {}
        self.pending, self.pending_streams = ([], []), []

    # Where the bucket copies and the collectives are enqueued (SOS_DDP_COMM; GPU gradients only):
    #   "inline" (default)  no communication stream: a bucket is closed whenever the producing stream changes, and its copy and its
    #                       collective (async_op=False: ProcessGroupNCCL then enqueues on the caller's stream, no internal stream,
    #                       no extra events) go onto the stream that produced it.  No cross-stream wait anywhere; the collective
    #                       sits in that stream's order and overlaps the OTHER streams' compute (the step runs three).
    #   "shared"            ONE communication stream for every model's buckets, synchronous collectives on it
    #   "own"               a communication stream per bucketer + async collectives on ProcessGroupNCCL's internal stream
    # Why: HIP multiplexes streams onto 4 hardware queues and the host enqueues ~100 ms ahead of the GPU, so a stream that waits for
    # "bucket k of the backward pass is full" blocks the hardware queue it shares with whatever the host enqueues later.  With a
    # communicator stream per model (rounds 3-4; fine with round 4's stream creation order: 0.998) round 5's first refresh had the
    # detector's WHOLE step start only when the denoiser's had ended (tools/probe/stream_timeline.py: its stream shared a queue with
    # a waiting communicator stream): forced-bucket path 0.944-0.952 of the plain step; GPU_MAX_HW_QUEUES=8 / 16 made it worse
    # (0.82 / 0.80: blocked queues still occupy the command processor).  Same box, `bench.py --force-buckets`, plain step 542.4
    # utt/s: "own" 512.8, "shared" 514.9, "inline" 541.4 (0.998); "inline" on ONE process group for both models
    # (SOS_SHARED_GROUP=1) 515.5 -- one communicator serialises the two models' collectives.
    COMM_MODE = os.environ.get("SOS_DDP_COMM", "inline")
    _shared_comm = {}

    def _comm_stream(self):
        if self.COMM_MODE == "own":
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            return self._comm
        dev = torch.cuda.current_device()
        st = GradBucketer._shared_comm.get(dev)
        if st is None:
            st = GradBucketer._shared_comm[dev] = torch.cuda.Stream()
        self._comm = st
        return st

    def _flush_pending(self):
        """ONE batched copy of the pending gradients into their bucket views.  GPU: on the communication stream, behind an
        event of every producing stream ("inline": on the producing stream itself); the gradients' and the bucket's memory is
        kept from the caching allocator until that stream's copy (and collective) has run.  CPU (gloo tests): in place."""
        if not self.pending[0]:
            return
        streams = {s_ for s_ in self.pending_streams if s_ is not None}
        if streams and self.COMM_MODE == "inline":
            (st,) = streams                    # ready() closes a bucket when the producing stream changes
            with torch.cuda.stream(st):
                torch._foreach_copy_(self.pending[0], self.pending[1])
        elif streams:
            comm = self._comm_stream()
            for st in streams:
                comm.wait_event(st.record_event())
            for g in self.pending[1]:
                g.record_stream(comm)
            with torch.cuda.stream(comm):
                torch._foreach_copy_(self.pending[0], self.pending[1])
        else:
            torch._foreach_copy_(self.pending[0], self.pending[1])
        self._flush_stream = next(iter(streams)) if streams else None
        self.pending, self.pending_streams = ([], []), []

    def _launch(self, flat, fill):
        # the bucket's gradients are gathered with ONE batched copy (a copy per tensor was 322 launches per step), then
        # its all-reduce starts while the backward pass goes on
        self._flush_pending()
        if not flat.is_cuda:
            if self.collective:
                self.handles.append(dist.all_reduce(flat[:fill], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if self.profile:
                    self._prof.append((None, None, fill * 4))
            return
        inline = self.COMM_MODE == "inline"
        st = self._flush_stream if inline else self._comm_stream()
        if not inline:
            flat.record_stream(st)
        elif self._flush_stream is not None:
            self._inline_streams.add(self._flush_stream)
        if self.collective:
            with torch.cuda.stream(st):
                e0 = e1 = None
                if self.profile:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                if self.COMM_MODE == "own":
                    # (async: the collective runs on ProcessGroupNCCL's internal stream; the bracket on `st` then spans from the
                    # enqueue to the point where `st` has waited for it)
                    h = dist.all_reduce(flat[:fill], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    self.handles.append(h)
                    if self.profile:
                        h.wait()
                else:       # synchronous: enqueued on `st` itself
                    dist.all_reduce(flat[:fill], op=dist.ReduceOp.SUM, group=self.group, async_op=False)
                if self.profile:
                    e1.record(st)
                    self._prof.append((e0, e1, fill * 4))

    def ready(self, name, g):
        """Called by the backward pass when a parameter gradient is final; returns the bucket view that
        will hold it (filled when the bucket is flushed)."""
        n = g.numel()
        cap = max(self.bucket_bytes // 4, n)
        here = torch.cuda.current_stream() if g.is_cuda else None
        if (self.COMM_MODE == "inline" and self.cur is not None and self.pending_streams and here is not None
                and self.pending_streams[-1] != here):
            self._launch(self.cur, self.cur_fill)         # the producing stream changed: this bucket ends here
            self.cur = None
        if self.cur is None or self.cur_fill + n > self.cur.numel():
            if self.cur is not None:
                self._launch(self.cur, self.cur_fill)
            self.cur = torch.empty(cap, dtype=torch.float32, device=g.device)
            self.cur_fill = 0
            self.buckets.append(self.cur)
        v = self.cur[self.cur_fill:self.cur_fill + n]
        self.pending[0].append(v)
        self.pending[1].append(g.reshape(-1))
        self.pending_streams.append(here)
        self.cur_fill += n
        self.views[name] = v.view(self.shapes[name])
        return self.views[name]

    def finalize(self):
        """Flush the last bucket, wait for every all-reduce and point p.grad at the reduced views (the
        1/world average is applied by the optimizer's grad_scale)."""
        if self.cur is not None and self.cur_fill:
            self._launch(self.cur, self.cur_fill)
        else:
            self._flush_pending()
        if self.profile and self.handles and not any(e is not None for e, _, _ in self._prof[-1:]):
            import time as _time
            t0 = _time.perf_counter()
            for h in self.handles:
                h.wait()
            self._prof_host_s += _time.perf_counter() - t0
        for h in self.handles:
            h.wait()
        if self.profile:
            self._prof_steps += 1
        if self._comm is not None:          # the consumer (the optimizer on the caller's stream) follows the copies and collectives
            torch.cuda.current_stream().wait_stream(self._comm)
        for st in self._inline_streams:     # "inline": they ran on the producing streams
            if st != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_stream(st)
        self._inline_streams = set()
        for name, v in self.views.items():
            self.params[name].grad = v
        self.reset_keep_views()

    def comm_reset(self):
        """Forget the communication brackets taken so far (bench.py: after the warm-up steps)."""
        self._prof, self._prof_steps, self._prof_host_s = [], 0, 0.0

    def comm_stats(self):
        """What the data-parallel path did per step since comm_reset() (SOS_DDP_PROFILE=1): buckets, bytes and the summed duration of
        the buckets' collectives -- HIP events on the stream each was enqueued on (synchronises the device: call it outside a timed
        region), or the host's time in the waits for CPU tensors (gloo)."""
        steps = max(1, self._prof_steps)
        ms = 1e3 * self._prof_host_s
        if any(e0 is not None for e0, _, _ in self._prof):
            torch.cuda.synchronize()
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self._prof if e0 is not None)
        return {"mode": self.COMM_MODE, "world": self.world, "steps": self._prof_steps, "collective": bool(self.collective),
                "buckets_per_step": len(self._prof) / steps, "bytes_per_step": sum(b for _, _, b in self._prof) / steps,
                "comm_ms_per_step": ms / steps, "bucket_bytes_cap": self.bucket_bytes}

    def reset_keep_views(self):
        self.cur, self.cur_fill, self.handles, self.buckets, self.views = None, 0, [], [], {}
        self.pending, self.pending_streams = ([], []), []


class GradSink(dict):
    """dict the backward pass fills; every insertion is routed through the bucketer."""

    def __init__(self, bucketer):
        super().__init__()
        self.bucketer = bucketer

    def __setitem__(self, k, v):
        super().__setitem__(k, self.bucketer.ready(k, v) if self.bucketer is not None else v)


def broadcast_module_state(net, src=0, group=None):
    """Same initial weights and buffers on every rank: rank `src`'s (nn.DataParallel replicates module 0's parameters
    and buffers onto the other GPUs every step, M1/agent.py:157-159; with one process per GPU once at start is enough,
    the averaged gradients keep the replicas identical and BatchNorm statistics stay per rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(net.parameters()) + list(net.buffers()):
        dist.broadcast(t.data, src=src, group=group)


# ------------------------------------------------------------------------------------------ agents
class BaseAgent(object):
    """M1/agent.py:25-150 without tensorboard / path plumbing."""

    def __init__(self, net, lr=1e-3, lr_step_size=15, model_dir=None, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.net = net.to(self.device)
        self.clock = TrainClock()
        self.model_dir = model_dir
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        force = dist.is_initialized() and os.environ.get("SOS_FORCE_BUCKETS") == "1"
        # One process group -- i.e. one RCCL communicator with its own stream -- PER MODEL: with the default group both models'
        # buckets queue on one communicator stream and the detector's 9 MB all-reduce waits behind the denoiser's 65 MB
        # (train_concurrent runs the two backward passes side by side).  Every rank constructs its agents in the same order,
        # so the new_group calls match up.  SOS_SHARED_GROUP=1: the default group for every model (A/B).
        self.group = None
        if dist.is_initialized() and (self.world > 1 or force) and os.environ.get("SOS_SHARED_GROUP") != "1":
            self.group = dist.new_group()
        broadcast_module_state(self.net, group=self.group)
        self.optimizer = FusedAdam(self.net.parameters(), lr)
        self.optimizer.grad_scale = 1.0 / self.world
        from .engine import guard_state
        self.optimizer.guard = guard_state(self.net)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, lr_step_size)
        self.bucketer = GradBucketer(list(self.net.named_parameters()), group=self.group) if (self.world > 1 or force) else None
        self.net.grad_sink_factory = (lambda: GradSink(self.bucketer)) if self.bucketer is not None else None

    # -- checkpoints: M1/agent.py:62-100
    def save_ckpt(self, name=None):
        path = os.path.join(self.model_dir, f"ckpt_epoch{self.clock.epoch}.pth" if name is None else f"{name}.pth")
        # the reference's four keys (M1/agent.py:62-78) + the overflow guard of the fp16 mode.  The optimizer's `step` entries
        # are written as APPLIED updates (attempts minus the steps the device-side guard skipped) and the guard's skipped-step
        # count as zero, so the pair stays consistent for this trainer and `step` means what it means to torch.optim.Adam.
        from .engine import guard_state
        guard = guard_state(self.net).detach().cpu().clone()          # (a host sync: checkpoints are outside the step loop)
        skipped = float(guard[3])
        osd = self.optimizer.state_dict()
        if skipped:
            # per state entry: the steps skipped since THAT parameter's state was created (FusedAdam._skip_base)
            order = [p for g_ in self.optimizer.param_groups for p in g_["params"]]
            base = {i: float(self.optimizer._skip_base[p]) for i, p in enumerate(order) if p in self.optimizer._skip_base}
            osd = dict(osd, state={k: (dict(v, step=v["step"] - (skipped - base.get(k, 0.0))) if "step" in v else v)
                                   for k, v in osd["state"].items()})
        guard[3] = 0.0
        guard[0] = 0.0
        torch.save({"clock": self.clock.make_checkpoint(),
                    "model_state_dict": {k: v.detach().cpu() for k, v in self.net.state_dict().items()},
                    "optimizer_state_dict": osd,
                    "scheduler_state_dict": self.scheduler.state_dict(),
                    "overflow_guard": guard}, path)
        return path

    def load_ckpt(self, name=None):
        name = name if name == "latest" else f"ckpt_epoch{name}"
        path = os.path.join(self.model_dir, f"{name}.pth")
        if not os.path.exists(path):
            raise ValueError("Checkpoint {} not exists.".format(path))
        ck = torch.load(path, map_location=self.device)
        self.net.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.optimizer._skip_base.clear()            # the checkpoint holds applied steps and a zero skipped count: every baseline is 0
        self.optimizer._calls = 1 if self.optimizer.state else 0
        for st in self.optimizer.state.values():     # map_location moved the step counters to the GPU: int(step) would
            if torch.is_tensor(st.get("step")):      # then synchronise once per parameter and step
                st["step"] = st["step"].detach().cpu()
        self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.clock.restore_checkpoint(ck["clock"])
        from .engine import guard_state
        g = guard_state(self.net)
        g.zero_()                                    # a reference checkpoint has no guard entry: full loss scale, nothing skipped
        if ck.get("overflow_guard") is not None:     # resumed fp16 run: the loss-scale back-off it had reached
            g.copy_(ck["overflow_guard"].to(g.device, torch.float32))

    # -- M1/agent.py:101-130
    def update_network(self, loss_dict):
        loss = sum(loss_dict.values())
        self.optimizer.zero_grad(set_to_none=True)
        if self.bucketer is not None:
            self.bucketer.reset()
        gate = getattr(self, "backward_gate", None)
        if gate is not None:           # train_concurrent: this model's backward waits for the big model's heavy phase to end
            torch.cuda.current_stream().wait_event(gate)
        loss.backward()
        if self.bucketer is not None:
            self.bucketer.finalize()
        self.optimizer.step()

    def update_learning_rate(self):
        self.scheduler.step()

    def train_func(self, data):
        self.net.train()
        outputs, losses = self.forward(data)
        self.update_network(losses)
        return outputs, losses

    def val_func(self, data):
        self.net.eval()
        with torch.no_grad():
            return self.forward(data)


_job_streams = {}


def _job_stream(device, k, prio=0):
    """The HIP stream of job k of train_concurrent on `device`: ONE per (device, k) for the life of the process, shared by every
    agent that is ever job k.  HIP multiplexes streams onto 4 hardware queues; every further stream that sits in a long wait
    (the second model waits tens of milliseconds for the first one's gates) is one more way for two streams to collide on a queue
    -- two sets of agents alive in one process (bench.py's alternating A/B) measured 2.8 % slower than either alone."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), k, prio)
    st = _job_streams.get(key)
    if st is None:
        st = _job_streams[key] = torch.cuda.Stream(device=device, priority=prio)
    return st


def train_concurrent(jobs):
    """One training step of several independent agents (the reference trains its two models in two separate
    processes), each on its own HIP stream: [(agent, batch), ...] -> [(outputs, losses), ...].
    Put the big model first.  If its network announces the end of its heavy backward phase (`after_stage2_backward` of
    denoiser.networks.JointModel), the other agents' backward + optimizer wait for that point: they then run under the big
    model's U-Net backward and optimizer -- many medium-sized, partly HBM-bound kernels that leave room -- instead of
    time-sharing the chip with its chip-filling 96-channel convolutions (which made every such kernel ~8 % longer and
    gave nothing back: 485 vs 477 utt/s fully overlapped vs back to back).  If it also announces the
    start of its BiLSTM (`before_lstm_forward`), the other agents' FORWARD starts there: the recurrence (8 workgroups
    stepping through T frames), the FC head, the losses and their backward up to the encoders' keep the chip mostly idle
    for ~4 ms per step (round 3; SOS_STREAM_OVERLAP=gated is the round-2 schedule: whole step behind the backward gate).
    SOS_STREAM_OVERLAP=full
    restores the unconstrained overlap."""
    cur = torch.cuda.current_stream()
    outs = []
    gate = gate_f = None
    mode = os.environ.get("SOS_STREAM_OVERLAP", "split")
    for k, (ag, data) in enumerate(jobs):
        prio = -1 if (k == 0 and os.environ.get("SOS_STREAM_PRIO") == "1") else 0     # A/B: big model on a high-priority stream
        ag.stream = _job_stream(ag.device, k, prio)
        ag.stream.wait_stream(cur)
        net = getattr(ag, "net", None)
        ag.backward_gate = None
        if k == 0 and len(jobs) > 1 and getattr(net, "ANNOUNCES_STAGE2_BACKWARD", False) and mode != "full":
            gate = torch.cuda.Event()
            net.after_stage2_backward = gate.record          # runs inside backward: records on this agent's stream
            if mode == "split" and getattr(net, "ANNOUNCES_LSTM_FORWARD", False):
                gate_f = torch.cuda.Event()
                net.before_lstm_forward = gate_f.record      # runs inside forward, once the encoders are enqueued
        elif gate_f is not None:
            ag.stream.wait_event(gate_f)                     # forward: under the big model's BiLSTM / FC head window
            ag.backward_gate = gate                          # backward + optimizer: under its U-Net backward
        elif gate is not None:
            ag.stream.wait_event(gate)
        try:
            with torch.cuda.stream(ag.stream):
                outs.append(ag.train_func(data))
        finally:
            ag.backward_gate = None
            if k == 0 and gate is not None:
                net.after_stage2_backward = None
                net.before_lstm_forward = None
    for ag, _ in jobs:
        cur.wait_stream(ag.stream)
    return outs


class DetectorAgent(BaseAgent):
    """MyAgent of M1/agent.py:153-206."""

    def forward(self, data):
        label = data["label"].to(self.device)
        audio = data["audio"].to(self.device)
        if "frames" in data:            # audio-visual variant: batch dict carries the video frames (B,3,Tv,H,W)
            output = self.net(audio, v=data["frames"].to(self.device))
        else:
            output = self.net(audio) if label.shape[1] == 60 else self.net(audio, label.shape[1])
        return output, {"bce": bce_with_logits_loss(output, label)}


class DenoiserAgent(BaseAgent):
    """MyAgent of M2/agent.py:148-190."""

    def forward(self, data):
        mixed, noise = data["mixed"].to(self.device), data["noise"].to(self.device)
        clean, full_noise = data["clean"].to(self.device), data["full_noise"].to(self.device)
        pred_noise, outputs = self.net(mixed, noise)
        rec = transform.batch_fast_icRM_sigmoid(mixed, outputs)
        return (pred_noise, outputs), {"stage1": mse_loss(pred_noise, full_noise), "stage2": mse_loss(rec, clean)}
