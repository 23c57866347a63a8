#!/usr/bin/env python3
"""bench.py -- throughput of the Listening-to-Sound-of-Silence hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it with
torch.distributed.run, one rank per GPU (RCCL); started WITHOUT a launcher (`WORLD_SIZE` unset) and N > 1 it launches the
N ranks itself (re-exec under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1`) and fails
loudly when fewer than N devices are visible.  `n_gpus` in the line is the size of the communicator, verified with an
all-reduce of ones.  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic 2 s clips resident in HBM:
  --mode train (default, BASELINE.json configs[1]): detector forward/backward/Adam (BCE) AND denoiser
               forward/backward/Adam (MSE + MSE through the mask apply) on the same B clips, 16-bit storage + f32
               accumulation, the two (independent) models on one HIP stream each;
               for N > 1 the gradients are averaged with bucketed RCCL all-reduces overlapped with backward
  --mode train-av (BASELINE.json configs[4]: batch = 128 over 4 GPUs, i.e. --gpus 4 at the default 32 clips per GPU): one training step
               of the AUDIO-VISUAL detector (audio branch + the video branch over 60 frames of 224 x 224 + fusion into the BiLSTM),
               forward / backward / Adam; the same bucketed RCCL all-reduce of the gradients for N > 1
  --mode infer: STFT -> detector -> bits->mask -> STFT(noise) -> JointModel -> mask apply -> ISTFT
  --mode infer-ragged (BASELINE.json configs[3]): the same chain over --batch (default 256) clips of DIFFERENT lengths,
               U(1 s, 10 s) with seed 99, per-clip geometry inside the kernels (pipeline.denoise_ragged)
Each rank processes its own batch (independent utterances): weak scaling.

The default run (N = 1, train) also times, AFTER the headline loop and outside its timed region, the other lines of the
round on the same box and attaches them as `secondary` (inference, ragged inference, training in bf16 -- the literal
BASELINE configs[1] dtype -- and in 'mixed'); `value` / `config` / `dtype` describe the headline loop only.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SAMPLES = 28000             # 2 s @ 14 kHz: the reference-true geometry
GFLOP_PER_UTT_INFER = 446.5
GFLOP_PER_CLIP_AV_FWD = 49.02 + 1471.0   # the detector's audio branch + the video branch at 60 x 224 x 224
DEFAULT_PRECISION = {"train": "fp16", "infer": "mixed", "infer-ragged": "mixed", "train-av": "fp16"}
# measured against the reference goldens (tests/test_gpu_nets.py, tests/test_gpu_train_nets.py); north_star bar: 1e-3
PARITY_NOTE = {
    # MEASURED values (max-abs error / max-abs reference per tensor, against the goldens generated from the imported reference);
    # "asserted" = the bound the GPU tests check
    "fp16": "measured vs reference goldens: eval n_pred/mask 6-7e-4 (asserted 1e-3), logits 1.5-1.8e-3 (asserted 3e-3), end-to-end "
            "waveform 1.2-4.8e-3 (asserted 5.5e-3; 9e-3 ragged; the storage model of the format alone gives 4.2e-3 on the -40 dB clip), "
            "SI-SDR delta <= 0.016 dB (asserted 0.05 dB); training: train-mode logits / n_pred / mask within 2x the storage-model "
            "deviation, summed loss of 25 Adam steps within 2.2 % of the bf16x3 parity mode; does NOT meet 1e-3 on every tensor "
            "(bf16x3 does); frame decisions of a 1x-cost 16-bit detector may flip within 3e-3 of the threshold (use 'mixed' for inference)",
    "mixed": "frame decisions equal the f32 reference's (two-pass detector: fp16 for every clip, bf16x3 again for the clips with a logit "
             "inside 9e-3 x max(1, max |logit|, max (|W2| a + |b2|)) of the threshold; logits of re-run clips 1-4e-5, of the others "
             "1.5-1.8e-3), everything else fp16: eval n_pred/mask 6-7e-4 (asserted 1e-3), waveform 1.1-4.8e-3, SI-SDR delta <= 0.016 dB",
    "bf16": "eval outputs 0.5-1.9e-2 rel vs reference goldens (tests assert 6e-2): does NOT meet the 1e-3 bar",
    "bf16x3": "eval outputs 1-4e-5 rel vs reference goldens (tests assert 1e-3)",
}


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(runs=5):
    """The oracle (CPU restatement of the reference path, torch fp32 + numpy f64) timed on the host cores with the
    protocol SURVEY.md 8-d / BASELINE.md 3 fix: BASELINE configs[0] = ONE 2 s clip through the whole inference pipeline
    (STFT -> detector -> bits->mask -> STFT -> JointModel -> mask apply -> ISTFT), 1 warm-up, median of `runs` runs --
    at 8, 32 and 128 torch threads (as far as the host has them), the BEST of which is reported with its thread count;
    plus the denoiser's training forward+backward at B=2 at that thread count (1 warm-up, best of 2)."""
    import statistics
    from oracle import frontend as ofe
    from oracle import nets as onet
    from sos_amd.dataset import synth_batch
    ncpu = os.cpu_count() or 1
    sd1 = onet.closed_form_state(onet.detector_spec(), seed=1)
    sd2 = onet.closed_form_state(onet.joint_spec(), seed=2)
    raw = synth_batch(0, 2)

    def infer(waves):
        S = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in waves]).astype(np.float32))
        with torch.no_grad():
            lo = onet.detector_forward(sd1, S, 60)
            bits = (torch.sigmoid(lo) >= 0.5).numpy().astype(np.uint8)
            noise = [w * ofe.convert_bitstreammask_to_audiomask(w, 14000 / 30.0, list(b)) for w, b in zip(waves, bits)]
            Sn = torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in noise]).astype(np.float32))
            n_pred, crm = onet.joint_forward(sd2, S, Sn)
        rec = onet.mask_apply(S, crm)
        return [ofe.fast_istft(r.permute(1, 2, 0).numpy()) for r in rec]

    def timed(fn, n, reduce=statistics.median):
        fn()                                   # warm-up
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return reduce(ts)

    per_threads = {}
    for nt in sorted({min(t, ncpu) for t in (8, 32, 128)}):
        torch.set_num_threads(nt)
        per_threads[nt] = timed(lambda: infer(raw["mixed"][:1]), runs)
    cores = min(per_threads, key=per_threads.get)
    t_inf = per_threads[cores]
    torch.set_num_threads(cores)

    # denoiser training step (forward + backward, train-mode BatchNorm) at B=2
    S = lambda a: torch.from_numpy(np.stack([ofe.fast_stft(w).transpose(2, 0, 1) for w in a]).astype(np.float32))  # noqa: E731
    mask = np.stack([ofe.convert_bitstreammask_to_audiomask(w, 14000 / 30.0, list(b)) for w, b in zip(raw["mixed"], raw["bits"])])
    batch = {"mixed": S(raw["mixed"]), "clean": S(raw["clean"] * (1 - mask)), "noise": S(raw["mixed"] * mask),
             "full_noise": S(raw["full_noise"])}
    sdt = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd2.items()}

    def train():
        for v in sdt.values():
            if v.requires_grad:
                v.grad = None
        _, losses = onet.denoiser_losses(sdt, batch, training=True)
        (losses["stage1"] + losses["stage2"]).backward()

    t_trn = timed(train, 2, min)
    return {"value": 1.0 / t_inf, "unit": "utterances/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "host_cpus": ncpu, "threads": cores,
            "infer_s_per_clip_by_threads": {str(k): round(v, 3) for k, v in per_threads.items()},
            "sample": f"BASELINE configs[0]: one 2 s clip (28000 samples) through the full inference pipeline, torch-CPU fp32 + numpy, "
                      f"1 warm-up + median of {runs} runs at {sorted(per_threads)} threads, best = {cores} threads: {t_inf:.2f} s/clip; "
                      f"denoiser fwd+bwd at B=2 at {cores} threads: best of 2 = {t_trn:.2f} s/step",
            "infer_s_per_clip_b1": t_inf, "denoiser_train_s_per_step_b2": t_trn,
            "denoiser_train_utt_per_s_b2": 2.0 / t_trn}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and become the launcher."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node (no CPU fallback, no oversubscription)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, cmd, env)


class Workload:
    """One (mode, precision) line: builds the resident inputs and models, exposes step()."""

    def __init__(self, mode, precision, B, rank, serial=False, graph=False, frames=None, train_detector=0, pipelined=False,
                 loader=0):
        import sos_amd
        from sos_amd import agent, pipeline, tools, transform
        from sos_amd.common import MyConfig
        from sos_amd.dataset import synth_batch
        from sos_amd.denoiser import networks as jnet
        from sos_amd.detector import networks as dnet
        self.mode, self.precision, self.B, self.graph = mode, precision, B, graph
        self.pipeline = pipeline
        sos_amd.set_precision(precision)
        torch.manual_seed(0)
        self.loader = None
        if mode == "train-av":
            # BASELINE configs[4]: the detector with its visual stream (M1/networks.py:54-77,110-118 + the fusion :135-142); synthetic
            # frames and spectrograms of the reference shapes, resident in HBM
            ag = agent.DetectorAgent(dnet.get_network(video=True), lr=1e-3)
            g = torch.Generator(device="cuda").manual_seed(1000 + rank)
            bav = {"audio": torch.randn(B, 2, 256, 178, device="cuda", generator=g),
                   "frames": torch.rand(B, 3, 60, 224, 224, device="cuda", generator=g),
                   "label": (torch.rand(B, 60, device="cuda", generator=g) > 0.3).float()}
            self.agents = (ag,)
            self.audio_seconds = B * 2.0
            self.lens = None
            self.step = self.eager = lambda: ag.train_func(bav)
            return
        det = dnet.get_network().cuda().eval()
        jm = jnet.get_network(MyConfig()).cuda().eval()
        if train_detector:
            # trained-like detector weights: `train_detector` Adam steps on fresh synthetic batches (fp16), so that
            # its logits have the confident / transition structure of a trained model instead of random-init noise around 0 --
            # what the two-pass detector's re-run fraction depends on
            from sos_amd.dataset import make_batch
            sos_amd.set_precision("fp16")
            ag0 = agent.DetectorAgent(det.train(), lr=1e-3)
            for it in range(train_detector):
                ag0.train_func(make_batch("detector", 5000 + 16 * it, 16))
            det = ag0.net.eval()
            del ag0
            sos_amd.set_precision(precision)
        self.det, self.jm = det, jm
        # B DISTINCT synthetic clips (rounds 1-4 tiled 8 clips 8x: the dominant kernel is power- and data-dependent,
        # profiles/r02_power_evidence.txt, so the batch must not repeat itself)
        raw = synth_batch(1000 * rank, B)
        if os.environ.get("SOS_BENCH_TILE8") == "1":      # diagnostic: the 8-clips-tiled-8x batch of rounds 1-4
            raw = {k: (np.tile(v[:8], (B // 8,) + (1,) * (np.ndim(v) - 1)) if isinstance(v, np.ndarray) else v) for k, v in raw.items()}
        mixed = torch.from_numpy(raw["mixed"]).cuda().contiguous()
        self.mixed = mixed
        self.audio_seconds = B * N_SAMPLES / 14000.0
        self.lens = None
        if mode == "train":
            # batch dicts of the reference schema (M1/dataset.py:348-352, M2/dataset.py:311-320), resident in HBM
            tile = lambda a: torch.from_numpy(a).cuda().contiguous()   # noqa: E731
            clean, full_noise, bits = tile(raw["clean"]), tile(raw["full_noise"]), tile(raw["bits"])
            mask, noise_sig = tools.bits_to_mask_batch(bits, 14000 / 30.0, N_SAMPLES, mixed)
            S = transform.stft_batch(torch.cat([mixed, clean * (1 - mask), noise_sig, full_noise]))
            if frames is not None and frames != S.shape[3]:
                # SURVEY.md 8-d secondary geometry (BASELINE-literal 16 kHz, n_fft 512 / hop 128 / win 512, Nyquist bin
                # dropped -> 2 x 256 x 251): throughput only, so the same spectrogram values are laid out on `frames`
                # columns (wrapped) instead of running a second front-end geometry; the step's work depends on the shape
                idx = torch.arange(frames, device=S.device) % S.shape[3]
                S = S[:, :, :, idx].contiguous()
            batch_jm = {"mixed": S[:B].contiguous(), "clean": S[B:2 * B].contiguous(), "noise": S[2 * B:3 * B].contiguous(),
                        "full_noise": S[3 * B:].contiguous()}
            batch_det = {"audio": batch_jm["mixed"], "label": bits.float()}
            ag_det = agent.DetectorAgent(det.train(), lr=1e-3)
            ag_jm = agent.DenoiserAgent(jm.train(), lr=1e-3)
            self.agents = (ag_det, ag_jm)

            def serial_step():
                # the two models back to back on ONE stream, the denoiser's branches in sequence: every kernel has the chip to
                # itself (run_timed's roofline pre-pass; --serial)
                prev, jnet.JointModel.BRANCH_STREAMS = jnet.JointModel.BRANCH_STREAMS, False
                try:
                    ag_det.train_func(batch_det)
                    ag_jm.train_func(batch_jm)
                finally:
                    jnet.JointModel.BRANCH_STREAMS = prev
            self.serial_step = serial_step

            def step():
                if serial:
                    serial_step()
                else:           # the two models are independent: one HIP stream each (+ the denoiser's branch stream)
                    agent.train_concurrent([(ag_jm, batch_jm), (ag_det, batch_det)])
            if loader:
                # --loader N: every step draws a FRESH batch from sos_amd.dataset.get_dataloader -- N worker
                # processes make the host draws, the producer thread uploads them from pinned memory and runs bits -> mask, the mix
                # at the SNR, the four STFTs and the cRM target on a side stream, two batches ahead (M2/dataset.py:44-50's
                # DataLoader(num_workers, pin_memory) + M2/dataset.py:144-320's __getitem__); the detector trains on the same clips
                from sos_amd.dataset import get_dataloader
                self.loader = get_dataloader("training", batch_size=B, num_workers=loader, model="denoiser", n_batches=10 ** 6,
                                             rank=rank, world_size=1, prefetch=2)
                it = iter(self.loader)

                def step():       # noqa: F811
                    b = next(it)
                    agent.train_concurrent([(ag_jm, b), (ag_det, {"audio": b["mixed"], "label": b["_bits"].float()})])
                self._loader_iter = it
            self.eager = step
            self.concurrent = not serial
        elif mode == "infer-ragged":
            # BASELINE configs[3]: lengths drawn uniformly from 1-10 s with seed 99, every rank its own draw
            lens = [int(v) for v in np.random.default_rng(99 + rank).uniform(14000, 140000, B)]
            pool = np.concatenate(list(synth_batch(2000 * rank, 20)["mixed"]))
            pool_t = torch.from_numpy(np.ascontiguousarray(pool)).cuda()
            clips = [pool_t[(4099 * i) % (len(pool) - 140000):][:n].contiguous() for i, n in enumerate(lens)]
            self.lens, self.audio_seconds = lens, sum(lens) / 14000.0
            graphed = pipeline.GraphedDenoiser(det, jm, max_graphs=16) if graph else None
            self.eager = lambda: pipeline.denoise_ragged(det, jm, clips)
            step = (lambda: graphed.denoise_mixed(clips)) if graph else self.eager
        else:
            graphed = pipeline.GraphedDenoiser(det, jm) if graph else None
            self.eager = lambda: pipeline.denoise(det, jm, mixed)
            step = (lambda: graphed(mixed, clone=False)) if graph else self.eager
            if pipelined and not graph:       # consecutive batches on two alternating streams (pipeline.PipelinedDenoiser)
                piped = pipeline.PipelinedDenoiser(det, jm)
                step = lambda: piped(mixed)   # noqa: E731  (the timed region ends with torch.cuda.synchronize(): both streams drained)
        self.step = step

    def close(self):
        """Stop the data loader's producer thread and worker processes (--loader workloads)."""
        if getattr(self, "loader", None) is not None:
            self._loader_iter.close()
            self.loader.close()
            self.loader = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gflop_per_utt(self):
        if self.mode == "train-av":
            return 3.0 * GFLOP_PER_CLIP_AV_FWD
        g = GFLOP_PER_UTT_INFER * (3.0 if self.mode == "train" else 1.0)   # algorithmic (reference) FLOPs: bf16x3's 3x MACs do not count
        if self.lens is not None:                                         # FLOPs scale with the frames of a clip: 2 s = 178 frames
            g *= sum(1 + n // 158 for n in self.lens) / (178.0 * self.B)
        return g


def _dominant(summ):
    """The dominant kernel launch signature of a step: the one that carries the most algorithmic FLOPs (per-launch FLOPs x
    launches; ties -> the larger total time).  Chosen by WORK, not by warm-up time: the first launches of a process include
    one-time host work between the event brackets (code-object loading, the tile search of a weight-gradient shape), which once
    made a 15 TFLOP/s first-layer gradient the 'dominant' signature of a run (roofline.frac 0.005)."""
    return max(summ, key=lambda k: (summ[k]["flops"] * summ[k]["launches"], summ[k]["total_ms"]))


def run_timed(wl, steps, warmup, barrier, profile=True):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize; returns (seconds, dominant launch
    signature, its HIP-event summary).  profile=False: no event brackets (secondary lines)."""
    import sos_amd
    from sos_amd import engine
    sos_amd.set_precision(wl.precision)
    dom = prof = None
    use_graph = wl.graph and wl.mode != "train"
    prepass = None
    if profile and wl.mode == "train" and getattr(wl, "concurrent", False):
        # Concurrent schedule (two model streams + the denoiser's branch stream): the dominant kernel time-shares the chip with
        # other streams' kernels, so a HIP-event bracket inside the timed region measures the co-run, not the kernel.  The
        # roofline figure is therefore taken in a SERIAL eager pre-pass of the same training step, outside the timed region
        # (one stream, every kernel alone on the chip; same launches, same data) -- what the hipGraph lines below have always
        # done.  The in-region bracket is still taken and reported next to it (`in_timed_region`).
        engine.PROFILER = engine.LaunchProfiler()
        wl.serial_step()
        dom = _dominant(engine.PROFILER.summary())
        wl.serial_step()                                # (clocks: a second step before the measured ones)
        engine.PROFILER = engine.LaunchProfiler(only=dom)
        for _ in range(3):
            wl.serial_step()
        prepass = engine.PROFILER.summary()[dom]
        engine.PROFILER = None
        for _ in range(max(1, warmup)):
            wl.step()
        engine.PROFILER = engine.LaunchProfiler(only=dom)
    elif profile and use_graph:
        # hipGraph replay: individual launches cannot be bracketed inside a replayed graph, so the dominant kernel is found
        # and timed (HIP events on the launch stream) in an EAGER pass of the same step before the graphs are captured
        engine.PROFILER = engine.LaunchProfiler()
        wl.eager()
        summ = engine.PROFILER.summary()
        dom = _dominant(summ)
        engine.PROFILER = engine.LaunchProfiler(only=dom)
        for _ in range(2):
            wl.eager()
        prof = engine.PROFILER.summary()[dom]
        engine.PROFILER = None
        for _ in range(max(1, warmup)):
            wl.step()                                   # captures the graphs
    elif profile:
        # warm-up; the first pass also finds the dominant kernel launch signature
        engine.PROFILER = engine.LaunchProfiler()
        for _ in range(max(1, warmup)):
            wl.step()
        summ = engine.PROFILER.summary()
        dom = _dominant(summ)
        engine.PROFILER = engine.LaunchProfiler(only=dom)
    else:
        for _ in range(max(1, warmup)):
            wl.step()
    for ag in getattr(wl, "agents", ()):            # data-parallel instrumentation: the timed steps only
        if getattr(ag, "bucketer", None) is not None:
            ag.bucketer.comm_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    if profile and not use_graph:
        prof = engine.PROFILER.summary()[dom]
    engine.PROFILER = None
    if prepass is not None:
        prof = dict(prepass, timed_in="serial pre-pass", in_region_avg_ms=prof["avg_ms"], in_region_launches=prof["launches"])
    return dt, dom, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU per step (default 64; 256 for infer-ragged; 32 for train-av)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "infer-ragged", "train-av"])
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "bf16x3", "mixed"],
                    help="16-bit storage type of activations/weights (MFMA rate is the same for bf16 and fp16); bf16x3 = "
                         "three-pass hi/lo split (3x the MACs, ~fp32 accuracy); mixed = detector in bf16x3 (frame decisions "
                         "equal the f32 reference's), everything else fp16.  Default: fp16 for train, mixed for inference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (inference, ragged, bf16 / mixed training)")
    ap.add_argument("--serial", action="store_true", help="train mode: run the two models back to back on one stream")
    ap.add_argument("--graph", action="store_true",
                    help="infer / infer-ragged: replay hipGraph-captured launch sequences (pipeline.GraphedDenoiser) instead "
                         "of eager launches")
    ap.add_argument("--pipelined", action="store_true",
                    help="infer: consecutive batches alternate between two HIP streams (pipeline.PipelinedDenoiser): the latency-bound "
                         "tail of one batch runs under the head of the next")
    ap.add_argument("--train-detector", type=int, default=0,
                    help="inference modes: train the detector for this many Adam steps on synthetic batches first (trained-like logits: "
                         "what the two-pass detector's re-run fraction depends on)")
    ap.add_argument("--loader", type=int, default=0, metavar="WORKERS",
                    help="train: draw a fresh batch from sos_amd.dataset.get_dataloader every step (WORKERS host processes, producer "
                         "thread, side stream, pinned uploads) instead of re-using one resident batch")
    ap.add_argument("--force-buckets", action="store_true",
                    help="world of one: run the data-parallel gradient path anyway (bucket copies + RCCL all-reduce of every "
                         "bucket on a 1-rank group) to measure its overhead on a single GPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.precision is None:
        args.precision = DEFAULT_PRECISION[args.mode]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args.gpus)                       # does not return
    # stdout carries exactly ONE line, the JSON record: RCCL prints a version banner through C stdio (flushed at exit, i.e.
    # BEHIND the record) whenever a communicator is created.  File descriptor 1 is pointed at stderr for the whole run and
    # the record is written to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dist = None
    n_gpus = 1
    if world > 1 or args.force_buckets:
        os.environ.setdefault("SOS_DDP_PROFILE", "1")     # HIP-event brackets around every bucket's collective (agent.GradBucketer.comm_stats)
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ["SOS_FORCE_BUCKETS"] = "1"
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # the communicator that will carry the gradients: count its members
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        n_gpus = int(round(float(ones.item())))
        if n_gpus != world:
            raise SystemExit(f"bench.py: the RCCL communicator has {n_gpus} members, expected {world}")

    import sos_amd

    if args.batch is None:
        args.batch = 256 if args.mode == "infer-ragged" else (32 if args.mode == "train-av" else 64)
    B = args.batch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    wl = Workload(args.mode, args.precision, B, rank, serial=args.serial, graph=args.graph, train_detector=args.train_detector,
                  pipelined=args.pipelined, loader=args.loader if args.mode == "train" else 0)
    dt, dom, prof = run_timed(wl, args.steps, args.warmup, barrier)
    ddp = None
    if dist is not None:
        # what the data-parallel path did: per-rank time of the K steps (min / max over the ranks: rank skew), and
        # per model the buckets, bytes and collective milliseconds per step -- HIP events on the stream each collective was
        # enqueued on (SOS_DDP_COMM: inline / own / shared), maximum over the ranks
        every = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, torch.tensor([dt], device="cuda", dtype=torch.float64))
        per_rank = [float(t.item()) for t in every]
        ddp = {"comm_mode": os.environ.get("SOS_DDP_COMM", "inline"), "world": world,
               "rank_step_ms_min": 1e3 * min(per_rank) / args.steps, "rank_step_ms_max": 1e3 * max(per_rank) / args.steps, "models": {}}
        for name, ag in zip(("denoiser", "detector") if args.mode == "train" else ("detector_av",), reversed(getattr(wl, "agents", ())) if args.mode == "train" else getattr(wl, "agents", ())):
            if getattr(ag, "bucketer", None) is None:
                continue
            st = ag.bucketer.comm_stats()
            cm = torch.tensor([st["comm_ms_per_step"]], device="cuda", dtype=torch.float64)
            dist.all_reduce(cm, op=dist.ReduceOp.MAX)
            ddp["models"][name] = {"buckets_per_step": st["buckets_per_step"], "bytes_per_step": st["bytes_per_step"],
                                   "comm_ms_per_step_max_over_ranks": float(cm.item()), "bucket_bytes_cap": st["bucket_bytes_cap"]}
        ddp["comm_ms_per_step"] = sum(m["comm_ms_per_step_max_over_ranks"] for m in ddp["models"].values())
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        value = n_gpus * B * args.steps / dt
        ach = prof["flops"] / (prof["avg_ms"] * 1e-3) / 1e12
        train = args.mode == "train"
        train_av = args.mode == "train-av"
        from sos_amd.denoiser import networks as _jn
        branch_streams = _jn.JointModel.BRANCH_STREAMS
        gflop = wl.gflop_per_utt()
        audio_seconds = wl.audio_seconds
        # HBM traffic of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, FETCH_SIZE doubled per the gfx950 correction); null when the signature has no PMC record
        traffic = traffic_source = None
        try:
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv96.json")))[-1]      # latest round
            pmc = json.load(open(pmc_path))
            if dom[0] == "conv" and tuple(dom[1:8]) == (5, 5, 1, 1, 1, 96, 96) and dom[8] == 64:
                traffic = pmc["hbm_bytes_per_launch"]
                # (PMC counters cannot be collected inside this timed run: the field is READ from the committed pass)
                traffic_source = ("profiles/" + os.path.basename(pmc_path) + " (stand-alone rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the "
                                  "same launch signature, tools/refresh_profiles.sh; not measured in this run)")
        except Exception:
            pass
        line = {
            "metric": ("utterances/sec (variable-length clips 1-10 s), " if args.mode == "infer-ragged" else "utterances/sec (2 s clips), ") +
                      ("training step: detector + denoiser forward/backward/Adam" if train
                       else "training step of the audio-visual detector (audio + 60 x 224 x 224 video frames): forward/backward/Adam" if train_av
                       else "inference pipeline STFT->detector->mask->denoiser->ISTFT"),
            "value": value, "unit": "utterances/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("training (BASELINE configs[1])" if train else
                                    "audio-visual training (BASELINE configs[4]: batch 128 = 4 GPUs x 32)" if train_av else
                                    "inference" if args.mode == "infer" else "variable-length inference (BASELINE configs[3])") +
                                   (f", batch={B} clips/GPU of 2 s @14 kHz (28000 samples, STFT 510/158/400 -> 2x256x178), "
                                    if args.mode != "infer-ragged" else
                                    f", batch={B} clips/GPU, lengths U(1 s, 10 s) seed 99 @14 kHz ({audio_seconds:.0f} s of audio, "
                                    f"{audio_seconds / 2.0:.0f} 2-s equivalents; STFT 510/158/400 -> 2x256xT, T = 89..887), clips of "
                                    "different lengths share launches (per-clip geometry in the kernels, no padding of the data), ") +
                                   ("detector with its video branch (7 Conv3d blocks + fusion into the BiLSTM), 60 frames of 224 x 224 per clip, "
                                    "random-init weights (manual_seed 0), BCE, Adam lr 1e-3, per-rank BatchNorm" if train_av else
                                    "detector + two-stage denoiser, random-init weights (manual_seed 0)"
                                    + (", detector BCE + denoiser 2xMSE, Adam lr 1e-3, per-rank BatchNorm" if train else "")),
                       "clips_per_gpu": B, "n_samples": N_SAMPLES, "mode": args.mode, "precision": args.precision,
                       "parity": PARITY_NOTE[args.precision],
                       "data_parallel": ddp,
                       "streams": (3 if branch_streams else 2) if (train and not args.serial) else 1,
                       "branch_streams": bool(train and not args.serial and branch_streams),
                       "forced_gradient_buckets": bool(args.force_buckets),
                       "data_loader_workers": int(args.loader) if train else 0,
                       "hipgraph": bool(args.graph),
                       "realtime_factor": value * audio_seconds / B,
                       "end_to_end_tflops": value * gflop / 1e3 / n_gpus},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("wgrad_kernel " if dom[0] == "wgrad" else "conv_mfma_kernel ") + str(dom),
                         "launches": prof["launches"],
                         "avg_ms": prof["avg_ms"], "flops_per_launch": prof["flops"],
                         # where the HIP-event brackets were taken: "timed region" (one launch stream, the kernel alone on the
                         # chip) or "serial pre-pass" (concurrent schedules: the same step run serially right before the timed
                         # region, run_timed); in_timed_region = the same signature bracketed inside the timed region, where it
                         # time-shares the chip with the other streams' kernels (a property of the schedule, not of the kernel)
                         "timed_in": prof.get("timed_in", "timed region")},
        }
        if "in_region_avg_ms" in prof:
            line["roofline"]["in_timed_region"] = {"avg_ms": prof["in_region_avg_ms"], "launches": prof["in_region_launches"],
                                                   "frac": prof["flops"] / (prof["in_region_avg_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
        if world == 1 and train and not args.no_secondary and not args.force_buckets and not args.serial:
            # the round's other lines on the same box, AFTER (and outside) the headline's timed region
            del wl
            torch.cuda.empty_cache()
            sec = {}

            def _roof(dom2, prof2):
                ach2 = prof2["flops"] / (prof2["avg_ms"] * 1e-3) / 1e12
                return {"bound": "mfma", "kernel": str(dom2), "avg_ms": prof2["avg_ms"], "launches": prof2["launches"],
                        "flops_per_launch": prof2["flops"], "achieved": ach2, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach2 / PEAK_BF16_TFLOPS, "timed_in": prof2.get("timed_in", "timed region")}

            for key, mode, prec, b, k, w, fr in (("infer_mixed_utt_s", "infer", "mixed", 64, 10, 3, "roofline"),
                                                 ("infer_mixed_trained_detector_utt_s", "infer", "mixed", 64, 10, 3, "trained"),
                                                 ("infer_fp16_utt_s", "infer", "fp16", 64, 10, 3, None),
                                                 # consecutive batches on two alternating streams (pipeline.PipelinedDenoiser)
                                                 ("infer_fp16_pipelined_utt_s", "infer", "fp16", 64, 10, 3, "pipelined"),
                                                 ("infer_mixed_pipelined_utt_s", "infer", "mixed", 64, 10, 3, "pipelined"),
                                                 ("ragged_mixed_clips_s", "infer-ragged", "mixed", 256, 3, 1, None),
                                                 ("ragged_mixed_hipgraph_clips_s", "infer-ragged", "mixed", 256, 3, 1, "graph"),
                                                 ("train_mixed_utt_s", "train", "mixed", 64, 10, 3, None),
                                                 # the ONE mode that meets the north_star's 1e-3 on every tensor (3x the MACs)
                                                 ("train_bf16x3_utt_s", "train", "bf16x3", 64, 3, 1, None),
                                                 ("infer_bf16x3_utt_s", "infer", "bf16x3", 64, 3, 1, None)):
                try:
                    w2 = Workload(mode, prec, b, rank, graph=fr == "graph", train_detector=60 if fr == "trained" else 0,
                                  pipelined=fr == "pipelined")
                    if prec == "mixed" and mode == "infer":
                        w2.pipeline.two_pass_stats(reset=True)
                    dt2, dom2, prof2 = run_timed(w2, k, w, barrier, profile=fr == "roofline")
                    sec[key] = round(b * k / dt2, 1)
                    if prec == "mixed" and mode == "infer":
                        # two-pass detector (pipeline.detect): clips whose fp16 logits came within the error band of the
                        # threshold and were re-run in the parity precision, over all clips of the run
                        mk, seen = w2.pipeline.two_pass_stats(reset=True)
                        sec[key.replace("_utt_s", "_rerun_fraction")] = round(mk / max(1, seen), 4)
                    if fr == "roofline":
                        sec[key.replace("_utt_s", "_roofline")] = _roof(dom2, prof2)
                    if mode == "infer-ragged":
                        sec["ragged_realtime_factor"] = round(b * k / dt2 * w2.audio_seconds / b, 0)
                    del w2
                    torch.cuda.empty_cache()
                except Exception as e:        # a secondary line never takes the headline down
                    sec[key] = None
                    sec[key + "_error"] = repr(e)[:200]

            def _alternate(make_a, make_b, rounds=3, steps=12, warm=6):
                """Two variants of the headline workload timed ALTERNATELY (A B A B A B; each run builds its workload afresh, `warm`
                untimed + `steps` timed steps, then frees it): the box's clocks drift over the minute the secondary lines take, so
                a ratio is the mean of B over the mean of A of interleaved runs, not a comparison with the headline of the record.
                (Round 5 first kept both workloads alive and alternated between them: that measured the branch streams 2.8 %
                SLOWER while separate processes on the same box had them 2.2 % faster -- tools/probe/ab_env.sh, 542.7 vs 554.6 --;
                two live workloads hold two sets of per-stream allocator pools.  One workload at a time, like a real job.)"""
                ta, tb = [], []
                w0 = make_a()
                run_timed(w0, 3, 8, barrier, profile=False)       # (thrown away: the first training run after the inference lines measures slow)
                w0.close()
                del w0
                torch.cuda.empty_cache()
                for _ in range(rounds):
                    for mk, acc in ((make_a, ta), (make_b, tb)):
                        wx = mk()
                        dtx, _, _ = run_timed(wx, steps, warm, barrier, profile=False)
                        acc.append(64 * steps / dtx)
                        wx.close()
                        del wx
                        torch.cuda.empty_cache()
                return sum(ta) / len(ta), sum(tb) / len(tb), ta, tb

            # the headline schedule (branch streams, the default since round 5) against the one-stream-per-model schedule of rounds 2-4
            try:
                from sos_amd.denoiser import networks as _jnet

                class _NoBranch(Workload):
                    def __init__(self, *a_, **k_):
                        super().__init__(*a_, **k_)
                        inner = self.step

                        def step():
                            prev, _jnet.JointModel.BRANCH_STREAMS = _jnet.JointModel.BRANCH_STREAMS, False
                            try:
                                inner()
                            finally:
                                _jnet.JointModel.BRANCH_STREAMS = prev
                        self.step = step
                va, vb, la, lb = _alternate(lambda: _NoBranch("train", "fp16", 64, rank), lambda: Workload("train", "fp16", 64, rank))
                sec["train_fp16_no_branch_streams_utt_s"] = round(va, 1)
                sec["train_fp16_branch_streams_utt_s"] = round(vb, 1)
                sec["train_fp16_branch_streams_ratio"] = round(vb / va, 4)
                sec["train_fp16_branch_streams_runs"] = {"without": [round(v, 1) for v in la], "with": [round(v, 1) for v in lb]}
            except Exception as e:
                sec["train_fp16_branch_streams_ratio"] = None
                sec["train_fp16_branch_streams_ratio_error"] = repr(e)[:200]
            # BASELINE configs[1] names bf16; the headline runs IEEE half on the same kernels at the same MFMA rate (8x less rounding
            # noise).  The two storage types alternated like every other ratio of this record (round 5's single
            # un-bracketed 10-step bf16 line read 506.5 against a 566.7 headline on the driver's box, 552.7 / 551.3 on another)
            try:
                va, vb, la, lb = _alternate(lambda: Workload("train", "fp16", 64, rank), lambda: Workload("train", "bf16", 64, rank))
                sec["train_bf16_utt_s"] = round(vb, 1)
                sec["train_bf16_ratio"] = round(vb / va, 4)
                sec["train_bf16_runs"] = {"fp16": [round(v, 1) for v in la], "bf16": [round(v, 1) for v in lb]}
            except Exception as e:
                sec["train_bf16_utt_s"] = None
                sec["train_bf16_utt_s_error"] = repr(e)[:200]
            # the headline step fed by the asynchronous data loader (a fresh batch per step) against the resident batch
            try:
                nw = max(2, min(8, (os.cpu_count() or 4) // 4))
                va, vb, la, lb = _alternate(lambda: Workload("train", "fp16", 64, rank), lambda: Workload("train", "fp16", 64, rank, loader=nw))
                sec["train_fp16_resident_batch_utt_s"] = round(va, 1)
                sec["train_fp16_loader_utt_s"] = round(vb, 1)
                sec["train_fp16_loader_ratio"] = round(vb / va, 4)
                sec["train_fp16_loader_workers"] = nw
                sec["train_fp16_loader_runs"] = {"resident": [round(v, 1) for v in la], "loader": [round(v, 1) for v in lb]}
            except Exception as e:
                sec["train_fp16_loader_utt_s"] = None
                sec["train_fp16_loader_utt_s_error"] = repr(e)[:200]
            # the audio-visual variant's per-GPU share of BASELINE configs[4] (32 clips of 60 x 224 x 224 frames + audio): one
            # training step of the detector with its video branch
            try:
                sos_amd.set_precision("fp16")
                from sos_amd import agent as _agent
                from sos_amd.detector import networks as _dnet
                torch.manual_seed(0)
                ag = _agent.DetectorAgent(_dnet.get_network(video=True), lr=1e-3)
                bav = {"audio": torch.randn(32, 2, 256, 178, device="cuda"), "frames": torch.rand(32, 3, 60, 224, 224, device="cuda"),
                       "label": (torch.rand(32, 60, device="cuda") > 0.3).float()}
                for _ in range(2):
                    ag.train_func(bav)
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    ag.train_func(bav)
                barrier()
                sec["audiovisual_train_fp16_b32_clips_s"] = round(32 * 3 / (time.perf_counter() - t0), 1)
                del ag, bav
                torch.cuda.empty_cache()
            except Exception as e:
                sec["audiovisual_train_fp16_b32_clips_s"] = None
                sec["audiovisual_train_fp16_b32_clips_s_error"] = repr(e)[:200]
            # the data-parallel gradient path on this one GPU (a 1-rank RCCL group: every bucket gathered and all-reduced,
            # one communicator per model): what the bucket copies + collective launches cost the headline step
            try:
                import torch.distributed as dist1
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
                dist1.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

                def _forced():
                    os.environ["SOS_FORCE_BUCKETS"] = "1"
                    try:
                        return Workload("train", "fp16", 64, rank)
                    finally:
                        os.environ.pop("SOS_FORCE_BUCKETS", None)

                try:       # (a world of one without SOS_FORCE_BUCKETS builds no bucketer: the plain single-process path)
                    va, vb, la, lb = _alternate(lambda: Workload("train", "fp16", 64, rank), _forced)
                finally:
                    dist1.destroy_process_group()
                sec["train_fp16_before_forced_buckets_utt_s"] = round(va, 1)
                sec["train_fp16_forced_buckets_utt_s"] = round(vb, 1)
                sec["train_fp16_forced_buckets_ratio"] = round(vb / va, 4)
                sec["train_fp16_forced_buckets_runs"] = {"without": [round(v, 1) for v in la], "with": [round(v, 1) for v in lb]}
            except Exception as e:
                sec["train_fp16_forced_buckets_utt_s"] = None
                sec["train_fp16_forced_buckets_utt_s_error"] = repr(e)[:200]
            # (the other-geometry line comes LAST: the first fp16 training run after it measured 4 % slow for dozens of steps -- cause
            # not tracked down)
            try:
                w2 = Workload("train", "fp16", 64, rank, frames=251)
                dt2, _, _ = run_timed(w2, 5, 2, barrier, profile=False)
                sec["train_fp16_16khz_2x256x251_utt_s"] = round(64 * 5 / dt2, 1)
                del w2
                torch.cuda.empty_cache()
            except Exception as e:
                sec["train_fp16_16khz_2x256x251_utt_s"] = None
                sec["train_fp16_16khz_2x256x251_utt_s_error"] = repr(e)[:200]
            sec["note"] = ("same box, after the headline loop: infer = B=64 2 s clips x 10 steps (infer_mixed_roofline: its dominant launch "
                           "signature bracketed with HIP events in that run; 'mixed' = two-pass detector: fp16 for every clip, bf16x3 again "
                           "for the clips with a logit inside the fp16 error band of the threshold -- *_rerun_fraction; "
                           "infer_mixed_trained_detector = the same with a detector trained for 60 Adam steps on synthetic batches first, "
                           "random-init logits all sit on one side of the threshold: no clip is marked; infer_*_pipelined = consecutive batches alternating between two "
                           "HIP streams, pipeline.PipelinedDenoiser: one batch's BiLSTM / FC / ISTFT tail under the next one's head); ragged = BASELINE configs[3], B=256 "
                           "U(1 s,10 s) x 3 steps (eager launches / replayed hipGraphs); train_* = the headline workload in another precision x 10 steps "
                           "(bf16x3 = the three-pass parity mode, 3x the MACs, the one mode within 1e-3 of the reference on every tensor: 3 steps, "
                           "also as infer_bf16x3); "
                           "train_fp16_16khz_2x256x251 = SURVEY.md 8-d's secondary (BASELINE-literal 16 kHz / STFT 512-128, Nyquist dropped) spectrogram "
                           "geometry, 1.41x the FLOPs per clip, throughput only; audiovisual_train = the detector with its video branch at BASELINE "
                           "configs[4]'s per-GPU share (32 clips of 60 x 224 x 224 frames) x 3 steps; train_fp16_forced_buckets = the headline "
                           "workload with the data-parallel gradient path forced in a world of one (1-rank RCCL groups, one per model); "
                           "train_fp16_branch_streams = the headline schedule, train_fp16_no_branch_streams = SOS_BRANCH_STREAMS=0 (the "
                           "schedule of rounds 2-4); *_ratio = mean of three runs with over mean of three runs without, timed alternately, "
                           "every run on a freshly built workload (6 warm-up + 12 timed steps per run; *_runs lists them); train_fp16_loader = the "
                           "headline step drawing a FRESH batch from sos_amd.dataset.get_dataloader every step (train_fp16_loader_workers "
                           "host processes make the draws; pinned uploads, bits -> mask, the mix, 4 STFTs and the cRM target on a side "
                           "stream, two batches ahead) against the resident batch of the headline, alternated the same way")
            line["secondary"] = sec
            sos_amd.set_precision(args.precision)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
