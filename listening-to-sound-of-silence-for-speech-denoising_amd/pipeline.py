"""On-device inference chain: STFT -> detector -> threshold -> bits->sample-mask -> noise-interval
STFT -> JointModel -> mask apply -> ISTFT, with no host round trip.

Restates M1/predict.py:38-233 (detector pass) + M2/predict.py:255-326,377-447 (denoiser pass);
the reference hands results over through JSON/WAV files on disk, here the hand-off is the
`bits` tensor in HBM."""
import os

import torch

from . import get_mode, get_precision, precision_scope
from . import tools
from . import transform

SR = 14000
FPS = 30.0
SIGMOID_THRESHOLD = 0.5     # M1/predict.py:30
MIN_FRAMES = 65             # ((T + 1) // 2 + 1) // 2 > 16, the reflect padding of InpaintNet's dilation-16 block


def n_video_frames(n_samples, sr=SR, fps=FPS):
    return int(round(n_samples / sr * fps))


# ---- the detector of the 'mixed' mode in two passes (round 5).  The only thing the pipeline takes from the detector is the SIGN
# of every frame's logit (M1/predict.py:117-119: sigmoid(logit) >= 0.5), and a 16-bit pass gets the sign wrong only where the
# logit lies inside its own error band around 0.  So: (1) the whole batch through the detector in fp16 (1x the MACs); (2) a device
# kernel marks the clips with ANY frame at |logit| < TWO_PASS_BAND x max(1, max_t |logit|) -- three times the fp16 logit tolerance
# the parity tests assert (tests/test_gpu_nets.py: 3e-3 of max |logit|; observed 1.5-1.8e-3); (3) ONLY the marked clips run again
# in the parity precision (bf16x3, 3x the MACs) through the kernels' ragged per-clip geometry -- unmarked clips get width 0 in
# every device table, their tiles exit at once, nothing is read back by the host (hipGraph-capturable); (4) marked clips take the
# parity logits, the others keep the fp16 ones.  Frame decisions equal the parity detector's: by recomputation inside the band,
# and outside it because an fp16 logit further than three tolerances from 0 has the reference's sign.  SOS_MIXED_TWO_PASS=0: every
# clip through the parity-precision detector (rounds 3-4).  The audio-visual variant and plain modes are untouched.
TWO_PASS = os.environ.get("SOS_MIXED_TWO_PASS", "1") != "0"
TWO_PASS_BAND = 3 * 3e-3
# Round 6: the band is relative to max(1, max_t |logit|, max_t scale_t) with scale_t = |W2| a_t + |b2| -- the magnitude
# of the TERMS of the last layer's dot product (one more 100 -> 1 launch in the fp16 pass).  An fp16 pass's logit error is
# proportional to the size of what is summed, not to the sum: a clip whose logits all hover near 0 by cancellation got a band of
# 0.009 from its own max |logit| while its error is that of its neighbours with large logits.  scale_t >= |logit_t|, so every clip
# the round-5 rule marked is still marked.  SOS_MIXED_BAND_SCALE=0: the round-5 rule (A/B).
BAND_SCALE = os.environ.get("SOS_MIXED_BAND_SCALE", "1") != "0"
_fixed_rag = {}


def detect(detector, S_mixed, n_frames, rag=None, return_mark=False, between=None):
    """Detector logits (B, n_frames) for the pipeline: the two-pass scheme above under set_precision('mixed'), one plain call
    otherwise.  rag: engine.Ragged of a variable-length batch.  between: optional callable run between the two passes (the
    pipeline starts the denoiser's encoder_x there: the second pass -- a few clips, or none -- cannot fill the chip)."""
    from . import engine as E
    # `between` fires where the chip starts to idle: before the (first) pass's BiLSTM if the detector offers the hook
    # (detector.networks.AudioVisualNet.forward(before_lstm=)), else after the pass
    hook = {}
    if between is not None and EARLY_X and "before_lstm" in getattr(getattr(detector.forward, "__code__", None), "co_varnames", ()):
        hook = dict(before_lstm=between)
        between = None
    if get_mode() != "mixed" or not TWO_PASS or getattr(detector, "video_feat", 0):
        lo = detector(s=S_mixed, v_num_frames=n_frames, rag=rag, **hook)
        return (lo, None) if return_mark else lo
    scale = None
    with precision_scope("fp16"):
        if BAND_SCALE and "return_scale" in getattr(getattr(detector.forward, "__code__", None), "co_varnames", ()):
            lo16, scale = detector(s=S_mixed, v_num_frames=n_frames, rag=rag, return_scale=True, **hook)
        else:
            lo16 = detector(s=S_mixed, v_num_frames=n_frames, rag=rag, **hook)
    if between is not None:
        between()
    base = rag
    if base is None:                                  # fixed-length batch: every clip has the same geometry (cached tables)
        B, T = S_mixed.shape[0], S_mixed.shape[3]
        key = (B, T, n_frames, str(S_mixed.device))
        base = _fixed_rag.get(key)
        if base is None:
            base = _fixed_rag[key] = E.Ragged([T] * B, S_mixed.device, n_vframes=[n_frames] * B)
            base.level(0), base.tab(base.n_vframes)
    mrag, mark = E.mask_ragged(base, lo16, TWO_PASS_BAND, [base.widths(0), base.n_vframes], scale=scale)
    lo3 = detector(s=S_mixed, v_num_frames=n_frames, rag=mrag)
    lo = torch.where(mark[:, None] != 0, lo3, lo16)
    return (lo, mark) if return_mark else lo


OVERLAP_X = os.environ.get("SOS_MIXED_OVERLAP_X", "1") != "0"      # A/B: encoder_x of the denoiser under the detector's low-occupancy tail
EARLY_X = os.environ.get("SOS_EARLY_X", "1") != "0"                # A/B: ... from the detector's BiLSTM on (every mode), not only between the two passes


def _begin_x(denoiser, S_mixed, rag=None, mode=None):
    """Start the denoiser's encoder_x on its branch stream if the model offers it (denoiser.networks.JointModel.begin_x).
    mode: the EFFECTIVE precision of the pipeline call, captured by denoise() / _denoise_group_padded() at their entry
    (get_precision(): honours an enclosing precision_scope, reads 'mixed' as 'fp16') -- the hook fires inside the detector's
    own scope (bf16x3 for the one-pass parity detector of 'mixed'), and the denoiser(...) call that later consumes the handle runs
    in the caller's mode: both sides of the hand-off must agree."""
    if not OVERLAP_X or not hasattr(denoiser, "begin_x"):
        return None
    with precision_scope(mode if mode is not None else get_precision()):
        return denoiser.begin_x(S_mixed, rag)


def two_pass_stats(device=None, reset=False):
    """(clips re-run in the parity precision, clips seen) by detect() on `device` so far; synchronises."""
    from . import engine as E
    c = E.band_count(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
    out = tuple(int(v) for v in c.tolist())
    if reset:
        c.zero_()
    return out


@torch.no_grad()
def denoise(detector, denoiser, mixed, sr=SR, fps=FPS, bits=None, return_all=False):
    """mixed f32 (B, N) on the GPU -> denoised f32 (B, hop*(T-1)).  `bits` (uint8 (B, n_frames),
    1 = non-silent) overrides the detector (M2/predict.py's `recovered_prediction` input)."""
    B, N = mixed.shape
    mode = get_precision()            # the denoiser's mode, taken OUTSIDE the detector's precision scope
    S_mixed = transform.stft_batch(mixed)
    logits = mark = None
    started = []
    if bits is None:
        logits, mark = detect(detector, S_mixed, n_video_frames(N, sr, fps), return_mark=True,
                              between=lambda: started.append(_begin_x(denoiser, S_mixed, mode=mode)))
        bits, _ = tools.threshold_bits(logits, SIGMOID_THRESHOLD)
    mask, noise_sig = tools.bits_to_mask_batch(bits, float(sr) / fps, N, mixed)
    S_noise = transform.stft_batch(noise_sig)
    n_pred, crm = denoiser(S_mixed, S_noise, **({"started": started[0]} if started and started[0] is not None else {}))
    S_out = transform.batch_fast_icRM_sigmoid(S_mixed, crm)
    out = transform.istft_batch(S_out)
    if return_all:
        # mark (two-pass detector of the 'mixed' mode only): int32 (B,), 1 = the clip's logits are the parity-precision pass's
        return dict(out=out, logits=logits, bits=bits, mask=mask, n_pred=n_pred, crm=crm, S_mixed=S_mixed,
                    S_noise=S_noise, S_out=S_out, mark=mark)
    return out


def _group_geometry(ns, device, sr, fps):
    """Per-clip frame / video-frame counts of a ragged group and their device tables (engine.Ragged)."""
    from . import engine as E
    T = [1 + n // transform.HOP_LENGTH for n in ns]
    nv = [n_video_frames(n, sr, fps) for n in ns]
    if min(T) < MIN_FRAMES:
        # the U-Net's dilation-16 block reflects 16 columns at a quarter of the resolution: nn.ReflectionPad2d raises for a
        # shorter input (M2/networks.py:105,181), and so does the single-clip path (sos_conv2d_fwd's descriptor check)
        raise ValueError(f"clips need at least {MIN_FRAMES} STFT frames ({MIN_FRAMES * transform.HOP_LENGTH} samples); got {min(ns)} samples")
    rag = E.Ragged(T, device, n_vframes=nv, n_samples=ns)
    rag.tab(ns), rag.tab(nv), rag.level(0)          # uploaded here, not inside a later stream capture
    return rag


def _denoise_group_padded(detector, denoiser, wave, rag, sr, fps):
    """The launch sequence of one ragged group on its padded waveform buffer (B, max samples); no host <-> device traffic
    (capturable in a hipGraph).  Returns the padded output (B, hop * (max T - 1)) and the detector's logits / bits."""
    ns, nv = rag.n_samples, rag.n_vframes
    t_ns, t_nv = rag.tab(ns), rag.tab(nv)
    mode = get_precision()            # the denoiser's mode, taken OUTSIDE the detector's precision scope
    S_mixed = transform.stft_batch(wave, clip_samples=t_ns)
    started = []
    logits = detect(detector, S_mixed, max(nv), rag=rag, between=lambda: started.append(_begin_x(denoiser, S_mixed, rag, mode)))
    bits, _ = tools.threshold_bits(logits, SIGMOID_THRESHOLD)
    mask, noise_sig = tools.bits_to_mask_batch(bits, float(sr) / fps, wave.shape[1], wave, clip_frames=t_nv, clip_samples=t_ns)
    S_noise = transform.stft_batch(noise_sig, clip_samples=t_ns)
    n_pred, crm = denoiser(S_mixed, S_noise, rag=rag, **({"started": started[0]} if started and started[0] is not None else {}))
    S_out = transform.batch_fast_icRM_sigmoid(S_mixed, crm)
    return transform.istft_batch(S_out, clip_frames=rag.level(0)), logits, bits


def _pad_group(clips):
    ns = [int(c.numel()) for c in clips]
    wave = torch.zeros((len(clips), max(ns)), dtype=torch.float32, device=clips[0].device)
    for b, c in enumerate(clips):
        wave[b, :ns[b]] = c
    return wave, ns


def _denoise_group(detector, denoiser, clips, sr, fps):
    """One launch sequence for clips of DIFFERENT lengths: buffers sized for the longest clip, every kernel takes the
    clips' own sample / frame / video-frame counts from device tables (engine.Ragged), so each clip is computed exactly
    as if it were run alone: reflect padding of the STFT at its own end, zero / reflect conv borders at its own last
    frame, its own stride-2 sizes and transposed-conv crops, BiLSTM reverse pass from its own last frame, overlap-add
    normalisation of its own frame count."""
    wave, ns = _pad_group(clips)
    rag = _group_geometry(ns, wave.device, sr, fps)
    out, logits, bits = _denoise_group_padded(detector, denoiser, wave, rag, sr, fps)
    return [out[b, :transform.HOP_LENGTH * (rag.T[b] - 1)] for b in range(len(clips))], dict(logits=logits, bits=bits, rag=rag)


def _ragged_groups(clips, max_batch, max_columns):
    """Clips sorted by length (longest first) and cut into groups of <= max_batch clips and <= max_columns
    (clips x frames of the group's longest) spectrogram columns: lists of indices into `clips`."""
    order = sorted(range(len(clips)), key=lambda i: -int(clips[i].numel()))
    groups, j = [], 0
    while j < len(order):
        t_long = 1 + int(clips[order[j]].numel()) // transform.HOP_LENGTH
        nb = max(1, min(max_batch, max_columns // t_long, len(order) - j))
        groups.append(order[j:j + nb])
        j += nb
    return groups


@torch.no_grad()
def denoise_ragged(detector, denoiser, clips, sr=SR, fps=FPS, max_batch=256, max_columns=65536, return_all=False):
    """Variable-length inference (BASELINE configs[3]): `clips` = list of 1-D f32 GPU tensors of ANY lengths.
    The reference denoises one file at a time at its own length (M2/predict.py:377-447: no padding, so reflect
    padding, frame count and video-frame count follow the clip).  Here clips of different lengths share launches:
    they are sorted by length (the BiLSTM kernel steps 16 clips in lockstep, so neighbours should be similar) and cut
    into groups of <= max_batch clips and <= max_columns (clips x frames of the longest) spectrogram columns -- the
    memory budget of a launch sequence, ~0.5 MB of live activations per column -- and every group runs as ONE batch
    with per-clip geometry (_denoise_group).  Outputs come back in input order, each hop*(T-1) samples long."""
    for c in clips:
        if c.dim() != 1:
            raise ValueError("denoise_ragged expects 1-D waveforms")
    outs, extra = [None] * len(clips), [None] * len(clips)
    for part in _ragged_groups(clips, max_batch, max_columns):
        ys, info = _denoise_group(detector, denoiser, [clips[i].contiguous().float() for i in part], sr, fps)
        for k, i in enumerate(part):
            outs[i] = ys[k]
            if return_all:
                extra[i] = dict(logits=info["logits"][k, :info["rag"].n_vframes[k]], bits=info["bits"][k, :info["rag"].n_vframes[k]])
    return (outs, extra) if return_all else outs


class PipelinedDenoiser:
    """Serving loop over CONSECUTIVE batches (M2/predict.py:405-447 denoises file after file): batch i and batch i + 1 alternate
    between two HIP streams, so that the latency-bound tail of batch i -- the denoiser's BiLSTM on 8 workgroups, the FC head, the mask
    apply, the ISTFT -- runs under the chip-filling head of batch i + 1 (STFT, the detector's convolutions).  Each call returns
    (output, event): the output is complete once `event` has been waited for (event.synchronize() on the host, or
    stream.wait_event(event) on a consumer stream); synchronize() drains both streams.  The batches themselves are computed exactly
    as by denoise() (same kernels, same order per batch: bit-identical outputs).
    Lifetimes: `out` is allocated from the worker stream's pool and handed to the caller's stream -- it is recorded on
    the stream that is current at the call (out.record_stream), so its block is not reused while that stream still reads it; a
    consumer on yet another stream must record it there itself.  The first batch builds the caches every later batch shares
    (packed weights, geometry tables, transform tables) on streams[0]: the second batch, the first on streams[1], waits for it."""

    def __init__(self, detector, denoiser, sr=SR, fps=FPS):
        self.detector, self.denoiser, self.sr, self.fps = detector, denoiser, sr, fps
        self.streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        self.k = 0
        self._first = None

    @torch.no_grad()
    def __call__(self, mixed):
        st = self.streams[self.k & 1]
        caller = torch.cuda.current_stream()
        st.wait_stream(caller)                                # the input exists
        if self.k == 1 and self._first is not None:
            st.wait_event(self._first)                        # cold start: batch 0 built the shared caches on the other stream
        with torch.cuda.stream(st):
            out = denoise(self.detector, self.denoiser, mixed, self.sr, self.fps)
            ev = torch.cuda.Event()
            ev.record(st)
        if self.k == 0:
            self._first = ev
        self.k += 1
        mixed.record_stream(st)
        out.record_stream(caller)
        return out, ev

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


class GraphedDenoiser:
    """hipGraph-captured inference chain (BASELINE configs[3]): the whole `denoise` launch sequence (~190 kernels
    for one (batch, length)) is captured once per shape and replayed, so a request costs one graph launch instead
    of ~190 host-side launches -- what bounds small-batch / streaming latency.  Shapes are exact (the reference
    pads nothing: M2/predict.py:377-447), the `max_graphs` most recently used graphs are kept.  Capture happens
    after two eager warm-up runs on a side stream (conv autotuning, weight packing and table uploads must not
    happen inside a capture).  Weights are read at replay time from the buffers packed at capture time, so the
    cache key carries the precision mode and the parameters' version counters (new weights -> new capture)."""

    def __init__(self, detector, denoiser, sr=SR, fps=FPS, max_graphs=16):
        self.detector, self.denoiser, self.sr, self.fps, self.max_graphs = detector, denoiser, sr, fps, max_graphs
        self._graphs = {}

    def reset(self):
        self._graphs.clear()

    def _capture(self, mixed):
        static_in = mixed.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                denoise(self.detector, self.denoiser, static_in, self.sr, self.fps)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = denoise(self.detector, self.denoiser, static_in, self.sr, self.fps)
        return [graph, static_in, static_out, 0]

    @torch.no_grad()
    def __call__(self, mixed, clone=True):
        if mixed.dim() != 2 or not mixed.is_cuda or mixed.dtype != torch.float32:
            raise ValueError("GraphedDenoiser expects a float32 (B, N) GPU tensor")
        # precision mode and weight versions are part of the key: a graph replays the buffers packed at capture time
        from . import get_mode as get_precision
        wv = tuple(t._version for m in (self.detector, self.denoiser) for t in list(m.parameters()) + list(m.buffers()))
        key = (tuple(mixed.shape), mixed.device.index, get_precision(), hash(wv))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                del self._graphs[min(self._graphs, key=lambda k: self._graphs[k][3])]
            entry = self._graphs[key] = self._capture(mixed.contiguous())
        self._tick = getattr(self, "_tick", 0) + 1
        entry[3] = self._tick
        entry[1].copy_(mixed)
        entry[0].replay()
        return entry[2].clone() if clone else entry[2]

    def _capture_mixed(self, wave, rag):
        static_in = wave.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                _denoise_group_padded(self.detector, self.denoiser, static_in, rag, self.sr, self.fps)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = _denoise_group_padded(self.detector, self.denoiser, static_in, rag, self.sr, self.fps)[0]
        return [graph, static_in, static_out, 0, rag]

    @torch.no_grad()
    def denoise_mixed(self, clips, max_batch=256, max_columns=65536):
        """BASELINE configs[3] as stated -- variable-length clips, hipGraph-captured forward: the ragged launch sequence of
        every group (pipeline.denoise_ragged's grouping, per-clip geometry in the kernels) is captured once per LENGTH MIX
        and replayed for new audio of the same lengths (a serving loop with fixed chunk sizes, a benchmark's repeated
        batch).  The per-clip tables are uploaded before the capture and kept alive with the graph."""
        from . import get_mode as get_precision
        for c in clips:
            if c.dim() != 1 or not c.is_cuda or c.dtype != torch.float32:
                raise ValueError("denoise_mixed expects 1-D float32 GPU waveforms")
        wv = tuple(t._version for m in (self.detector, self.denoiser) for t in list(m.parameters()) + list(m.buffers()))
        outs = [None] * len(clips)
        for part in _ragged_groups(clips, max_batch, max_columns):
            wave, ns = _pad_group([clips[i] for i in part])
            key = ("mixed", tuple(ns), wave.device.index, get_precision(), hash(wv))
            entry = self._graphs.get(key)
            if entry is None:
                if len(self._graphs) >= self.max_graphs:
                    del self._graphs[min(self._graphs, key=lambda k: self._graphs[k][3])]
                entry = self._graphs[key] = self._capture_mixed(wave, _group_geometry(ns, wave.device, self.sr, self.fps))
            self._tick = getattr(self, "_tick", 0) + 1
            entry[3] = self._tick
            entry[1].copy_(wave)
            entry[0].replay()
            T = entry[4].T
            for k, i in enumerate(part):
                outs[i] = entry[2][k, :transform.HOP_LENGTH * (T[k] - 1)].clone()
        return outs

    def denoise_ragged(self, clips, max_batch=64):
        """Equal-length buckets of `clips`, each replayed from its graph (streams of fixed-size chunks: the shapes repeat,
        so the graphs are reused; pipeline.denoise_ragged is the path for arbitrary mixed lengths)."""
        order = {}
        for i, c in enumerate(clips):
            if c.dim() != 1:
                raise ValueError("denoise_ragged expects 1-D waveforms")
            order.setdefault(int(c.numel()), []).append(i)
        outs = [None] * len(clips)
        for n, idx in sorted(order.items()):
            for j in range(0, len(idx), max_batch):
                part = idx[j:j + max_batch]
                y = self(torch.stack([clips[i] for i in part]))
                for k, i in enumerate(part):
                    outs[i] = y[k]
        return outs
