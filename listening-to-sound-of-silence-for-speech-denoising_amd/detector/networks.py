"""Mirror of M1/networks.py (model_1_silent_interval_detection/audioonly_model/networks.py):
`get_network()` -> AudioVisualNet with the reference's module tree / state_dict keys
 and `forward(s, v_num_frames=60)` executed by the HIP kernels."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import detector_precision, precision_scope
from .. import common_nets as CN
from .. import engine as E
from .. import train_ops as TO


def get_network(video=False):
    """M1/networks.py:8-9.  video=True: the audio-visual variant (the reference's commented-out video branch, :87-89,
    :135-142, on its live Conv3dBlock / make_video_branch classes)."""
    return AudioVisualNet(video=video)


class _TrainFn(torch.autograd.Function):
    """Training-mode forward (batch-statistics BN, running stats updated in place) with the
    hand-written HIP backward; the parameters are passed so autograd routes their gradients."""

    @staticmethod
    def forward(ctx, net, s, n, v, *params):
        mode = detector_precision()             # 'bf16x3' under set_precision('mixed'); the backward re-enters it
        with precision_scope(mode):
            out, tape = net._forward_train(s, n, v)
        tape["mode"] = mode
        ctx.net, ctx.tape = net, tape
        return out

    @staticmethod
    def backward(ctx, g):
        with E.deferred_reductions():         # weight gradients reduce on a side stream; joined when the scope closes
            grads = ctx.net._backward(ctx.tape, g)
        ctx.tape = None
        if getattr(ctx.net, "grad_sink_factory", None) is not None:
            # data-parallel: the gradients sit in the all-reduce buckets, whose collectives are still running; the
            # bucketer assigns p.grad itself once they are done (GradBucketer.finalize) -- autograd gets nothing to
            # accumulate (no clone racing with the in-place collective)
            return (None, None, None, None) + (None,) * len(list(ctx.net.parameters()))
        return (None, None, None, None) + tuple(grads[name].reshape(p.shape) for name, p in ctx.net.named_parameters())


class AudioVisualNet(nn.Module):
    """M1/networks.py:80-155 (audio-only: the video branch is commented out in the reference)."""

    def __init__(self, freq_bins=256, time_bins=178, nf=96, video=False):
        super().__init__()
        audio_kernel_sizes = [(1, 7), (7, 1)] + [(5, 5)] * 9
        audio_dilations = [(1, 1), (1, 1), (1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (32, 1), (1, 1), (2, 2), (4, 4)]
        self.encoder_audio = CN.make_encoder(audio_kernel_sizes, audio_dilations, nf=48, outf=8)
        self.video_feat = 256 if video else 0
        self.lstm = nn.LSTM(input_size=8 * freq_bins + self.video_feat, hidden_size=100, bidirectional=True)
        self.fc1 = nn.Sequential(nn.Linear(200, 100), nn.ReLU(True), nn.Linear(100, 1))
        if video:       # registered after fc1, like assigning it to the reference module after construction
            self.encoder_video = CN.make_video_branch(CN.VIDEO_KERNEL_SIZES, CN.VIDEO_STRIDES, nf=128, outf=256)
        self.freq_bins = freq_bins
        self._caches = {}            # inference plans, one per precision mode (see _cache)
        self._tcache = E.PlanCache(record=True)

    @property
    def _cache(self):
        """The inference plan cache of the CURRENT precision mode.  One per mode: the two-pass detector of the 'mixed' pipeline
        (pipeline.detect_two_pass) runs this network in fp16 and, for the clips it marks, in bf16x3 within the same call -- with a
        single cache every pass would re-pack all the weights."""
        from .. import get_precision
        c = self._caches.get(get_precision())
        if c is None:
            c = self._caches[get_precision()] = E.PlanCache()
        return c

    def _build_plan(self):
        x3 = E.is_x3()
        return dict(x3=x3,
                    enc=CN.encoder_plan(self.encoder_audio, x3),
                    vid=CN.video_plan(self.encoder_video, x3) if self.video_feat else None,
                    lstm=CN.lstm_plan(self.lstm, 8 * self.freq_bins + self.video_feat, x3),
                    fc0=CN.linear_plan(self.fc1[0], E.pad_to(200, 16), x3),
                    fc2=CN.linear_plan(self.fc1[2], E.pad_to(100, 16), x3),
                    fc2abs=CN.linear_plan(self.fc1[2], E.pad_to(100, 16), x3, absolute=True))

    # ------------------------------------------------------------------ training path
    def _build_train_plan(self):
        x3 = E.is_x3()
        return dict(x3=x3, enc=TO.encoder_train_plan(self.encoder_audio, x3),
                    vid=TO.video_train_plan(self.encoder_video, x3) if self.video_feat else None,
                    lstm=TO.lstm_train_plan(self.lstm, 8 * self.freq_bins + self.video_feat, x3),
                    fc0=TO.linear_train_plan(self.fc1[0], E.pad_to(200, 16), x3),
                    fc2=TO.linear_train_plan(self.fc1[2], E.pad_to(100, 16), x3))

    def _forward_train(self, s, n, v=None):
        plan = self._tcache.get(self, self._build_train_plan)
        x3 = plan["x3"]
        dev = s.device
        B, _, F, T = s.shape
        nseg = 3 if x3 else 1
        nfeat = 8 * F + self.video_feat
        a = CN.pack_encoder_input(plan["enc"], s, x3)
        feat = torch.empty((B, n, nseg * nfeat), dtype=E.act_dtype(), device=dev)
        gather = CN.nearest_index(T, n, dev)
        fspec = dict(t=feat, row=nseg * nfeat, third=nfeat, c_off=0, H=F, W=T, Wo=n, gather=gather, x3=x3)
        tape_enc = TO.encoder_forward_train(plan["enc"], a, fspec, x3)
        tape_vid = None
        if self.video_feat:
            frames = E.pack_input(v.float().permute(0, 2, 1, 3, 4).reshape(B * n, 3, v.shape[3], v.shape[4]), x3)
            tape_vid = TO.video_forward_train(plan["vid"], frames, B, n, feat, nseg * nfeat, nfeat, 8 * F, x3)
        h, tape_lstm = TO.lstm_forward_train(plan["lstm"], (feat, B, 1, n, nfeat, nseg), B, n, x3, dev)
        f0, f2 = plan["fc0"], plan["fc2"]
        m = E.Act(B, 1, n, E.pad_to(f0["cout"], 16), x3, dev, zero=True)
        E.conv_to_act(h, 0, f0["cin_store"], f0["w"], 1, 1, f0["cout"], f0["scale"], f0["shift"], L.ACT_RELU, m,
                      cout_store=m.cs, Ho=1, Wo=n)
        out = torch.empty((B, n), dtype=torch.float32, device=dev)
        E.conv(m, 0, f2["cin_store"], f2["w"], 1, 1, 1, f2["scale"], f2["shift"], L.ACT_NONE, out=out,
               out_dtype=L.DT_F32, sb=n, sh=0, sw=1, sc=1, Ho=1, Wo=n)
        tape = dict(plan=plan, enc=tape_enc, vid=tape_vid, lstm=tape_lstm, h=h, m=m, gather=gather, dims=(B, F, T, n), x3=x3)
        return out, tape

    def _backward(self, tape, g):
        g = g.contiguous().float()
        E.check_tape_weights(self, tape)
        with precision_scope(tape.get("mode")), E.backward_scale(g, guard=E.guard_state(self)):   # fp16 mode: loss scale for this pass (a no-op in the bf16 modes)
            return self._backward_scaled(tape, g)

    def _backward_scaled(self, tape, g):
        plan, x3 = tape["plan"], tape["x3"]
        B, F, T, n = tape["dims"]
        dev = g.device
        factory = getattr(self, "grad_sink_factory", None)
        grads = factory() if factory is not None else {}
        dz2 = E.Act(B, 1, n, 16, x3, dev, zero=True)
        TO.pack_grad(g, None, L.ACT_NONE, B, n, 1, n, 1, 1, dz2)
        d_m = TO.linear_backward(plan["fc2"], tape["m"], dz2, grads, "fc1.2", x3, dev)
        dz0 = E.Act(B, 1, n, tape["m"].cs, x3, dev, zero=True)
        TO.act_bwd_from_y(d_m, tape["m"], L.ACT_RELU, dz0, plan["fc0"]["cout"])
        dh = TO.linear_backward(plan["fc0"], tape["h"], dz0, grads, "fc1.0", x3, dev)
        dfeat = TO.lstm_backward(plan["lstm"], tape["lstm"], dh, grads, "lstm", B, n, x3, dev)
        lo, hi = CN.nearest_ranges(T, n, dev)           # cached device tables: no host sync in the backward pass
        nseg = 3 if x3 else 1
        nfeat = 8 * F + self.video_feat
        dy = TO.feat_grad_to_nhwc(dfeat, nseg * nfeat, nfeat, 0, 8, B, F, T, n, x3,
                                  lo, hi)
        TO.encoder_backward(plan["enc"], tape["enc"], dy, grads, "encoder_audio", x3)
        if self.video_feat:
            TO.video_backward(plan["vid"], tape["vid"], dfeat, nseg * nfeat, nfeat, 8 * F, grads, "encoder_video", B, n, x3)
        return grads

    def forward(self, s, v_num_frames=60, v=None, rag=None, before_lstm=None, return_scale=False):
        """s (B,2,F,T) -> logits (B, v_num_frames).  Audio-visual variant: v (B,3,Tv,H,W) video frames; the audio
        features are resized to Tv frames (M1/networks.py:138) and v_num_frames is ignored.
        rag (eval only): engine.Ragged with the clips' own STFT frame counts (rag.T) and video-frame counts
        (rag.n_vframes) of a variable-length batch; s is then (B,2,F,max T), the logits (B, max n) with clip b's first
        rag.n_vframes[b] entries valid -- each clip computed exactly as if it were run alone at its own length."""
        L.require_cuda(s, v)
        if s.dim() != 4 or s.shape[1] != 2 or s.shape[2] != self.freq_bins:
            raise ValueError(f"expected (B, 2, {self.freq_bins}, T) input, got {tuple(s.shape)}")
        if self.video_feat:
            if v is None or v.dim() != 5 or v.shape[0] != s.shape[0] or v.shape[1] != 3:
                raise ValueError("the audio-visual variant needs video frames v of shape (B, 3, Tv, H, W)")
            v_num_frames = v.shape[2]
        elif v is not None:
            raise ValueError("this network was built without the video branch (get_network(video=True))")
        if rag is not None and (self.training or self.video_feat):
            raise ValueError("ragged batches are an inference feature of the audio-only network")
        if return_scale and self.training:
            raise ValueError("return_scale is an inference feature")
        if self.training:
            return _TrainFn.apply(self, s.contiguous().float(), int(v_num_frames),
                                  v.contiguous().float() if v is not None else None, *self.parameters())
        # before_lstm (eval only): called once the encoder's convolutions are enqueued -- what follows (the recurrence over the
        # frames on 8 workgroups, the FC head) leaves the chip mostly idle, the pipeline starts the denoiser's encoder_x there
        with precision_scope(detector_precision()):
            return self._forward_eval(s, int(v_num_frames), v, rag, before_lstm, return_scale)

    def _forward_eval(self, s, n, v, rag, before_lstm=None, return_scale=False):
        plan = self._cache.get(self, self._build_plan)
        x3 = plan["x3"]
        dev = s.device
        B, _, F, T = s.shape
        a = CN.pack_encoder_input(plan["enc"], s, x3, rag)
        nseg = 3 if x3 else 1
        nfeat = 8 * F + self.video_feat
        lengths = None
        if rag is None:
            feat = torch.empty((B, n, nseg * nfeat), dtype=E.act_dtype(), device=dev)
            gather = CN.nearest_index(T, n, dev)
            CN.run_encoder(plan["enc"], a, feat, nseg * nfeat, nfeat, 0, x3, w_gather=gather, T_out=n)
        else:
            if len(rag.T) != B or max(rag.T) != T or max(rag.n_vframes) != n:
                raise ValueError("rag does not describe this batch")
            feat = torch.zeros((B, n, nseg * nfeat), dtype=E.act_dtype(), device=dev)      # padding rows stay finite
            gather = CN.nearest_index_ragged(rag, n, dev)                                  # (B, n): per-clip T_c -> n_c
            lengths = rag.tab(rag.n_vframes)
            CN.run_encoder(plan["enc"], a, feat, nseg * nfeat, nfeat, 0, x3, w_gather=gather, T_out=n, rag=rag,
                           rag_out=(lengths, n))
        if self.video_feat:
            # frames as a batch of B*Tv images; the video features land next to the audio ones (channel concat)
            frames = E.pack_input(v.float().permute(0, 2, 1, 3, 4).reshape(B * n, 3, v.shape[3], v.shape[4]), x3)
            CN.run_video_branch(plan["vid"], frames, B, n, feat, nseg * nfeat, nfeat, 8 * F, x3)
        if before_lstm is not None:
            before_lstm()
        h = CN.run_lstm(plan["lstm"], (feat, B, 1, n, nfeat, nseg), B, n, x3, dev, lengths=lengths)
        f0, f2 = plan["fc0"], plan["fc2"]
        m = E.Act(B, 1, n, E.pad_to(f0["cout"], 16), x3, dev)
        E.conv_to_act(h, 0, f0["cin_store"], f0["w"], 1, 1, f0["cout"], f0["scale"], f0["shift"], L.ACT_RELU, m,
                      cout_store=m.cs, Ho=1, Wo=n)
        out = torch.empty((B, n), dtype=torch.float32, device=dev)
        E.conv(m, 0, f2["cin_store"], f2["w"], 1, 1, 1, f2["scale"], f2["shift"], L.ACT_NONE, out=out,
               out_dtype=L.DT_F32, sb=n, sh=0, sw=1, sc=1, Ho=1, Wo=n)
        if not return_scale:
            return out
        fa = plan["fc2abs"]
        scale = torch.empty((B, n), dtype=torch.float32, device=dev)
        E.conv(m, 0, fa["cin_store"], fa["w"], 1, 1, 1, fa["scale"], fa["shift"], L.ACT_NONE, out=scale,
               out_dtype=L.DT_F32, sb=n, sh=0, sw=1, sc=1, Ho=1, Wo=n)
        return out, scale
