#!/usr/bin/env python3
"""Regenerates the shipped conv tiling table (sos_amd/tune_table_gfx950.txt) on an MI355X: runs the BASELINE workloads
once with timing-based autotuning on, so that every launch shape they contain gets its measured-best tiling, and writes
the table to gpurun_out/tune_table_gfx950.txt (copy it into the package and commit it).

    gpurun -- python tools/make_tune_table.py
    gpurun -- python tools/make_tune_table.py --extend-av      (adds the audio-visual variant's shapes to the committed table)

Every later process loads the committed table and never times anything: all processes (and all ranks of a
data-parallel job) then run identical tilings, i.e. identical summation orders (engine.py, conv.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "tune_table_gfx950.txt")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
os.environ["SOS_CONV_TUNE"] = "1"
os.environ["SOS_CONV_TUNE_CACHE"] = OUT
os.environ["SOS_CONV_TUNE_TABLE"] = "0"        # only the table being built (OUT), never the shipped one underneath
os.environ["SOS_FOLD_FUSED"] = "0"             # the U-Net's folded data-gradient launches are never timed (their output is live): tune
                                               # the unfused launches of the SAME shape keys instead
EXTEND_AV = "--extend-av" in sys.argv        # keep the committed table, add the audio-visual variant's shapes
RETUNE_X3 = "--retune-x3" in sys.argv        # keep the committed table, re-measure the three-segment shapes the 16-row kernel now takes
RETUNE_NPOT = "--retune-npot" in sys.argv    # keep the committed table, re-measure the shapes that gained non-power-of-two tile candidates
RETUNE_16ROW = "--retune-16row" in sys.argv  # keep the committed table, re-measure the 16- / 48-channel-input shapes (round 4: the 16-row
                                             # kernel's 512-pixel workgroups; the first layers with their horizontal taps on the channel axis)
RETUNE_WIDE = "--retune-wide" in sys.argv    # keep the committed table, re-measure the B = 64 single-segment shapes (the bench's training / fp16
                                             # inference step) over up to 400 candidates with sos_conv2d_tune's two-stage decision (round 4)
RETUNE_WIDE_ALL = "--retune-wide-all" in sys.argv   # every entry of the committed table re-measured that way (all batch sizes, the
                                             # three-segment detector, the audio-visual variant)
RETUNE_WGRAD = "--retune-wgrad" in sys.argv  # measure the weight-gradient plans of the B = 64 training step (sos_wgrad_tune, round 4) ->
                                             # gpurun_out/wgrad_table_gfx950.txt; the conv table is loaded as shipped and left alone
RETUNE_PT3 = "--retune-pt3" in sys.argv      # keep the committed table, re-measure the B = 64 96 -> 96 stride-1 shapes, which gained the
                                             # 384-slot tile candidates (round 6: conv_mfma_kernel<3, ks, false, false, 3>)
OUTW = os.path.join(ROOT, "gpurun_out", "wgrad_table_gfx950.txt")
if RETUNE_WGRAD:
    os.environ["SOS_CONV_TUNE_TABLE"] = "1"
    os.environ.pop("SOS_CONV_TUNE_CACHE", None)
    os.environ.pop("SOS_FOLD_FUSED", None)
    os.environ["SOS_WGRAD_TUNE_TABLE"] = "0"
    os.environ["SOS_WGRAD_TUNE_CACHE"] = OUTW
    os.environ.setdefault("SOS_CONV_TUNE_VERBOSE", "1")
    if os.path.exists(OUTW):
        os.remove(OUTW)
SHIPPED = os.path.join(ROOT, "listening-to-sound-of-silence-for-speech-denoising_amd", "tune_table_gfx950.txt")
if os.path.exists(OUT) and not RETUNE_WGRAD:
    os.remove(OUT)
if EXTEND_AV:
    import shutil
    shutil.copy(SHIPPED, OUT)
    shutil.copy(SHIPPED, OUT + ".f16")       # the fp16 build reads / writes its own cache file (engine._load_tune_cache)
if RETUNE_X3:
    # shape key columns (conv.hip, shape_key): 4 cin, 5 segments, 14 out dtype (1 = hi|hi|lo), 15 dense NHWC, 18 cout
    lines = open(SHIPPED).read().splitlines()
    keep = [lines[0]]
    for ln in lines[1:]:
        v = ln.split()
        elig = len(v) >= 23 and v[5] == "3" and v[4] in ("16", "48") and v[14] == "1" and v[15] == "1" and (int(v[18]) <= 16 or 32 < int(v[18]) <= 48)
        if not elig:
            keep.append(ln)
    open(OUT, "w").write("\n".join(keep) + "\n")
    print("dropped", len(lines) - len(keep), "entries to re-measure")

if RETUNE_16ROW:
    os.environ.setdefault("SOS_CONV_TUNE_CANDIDATES", "48")       # the 512-pixel candidates sit anywhere in the cost-ordered list
    lines = open(SHIPPED).read().splitlines()
    keep = [lines[0]] + [ln for ln in lines[1:] if ln.split()[4] not in ("16", "48")]      # column 4: cin (conv.hip, shape_key)
    for path in (OUT, OUT + ".f16"):
        open(path, "w").write("\n".join(keep) + "\n")
    print("dropped", len(lines) - len(keep), "entries to re-measure")

if RETUNE_WIDE_ALL:
    os.environ.setdefault("SOS_CONV_TUNE_CANDIDATES", "400")
    os.environ.setdefault("SOS_CONV_TUNE_VERBOSE", "1")

if RETUNE_WIDE:
    os.environ.setdefault("SOS_CONV_TUNE_CANDIDATES", "400")
    os.environ.setdefault("SOS_CONV_TUNE_VERBOSE", "1")
    lines = open(SHIPPED).read().splitlines()
    keep = [lines[0]] + [ln for ln in lines[1:] if not (ln.split()[0] == "64" and ln.split()[5] == "1")]   # columns 0: B, 5: segments
    for path in (OUT, OUT + ".f16"):
        open(path, "w").write("\n".join(keep) + "\n")
    print("dropped", len(lines) - len(keep), "entries to re-measure")

if RETUNE_PT3:
    os.environ.setdefault("SOS_CONV_TUNE_CANDIDATES", "400")
    os.environ.setdefault("SOS_CONV_TUNE_VERBOSE", "1")
    lines = open(SHIPPED).read().splitlines()
    # columns (conv.hip, shape_key): 0 B, 4 cin, 5 segments, 6 cout_pad, 9 stride, 14 out dtype
    keep = [lines[0]] + [ln for ln in lines[1:] if not (ln.split()[0] == "64" and ln.split()[4] == "96" and ln.split()[5] == "1" and
                                                         ln.split()[6] == "96" and ln.split()[9] == "1" and ln.split()[14] == "0")]
    for path in (OUT, OUT + ".f16"):
        open(path, "w").write("\n".join(keep) + "\n")
    print("dropped", len(lines) - len(keep), "entries to re-measure")

if RETUNE_NPOT:
    # shape key columns (conv.hip, shape_key): 9 stride, 10 dil_h, 11 dil_w, 12 Ho, 13 Wo.  enumerate_cfgs adds the non-power-of-two
    # tiles for stride-1 shapes whose strided extent per residue class is <= 48 in either direction
    lines = open(SHIPPED).read().splitlines()
    keep = [lines[0]]
    for ln in lines[1:]:
        v = [int(x) for x in ln.split()]
        hc, wc = -(-v[12] // v[10]), -(-v[13] // v[11])
        if not (v[9] == 1 and (hc <= 48 or wc <= 48) and v[12] > 1):
            keep.append(ln)
    for path in (OUT, OUT + ".f16"):
        open(path, "w").write("\n".join(keep) + "\n")
    print("dropped", len(lines) - len(keep), "entries to re-measure")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sos_amd  # noqa: E402
from sos_amd import agent, pipeline, tools, transform  # noqa: E402
from sos_amd.common import MyConfig  # noqa: E402
from sos_amd.dataset import synth_batch  # noqa: E402
from sos_amd.denoiser import networks as jnet  # noqa: E402
from sos_amd.detector import networks as dnet  # noqa: E402


def workloads(B):
    torch.manual_seed(0)
    det = dnet.get_network().cuda()
    jm = jnet.get_network(MyConfig()).cuda()
    raw = synth_batch(0, min(B, 8))
    rep = (B + len(raw["mixed"]) - 1) // len(raw["mixed"])
    tile = lambda a: torch.from_numpy(np.tile(a, (rep, 1))[:B]).cuda().contiguous()   # noqa: E731
    mixed, clean, full_noise, bits = tile(raw["mixed"]), tile(raw["clean"]), tile(raw["full_noise"]), tile(raw["bits"])
    mask, noise_sig = tools.bits_to_mask_batch(bits, 14000 / 30.0, 28000, mixed)
    S = transform.stft_batch(torch.cat([mixed, clean * (1 - mask), noise_sig, full_noise]))
    bj = {"mixed": S[:B].contiguous(), "clean": S[B:2 * B].contiguous(), "noise": S[2 * B:3 * B].contiguous(),
          "full_noise": S[3 * B:].contiguous()}
    bd = {"audio": bj["mixed"], "label": bits.float()}
    pipeline.denoise(det.eval(), jm.eval(), mixed)                       # inference shapes
    agent.DetectorAgent(det.train(), lr=1e-3).train_func(bd)             # training shapes (forward, dgrad)
    agent.DenoiserAgent(jm.train(), lr=1e-3).train_func(bj)
    torch.cuda.synchronize()


def av_workloads():
    """Audio-visual variant (BASELINE configs[4] geometry): inference at 1 / 4 / 16 clips, a training step at 8 / 32."""
    torch.manual_seed(0)
    net = dnet.get_network(video=True).cuda().eval()
    for B in (1, 4, 16):
        with torch.no_grad():
            net(torch.randn(B, 2, 256, 178, device="cuda"), v=torch.rand(B, 3, 60, 224, 224, device="cuda"))
    for B in (8, 32):
        ag = agent.DetectorAgent(dnet.get_network(video=True), lr=1e-3)
        ag.train_func({"audio": torch.randn(B, 2, 256, 178, device="cuda"), "frames": torch.rand(B, 3, 60, 224, 224, device="cuda"),
                       "label": (torch.rand(B, 60, device="cuda") > 0.3).float()})
        del ag
    torch.cuda.synchronize()


def x3_detector_workloads():
    """The detector in the three-pass mode (what 'mixed' runs): inference at the table's batch sizes + a training step."""
    for B in (64, 32, 16, 8, 4, 2, 1):
        torch.manual_seed(0)
        det = dnet.get_network().cuda().eval()
        S = torch.randn(B, 2, 256, 178, device="cuda")
        with torch.no_grad():
            det(S, 60)
        if B in (64, 2, 1):
            agent.DetectorAgent(det.train(), lr=1e-3).train_func({"audio": S, "label": (torch.rand(B, 60, device="cuda") > 0.3).float()})
        torch.cuda.synchronize()
        print("re-tuned bf16x3 detector, B =", B, flush=True)


def main():
    if RETUNE_X3:
        sos_amd.set_precision("bf16x3")
        x3_detector_workloads()
        from sos_amd import _lib
        _lib.lib().sos_conv2d_tune_save(OUT.encode())
        print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")
        return
    if RETUNE_16ROW:
        for prec, batches in (("bf16", (64, 32, 16, 8, 4, 2, 1)), ("bf16x3", (64, 2, 1))):
            sos_amd.set_precision(prec)
            for B in batches:
                workloads(B)
                print("re-tuned", prec, "B =", B, flush=True)
        sos_amd.set_precision("bf16x3")
        x3_detector_workloads()
        from sos_amd import _lib
        sos_amd.set_precision("bf16")
        _lib.lib().sos_conv2d_tune_save(OUT.encode())
        print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")
        return
    if RETUNE_WGRAD:
        sos_amd.set_precision("bf16")
        workloads(64)
        from sos_amd import _lib
        _lib.lib().sos_wgrad_tune_save(OUTW.encode())
        print("wrote", OUTW, sum(1 for _ in open(OUTW)) - 1, "entries")
        return
    if RETUNE_WIDE or RETUNE_PT3:
        sos_amd.set_precision("bf16")
        workloads(64)
        from sos_amd import _lib
        _lib.lib().sos_conv2d_tune_save(OUT.encode())
        print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")
        return
    if RETUNE_NPOT or RETUNE_WIDE_ALL:
        for prec, batches in (("bf16", (64, 32, 16, 8, 4, 2, 1)), ("bf16x3", (64, 2, 1))):
            sos_amd.set_precision(prec)
            for B in batches:
                workloads(B)
                print("re-tuned", prec, "B =", B, flush=True)
        sos_amd.set_precision("bf16x3")
        x3_detector_workloads()
        sos_amd.set_precision("fp16")
        av_workloads()
        from sos_amd import _lib
        sos_amd.set_precision("bf16")
        _lib.lib().sos_conv2d_tune_save(OUT.encode())
        print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")
        return
    if EXTEND_AV:
        sos_amd.set_precision("fp16")
        av_workloads()
        from sos_amd import _lib
        _lib.lib().sos_conv2d_tune_save(OUT.encode())
        print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")
        return
    # the table is shared by both builds (same kernels, same shapes): tune on the bf16 one; bf16x3 has its own
    # (three-segment) shapes
    for prec, batches in (("bf16", (64, 32, 16, 8, 4, 2, 1)), ("bf16x3", (64, 2, 1))):
        sos_amd.set_precision(prec)
        for B in batches:
            workloads(B)
            print("tuned", prec, "B =", B, flush=True)
    from sos_amd import _lib
    sos_amd.set_precision("bf16")
    _lib.lib().sos_conv2d_tune_save(OUT.encode())
    print("wrote", OUT, sum(1 for _ in open(OUT)) - 1, "entries")


if __name__ == "__main__":
    main()
