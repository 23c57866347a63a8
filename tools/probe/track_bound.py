#!/usr/bin/env python3
"""The quantity tests/test_gpu_agent.py::test_fp16_training_tracks_the_parity_mode bounds (relative difference of the summed
denoiser loss, fp16 vs bf16x3, 25 steps from the same weights on the same batches), printed so that the bound can be set at 1.5x
the worst value over the shipped and two forced tilings (VERDICT r4 #2c): run under SOS_CONV_FORCE_CFG=3 / 7 as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sos_amd  # noqa: E402
from sos_amd import agent  # noqa: E402
from sos_amd.common import MyConfig  # noqa: E402
from sos_amd.dataset import make_batch  # noqa: E402
from sos_amd.denoiser import networks as jnet  # noqa: E402

curves = {}
for precision in ("bf16x3", "fp16"):
    sos_amd.set_precision(precision)
    torch.manual_seed(0)
    ag = agent.DenoiserAgent(jnet.get_network(MyConfig()), lr=1e-3)
    c = []
    for it in range(25):
        _, ls = ag.train_func(make_batch("denoiser", 7000 + 8 * it, 8))
        c.append(float(ls["stage1"].detach()) + float(ls["stage2"].detach()))
    curves[precision] = np.array(c)
    del ag
rel = np.abs(curves["fp16"] - curves["bf16x3"]) / curves["bf16x3"]
print("FORCE_CFG", os.environ.get("SOS_CONV_FORCE_CFG"), "rel[:5].max", float(rel[:5].max()), "rel.max", float(rel.max()),
      "per step", np.round(rel, 4).tolist())
