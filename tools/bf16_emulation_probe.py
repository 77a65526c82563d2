#!/usr/bin/env python3
"""bf16 extraction mode against (a) the fp32 reference goldens and (b) the oracle run with the mode's roundings emulated
(`oracle.spatial_cnn.resnet_trunk_bf16_emulation`), per output, as fractions of the output's range.
  python tools/bf16_emulation_probe.py > profiles/r04_bf16_emulation_probe.txt"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from computervision_codes_amd import shapes, synth  # noqa: E402
from computervision_codes_amd.spatial_cnn import VideoNas  # noqa: E402
from oracle import spatial_cnn as o_cnn  # noqa: E402

print("fixture               output      hip-vs-fp32-reference   hip-vs-emulation   emulation-vs-fp32-reference   (max abs / range)")
for name in ("cnn_resnet50_small", "cnn_resnet50_224", "cnn_resnet18_224", "cnn_resnet50_256x448", "cnn_resnet18_odd"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cfg = ast.literal_eval(str(z["cfg"]))
    args = types.SimpleNamespace(network=cfg["network"], loss_type="all", student_dim=shapes.resnet_feat_dim(cfg["network"]), teacher_dim=1536, train=False)
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes(cfg["network"]), seed=cfg["seed"])
    m = VideoNas(args=args, dtype=torch.bfloat16).eval().load_state_dict(sd)
    frames = synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"])
    with torch.no_grad():
        emu = o_cnn.spatial_cnn_forward(sd, synth.normalize_frames(frames), cfg["network"], emulate_bf16=True)
    out = m.extract_u8(frames.cuda())
    for got, want, key in ((out[0][1], emu[0][1], "logit_i"), (out[1][1], emu[1][1], "logit_v"), (out[2][1], emu[2][1], "logit_t"),
                           (out[3][1], emu[3][1], "logit_ivt"), (out[3][0], emu[3][0], "feat")):
        ref = torch.from_numpy(z[key])
        rng = float(ref.abs().max())
        g = got.float().cpu()
        print(f"{name:21s} {key:10s}  {float((g - ref).abs().max()) / rng:21.2e}  {float((g - want).abs().max()) / rng:17.2e}  {float((want - ref).abs().max()) / rng:28.2e}")
