#!/usr/bin/env python3
"""Swin-B + 4 Q2L decoders (BASELINE configs[2]), bf16: frames/s by frames per forward.  The GEMMs run 256 x 256 tiles on 256 CUs; stage 2 has
576 B rows (384^2) / 196 B (224^2), stage 3 and the decoders 144 B / 49 B, and proj / fc2 / linear2 have 2 - 4 column tiles: at B = 128 those launches
are 2.25 or 1.125 rounds of the chip.  python tools/swin_batch_sweep.py > profiles/r04_swin_batch_sweep.txt"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_transformer import build_q2l
dev = torch.device("cuda")
for name, img, batches in (("swin_B_384_22k", 384, (96, 112, 113, 114, 128, 170, 226, 227)), ("swin_B_224_22k", 224, (128, 256, 332, 334, 336, 512, 668))):
    args = types.SimpleNamespace(backbone=name, img_size=img, hidden_dim=1024, loss_type="all")
    m = build_q2l(args, dtype=torch.bfloat16, device="cuda").eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.q2l_param_shapes(name, img, 1024, "all"), seed=7))
    for b in batches:
        frames = bench.device_frames(b, img, img, 7, dev)
        tf = [synth.synthetic_features(b, 512, seed=7 + k)[0].to(dev) for k in (1, 2, 3)]
        ms = bench._time_call(lambda: m(frames, *tf), iters=5)
        tok2 = b * (img // 16) ** 2
        print(f"{name} frames {b:4d}: {ms:8.3f} ms  {b / ms * 1e3:8.1f} frames/s   stage-2 row tiles {tok2 / 256:7.2f}, stage-3 / decoder row tiles {tok2 / 4 / 256:6.2f}", flush=True)
        del frames, tf
    del m
