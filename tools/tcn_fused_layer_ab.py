#!/usr/bin/env python3
"""Throughput mode of the temporal head (4 stages, D = 512, bf16, T = 256 frames per video, hipGraph replay): every DilatedResidualLayer as ONE launch
(`mt4_tcn_layer_fused_bf16`, hidden map in LDS) against the two launches per layer, by the number of videos per forward.
  python tools/tcn_fused_layer_ab.py > profiles/r04_tcn_fused_layer_ab.txt"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import shapes, synth  # noqa: E402
from computervision_codes_amd.graph import GraphedForward  # noqa: E402
from computervision_codes_amd.temporal_tenco import VideoNas  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


args = types.SimpleNamespace(fpn=True, output=False, hier=False, mask=True)
sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, 3, 512, 512, 100, fpn=True), seed=47)
m = VideoNas(args, 11, 10, 3, 512, 512, 100, dtype=torch.bfloat16).eval().load_state_dict(sd)
print("videos  T   tiles(64 frames)   two launches per layer: ms / videos/s     one launch per layer: ms / videos/s")
for B, T in ((8, 256), (16, 256), (24, 256), (32, 256), (48, 256), (64, 256), (128, 256), (256, 256), (8, 2000), (32, 2000)):
    x = torch.stack([synth.synthetic_features(T, 512, seed=47 + i)[0] for i in range(min(B, 8))]).repeat((B + 7) // 8, 1, 1)[:B].cuda().contiguous()
    row = []
    for gate in (10 ** 9, 0):
        m.fused_layer_min_tiles, m.fused_layer_max_tiles = gate, 10 ** 9
        g = GraphedForward(lambda xx: m(xx, False), [x])
        row.append(timeit(lambda: g(x)))
        del g
    print(f"{B:6d} {T:5d} {B * ((T + 63) // 64):8d}            {row[0]:8.3f} / {B / row[0] * 1e3:9.0f}                     {row[1]:8.3f} / {B / row[1] * 1e3:9.0f}", flush=True)
