#!/usr/bin/env python3
"""Temporal_tenco on whole videos (T = 1000 ... 4000, the reference's unit: `Temporal_tenco/run.py:369-379` runs batch 1 on full videos, CholecT45
averages ~2000 frames): the implicit-GEMM path with every generic tile id forced (`VideoNas.tile`), fp32 and bf16, hipGraph replay, against
the automatic choice and the latency path.  Same box, one process.
  python tools/tcn_long_sweep.py [--T 2000 ...] [--tiles 1 2 3 ...]"""
import argparse, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.graph import GraphedForward
from computervision_codes_amd.temporal_tenco import VideoNas

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, nargs="+", default=[2000])
ap.add_argument("--tiles", type=int, nargs="+", default=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 35, 36, 37])
a = ap.parse_args()
dev = torch.device("cuda:0")


def med(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


args = types.SimpleNamespace(fpn=True, output=False, hier=False, mask=True)
sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, 3, 512, 512, 100, fpn=True), seed=47)
for T in a.T:
    x = synth.synthetic_features(T, 512, seed=47).to(dev)
    for dt in (torch.float32, torch.bfloat16):
        name = str(dt).split(".")[-1]
        ref = None
        for path, tile in [("igemm", 0), ("tcn", 0)] + [("igemm", t) for t in a.tiles]:
            m = VideoNas(args, 11, 10, 3, 512, 512, 100, dtype=dt, path=path).eval().load_state_dict(sd)
            m.tile = tile
            try:
                g = GraphedForward(lambda xx: m(xx, False), [x])
                ms = med(lambda: g(x))
                y = g(x)[0][0].float()
                if ref is None:
                    ref = y.clone()
                err = (y - ref).abs().max().item()
                print(f"T={T} {name} {path} tile {tile:2d}: {ms:.4f} ms   max|dlogit| vs auto {err:.2e}", flush=True)
            except Exception as e:
                print(f"T={T} {name} {path} tile {tile:2d}: {type(e).__name__} {str(e)[:80]}", flush=True)
