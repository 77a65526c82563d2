#!/usr/bin/env python3
"""Temporal_tenco latency (GPU box): latency path (csrc/tcn_kernels.hip) against the implicit-GEMM path, hipGraph replay and eager,
fp32 and bf16, plus the per-launch device time of single layers.
  python tools/tcn_bench.py [--T 256 ...] [--layers]"""
import argparse, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops, shapes, synth
from computervision_codes_amd.graph import GraphedForward
from computervision_codes_amd.temporal_tenco import VideoNas

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, nargs="+", default=[256])
ap.add_argument("--layers", action="store_true", help="per-launch timing of single convs (graph of 20 back-to-back launches)")
a = ap.parse_args()
dev = torch.device("cuda:0")


def med(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for T in a.T:
    for name, num_R, dim in (("tenco4", 3, 512), ("config1", 0, 2048)):
        fpn = num_R > 0
        args = types.SimpleNamespace(fpn=fpn, output=False, hier=False, mask=True)
        sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, num_R, 512, dim, 100, fpn=fpn), seed=47)
        x = synth.synthetic_features(T, dim, seed=47).to(dev)
        for dt in (torch.float32, torch.bfloat16):
            row = []
            for path in ("tcn", "igemm"):
                m = VideoNas(args, 11, 10, num_R, 512, dim, 100, dtype=dt, path=path).eval().load_state_dict(sd)
                eager = med(lambda: m(x, False))
                g = GraphedForward(lambda xx: m(xx, False), [x])
                row.append(f"{path}: graph {med(lambda: g(x)):.4f} ms eager {eager:.4f} ms")
            print(f"{name} T={T} {str(dt).split('.')[-1]}: " + " | ".join(row), flush=True)

if a.layers:
    T = a.T[0]
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(1, T, 512, device=dev).to(dt)
        w3 = ops.pack_conv_weight(torch.randn(512, 512, 1, 3, device=dev) / 40, None, dt)
        w1 = ops.pack_conv_weight(torch.randn(512, 512, 1, 1, device=dev) / 22, None, dt)
        bias = torch.zeros(512, device=dev)
        for d in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            def chain():
                for _ in range(20):
                    ops.tcn_conv(x, w3, bias, taps=3, dilation=d, relu=True)
            g = GraphedForward(lambda xx: chain(), [x])
            t3 = med(lambda: g(x)) / 20 * 1e3
            print(f"{str(dt).split('.')[-1]} T={T} k3 d={d}: {t3:.2f} us per launch (20 back-to-back in a graph)", flush=True)

        def chain1():
            for _ in range(20):
                ops.tcn_conv(x, w1, bias, taps=1, residual=x)
        g = GraphedForward(lambda xx: chain1(), [x])
        print(f"{str(dt).split('.')[-1]} T={T} 1x1+res: {med(lambda: g(x)) / 20 * 1e3:.2f} us per launch", flush=True)
