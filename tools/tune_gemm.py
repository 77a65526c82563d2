#!/usr/bin/env python3
"""Time every tile instantiation of mt4_conv_nhwc in GEMM mode on the Swin-B linear-layer shapes (GPU box only).
usage: python tools/tune_gemm.py [--batch 128] [--img 384]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--img", type=int, default=384); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
ntiles = _lib.lib.mt4_conv_tile_count()
res = a.img // 4
for s, c in enumerate((128, 256, 512, 1024)):
    m = a.batch * (res >> s) ** 2
    for name, n, k, act, has_res in (("qkv", 3 * c, c, None, False), ("proj", c, c, None, True), ("fc1", 4 * c, c, "gelu", False), ("fc2", c, 4 * c, None, True)):
        x = torch.randn(m, 1, 1, k, device=dev).to(torch.bfloat16)
        w = ops.pack_conv_weight(torch.randn(n, k, 1, 1, device=dev) * 0.05, None, torch.bfloat16)
        bias = torch.zeros(n, device=dev)
        r = torch.randn(m, 1, 1, n, device=dev).to(torch.bfloat16) if has_res else None
        line = []
        for t in range(ntiles + 1):
            ts = []
            try:
                for it in range(a.iters + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); ops.conv_nhwc(x, w, bias, kh=1, kw=1, residual=r, act=act, tile=t); e1.record(); torch.cuda.synchronize()
                    if it: ts.append(e0.elapsed_time(e1))
                line.append(min(ts))
            except Exception:
                line.append(None)
        auto = line[0]
        best = min((v, i) for i, v in enumerate(line) if i and v is not None)
        print(f"stage{s + 1} {name:5s} M{m:8d} N{n:5d} K{k:5d} auto {auto:.4f} best t{best[1]} {best[0]:.4f} ({100 * (auto - best[0]) / best[0]:+.1f}%)  " +
              " ".join(f"t{i}={v:.3f}" for i, v in enumerate(line) if i and v is not None and v < 1.15 * best[0]), flush=True)
        del x, w, r
