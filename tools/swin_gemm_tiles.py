#!/usr/bin/env python3
"""Every large tile of the generic kernel on the GEMM shapes of a Swin-B/384 forward at batch 128 whose 256 x 256 tile count is a poor multiple of the
256 CUs (proj / fc2: N = C gives 2.25 rounds at stage 2), with residual ("-": without) (GPU box): python tools/swin_gemm_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
dev = torch.device("cuda:0")
SHAPES = [("s0 qkv -", 128 * 96 * 96, 128, 384), ("s1 qkv -", 128 * 48 * 48, 256, 768), ("s2 qkv -", 128 * 24 * 24, 512, 1536), ("s2 fc1 -", 128 * 24 * 24, 512, 2048),
          ("s0 proj", 128 * 96 * 96, 128, 128), ("s1 proj", 128 * 48 * 48, 256, 256), ("s2 qkv", 128 * 24 * 24, 512, 1536), ("s2 proj", 128 * 24 * 24, 512, 512),
          ("s2 fc1", 128 * 24 * 24, 512, 2048), ("s2 fc2", 128 * 24 * 24, 2048, 512), ("s3 qkv", 128 * 12 * 12, 1024, 3072), ("s3 proj", 128 * 12 * 12, 1024, 1024),
          ("s3 fc1", 128 * 12 * 12, 1024, 4096), ("s3 fc2", 128 * 12 * 12, 4096, 1024)]
TILES = [0, 17, 15, 13, 18, 19, 14, 16, 1, 7, 20, 2, 12]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print("shape                M      K     N   256x256 tiles / 256 |  us per launch by tile id: " + " ".join(f"{t:6d}" for t in TILES))
for name, m, k, n in SHAPES:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05)
    wp = ops.pack_linear_weight(w, torch.bfloat16)
    bias = torch.zeros(n, device=dev)
    res = torch.randn(m, n, device=dev).to(torch.bfloat16)
    row = []
    for t in TILES:
        try:
            row.append(timeit(lambda: ops.conv_nhwc(x.view(m, 1, 1, k), wp, bias, kh=1, kw=1, residual=None if name.endswith("-") else res, tile=t)) * 1e3)
        except Exception:
            row.append(float("nan"))
    rounds = ((m + 255) // 256) * ((n + 255) // 256) / 256
    print(f"{name:10s} {m:9d} {k:6d} {n:5d}   {rounds:8.2f}                 | " + " ".join(f"{v:6.1f}" for v in row), flush=True)
