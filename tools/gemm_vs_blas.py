#!/usr/bin/env python3
"""plain bf16 GEMMs of the Swin-B/384 forward (batch 128): mt4_conv_nhwc's generic kernel against torch.matmul (hipBLASLt / rocBLAS) -- a
yardstick for the hand-written kernel, not a product path (GPU box): python tools/gemm_vs_blas.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
dev = torch.device("cuda:0")
# (name, M, K, N): tokens of 128 frames x (384 / 4 / 2^s)^2 per stage
SHAPES = [("s0 qkv", 128 * 96 * 96, 128, 384), ("s0 fc1", 128 * 96 * 96, 128, 512), ("s1 qkv", 128 * 48 * 48, 256, 768), ("s1 fc1", 128 * 48 * 48, 256, 1024),
          ("s2 qkv", 128 * 24 * 24, 512, 1536), ("s2 proj", 128 * 24 * 24, 512, 512), ("s2 fc1", 128 * 24 * 24, 512, 2048), ("s2 fc2", 128 * 24 * 24, 2048, 512),
          ("s3 qkv", 128 * 12 * 12, 1024, 3072), ("s3 fc1", 128 * 12 * 12, 1024, 4096), ("s3 fc2", 128 * 12 * 12, 4096, 1024),
          ("resnet l3 conv1", 1336 * 196, 1024, 256), ("resnet l4 conv3", 1336 * 49, 512, 2048), ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, m, k, n in SHAPES:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05)
    wp = ops.pack_linear_weight(w, torch.bfloat16)
    wb = w.to(torch.bfloat16)
    bias = torch.zeros(n, device=dev)
    ms_a = timeit(lambda: ops.linear(x, wp, bias))
    ms_b = timeit(lambda: torch.nn.functional.linear(x, wb))
    fl = 2.0 * m * k * n
    print(f"{name:16s} M {m:8d} K {k:5d} N {n:5d}  mt4 {ms_a:7.3f} ms {fl / ms_a * 1e-9:7.0f} TF/s   torch {ms_b:7.3f} ms {fl / ms_b * 1e-9:7.0f} TF/s   ratio {ms_a / ms_b:5.2f}", flush=True)
