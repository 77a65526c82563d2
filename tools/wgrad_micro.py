#!/usr/bin/env python3
"""weight-gradient launches of one ResNet-50 training step (batch 64, 256 x 448 frames), timed one by one (GPU box):
python tools/wgrad_micro.py [--only i,j,...] [--iters n]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops

# (name, H, W (input), Cin, Cout, K, stride, launches per step)
SHAPES = [("l1 64->64 1x1", 64, 112, 64, 64, 1, 1, 1), ("l1 256->64 1x1", 64, 112, 256, 64, 1, 1, 2), ("l1 64->64 3x3", 64, 112, 64, 64, 3, 1, 3),
          ("l1 64->256 1x1", 64, 112, 64, 256, 1, 1, 4), ("l2 256->128 1x1", 64, 112, 256, 128, 1, 1, 1), ("l2 128 3x3 s2", 64, 112, 128, 128, 3, 2, 1),
          ("l2 128->512 1x1", 32, 56, 128, 512, 1, 1, 4), ("l2 512->128 1x1", 32, 56, 512, 128, 1, 1, 3), ("l2 128 3x3", 32, 56, 128, 128, 3, 1, 3),
          ("l2 ds 256->512 s2", 64, 112, 256, 512, 1, 2, 1), ("l3 512->256 1x1", 32, 56, 512, 256, 1, 1, 1), ("l3 256 3x3 s2", 32, 56, 256, 256, 3, 2, 1),
          ("l3 256->1024 1x1", 16, 28, 256, 1024, 1, 1, 6), ("l3 1024->256 1x1", 16, 28, 1024, 256, 1, 1, 5), ("l3 256 3x3", 16, 28, 256, 256, 3, 1, 5),
          ("l3 ds 512->1024 s2", 32, 56, 512, 1024, 1, 2, 1), ("l4 1024->512 1x1", 16, 28, 1024, 512, 1, 1, 1), ("l4 512 3x3 s2", 16, 28, 512, 512, 3, 2, 1),
          ("l4 512->2048 1x1", 8, 14, 512, 2048, 1, 1, 3), ("l4 2048->512 1x1", 8, 14, 2048, 512, 1, 1, 2), ("l4 512 3x3", 8, 14, 512, 512, 3, 1, 2),
          ("l4 ds 1024->2048 s2", 16, 28, 1024, 2048, 1, 2, 1)]

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
only = [int(v) for v in args.only.split(",") if v]
dev = torch.device("cuda:0")
B = args.batch
total = 0.0
for idx, (name, h, w, cin, cout, k, s, n) in enumerate(SHAPES):
    if only and idx not in only:
        continue
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    x = torch.randn(B, h, w, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, ho, wo, cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(cout, ops.packed_k(cin, k, k, torch.float32), device=dev)
    for _ in range(3):
        ops.wgrad_conv2d_bf16(dy, x, dw, k, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        ops.wgrad_conv2d_bf16(dy, x, dw, k, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    fl = 2.0 * B * ho * wo * cin * cout * k * k
    by = (x.numel() + dy.numel()) * 2
    total += us * n
    print(f"{idx:2d} {name:22s} x{n}  {us:7.1f} us  {fl / us * 1e-6:6.0f} TFLOP/s  {by / us * 1e-6:5.2f} TB/s (tensors once)", flush=True)
print(f"per step: {total / 1e3:.3f} ms")
