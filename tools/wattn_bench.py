#!/usr/bin/env python3
"""MFMA window attention micro-benchmark (GPU box): python tools/wattn_bench.py [--windows 8192] [--heads 4] [--n 144] [--mask 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=8192); ap.add_argument("--heads", type=int, default=4); ap.add_argument("--n", type=int, default=144)
ap.add_argument("--mask", type=int, default=0, help="number of distinct window masks (0 = unshifted block)"); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--rel", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
C = a.heads * 32
qkv = torch.randn(a.windows * a.n, 3 * C, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
bias = ops.pad_attention_bias(torch.randn(a.heads, a.n, a.n, device=dev) * 0.1)
mask = None
if a.mask:
    m = torch.zeros(a.mask, a.n, a.n, device=dev)
    edge = [i for i in range(a.mask) if (i % 8 == 7 or i // 8 == 7)] if a.mask == 64 else list(range(a.mask))
    for i in edge:
        m[i, : a.n // 2, a.n // 2:] = -100.0
        m[i, a.n // 2:, : a.n // 2] = -100.0
    mask = ops.pad_attention_bias(m, key_pad_value=0.0)
ws = int(round(a.n ** 0.5))
table = (torch.randn(a.heads, (2 * ws - 1) ** 2, device=dev) * 0.1).contiguous()
region = None
if a.mask:
    region = torch.zeros(a.mask, a.n, dtype=torch.int32, device=dev)
    for i in edge:
        region[i, a.n // 2:] = 1
ts = []
for _ in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if a.rel:
        ops.window_attention_rel_bf16(q, k, v, batch=a.windows, heads=a.heads, ws=ws, q_stride=3 * C, k_stride=3 * C, v_stride=3 * C, scale=32 ** -0.5,
                                      rel_table=table, region=region)
    else:
        ops.window_attention_bf16(q, k, v, batch=a.windows, heads=a.heads, n=a.n, q_stride=3 * C, k_stride=3 * C, v_stride=3 * C, scale=32 ** -0.5,
                                  bias_padded=bias, mask_padded=mask)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = min(ts[2:])
fl = a.windows * a.heads * 2 * (a.n * a.n * 32) * 2
by = a.windows * a.n * C * 2 * 4
print("rel" if a.rel else "exp", f"windows {a.windows} heads {a.heads} n {a.n} masks {a.mask}: {ms:.4f} ms  {fl / ms / 1e9:.1f} TF  {by / ms / 1e9:.2f} TB/s (q,k,v,out)")
