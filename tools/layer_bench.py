#!/usr/bin/env python3
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace on a single kernel).
usage: python tools/layer_bench.py Cin Cout k stride Hout [--tile T] [--res] [--batch 256] [--iters 20] [--dtype bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("cin", type=int); ap.add_argument("cout", type=int); ap.add_argument("k", type=int); ap.add_argument("stride", type=int)
ap.add_argument("hout", type=int); ap.add_argument("--tile", type=int, default=0); ap.add_argument("--res", action="store_true")
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dev = torch.device("cuda:0")
hin = a.hout * a.stride
x = torch.randn(a.batch, hin, hin, a.cin, device=dev).to(dt)
w = ops.pack_conv_weight(torch.randn(a.cout, a.cin, a.k, a.k, device=dev) * 0.05, None, dt)
bias = torch.zeros(a.cout, device=dev)
res = torch.randn(a.batch, a.hout, a.hout, a.cout, device=dev).to(dt) if a.res else None
y = torch.empty(a.batch, a.hout, a.hout, a.cout, device=dev, dtype=dt)
ts = []
for it in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_nhwc(x, w, bias, kh=a.k, kw=a.k, stride=(a.stride, a.stride), pad=(a.k // 2, a.k // 2), residual=res, relu=True, tile=a.tile, out=y)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
fl = 2 * a.batch * a.hout * a.hout * a.cout * a.cin * a.k * a.k
byts = (x.numel() + y.numel() * (2 if a.res else 1)) * x.element_size()
ms = min(ts[2:])
print(f"cin{a.cin} cout{a.cout} k{a.k} s{a.stride} ho{a.hout} tile{a.tile} res{int(a.res)}: {ms:.4f} ms  {fl/ms/1e9:.1f} TF  {byts/ms/1e9:.2f} TB/s")
