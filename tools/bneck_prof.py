#!/usr/bin/env python3
"""a few launches of the fused layer1 Bottleneck for rocprofv3 (--kernel-trace --stats, or --pmc passes): 1336 frames at 56 x 56"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B = 1336
x = torch.randn((B, 56, 56, 256), device=dev).to(bf)
mk = lambda co, ci, k: (ops.pack_conv_weight(torch.randn((co, ci, k, k), device=dev) * (ci * k * k) ** -0.5, None, bf), torch.randn(co, device=dev) * 0.1)
c1, c2, c3 = mk(64, 256, 1), mk(64, 64, 3), mk(256, 64, 1)
y = torch.empty((B, 56, 56, 256), device=dev, dtype=bf)
pk = ops.bottleneck_pack(c1, c2, c3, None)
for _ in range(3):
    ops.bottleneck_fused(x, pk, out=y)
torch.cuda.synchronize()
print("done")
