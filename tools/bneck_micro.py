#!/usr/bin/env python3
"""per-launch time of the fused layer1 Bottleneck against its three (four) stand-alone launches, 1336 frames at 56 x 56 (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from computervision_codes_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1336
for ds in (False, True):
    cin = 64 if ds else 256
    x = torch.randn((B, 56, 56, cin), device=dev).to(bf)
    mk = lambda co, ci, k: (ops.pack_conv_weight(torch.randn((co, ci, k, k), device=dev) * (ci * k * k) ** -0.5, None, bf), torch.randn(co, device=dev) * 0.1)
    c1, c2, c3 = mk(64, cin, 1), mk(64, 64, 3), mk(256, 64, 1)
    cd = mk(256, cin, 1) if ds else None
    y = torch.empty((B, 56, 56, 256), device=dev, dtype=bf)

    def unfused():
        idt = ops.conv_nhwc(x, cd[0], cd[1], kh=1, kw=1, relu=False) if ds else x
        o = ops.conv_nhwc(x, c1[0], c1[1], kh=1, kw=1, relu=True)
        o = ops.conv_nhwc(o, c2[0], c2[1], kh=3, kw=3, pad=(1, 1), relu=True)
        return ops.conv_nhwc(o, c3[0], c3[1], kh=1, kw=1, residual=idt, relu=True, out=y)
    t_u = bench._time_call(unfused, iters=10)
    pk = ops.bottleneck_pack(c1, c2, c3, cd)
    t_f = bench._time_call(lambda: ops.bottleneck_fused(x, pk, out=y), iters=10)
    gb = B * 3136 * (cin + 256) * 2 / 1e9
    print(f"ds={int(ds)} Cin={cin}: unfused {t_u:.3f} ms | fused {t_f:.3f} ms = {gb / t_f:.2f} TB/s of x-once + y-once ({gb:.2f} GB)", flush=True)
