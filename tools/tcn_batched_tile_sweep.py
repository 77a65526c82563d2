#!/usr/bin/env python3
"""Throughput mode of the temporal head (several videos per forward: M = B x T rows, N = 512): every tile of `mt4_conv_nhwc` on the two GEMM
shapes of a DilatedResidualLayer (k3 dilated 512 -> 512, K = 1536; 1x1 512 -> 512 + residual), bf16 and fp32.
  python tools/tcn_batched_tile_sweep.py > profiles/r04_tcn_batched_tile_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
ntiles = _lib.lib.mt4_conv_tile_count()
for dt in (torch.bfloat16, torch.float32):
    for B in (8, 32, 128):
        T = 256
        x = torch.randn(B, 1, T, 512, device=dev).to(dt)
        for name, k, d in (("dilated k3 d=4", 3, 4), ("1x1 + residual", 1, 1)):
            w = ops.pack_conv_weight(torch.randn(512, 512, 1, k, device=dev) * 0.03, None, dt)
            bias = torch.zeros(512, device=dev)
            res = x if k == 1 else None
            out = {}
            for t in [0, -1] + list(range(1, ntiles + 1)):
                try:
                    ts = []
                    for it in range(6):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        ops.conv_nhwc(x, w, bias, kh=1, kw=k, pad=(0, d * (k // 2)), dil=(1, d), residual=res, relu=(k == 3), tile=t)
                        e1.record(); torch.cuda.synchronize()
                        if it:
                            ts.append(e0.elapsed_time(e1) * 1e3)
                    out[t] = min(ts)
                except Exception:
                    pass
            best = sorted((v, t) for t, v in out.items() if t > 0)[:4]
            gf = 2 * B * T * 512 * 512 * k / 1e9
            print(f"{str(dt)[6:]:8s} B={B:3d} rows={B * T:6d} {name:15s}: auto {out.get(0, float('nan')):7.1f} us  latency-auto {out.get(-1, float('nan')):7.1f} us | best "
                  + "  ".join(f"t{t}={v:.1f}us ({gf / v * 1e-3:.0f} TF)" for v, t in best), flush=True)
