#!/usr/bin/env python3
"""3x3 stride-1 bf16 layers of the ResNet trunks at both frame sizes: generic tiles against the patch-kernel tiles (GPU box only).
usage: python tools/tune_patch3x3.py [--iters 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=3); a = ap.parse_args()
dev = torch.device("cuda:0")
GENERIC = (1, 13, 17, 18, 19, 20)
PATCH = (23, 24, 26, 30, 31, 32)
cases = []
for (b, hw) in ((1336, ((56, 56), (28, 28), (14, 14), (7, 7))), (584, ((64, 112), (32, 56), (16, 28), (8, 14)))):
    for c, (h, w) in zip((64, 128, 256, 512), hw):
        cases.append((b, h, w, c))
for (b, h, w, c) in cases:
    x = torch.randn(b, h, w, c, device=dev).to(torch.bfloat16)
    wp = ops.pack_conv_weight(torch.randn(c, c, 3, 3, device=dev) * 0.05, None, torch.bfloat16)
    bias = torch.zeros(c, device=dev)
    res = {}
    for t in GENERIC + PATCH:
        ts = []
        try:
            for it in range(a.iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.conv_nhwc(x, wp, bias, kh=3, kw=3, pad=(1, 1), relu=True, tile=t); e1.record(); torch.cuda.synchronize()
                if it: ts.append(e0.elapsed_time(e1))
            res[t] = min(ts)
        except Exception:
            res[t] = None
    bg = min((v, t) for t, v in res.items() if t in GENERIC and v is not None)
    bp = min(((v, t) for t, v in res.items() if t in PATCH and v is not None), default=(None, None))
    print(f"B{b} {h}x{w} C{c}: generic best t{bg[1]}={bg[0]:.4f}  patch best t{bp[1]}={bp[0]:.4f}  | " +
          " ".join(f"t{t}={v:.4f}" if v else f"t{t}=-" for t, v in res.items()), flush=True)
