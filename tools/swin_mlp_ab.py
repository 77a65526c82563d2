#!/usr/bin/env python3
"""same-box A/B: Swin-B + Q2L (BASELINE configs[2]) with the Mlp + shortcut of stages 0 / 1 as one launch each (`mt4_chain_gemm_bf16`, MLP form)
against fc1 -> fc2 (MT4_NO_MLP_CHAIN=1), alternating in one process; bench.py's own swin_bench()"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
for rep in range(2):
    for off in ("1", ""):
        if off:
            os.environ["MT4_NO_MLP_CHAIN"] = off
        else:
            os.environ.pop("MT4_NO_MLP_CHAIN", None)
        r = bench.swin_bench(dev)
        print(f"rep {rep} mlp chain {'off' if off else 'on '}: " + "  ".join(f"{k} {v['frames_per_s']:.0f} f/s ({v['mfma_frac']:.3f})" for k, v in r.items()), flush=True)
