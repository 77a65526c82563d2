#!/usr/bin/env python3
"""`mt4_tcn_linear_ln_f32` against LayerNorm + Linear as two launches of the latency path, per MS-TCT shape (256 rows, fp32): 20 dependent-free launches
per hipGraph replay, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops
from computervision_codes_amd.graph import GraphedForward

dev = torch.device("cuda")
print("rows  Cin  Cout     linear    layernorm + linear    linear_ln (stats_in)   linear + stats_out   (us per launch / pair)")
for cin, cout in ((256, 768), (256, 2048), (384, 1152), (384, 3072), (576, 1728), (576, 4608), (864, 2592), (864, 6912)):
    x = torch.randn(256, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    b, g, be = torch.randn(cout, device=dev), torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    wp = ops.pack_linear_weight(w, torch.float32)
    wf, cs, bf = ops.fold_layernorm(w.cpu(), b.cpu(), g.cpu(), be.cpu())
    wfp, cs, bf = ops.pack_linear_weight(wf.to(dev), torch.float32), cs.to(dev), bf.to(dev)

    st = torch.randn(cin // 16, 256, 2, device=dev).abs()

    def lin(xx):
        with ops.latency_tiles():
            for _ in range(20):
                y = ops.linear(xx, wp, b)
        return y

    def two(xx):
        with ops.latency_tiles():
            for _ in range(20):
                y = ops.linear(ops.layernorm(xx, g, be), wp, b)
        return y


    def one_st(xx):
        for _ in range(20):
            y = ops.linear_ln(xx, wfp, cs, bf, stats_in=st)
        return y

    def lin_st(xx):
        for _ in range(20):
            y = ops.linear_stats(xx, wp, b)[0]
        return y

    row = []
    for fn in (lin, two, one_st, lin_st):
        gr = GraphedForward(fn, [x])
        for _ in range(5):
            gr(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gr(x)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 400 * 1e3)
    print(f"{256:4d} {cin:4d} {cout:5d}   {row[0]:8.2f}   {row[1]:12.2f}   {row[2]:18.2f}   {row[3]:18.2f}", flush=True)
