#!/usr/bin/env python3
"""probe: eager vs hipGraph-replayed extraction at the reference's small batches (8 / 32 frames of 256x448), bf16 and fp32"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.graph import GraphedForward
from computervision_codes_amd.spatial_cnn import VideoNas
dev = torch.device("cuda:0")
for net in ("resnet18", "resnet50"):
    args = types.SimpleNamespace(network=net, loss_type="all", student_dim=None, teacher_dim=1536, train=False)
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes(net), seed=1)
    for dt in (torch.bfloat16, torch.float32):
        m = VideoNas(args=args, dtype=dt).eval().load_state_dict(sd)
        for B in (8, 32):
            fr = synth.synthetic_frames(B, 256, 448, seed=3).to(dev)
            def timeit(fn, n=30):
                for _ in range(5): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n): fn()
                torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
            eager = timeit(lambda: m.extract_u8(fr))
            g = GraphedForward(lambda f: m.extract_u8(f), [fr])
            graph = timeit(lambda: g(fr))
            print(f"{net} {str(dt)[6:]:8s} B={B:3d}: eager {eager:.3f} ms ({B / eager * 1e3:.0f} fps)  graph {graph:.3f} ms ({B / graph * 1e3:.0f} fps)")
