#!/usr/bin/env python3
"""probe: does running two half-batches on two HIP streams overlap compute-bound and HBM-bound layers?  (GPU box)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_cnn import VideoNas
dev = torch.device("cuda:0")
args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
m = VideoNas(args=args, dtype=torch.bfloat16).eval().load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=1))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1336
base = synth.synthetic_frames(16, 224, 224, seed=3).to(dev)
frames = base.repeat(B // 16 + 1, 1, 1, 1)[:B].contiguous()
def run_single(n):
    for _ in range(n): m.extract_u8(frames)
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
streams = [torch.cuda.Stream() for _ in range(NS)]
h = B // NS
parts = [frames[i * h:(i + 1) * h].contiguous() for i in range(NS)]
def run_two(n):
    for _ in range(n):
        for st, f in zip(streams, parts):
            with torch.cuda.stream(st): m.extract_u8(f)
for fn, name in ((run_single, "one stream"), (run_two, f"{NS} streams")):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(8); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"{name}: {dt * 1e3:.3f} ms per {B} frames = {B / dt:.0f} frames/s")
