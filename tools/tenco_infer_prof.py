#!/usr/bin/env python3
"""Per-video inference of the Temporal_tenco head under rocprofv3 (GPU box): N hipGraph replays of one forward.
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/tenco_infer_prof.py [--T 256] [--dtype f32|bf16] [--config tenco4|config1]
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/tenco_infer_prof.py --replays 3      (and WRITE_SIZE)"""
import argparse, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.graph import GraphedForward
from computervision_codes_amd.temporal_tenco import VideoNas

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=256)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--config", default="tenco4", choices=["tenco4", "config1"])
ap.add_argument("--replays", type=int, default=50)
ap.add_argument("--videos", type=int, default=0, help="throughput mode: this many videos per forward ([B, T, dim] input)")
ap.add_argument("--layer", default="auto", choices=["auto", "fused", "two"], help="throughput mode, bf16: one launch per DilatedResidualLayer / two / the model's window")
ap.add_argument("--tile", type=int, default=0, help="implicit-GEMM path: force this tile id (0 = the library's choice)")
a = ap.parse_args()
num_R, dim = (3, 512) if a.config == "tenco4" else (0, 2048)
fpn = num_R > 0
args = types.SimpleNamespace(fpn=fpn, output=False, hier=False, mask=True)
sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, num_R, 512, dim, 100, fpn=fpn), seed=47)
m = VideoNas(args, 11, 10, num_R, 512, dim, 100, dtype=torch.float32 if a.dtype == "f32" else torch.bfloat16).eval().load_state_dict(sd)
m.tile = a.tile
if a.layer != "auto":
    m.fused_layer_min_tiles, m.fused_layer_max_tiles = (0, 10 ** 9) if a.layer == "fused" else (10 ** 9, 10 ** 9)
x = synth.synthetic_features(a.T, dim, seed=47).cuda()
if a.videos:
    x = x.repeat(a.videos, 1, 1).contiguous()
g = GraphedForward(lambda xx: m(xx, False), [x])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.replays):
    g(x)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"config": a.config, "T": a.T, "dtype": a.dtype, "replays": a.replays, "videos": max(a.videos, 1), "layer": a.layer,
                  "ms_per_forward": round(e0.elapsed_time(e1) / a.replays, 4)}))
