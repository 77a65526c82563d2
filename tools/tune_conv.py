#!/usr/bin/env python3
"""Time every tile instantiation of mt4_conv_nhwc on the distinct conv shapes of a trunk (GPU box only).
usage: python tools/tune_conv.py [--network resnet50] [--batch 256] [--dtype bf16] [--out gpurun_out/tune.json]"""
import argparse, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops, _lib
from computervision_codes_amd.spatial_cnn import VideoNas

ap = argparse.ArgumentParser()
ap.add_argument("--network", default="resnet50"); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--size", type=int, default=224)
ap.add_argument("--out", default="gpurun_out/tune.json"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dev = torch.device("cuda:0")
args = types.SimpleNamespace(network=a.network, loss_type="all", student_dim=None, teacher_dim=1536, train=False)
plan = VideoNas(args=args, device="cpu").conv_plan(a.size, a.size)
seen, rows = set(), []
ntiles = _lib.lib.mt4_conv_tile_count()
for rec in plan:
    if rec["name"] == "stem":
        continue
    key = (rec["Cin"], rec["Cout"], rec["kh"], rec["stride"], rec["Ho"])
    if key in seen:
        continue
    seen.add(key)
    s, k = rec["stride"], rec["kh"]
    hin = rec["Ho"] * s if k == 1 else (rec["Ho"] - 1) * s + 1 + (0 if s == 1 else 0)
    hin = rec["Ho"] * s
    x = torch.randn(a.batch, hin, hin, rec["Cin"], device=dev).to(dt)
    w = ops.pack_conv_weight(torch.randn(rec["Cout"], rec["Cin"], k, k, device=dev) * 0.05, None, dt)
    bias = torch.zeros(rec["Cout"], device=dev)
    has_res = rec["name"].endswith("conv3")
    res = torch.randn(a.batch, rec["Ho"], rec["Wo"], rec["Cout"], device=dev).to(dt) if has_res else None
    fl = 2 * a.batch * rec["Ho"] * rec["Wo"] * rec["Cout"] * rec["Cin"] * k * k
    best = None
    line = {"name": rec["name"], "key": key}
    for t in range(0, ntiles + 1):
        ts = []
        try:
            for it in range(a.iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.conv_nhwc(x, w, bias, kh=k, kw=k, stride=(s, s), pad=(k // 2, k // 2), residual=res, relu=True, tile=t)
                e1.record(); torch.cuda.synchronize()
                if it: ts.append(e0.elapsed_time(e1))
        except Exception as e:
            line[f"t{t}"] = None
            continue
        ms = min(ts)
        line[f"t{t}"] = round(ms, 4)
    rows.append(line)
    tl = " ".join(f"t{t}={line[f't{t}']}" for t in range(ntiles + 1))
    print(f"{rec['name']:16s} Cin{rec['Cin']:5d} Cout{rec['Cout']:5d} k{k} s{s} Ho{rec['Ho']:4d} res={int(has_res)} GF={fl/1e9:7.1f} | {tl}", flush=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(rows, open(a.out, "w"), indent=1)
