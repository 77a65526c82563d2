#!/usr/bin/env python3
"""DDP training-step throughput, flat-gradient all-reduce over RCCL once per step.
  --model tenco (default): Temporal_tenco, one video per rank per step;  --model spatial: the Spatial_cnn student distillation step
  (BASELINE configs[4]), one batch of 8 frames 256x448 per rank per step (frame-DDP).
  python tools/ddp_train_bench.py [--model spatial --network resnet50] [--T 1000] [--steps 10]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_train_bench.py"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.tenco_train import TencoTrainer
ap = argparse.ArgumentParser(); ap.add_argument("--T", type=int, default=1000); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-gc", action="store_true"); ap.add_argument("--graph", action="store_true")
ap.add_argument("--model", default="tenco", choices=["tenco", "spatial"]); ap.add_argument("--network", default="resnet50")
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
if a.model == "spatial":
    from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
    dev = f"cuda:{local}"
    tr = SpatialCnnTrainer(a.network, lr=0.01, device=dev).load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(a.network), seed=1))
    B = a.batch
    frames = synth.synthetic_frames(B, 256, 448, seed=10 + rank).to(dev)
    z = torch.cat([torch.from_numpy((synth.uniform01(5 + rank, i, B * k) < 0.15).reshape(B, k).astype(np.float32)) for i, k in
                   enumerate((6, 10, 15, 100))], 1).to(dev)
    tp = [synth.synthetic_features(B, k, seed=11 + i)[0].to(dev) for i, k in enumerate((6, 10, 15))]
    tf = [synth.synthetic_features(B, 1536, seed=21 + i)[0].to(dev) for i in range(3)]
    for _ in range(2):
        tr.train_step(frames, z, tp, tf, use_graph=a.graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        terms = tr.train_step(frames, z, tp, tf, use_graph=a.graph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        print(json.dumps({"spatial_train_ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(world * B / dt, 1), "network": a.network,
                          "frames_per_rank": B, "n_gpus": world, "grad_allreduce_MB": round(tr.G.numel() * 4 / 1e6, 1), "loss": round(terms["loss"], 4)}))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0)
tr = TencoTrainer(lr=0.01, device=f"cuda:{local}").load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(), seed=1))
x = synth.synthetic_features(a.T, 512, seed=10 + rank).to(f"cuda:{local}")
labels = {s: torch.from_numpy((synth.uniform01(3 + rank, i, a.T * k) < 0.1).reshape(a.T, k).astype(np.int64)) for i, (s, k) in
          enumerate((("", 100), ("_i", 6), ("_v", 10), ("_t", 15)))}
zl = tr.prepare_labels(labels)   # resident on the device, like the features
for _ in range(2):
    tr.train_step(x, zl, use_graph=a.graph)
torch.cuda.synchronize()
if a.no_gc:
    import gc
    gc.disable()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _ = tr.train_step(x, zl, use_graph=a.graph)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
if rank == 0:
    print(json.dumps({"tenco_train_ms_per_step": round(dt * 1e3, 3), "videos_per_s": round(world / dt, 2), "T": a.T, "n_gpus": world,
                      "grad_allreduce_MB": round(tr.G.numel() * 4 / 1e6, 1), "loss": round(loss, 4)}))
if world > 1:
    dist.destroy_process_group()
