#!/usr/bin/env python3
"""Temporal_tenco DDP training-step throughput: one video per rank per step, flat-gradient all-reduce over RCCL.
  python tools/ddp_train_bench.py [--T 1000] [--steps 10]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/ddp_train_bench.py"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.tenco_train import TencoTrainer
ap = argparse.ArgumentParser(); ap.add_argument("--T", type=int, default=1000); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-gc", action="store_true"); ap.add_argument("--graph", action="store_true")
a = ap.parse_args()
world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
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
