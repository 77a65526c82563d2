"""worker of tests/test_gpu_ddp.py: one DDP step of the spatial student on rank-specific data (both ranks on cuda:0, gloo transport)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.distributed as dist
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer


def batch(rank, B=2, H=64, W=64):
    fr = synth.synthetic_frames(B, H, W, seed=100 + rank)
    labels = [torch.from_numpy((synth.uniform01(50 + rank, i, B * k) < 0.2).reshape(B, k).astype(np.int64)) for i, k in enumerate((6, 10, 15, 100))]
    tp = [synth.synthetic_features(B, k, seed=60 + rank + i)[0] for i, k in enumerate((6, 10, 15))]
    tf = [synth.synthetic_features(B, 1536, seed=70 + rank + i)[0] for i in range(3)]
    return fr, labels, tp, tf


if __name__ == "__main__":
    out_dir, overlap = sys.argv[1], sys.argv[2] in ("1", "graph")
    graph = sys.argv[2] == "graph"
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet18"), seed=9)
    tr = SpatialCnnTrainer("resnet18", lr=0.05, overlap=overlap).load_state_dict(sd)
    fr, labels, tp, tf = batch(rank)
    tr.train_step(fr.cuda(), labels, tp, tf, use_graph=graph)     # graph: segmented replay, a bucket's all-reduce issued between two segments
    if rank == 0:
        sd = tr.state_dict()
        sd["__bucket_order__"] = list(getattr(tr, "bucket_order", []))
        torch.save(sd, os.path.join(out_dir, f"ddp_overlap{sys.argv[2]}.pth"))
    dist.barrier()
    dist.destroy_process_group()
