#!/usr/bin/env python3
"""a few bf16-operand ResNet-50 training steps (b64, 256x448) for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_cnn_train import SpatialCnnTrainer
dev = torch.device("cuda:0")
B = 64
sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=47)
frames = bench.device_frames(B, 256, 448, 3, dev)
z = (torch.rand(B, 131, device=dev) < 0.1).float()
tp = [torch.randn(B, k, device=dev) for k in (6, 10, 15)]
tf = [torch.randn(B, 1536, device=dev) for _ in range(3)]
tr = SpatialCnnTrainer("resnet50", lr=0.001, operand_dtype=torch.bfloat16).load_state_dict(sd)
tr.exchange = False
for _ in range(4):
    tr.train_step(frames, z, tp, tf)
torch.cuda.synchronize()
print("done")
