#!/usr/bin/env python3
"""Temporal_tenco training steps on one whole video (hipGraph replay) for rocprofv3 --kernel-trace (GPU box): python tools/tenco_train_prof.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.tenco_train import TencoTrainer
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tr = TencoTrainer(lr=0.01, device=str(dev)).load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(), seed=47))
xt = synth.synthetic_features(T, 512, seed=47).to(dev)
zl = tr.prepare_labels({s: torch.from_numpy((synth.uniform01(3, i, T * k) < 0.1).reshape(T, k).astype(np.int64))
                        for i, (s, k) in enumerate((("", 100), ("_i", 6), ("_v", 10), ("_t", 15)))})
for _ in range(3):
    tr.train_step(xt, zl, use_graph=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10):
    tr.train_step(xt, zl, use_graph=True)
e1.record(); torch.cuda.synchronize()
print("T", T, "ms/step", round(e0.elapsed_time(e1) / 10, 3))
