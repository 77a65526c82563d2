#!/usr/bin/env python3
"""one Swin-L + Q2L teacher training step under rocprofv3 (GPU box): rocprofv3 --kernel-trace --stats ... -- python3 tools/q2l_train_prof.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.q2l_train_bench(torch.device("cuda:0"))))
