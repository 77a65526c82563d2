#!/usr/bin/env python3
"""MI355X entry point for `MT4MTLKD/Spatial_cnn/run.py` (same flags, paths and files; see computervision_codes_amd/drivers.py):
-t trains the student (frame-DDP under torchrun), -e evaluates the test split and writes the closing report (`run.py:503-560`:
per-category AP, the mean-AP row with I / V / T disentangled from the triplet head, top-5 / 10 / 20); the extraction pass is test.py."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from computervision_codes_amd.drivers import spatial_cnn_run  # noqa: E402

if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.init_process_group(os.environ.get("MT4_DIST_BACKEND", "nccl"))   # RCCL; "gloo" = several ranks on one GPU (tests)
    spatial_cnn_run(sys.argv[1:])
