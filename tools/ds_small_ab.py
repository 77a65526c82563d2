"""ResNet-50 bf16 extraction rate at small batches with / without the conv3 + downsample launch (the one-GEMM form always runs the 256 x 256
tile; the two-launch form picks smaller tiles for few pixels).  python tools/ds_small_ab.py"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_cnn import VideoNas

dev = torch.device("cuda:0")
args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
m = VideoNas(args=args, dtype=torch.bfloat16, device=str(dev)).eval()
m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=1234))
for n in (1, 8, 32, 128, 512):
    frames = synth.synthetic_frames(n, 224, 224, seed=3).to(dev)
    row = []
    for fuse in (True, False):
        m.fuse_downsample = fuse
        for _ in range(3):
            m.extract_u8(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 20
        for _ in range(it):
            m.extract_u8(frames)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / it * 1e3)
    print(f"batch {n}: one-GEMM {row[0]:.3f} ms, two launches {row[1]:.3f} ms", flush=True)
