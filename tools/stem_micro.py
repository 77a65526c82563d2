"""stem + max-pool launch against the two launches, per 584 frames of 256 x 448 and 1336 frames of 224 x 224 (GPU box).  python tools/stem_micro.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from computervision_codes_amd import ops, synth
from computervision_codes_amd.spatial_cnn import IMAGENET_MEAN, IMAGENET_STD
dev = torch.device("cuda:0")
for (n, h, w) in ((584, 256, 448), (1336, 224, 224)):
    frames = synth.synthetic_frames(64, h, w, seed=3).to(dev).repeat(n // 64 + 1, 1, 1, 1)[:n].contiguous()
    g = torch.Generator().manual_seed(1)
    wp = ops.stem_s2d_weight((torch.randn((64, 3, 7, 7), generator=g) * 0.1).to(dev), None)
    bias = torch.zeros(64, device=dev)
    xs = ops.preprocess_u8_s2d(frames, IMAGENET_MEAN, IMAGENET_STD)
    t_f = bench._time_call(lambda: ops.stem_maxpool(xs, wp, bias), iters=10)
    t_u = bench._time_call(lambda: ops.maxpool3x3s2(ops.conv_nhwc(xs, wp, bias, kh=4, kw=1, relu=True, run_pixels=4, out_hw=(h // 2, w // 2))), iters=10)
    print(f"{n}x{h}x{w}: fused {t_f:.3f} ms, two launches {t_u:.3f} ms", flush=True)
