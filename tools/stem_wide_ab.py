"""same-box A/B of the stem + max-pool launch on 256 x 448 frames (two column segments) against the two launches (MT4_NO_STEM_POOL_FUSE): ResNet-50 and
ResNet-18, 584 frames per stream x 2 streams.  python tools/stem_wide_ab.py"""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import shapes, synth
from computervision_codes_amd.spatial_cnn import VideoNas

dev = torch.device("cuda:0")
frames = synth.synthetic_frames(64, 256, 448, seed=3).to(dev).repeat(19, 1, 1, 1)[:1168].contiguous()
for net in ("resnet50", "resnet18"):
    args = types.SimpleNamespace(network=net, loss_type="all", student_dim=shapes.resnet_feat_dim(net), teacher_dim=1536, train=False)
    m = VideoNas(args=args, dtype=torch.bfloat16, device=str(dev)).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(net), seed=1234))
    outs = {}
    for rep in range(2):
        for fuse in (True, False):
            m.fuse_stem_pool = fuse
            for _ in range(3):
                o = m.extract_u8(frames, streams=2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                o = m.extract_u8(frames, streams=2)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 8 * 1e3
            outs[fuse] = o
            print(f"{net} 256x448 stem+pool fused={int(fuse)}: {1168 / ms * 1e3:.0f} frames/s ({ms:.2f} ms)", flush=True)
    same = torch.equal(outs[True][3][0], outs[False][3][0]) and all(torch.equal(outs[True][i][1], outs[False][i][1]) for i in range(4))
    print(f"{net}: outputs bit-identical = {same}", flush=True)
