#!/usr/bin/env python3
"""Same-box A/B of the chained launch (`mt4_chain_gemm_bf16`) against the two launches it replaces, as a function of the batch: ResNet-50 bf16
extraction (conv3 + next conv1 in layers 2 / 3) and Swin-B/384 (Mlp of stages 0 / 1).  The kernel runs one 128-row workgroup per CU with no
tile choice, so small batches cannot fill the chip: this sweep sets `ops.CHAIN_MIN_TILES`.
  python tools/chain_small_batch_ab.py > profiles/r04_chain_small_batch_ab.txt"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import ops, shapes, synth  # noqa: E402
from computervision_codes_amd.graph import GraphedForward  # noqa: E402
from computervision_codes_amd.spatial_cnn import VideoNas  # noqa: E402
from computervision_codes_amd.spatial_transformer import build_q2l  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
    m = VideoNas(args=args, dtype=torch.bfloat16).eval().load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=1))
    print("# ResNet-50 bf16 224x224, hipGraph replay of extract_u8, ms per batch (median of 20); tiles = 128-row tiles of the layer2 / layer3 chained launches")
    print("batch  tiles_l2  tiles_l3  chain_off  chain_l3  chain_l23")
    for b in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1336):
        fr = synth.synthetic_frames(min(b, 16), 224, 224, seed=2).to(dev).repeat((b + 15) // 16, 1, 1, 1)[:b].contiguous()
        row = []
        for chain in ((), (3,), (2, 3)):
            m.chain_layers = chain
            ops.CHAIN_MIN_TILES = 0
            g = GraphedForward(lambda x: m.extract_u8(x), [fr])
            row.append(timeit(lambda: g(fr)))
            del g
        print(f"{b:5d}  {(b * 784 + 127) // 128:8d}  {(b * 196 + 127) // 128:8d}  {row[0]:9.4f}  {row[1]:8.4f}  {row[2]:9.4f}", flush=True)
    print("# Swin-B/384 + one decoder, bf16, eager, ms per batch; tiles = 128-row tiles of the stage-0 / stage-1 MLP launches")
    print("batch  tiles_s0  tiles_s1  mlp_two_launches  mlp_chain")
    a = types.SimpleNamespace(backbone="swin_B_384_22k", img_size=384, hidden_dim=1024, loss_type="i")
    q = build_q2l(a, dtype=torch.bfloat16).eval().load_state_dict(synth.fill_from_shapes(shapes.q2l_param_shapes("swin_B_384_22k", 384, 1024, "i"), seed=3))
    for b in (1, 2, 4, 8, 16, 32, 64, 128):
        fr = synth.synthetic_frames(min(b, 4), 384, 384, seed=4).to(dev).repeat((b + 3) // 4, 1, 1, 1)[:b].contiguous()
        row = []
        for tiles in (10 ** 9, 0):
            ops.CHAIN_MIN_TILES = tiles
            g = GraphedForward(lambda x: q(x), [fr])
            row.append(timeit(lambda: g(fr), iters=10))
            del g
        print(f"{b:5d}  {b * 9216 // 128:8d}  {b * 2304 // 128:8d}  {row[0]:16.4f}  {row[1]:9.4f}", flush=True)


if __name__ == "__main__":
    main()
