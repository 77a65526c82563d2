#!/usr/bin/env python3
"""`mt4_chain_gemm_bf16` against the two launches it replaces, at the shapes it is used on (GPU box, one process): ResNet-50 layer3 / layer2
conv3 + next conv1 at 1336 frames of 224 x 224, Swin-B/384 stage-0 / stage-1 MLP at batch 128.  Median of 20 timed calls each, outputs
compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervision_codes_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16


def med(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def case(name, m, k1, n1, n2, conv):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((m, k1), device=dev, generator=g).to(bf)
    w1 = ops.pack_conv_weight(torch.randn((n1, k1, 1, 1), device=dev, generator=g) * k1 ** -0.5, None, bf)
    w2 = ops.pack_conv_weight(torch.randn((n2, n1, 1, 1), device=dev, generator=g) * n1 ** -0.5, None, bf)
    b1, b2 = torch.randn(n1, device=dev, generator=g) * 0.1, torch.randn(n2, device=dev, generator=g) * 0.1
    w1f, w2f = ops.pack_fragments(w1), ops.pack_fragments(w2)
    if conv:
        r1 = torch.randn((m, n1), device=dev, generator=g).to(bf)
        y1 = torch.empty((m, n1), dtype=bf, device=dev)
        y2 = torch.empty((m, n2), dtype=bf, device=dev)
        t_a = med(lambda: ops.conv_nhwc(x.view(1, 1, m, k1), w1, b1, kh=1, kw=1, residual=r1.view(1, 1, m, n1), relu=True, out=y1.view(1, 1, m, n1)))
        t_b = med(lambda: ops.conv_nhwc(y1.view(1, 1, m, n1), w2, b2, kh=1, kw=1, relu=True, out=y2.view(1, 1, m, n2)))
        c1, c2 = torch.empty_like(y1), torch.empty_like(y2)
        t_c = med(lambda: ops.chain_gemm(x, w1f, b1, w2f, b2, r1=r1, y1=c1, y2=c2))
        same = torch.equal(c1, y1) and torch.equal(c2, y2)
        gb = (m * (k1 + 2 * n1 + n2) * 2) / 1e9
    else:
        sc = torch.randn((m, n2), device=dev, generator=g).to(bf)
        h = torch.empty((m, n1), dtype=bf, device=dev)
        y2 = torch.empty((m, n2), dtype=bf, device=dev)
        t_a = med(lambda: ops.linear(x, w1, b1, act="gelu", out=h))
        t_b = med(lambda: ops.linear(h, w2, b2, residual=sc, out=y2))
        c2 = torch.empty_like(y2)
        t_c = med(lambda: ops.chain_gemm(x, w1f, b1, w2f, b2, r2=sc, y2=c2))
        same = torch.equal(c2, y2)
        gb = (m * (k1 + 2 * n2) * 2) / 1e9
    gf = 2.0 * m * n1 * (k1 + n2) / 1e9
    print(f"{name}: two launches {t_a:.3f} + {t_b:.3f} = {t_a + t_b:.3f} ms   chained {t_c:.3f} ms ({gf / t_c:.0f} TFLOP/s, {gb / t_c * 1e3:.0f} GB/s of its own bytes)"
          f"   bit-identical {same}", flush=True)


case("layer3 conv3 + next conv1 (M = 1336 x 196, 256 -> 1024 -> 256)", 1336 * 196, 256, 1024, 256, True)
case("layer2 conv3 + next conv1 (M = 1336 x 784, 128 -> 512 -> 128)", 1336 * 784, 128, 512, 128, True)
case("layer2.3 conv3 + layer3.0 conv1 (M = 1336 x 784, 128 -> 512 -> 256)", 1336 * 784, 128, 512, 256, True)
case("Swin-B/384 stage-0 MLP (M = 128 x 9216, C = 128)", 128 * 9216, 128, 512, 128, False)
case("Swin-B/384 stage-1 MLP (M = 128 x 2304, C = 256)", 128 * 2304, 256, 1024, 256, False)
