"""GPU: each C-ABI kernel against the CPU oracle op (torch fp32 on the host) on seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _conv_case(cuda, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, dtype, relu, use_res, tile, seed):
    from computervision_codes_amd import ops
    x = _rand((B, Cin, H, W), seed)
    w = _rand((Cout, Cin, kh, kw), seed + 1, scale=(3.0 / (Cin * kh * kw)) ** 0.5)
    bias = _rand((Cout,), seed + 2, 0.1)
    if dtype == torch.bfloat16:  # oracle sees the same rounded operands
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    ref = F.conv2d(x, w, bias, stride=stride, padding=pad, dilation=dil)
    res = None
    if use_res:
        res = _rand(tuple(ref.shape), seed + 3)
        if dtype == torch.bfloat16:
            res = res.bfloat16().float()
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda, dtype)
    wp = ops.pack_conv_weight(w.to(cuda), None, dtype)
    rd = res.permute(0, 2, 3, 1).contiguous().to(cuda, dtype) if use_res else None
    y = ops.conv_nhwc(xd, wp, bias.to(cuda), kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, residual=rd, relu=relu, tile=tile)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (err, ref.abs().max().item())


CONV_SHAPES = [
    # B, H, W, Cin, Cout, kh, kw, stride, pad, dil
    (2, 14, 14, 64, 64, 1, 1, (1, 1), (0, 0), (1, 1)),      # 1x1, FAST for both dtypes
    (2, 14, 14, 64, 128, 3, 3, (1, 1), (1, 1), (1, 1)),     # 3x3 pad 1
    (2, 15, 13, 64, 64, 3, 3, (2, 2), (1, 1), (1, 1)),      # strided 3x3, odd sizes
    (1, 9, 9, 128, 256, 1, 1, (2, 2), (0, 0), (1, 1)),      # strided 1x1 downsample
    (1, 1, 77, 64, 64, 1, 3, (1, 1), (0, 4), (1, 4)),       # dilated conv1d, ragged T
    (1, 1, 50, 64, 64, 1, 3, (1, 1), (0, 64), (1, 64)),     # dilation > T: outer taps read only padding
    (1, 1, 40, 48, 100, 1, 1, (1, 1), (0, 0), (1, 1)),      # GENERIC path (Cin*es % 128 != 0), Cout % 4 == 0 but ragged
    (3, 1, 33, 32, 131, 1, 1, (1, 1), (0, 0), (1, 1)),      # Cout % 4 != 0 -> scalar epilogue
    (2, 7, 7, 512, 512, 3, 3, (1, 1), (1, 1), (1, 1)),      # layer4 shape, long K
    (70, 1, 1, 64, 64, 3, 3, (1, 1), (1, 1), (1, 1)),       # padded kernel over 1x1 images: only the centre tap is inside
]


PATCH_SHAPES = [
    # B, H, W, Cin, Cout: 3x3 / stride 1 / pad 1 bf16 layers of the ResNet trunks and ragged variants
    (3, 56, 56, 64, 64),      # layer1 conv2: one channel slice, several rows of the image per tile
    (5, 28, 28, 128, 128),    # layer2: two slices (double-buffered patch)
    (9, 14, 14, 256, 256),    # layer3: a tile spans more than one image
    (11, 7, 7, 512, 512),     # layer4: 5 images per tile, 8 slices
    (2, 13, 17, 64, 72),      # odd sizes, ragged M, Cout not a multiple of the tile
    (1, 3, 90, 64, 40),       # wide rows: patch of 256 + 182 rows, 7 DMA pieces
    (300, 1, 1, 64, 64),      # 1x1 images: only the centre tap is ever valid
    (4, 2, 2, 192, 136),      # three slices, tiny images
]


PATCH_TILES = {  # id: (BM, BN, waves, weight stages)   (igemm_conv.hip launch_patch_tile; the other ids of 21..32 are retired variants)
    23: (256, 256, 16, 2), 24: (256, 64, 8, 2), 26: (256, 128, 16, 2), 30: (128, 128, 4, 2), 32: (256, 128, 8, 2)}


def _patch_tile_fits(tile, W, Cin):
    """launch_patch3x3's own rule: the patch (double-buffered over channel slices) + the weight ring fit 160 KB of LDS, the next
    slice's patch pieces fit the taps they ride along with, and the weights-resident variants take single-slice layers only"""
    bm, bn, waves, ws = PATCH_TILES[tile]
    spt = Cin // 64
    pra = (bm + 2 * W + 2 + 7) // 8 * 8
    lds = (2 if spt > 1 else 1) * pra * 128 + ws * bn * 128
    if ws == 9 and spt != 1:
        return False
    return lds <= 160 * 1024 and (spt == 1 or -(-pra // (waves * 8)) <= 11 - ws)


@pytest.mark.parametrize("tile", sorted(PATCH_TILES))
@pytest.mark.parametrize("shape", PATCH_SHAPES)
def test_conv3x3_patch_kernel(cuda, shape, tile):
    """the 3x3 patch kernel (input patch staged once per channel slice, tap validity masked on the fragments) walks K in the
    generic kernel's order with the same MFMA chain: bit-identical to it, and within bf16 tolerance of F.conv2d"""
    from computervision_codes_amd import ops
    B, H, W, Cin, Cout = shape
    if not _patch_tile_fits(tile, W, Cin):   # (auto selection falls back to the generic tiles in these cases)
        with pytest.raises(RuntimeError):
            _conv_case(cuda, B, H, W, Cin, Cout, 3, 3, (1, 1), (1, 1), (1, 1), torch.bfloat16, relu=False, use_res=False, tile=tile, seed=21)
        return
    for relu in (False, True):
        _conv_case(cuda, B, H, W, Cin, Cout, 3, 3, (1, 1), (1, 1), (1, 1), torch.bfloat16, relu=relu, use_res=relu, tile=tile, seed=21 + relu)
    x = _rand((B, H, W, Cin), 31).to(cuda, torch.bfloat16)
    wp = ops.pack_conv_weight(_rand((Cout, Cin, 3, 3), 32, 0.05).to(cuda), None, torch.bfloat16)
    bias = _rand((Cout,), 33, 0.1).to(cuda)
    ref = ops.conv_nhwc(x, wp, bias, kh=3, kw=3, pad=(1, 1), relu=True, tile=2)
    got = ops.conv_nhwc(x, wp, bias, kh=3, kw=3, pad=(1, 1), relu=True, tile=tile)
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16))
    res = _rand((B, H, W, Cout), 34).to(cuda, torch.bfloat16)   # the second conv of a BasicBlock: + identity, then ReLU
    ref = ops.conv_nhwc(x, wp, bias, kh=3, kw=3, pad=(1, 1), residual=res, relu=True, tile=2)
    got = ops.conv_nhwc(x, wp, bias, kh=3, kw=3, pad=(1, 1), residual=res, relu=True, tile=tile)
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tile", [35, 36, 37, 38])
def test_conv_ksplit_tiles(cuda, tile, dtype):
    """tiles 35 / 36: four K-split wave groups per workgroup (each runs every 4th K-step of the same output tile, partial tiles added
    in LDS in fixed order): TCN-shaped layers -- long K, few pixels -- against F.conv2d; K-step counts that do not divide by 4, a single
    K-step, ragged T, the 131-wide heads (direct epilogue) and residual + ReLU; deterministic and independent of what else is in the batch"""
    from computervision_codes_amd import ops
    _conv_case(cuda, 1, 1, 256, 512, 512, 1, 3, (1, 1), (0, 4), (1, 4), dtype, relu=True, use_res=False, tile=tile, seed=51)     # 48 / 24 K-steps
    _conv_case(cuda, 1, 1, 77, 512, 512, 1, 1, (1, 1), (0, 0), (1, 1), dtype, relu=False, use_res=True, tile=tile, seed=52)      # ragged T
    _conv_case(cuda, 2, 1, 100, 448, 131, 1, 1, (1, 1), (0, 0), (1, 1), dtype, relu=False, use_res=False, tile=tile, seed=53)    # 14 / 7 steps, ragged Cout
    _conv_case(cuda, 1, 1, 40, 64, 96, 1, 3, (1, 1), (0, 1), (1, 1), dtype, relu=True, use_res=True, tile=tile, seed=54)         # 6 / 3 steps
    _conv_case(cuda, 1, 1, 40, 32 * (4 // (2 if dtype == torch.bfloat16 else 1)), 64, 1, 1, (1, 1), (0, 0), (1, 1), dtype, relu=False,
               use_res=False, tile=tile, seed=55)                                                                             # one K-step
    x = _rand((3, 1, 200, 512), 56).to(cuda, dtype)
    wp = ops.pack_conv_weight(_rand((512, 512, 1, 3), 57, 0.03).to(cuda), None, dtype)
    bias = _rand((512,), 58, 0.1).to(cuda)
    kw = dict(kh=1, kw=3, pad=(0, 2), dil=(1, 2), relu=True)
    a = ops.conv_nhwc(x, wp, bias, tile=tile, **kw)
    b = ops.conv_nhwc(x, wp, bias, tile=tile, **kw)
    one = ops.conv_nhwc(x[1:2].contiguous(), wp, bias, tile=tile, **kw)
    assert torch.equal(a, b) and torch.equal(a[1:2], one)
    ref = ops.conv_nhwc(x, wp, bias, tile=6, **kw)        # sequential K order: equal up to fp32 reassociation / one bf16 ulp
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    assert (a.float() - ref.float()).abs().max().item() <= tol * max(1.0, ref.float().abs().max().item())


def test_conv_input_larger_than_2gib(cuda):
    """the LDS-DMA path addresses x with 32-bit offsets from a per-tile origin: a 2.3 GB input gives the bytes of its two halves run
    separately (1x1 GEMM rows, 3x3 pad 1, strided 1x1), and takes the same time class as the halves (not the register-staged path)"""
    from computervision_codes_amd import ops
    g = torch.Generator(device=cuda).manual_seed(3)
    B = 1440
    x = (torch.rand((B, 56, 56, 256), device=cuda, generator=g) - 0.5).to(torch.bfloat16)
    assert x.numel() * 2 > 2 ** 31
    for (cout, k, stride, pad) in ((64, 1, 1, 0), (64, 3, 1, 1), (128, 1, 2, 0)):
        wp = ops.pack_conv_weight(_rand((cout, 256, k, k), 41, 0.05).to(cuda), None, torch.bfloat16)
        bias = _rand((cout,), 42, 0.1).to(cuda)
        kw = dict(kh=k, kw=k, stride=(stride, stride), pad=(pad, pad), relu=True)
        whole = ops.conv_nhwc(x, wp, bias, **kw)
        h = B // 2
        lo = ops.conv_nhwc(x[:h].contiguous(), wp, bias, **kw)
        hi = ops.conv_nhwc(x[h:].contiguous(), wp, bias, **kw)
        assert torch.equal(whole[:h].view(torch.int16), lo.view(torch.int16)) and torch.equal(whole[h:].view(torch.int16), hi.view(torch.int16)), (cout, k)
        del whole, lo, hi
    torch.cuda.empty_cache()


def test_conv3x3_patch_kernel_refuses_other_geometry(cuda):
    from computervision_codes_amd import ops
    x = _rand((1, 8, 8, 64), 1).to(cuda, torch.bfloat16)
    wp = ops.pack_conv_weight(_rand((64, 64, 3, 3), 2, 0.05).to(cuda), None, torch.bfloat16)
    for retired in (21, 22, 25, 27, 28, 29, 31):      # ids of the tuning record whose instantiations were removed
        with pytest.raises(RuntimeError):
            ops.conv_nhwc(x, wp, None, kh=3, kw=3, pad=(1, 1), tile=retired)
    with pytest.raises(RuntimeError):
        ops.conv_nhwc(x, wp, None, kh=3, kw=3, stride=(2, 2), pad=(1, 1), tile=24)
    with pytest.raises(RuntimeError):
        ops.conv_nhwc(x.float(), ops.pack_conv_weight(_rand((64, 64, 3, 3), 2, 0.05).to(cuda), None, torch.float32), None, kh=3, kw=3, pad=(1, 1), tile=23)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_nhwc_auto_tile(cuda, shape, dtype):
    _conv_case(cuda, *shape, dtype=dtype, relu=True, use_res=True, tile=0, seed=11)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tile", list(range(1, 21)))
def test_conv_nhwc_every_tile(cuda, tile, dtype):
    # M = 2*13*11 = 286 (ragged vs every BM), N = 96 (ragged vs 64/128), both K modes
    _conv_case(cuda, 2, 13, 11, 64, 96, 3, 3, (1, 1), (1, 1), (1, 1), dtype, relu=False, use_res=False, tile=tile, seed=5)
    _conv_case(cuda, 2, 13, 11, 24, 96, 3, 3, (1, 1), (1, 1), (1, 1), dtype, relu=True, use_res=True, tile=tile, seed=6)
    # long K (18-72 K-steps): exercises the multi-stage DMA ring's steady state and its tail
    _conv_case(cuda, 1, 9, 9, 256, 64, 3, 3, (1, 1), (1, 1), (1, 1), dtype, relu=True, use_res=False, tile=tile, seed=7)
    _conv_case(cuda, 1, 1, 40, 128, 32, 1, 3, (1, 1), (0, 2), (1, 2), dtype, relu=False, use_res=True, tile=tile, seed=8)
    if tile >= 13:   # 8-wave tiles: several 256-row tiles with ragged edges in both directions, 4-pass epilogue
        _conv_case(cuda, 3, 17, 19, 64, 320, 3, 3, (1, 1), (1, 1), (1, 1), dtype, relu=True, use_res=True, tile=tile, seed=9)


def test_conv_bf16_in_f32_out(cuda):
    from computervision_codes_amd import ops
    x = _rand((1, 64, 8, 8), 3).bfloat16().float()
    w = _rand((32, 64, 1, 1), 4, 0.2).bfloat16().float()
    ref = F.conv2d(x, w)
    y = ops.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().to(cuda, torch.bfloat16), ops.pack_conv_weight(w.to(cuda), None, torch.bfloat16),
                      None, kh=1, kw=1, out_dtype=torch.float32)
    assert y.dtype == torch.float32
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-4


def test_conv_rejects_bad_alignment(cuda):
    from computervision_codes_amd import _lib, ops
    x = torch.zeros(1, 4, 4, 6, device=cuda)  # Cin*4 % 16 != 0
    with pytest.raises((_lib.Mt4Error, AssertionError)):
        ops.conv_nhwc(x, torch.zeros(8, 8, device=cuda), None, kh=1, kw=1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(224, 224), (64, 96), (37, 50)])
def test_stem_path_matches_conv7x7(cuda, hw, dtype):
    """preprocess (u8 -> normalised padded NHWC4) + stem conv == Normalize + Conv2d(3,64,7,2,3)."""
    from computervision_codes_amd import ops, synth
    h, w = hw
    fr = synth.synthetic_frames(2, h, w, seed=9)
    xn = synth.normalize_frames(fr)
    wt = _rand((64, 3, 7, 7), 21, (3.0 / 147) ** 0.5)
    bias = _rand((64,), 22, 0.1)
    if dtype == torch.bfloat16:
        xn_ref, wt_ref = xn.bfloat16().float(), wt.bfloat16().float()
    else:
        xn_ref, wt_ref = xn, wt
    ref = F.relu(F.conv2d(xn_ref, wt_ref, bias, stride=2, padding=3))
    xp = ops.preprocess_u8(fr.to(cuda), synth.IMAGENET_MEAN, synth.IMAGENET_STD, dtype)
    xp2 = ops.pad_nchw(xn.to(cuda), dtype)
    assert torch.equal(xp[..., :3].float().cpu()[:, 3:3 + h, 3:3 + w], xn_ref.permute(0, 2, 3, 1)) or \
        (xp[..., :3].float().cpu()[:, 3:3 + h, 3:3 + w] - xn_ref.permute(0, 2, 3, 1)).abs().max() < (1e-6 if dtype == torch.float32 else 2e-2)
    assert (xp.float() - xp2.float()).abs().max().item() < (1e-6 if dtype == torch.float32 else 2e-2)
    assert xp[:, :3].abs().max().item() == 0 and xp[..., 3].abs().max().item() == 0
    wp = ops.pack_stem_weight(wt.to(cuda), None, dtype)
    b, hp, wpd, _ = xp.shape
    y = ops.conv_nhwc(xp.view(b, hp, wpd // 2, 8), wp, bias.to(cuda), kh=7, kw=4, stride=(2, 1), relu=True)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_avgpool_linear(cuda, dtype):
    from computervision_codes_amd import ops
    x = _rand((3, 64, 13, 18), 31).to(dtype).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda, dtype)
    got = ops.maxpool3x3s2(xd).float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.max_pool2d(x, 3, 2, 1))  # max is exact in any dtype
    pooled = ops.global_avgpool(xd).cpu()
    assert (pooled - x.mean(dim=(2, 3))).abs().max().item() < 1e-5
    f = _rand((5, 2048), 32)
    w = _rand((131, 2048), 33, 0.05)
    b = _rand((131,), 34)
    y = ops.linear_f32(f.to(cuda), w.to(cuda), b.to(cuda)).cpu()
    assert (y - F.linear(f, w, b)).abs().max().item() < 1e-4


@pytest.mark.parametrize("b,h,w", [(3, 56, 56), (2, 16, 24), (1, 9, 15), (2, 8, 14), (1, 1, 1)])
@pytest.mark.parametrize("ds", [False, True])
def test_bottleneck_fused_bit_identical_to_three_launches(cuda, b, h, w, ds):
    """`mt4_bottleneck_fused_bf16` (layer1 of ResNet-50 in one launch, intermediates in LDS) == conv1 -> conv2 -> conv3 (+ downsample) through
    `mt4_conv_nhwc`, bit for bit: full tiles, ragged edge tiles, images smaller than a tile"""
    from computervision_codes_amd import ops
    bf = torch.bfloat16
    cin = 64 if ds else 256
    g = torch.Generator().manual_seed(7 + h)
    x = torch.randn((b, h, w, cin), generator=g).to(cuda).to(bf)

    def mk(cout, ci, k, s):
        wt = (torch.randn((cout, ci, k, k), generator=g) * s).to(cuda)
        return ops.pack_conv_weight(wt, None, bf), (torch.randn(cout, generator=g) * 0.3).to(cuda)
    c1, c2, c3 = mk(64, cin, 1, cin ** -0.5), mk(64, 64, 3, 1 / 24), mk(256, 64, 1, 1 / 8)
    cd = mk(256, cin, 1, cin ** -0.5) if ds else None
    idt = ops.conv_nhwc(x, cd[0], cd[1], kh=1, kw=1, relu=False) if ds else x
    o = ops.conv_nhwc(x, c1[0], c1[1], kh=1, kw=1, relu=True)
    o = ops.conv_nhwc(o, c2[0], c2[1], kh=3, kw=3, pad=(1, 1), relu=True)
    ref = ops.conv_nhwc(o, c3[0], c3[1], kh=1, kw=1, residual=idt, relu=True)
    y = ops.bottleneck_fused(x, ops.bottleneck_pack(c1, c2, c3, cd))
    assert float(ref.float().abs().max()) > 0.5
    assert torch.equal(y.view(torch.int16), ref.view(torch.int16)), float((y.float() - ref.float()).abs().max())


@pytest.mark.parametrize("b,h,w", [(3, 224, 224), (2, 68, 64), (1, 8, 32), (5, 30, 96), (3, 256, 448), (2, 34, 256), (1, 6, 320)])
def test_stem_maxpool_fused_bit_identical_to_two_launches(cuda, b, h, w):
    """`mt4_stem_maxpool_bf16` (conv1 / bn1 / relu / maxpool of `resnet.py:145-149` in one launch on the space-to-depth frame) == the stem through
    `mt4_conv_nhwc` followed by `mt4_maxpool3x3s2_nhwc`, bit for bit: full frames, an odd number of pooled rows, frames smaller than a tile, and the
    two-column-segment form of frames wider than 224 pixels (the reference's 256 x 448)"""
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_cnn import IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator().manual_seed(11 + h)
    frames = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8).to(cuda)
    wt = (torch.randn((64, 3, 7, 7), generator=g) * 0.1).to(cuda)
    scale = (torch.rand(64, generator=g) + 0.5).to(cuda)
    bias = (torch.randn(64, generator=g) * 0.3).to(cuda)
    wp = ops.stem_s2d_weight(wt, scale)
    assert ops.stem_maxpool_supported(h, w)
    xs = ops.preprocess_u8_s2d(frames, IMAGENET_MEAN, IMAGENET_STD)
    conv = ops.conv_nhwc(xs, wp, bias, kh=4, kw=1, relu=True, run_pixels=4, out_hw=(h // 2, w // 2))
    ref = ops.maxpool3x3s2(conv)
    y = ops.stem_maxpool(xs, wp, bias)
    assert y.shape == ref.shape and float(ref.float().abs().max()) > 0.5
    assert torch.equal(y.view(torch.int16), ref.view(torch.int16)), float((y.float() - ref.float()).abs().max())


@pytest.mark.parametrize("b,h2,w2,mid,c2,cout,s", [(3, 28, 28, 128, 256, 512, 2), (2, 27, 13, 256, 512, 1024, 2), (5, 7, 7, 64, 128, 256, 1), (1, 56, 56, 128, 256, 512, 2)])
def test_conv_with_second_k_source_is_conv3_plus_downsample(cuda, b, h2, w2, mid, c2, cout, s):
    """`mt4_conv_desc.x2`: conv3 + bn3 and the downsample branch of a strided Bottleneck (`resnet.py:112-119`) as one GEMM over K = planes + Cin.
    Against fp32 torch on the same bf16 operands (one rounding at the end), and within bf16 rounding of the identity of the two-launch form"""
    from computervision_codes_amd import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5 + h2)
    ho, wo = (h2 - 1) // s + 1, (w2 - 1) // s + 1
    t2 = torch.randn((b, ho, wo, mid), generator=g).to(bf)
    xb = torch.randn((b, h2, w2, c2), generator=g).to(bf)
    w3 = (torch.randn((cout, mid, 1, 1), generator=g) * mid ** -0.5)
    wd = (torch.randn((cout, c2, 1, 1), generator=g) * c2 ** -0.5)
    b3, bd = torch.randn(cout, generator=g) * 0.3, torch.randn(cout, generator=g) * 0.3
    w3p, wdp = ops.pack_conv_weight(w3.to(cuda), None, bf), ops.pack_conv_weight(wd.to(cuda), None, bf)
    y = ops.conv_nhwc(t2.to(cuda), torch.cat([w3p, wdp], 1).contiguous(), (b3 + bd).to(cuda), kh=1, kw=1, relu=True, second=(xb.to(cuda), s))
    ref = torch.relu(t2.float() @ w3.view(cout, mid).to(bf).float().T + xb[:, ::s, ::s].float() @ wd.view(cout, c2).to(bf).float().T + (b3 + bd))
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < 2 ** -8            # half a bf16 ulp of the result + accumulation order
    idt = ops.conv_nhwc(xb.to(cuda), wdp, bd.to(cuda), kh=1, kw=1, stride=(s, s), relu=False)
    two = ops.conv_nhwc(t2.to(cuda), w3p, b3.to(cuda), kh=1, kw=1, residual=idt, relu=True)
    assert float(((y.float() - two.float()).abs() / (two.float().abs() + 1.0)).max()) < 2 ** -6


@pytest.mark.parametrize("b,h,w", [(3, 56, 56), (2, 16, 24), (1, 9, 15), (2, 8, 14), (1, 1, 1)])
def test_bottleneck_fused_next_bit_identical(cuda, b, h, w):
    """`mt4_bottleneck_fused_next_bf16`: the last identity block of layer1 + the following block's conv1 in one launch; the block's output at the
    even pixels and the conv1 output equal the separate launches bit for bit (full, ragged and odd-sized frames)"""
    from computervision_codes_amd import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(3 + h)
    x = torch.randn((b, h, w, 256), generator=g).to(cuda).to(bf)

    def mk(cout, ci, k, s):
        wt = (torch.randn((cout, ci, k, k), generator=g) * s).to(cuda)
        return ops.pack_conv_weight(wt, None, bf), (torch.randn(cout, generator=g) * 0.3).to(cuda)
    c1, c2, c3, cn = mk(64, 256, 1, 1 / 16), mk(64, 64, 3, 1 / 24), mk(256, 64, 1, 1 / 8), mk(128, 256, 1, 1 / 16)
    packed = ops.bottleneck_pack(c1, c2, c3, None)
    ref_y = ops.bottleneck_fused(x, packed)
    ref_t = ops.conv_nhwc(ref_y, cn[0], cn[1], kh=1, kw=1, relu=True)
    y_even, t = ops.bottleneck_fused_next(x, packed, ops.bottleneck_pack_next(cn))
    assert float(ref_t.float().abs().max()) > 0.5
    assert torch.equal(y_even.view(torch.int16), ref_y[:, ::2, ::2].contiguous().view(torch.int16))
    assert torch.equal(t.view(torch.int16), ref_t.view(torch.int16)), float((t.float() - ref_t.float()).abs().max())


@pytest.mark.parametrize("b,h,w", [(700, 28, 28), (300, 27, 29), (160, 32, 56)])
def test_conv3x3_expand_bit_identical_to_two_launches(cuda, b, h, w):
    """`mt4_conv_desc.fuse_expand`: conv2 + bn2 + ReLU and conv3 + bn3 + add + ReLU of a layer2 identity Bottleneck in one launch (the 128-channel map
    stays in LDS) == the two launches, bit for bit: both patch tiles (rows of <= 31 and of > 31 pixels), ragged last tile"""
    from computervision_codes_amd import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(9 + h)
    t1 = torch.randn((b, h, w, 128), generator=g).to(cuda).to(bf)
    res = torch.randn((b, h, w, 512), generator=g).to(cuda).to(bf)
    w2 = ops.pack_conv_weight((torch.randn((128, 128, 3, 3), generator=g) / 34).to(cuda), None, bf)
    w3 = ops.pack_conv_weight((torch.randn((512, 128, 1, 1), generator=g) / 11).to(cuda), None, bf)
    b2, b3 = (torch.randn(128, generator=g) * 0.3).to(cuda), (torch.randn(512, generator=g) * 0.3).to(cuda)
    t2 = ops.conv_nhwc(t1, w2, b2, kh=3, kw=3, pad=(1, 1), relu=True)
    ref = ops.conv_nhwc(t2, w3, b3, kh=1, kw=1, residual=res, relu=True)
    y = ops.conv3x3_expand(t1, w2, b2, ops.pack_fragments(w3), b3, res)
    assert y is not None and float(ref.float().abs().max()) > 0.5
    assert torch.equal(y.view(torch.int16), ref.view(torch.int16)), float((y.float() - ref.float()).abs().max())
    small = ops.conv3x3_expand(t1[:2].contiguous(), w2, b2, ops.pack_fragments(w3), b3, res[:2].contiguous())
    assert small is None          # few tiles: the caller runs the two launches


@pytest.mark.parametrize("b,h,w", [(2, 224, 224), (3, 6, 10), (1, 2, 2), (2, 32, 34), (2, 256, 448)])
def test_preprocess_u8_s2d_matches_torch(cuda, b, h, w):
    """`mt4_preprocess_u8_s2d` (ToTensor + Normalize of `Spatial_cnn/dataloader.py:153-162`, 3-pixel zero padding, 2 x 2 space-to-depth) against the
    same arithmetic in torch, bit for bit -- every border column and row (the kernel reads pixel pairs with clamped addresses)"""
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_cnn import IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator().manual_seed(17 + w)
    frames = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8)
    got = ops.preprocess_u8_s2d(frames.to(cuda), IMAGENET_MEAN, IMAGENET_STD).cpu()
    mean, std = torch.tensor(IMAGENET_MEAN, dtype=torch.float32), torch.tensor(IMAGENET_STD, dtype=torch.float32)
    x = ((frames.float() / 255.0 - mean) / std).to(torch.bfloat16)
    hp, wp = h + 6, w + 6
    pad = torch.zeros((b, hp, wp, 3), dtype=torch.bfloat16)
    pad[:, 3:3 + h, 3:3 + w] = x
    s2d = pad.view(b, hp // 2, 2, wp // 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(b, hp // 2, wp // 2, 12)
    ref = torch.zeros((b, hp // 2, wp // 2, 16), dtype=torch.bfloat16)
    ref[..., :12] = s2d
    assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b,t,c", [(1, 301, 64), (2, 26, 512), (3, 7, 8)])
def test_avgpool1d_and_linear_interpolation_rows(cuda, dtype, b, t, c):
    """`--hier` pieces of Temporal_tenco on frame-major rows: nn.AvgPool1d(7, 3) and F.interpolate(mode='linear') over time, against torch"""
    import torch.nn.functional as F
    from computervision_codes_amd import ops
    x = torch.randn(b, t, c, generator=torch.Generator().manual_seed(5)).to(dtype)
    tol = 1e-6 if dtype == torch.float32 else 2 ** -8
    y = ops.avgpool1d_rows(x.to(cuda), 7, 3)
    ref = F.avg_pool1d(x.float().permute(0, 2, 1), 7, 3).permute(0, 2, 1)
    assert tuple(y.shape) == tuple(ref.shape) == (b, (t - 7) // 3 + 1, c)
    assert (y.float().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    for t_out in (t, 3 * t + 2, max(1, t // 2)):
        u = ops.interp_linear_rows(x.to(cuda), t_out)
        ref = F.interpolate(x.float().permute(0, 2, 1), size=t_out, mode="linear").permute(0, 2, 1)
        assert tuple(u.shape) == (b, t_out, c)
        assert (u.float().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        if t_out == t:
            assert torch.equal(u.cpu(), x)                 # to the same length: the identity (what the non-hier FPN relies on)
