"""GPU parity proper: the HIP-backed model mirrors vs (a) the golden outputs of the REFERENCE modules and
(b) the CPU oracle, through the C-ABI, on the same synthetic weights and inputs.

Tolerances (BASELINE.json north_star): per-frame logits within 1e-3 in fp32 mode; per-head argmax and
top-5 sets identical.  bf16 mode (throughput mode) is checked at 5e-2 of the logit range and reported
as such -- 53 stacked bf16 convs cannot hold 1e-3 (SURVEY 7 hard part (ii))."""
import types

import numpy as np
import pytest
import torch

from computervision_codes_amd import shapes, synth
from conftest import load_golden

pytestmark = pytest.mark.gpu

TENCO = ["tenco_tiny", "tenco_ragged", "tenco_config1", "tenco_4stage", "tenco_hier", "tenco_hier_4stage"]
CNN = ["cnn_resnet18_odd", "cnn_resnet50_small", "cnn_resnet18_224", "cnn_resnet50_224", "cnn_resnet50_256x448"]


def _maxerr(a, ref):
    return (a.detach().float().cpu() - torch.from_numpy(np.asarray(ref))).abs().max().item()


def _same_topk(a, ref, k):
    a = a.detach().float().cpu()
    ref = torch.from_numpy(np.asarray(ref))
    k = min(k, a.shape[-1])
    ia = a.topk(k, dim=-1).indices.sort(dim=-1).values
    ir = ref.topk(k, dim=-1).indices.sort(dim=-1).values
    return torch.equal(ia, ir)


@pytest.mark.parametrize("name", TENCO)
def test_tenco_vs_reference_golden(cuda, name):
    from computervision_codes_amd.temporal_tenco import VideoNas
    z, cfg = load_golden(name)
    args = types.SimpleNamespace(fpn=cfg["fpn"], output=False, hier=cfg.get("hier", False), mask=True)   # hier: pooled levels of their own lengths
    m = VideoNas(args, cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"], 100).eval()
    table = shapes.tenco_shapes(cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"], 100,
                                fpn=cfg["fpn"])
    m.load_state_dict(synth.fill_from_shapes(table, seed=cfg["seed"]))
    x = synth.synthetic_features(cfg["T"], cfg["dim"], seed=cfg["seed"]).to(cuda)
    out = m(x, False)
    torch.cuda.synchronize()
    for gi, g in enumerate(("ivt", "i", "v", "t")):
        assert len(out[gi]) == sum(1 for k in z.files if k.startswith(f"logit_{g}_"))
        for li, o in enumerate(out[gi]):
            ref = z[f"logit_{g}_{li}"]
            assert tuple(o.shape) == ref.shape
            assert _maxerr(o, ref) < 1e-3, (g, li, _maxerr(o, ref))
            # per-frame argmax over classes (dim 1 of [1,K,T]) bit-exact
            assert torch.equal(o.float().cpu().argmax(1), torch.from_numpy(ref).argmax(1))
    for li, f in enumerate(out[4]):
        if f"feat_{li}" in z:
            assert _maxerr(f, z[f"feat_{li}"]) < 1e-3
        else:
            flat = f.float().cpu().contiguous().flatten()
            assert _maxerr(flat[:: max(1, flat.numel() // 4096)], z[f"feat_{li}_sample"]) < 1e-3


@pytest.mark.parametrize("name", ["tenco_config1", "tenco_4stage", "tenco_ragged"])
def test_tenco_bf16_argmax_agreement_vs_reference_golden(cuda, name):
    """The per-frame readouts of the mode `bench.py` quotes `ms_per_video_bf16` in, against the REFERENCE goldens, every head and FPN level:
    DECLARED logit error <= 2.5 % of the level's logit range (measured <= 1.5 %), per-frame argmax agreement >= 94 % and top-5 set agreement
    >= 90 % of the frames (measured 96.1-100 % / 92.6-100 %, profiles/r03_bf16_agreement_probe.txt; the synthetic weights leave a median
    top-1 margin of 0.1-0.7 against a bf16 logit error of 0.03-0.06, so the disagreeing frames are near-ties).  The 1e-3 / bit-exact-argmax
    claim belongs to the fp32 mode (test_tenco_fp32_vs_reference_golden)."""
    from computervision_codes_amd.temporal_tenco import VideoNas
    z, cfg = load_golden(name)
    args = types.SimpleNamespace(fpn=cfg["fpn"], output=False, hier=False, mask=True)
    m = VideoNas(args, cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"], 100, dtype=torch.bfloat16).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"],
                                                                 100, fpn=cfg["fpn"]), seed=cfg["seed"]))
    out = m(synth.synthetic_features(cfg["T"], cfg["dim"], seed=cfg["seed"]).to(cuda), False)
    seen = 0
    for gi, g in enumerate(("ivt", "i", "v", "t")):
        for li, o in enumerate(out[gi]):
            ref = torch.from_numpy(z[f"logit_{g}_{li}"])[0].t()                    # [T, K]
            got = o.float().cpu()[0].t()
            assert (got - ref).abs().max().item() <= 2.5e-2 * ref.abs().max().item(), (g, li)
            am, t5 = _agreement(got, ref.numpy())
            assert am >= 0.94 and t5 >= 0.90, (name, g, li, am, t5)
            seen += 1
    assert seen == (16 if cfg["fpn"] else 1)


@pytest.mark.parametrize("name", ["tenco_config1", "tenco_4stage"])
def test_tenco_bf16_mode(cuda, name):
    """throughput mode of the TCN: bf16 weights / activations, fp32 accumulate, fp32 logits -- within 5 % of the logit range of the
    reference's fp32 result (the 1e-3 parity claim belongs to the fp32 mode)"""
    from computervision_codes_amd.temporal_tenco import VideoNas
    z, cfg = load_golden(name)
    args = types.SimpleNamespace(fpn=cfg["fpn"], output=False, hier=False, mask=True)
    m = VideoNas(args, cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"], 100, dtype=torch.bfloat16).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.tenco_shapes(cfg["num_layers_PG"], cfg["num_layers_R"], cfg["num_R"], cfg["num_f_maps"], cfg["dim"],
                                                                 100, fpn=cfg["fpn"]), seed=cfg["seed"]))
    x = synth.synthetic_features(cfg["T"], cfg["dim"], seed=cfg["seed"]).to(cuda)
    out = m(x, False)
    ref = z["logit_ivt_0"]
    assert out[0][0].dtype == torch.float32 and tuple(out[0][0].shape) == ref.shape
    rng = float(np.abs(ref).max())
    assert _maxerr(out[0][0], ref) < 5e-2 * rng, (_maxerr(out[0][0], ref), rng)


def test_tenco_long_video_vs_oracle(cuda):
    """T = 1500 (a full-video length, not in the fixtures): HIP vs the CPU oracle directly."""
    from computervision_codes_amd.temporal_tenco import VideoNas
    from oracle import tenco as o_tenco
    cfgk = dict(num_layers_PG=11, num_layers_R=10, num_R=3)
    args = types.SimpleNamespace(fpn=True, output=False, hier=False, mask=True)
    table = shapes.tenco_shapes(11, 10, 3, 64, 64, 100, fpn=True)
    sd = synth.fill_from_shapes(table, seed=3)
    m = VideoNas(args, 11, 10, 3, 64, 64, 100).eval().load_state_dict(sd)
    x = synth.synthetic_features(1500, 64, seed=3)
    with torch.no_grad():
        ref = o_tenco.tenco_forward(sd, x, fpn=True, **cfgk)
    out = m(x.to(cuda), False)
    for gi in range(4):
        for o, r in zip(out[gi], ref[gi]):
            assert _maxerr(o, r.numpy()) < 1e-3


def _cnn_model(cfg, dtype):
    from computervision_codes_amd.spatial_cnn import VideoNas
    args = types.SimpleNamespace(network=cfg["network"], loss_type="all", student_dim=shapes.resnet_feat_dim(cfg["network"]),
                                 teacher_dim=1536, train=False)
    m = VideoNas(args=args, dtype=dtype).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes(cfg["network"]), seed=cfg["seed"]))
    return m


@pytest.mark.parametrize("name", CNN)
def test_spatial_cnn_fp32_vs_reference_golden(cuda, name):
    z, cfg = load_golden(name)
    m = _cnn_model(cfg, torch.float32)
    frames = synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"])
    img = synth.normalize_frames(frames).to(cuda)
    (k0, li), (k1, lv), (k2, lt), (feat, livt) = m(img)
    assert k0 == 0 and k1 == 0 and k2 == 0
    for o, key, k in ((li, "logit_i", 6), (lv, "logit_v", 10), (lt, "logit_t", 15), (livt, "logit_ivt", 100)):
        assert tuple(o.shape) == z[key].shape
        assert _maxerr(o, z[key]) < 1e-3, (key, _maxerr(o, z[key]))
        assert torch.equal(o.float().cpu().argmax(1), torch.from_numpy(z[key]).argmax(1))
        assert _same_topk(o, z[key], 5)
    assert _maxerr(feat, z["feat"]) < 1e-3
    # uint8 fast path: same numbers (normalisation done on the GPU)
    (_, _), (_, _), (_, _), (feat2, livt2) = m.extract_u8(frames.to(cuda))
    assert _maxerr(livt2, z["logit_ivt"]) < 1e-3 and _maxerr(feat2, z["feat"]) < 1e-3


@pytest.mark.parametrize("name", ["cnn_resnet50_small", "cnn_resnet50_224", "cnn_resnet18_224"])
def test_spatial_cnn_bf16_mode(cuda, name):
    """throughput mode: bf16 activations/weights, fp32 accumulate + fp32 epilogue.  Tolerance 5e-2 of the
    logit range (NOT the 1e-3 parity claim, which the fp32 mode carries)."""
    z, cfg = load_golden(name)
    m = _cnn_model(cfg, torch.bfloat16)
    frames = synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"]).to(cuda)
    (_, li), (_, lv), (_, lt), (feat, livt) = m.extract_u8(frames)
    for o, key in ((li, "logit_i"), (lv, "logit_v"), (lt, "logit_t"), (livt, "logit_ivt")):
        rng = float(np.abs(z[key]).max())
        assert _maxerr(o, z[key]) < 5e-2 * rng, (key, _maxerr(o, z[key]), rng)
    assert _maxerr(feat, z["feat"]) < 5e-2 * float(np.abs(z["feat"]).max())


@pytest.mark.parametrize("name", ["cnn_resnet50_small", "cnn_resnet50_224", "cnn_resnet18_224", "cnn_resnet50_256x448"])
def test_spatial_cnn_bf16_mode_vs_rounding_emulating_oracle(cuda, name):
    """The bf16 throughput mode against an oracle run that rounds to bf16 exactly where the kernels do (`oracle.spatial_cnn.
    resnet_trunk_bf16_emulation`: folded weights, the normalised frame, every stored activation; conv3 + downsample of the strided Bottlenecks as
    one fp32 sum) and is fp32 otherwise.  What remains is the fp32 summation order inside a convolution -- and the bf16 rounding ties it flips:
    ~0.2 % of a layer's activations land within the summation error of a rounding midpoint and come out one bf16 ulp apart, which over 50
    layers is a noise floor of ~1e-3 of the output range (measured 7e-4 - 2e-3 on the 224 x 224 / 256 x 448 fixtures, 2.5e-3 - 4.6e-3 on the
    64 x 96 one, whose last maps are 2 x 3 pixels: profiles/r04_bf16_emulation_probe.txt).  DECLARED: 3e-3 of the logit / feature range on the
    full-size fixtures (6e-3 on the small one) -- 15 x tighter than the 5e-2 the mode is given against the fp32 reference: a wrong tap, a
    dropped residual or a branch handled differently from the kernels shows up here.  On the full-size fixtures the emulation also EXPLAINS
    the mode's deviation from the fp32 reference: hip-vs-emulation stays under 0.6 x hip-vs-reference on every output (measured 0.2 - 0.4)."""
    from oracle import spatial_cnn as o_cnn
    z, cfg = load_golden(name)
    m = _cnn_model(cfg, torch.bfloat16)
    frames = synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"])
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes(cfg["network"]), seed=cfg["seed"])
    with torch.no_grad():
        emu = o_cnn.spatial_cnn_forward(sd, synth.normalize_frames(frames), cfg["network"], emulate_bf16=True)
    out = m.extract_u8(frames.to(cuda))
    full = min(cfg["H"], cfg["W"]) >= 224
    for got, want, key in ((out[0][1], emu[0][1], "logit_i"), (out[1][1], emu[1][1], "logit_v"), (out[2][1], emu[2][1], "logit_t"),
                           (out[3][1], emu[3][1], "logit_ivt"), (out[3][0], emu[3][0], "feat")):
        rng = float(np.abs(z[key]).max())
        d_emu = (got.float().cpu() - want).abs().max().item()
        d_ref = _maxerr(got, z[key])
        assert d_emu <= (3e-3 if full else 6e-3) * rng, (key, d_emu / rng)
        if full:
            assert d_emu < 0.6 * d_ref, (key, d_emu / rng, d_ref / rng)


def _agreement(a, ref):
    a, ref = a.float().cpu(), torch.as_tensor(np.asarray(ref)).float()
    k = min(5, a.shape[1])
    argmax = float((a.argmax(1) == ref.argmax(1)).float().mean())
    top5 = float((a.topk(k, 1).indices.sort(1).values == ref.topk(k, 1).indices.sort(1).values).all(1).float().mean())
    return argmax, top5


def test_bf16_argmax_top5_agreement(cuda):
    """The discrete readouts in the mode the headline number is quoted in.  bf16 extraction against the REFERENCE goldens: per-head argmax
    and top-5 index sets identical on every golden frame (measured 100 % on all four fixtures, tools/argmax_probe.py); and at
    bench.py's full step size, 1336 distinct frames, bf16 against the fp32 parity mode: measured 100 % / 100 % on every head (median
    top-1 margin of the triplet head 1.86 against a bf16 logit error <= 0.08); asserted floor 99 % argmax, 98 % top-5 sets."""
    for name in ("cnn_resnet50_224", "cnn_resnet18_224", "cnn_resnet50_256x448", "cnn_resnet50_small"):
        z, cfg = load_golden(name)
        out = _cnn_model(cfg, torch.bfloat16).extract_u8(synth.synthetic_frames(cfg["B"], cfg["H"], cfg["W"], seed=cfg["seed"]).to(cuda))
        for gi, key in enumerate(("logit_i", "logit_v", "logit_t", "logit_ivt")):
            assert _agreement(out[gi][1], z[key]) == (1.0, 1.0), (name, key, _agreement(out[gi][1], z[key]))
    n = 1336
    _, cfg = load_golden("cnn_resnet50_224")
    base = synth.synthetic_frames(64, 224, 224, seed=11).to(cuda)
    idx = torch.arange(n, device=cuda)
    frames = (base[idx % 64] ^ ((idx // 64 * 37) % 256).to(torch.uint8)[:, None, None, None]).contiguous()
    o16 = _cnn_model(cfg, torch.bfloat16).extract_u8(frames)
    m32 = _cnn_model(cfg, torch.float32)
    for gi in range(4):
        l32 = torch.cat([m32.extract_u8(frames[s:s + 167].contiguous())[gi][1] for s in range(0, n, 167)])
        am, t5 = _agreement(o16[gi][1], l32.cpu().numpy())
        assert am >= 0.99 and t5 >= 0.98, (gi, am, t5)


def test_spatial_cnn_fused_layer1_bottlenecks_are_bit_identical(cuda):
    """ResNet-50 bf16 (the default path): every layer1 Bottleneck is ONE launch (`mt4_bottleneck_fused_bf16`: conv1 + conv2 + conv3 (+ downsample),
    intermediates in LDS); features and logits equal the one-launch-per-conv path bit for bit at 224x224, the reference's 256x448 and a small
    ragged size; the launch grouping bench.py accounts with matches what runs"""
    from computervision_codes_amd import ops
    _, cfg = load_golden("cnn_resnet50_224")
    m = _cnn_model(cfg, torch.bfloat16)
    assert m.fuse_bottleneck
    for (n, h, w) in ((5, 224, 224), (3, 256, 448), (2, 64, 96), (1, 36, 60)):
        frames = synth.synthetic_frames(n, h, w, seed=13).to(cuda)
        calls = []
        orig = ops.bottleneck_fused
        ops.bottleneck_fused = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
        try:
            a = m.extract_u8(frames)
        finally:
            ops.bottleneck_fused = orig
        assert len(calls) == 2      # (the third layer1 block runs through ops.bottleneck_fused_next)
        m.fuse_bottleneck = False
        b = m.extract_u8(frames)
        m.fuse_bottleneck = True
        assert torch.equal(a[3][0], b[3][0]) and all(torch.equal(a[i][1], b[i][1]) for i in range(4)), (n, h, w)
    groups = m.launch_groups(224, 224)
    # (+ the heads GEMM = 36 launches; the three strided blocks run conv3 + downsample as one launch, layer2.0's conv1 rides in layer1.2's launch,
    #  conv3 + the next block's conv1 of the identity blocks of layers 2 and 3 are one `mt4_chain_gemm_bf16` launch each: 7 pairs)
    assert len(groups) == 35 and [len(g) for g in groups[1:4]] == [4, 3, 4] and sum(len(g) for g in groups) == 53 and sum(len(g) == 2 for g in groups) == 10
    m.chain_layers = ()        # round 2's grouping: layer2's three identity blocks run conv2 + conv3 as one launch
    g2 = m.launch_groups(224, 224)
    assert len(g2) == 39 and sum(len(g) for g in g2) == 53 and sum(len(g) == 2 for g in g2) == 6


def test_spatial_cnn_batch_independence(cuda):
    """frames are independent units (the multi-GPU sharding premise): a frame's output does not depend on
    its batch neighbours or its position in the batch -- bit-exact."""
    _, cfg = load_golden("cnn_resnet18_odd")
    m = _cnn_model(cfg, torch.float32)
    frames = synth.synthetic_frames(6, 64, 64, seed=77).to(cuda)
    (_, _), (_, _), (_, _), (feat_all, l_all) = m.extract_u8(frames)
    (_, _), (_, _), (_, _), (feat_2, l_2) = m.extract_u8(frames[4:6].contiguous())
    assert torch.equal(feat_all[4:6], feat_2) and torch.equal(l_all[4:6], l_2)


def test_extract_video_device_prefetch_matches_serial(cuda):
    """the per-video loop of the extraction driver with the next span loaded on the helper thread: same bytes as the serial loop, spans
    requested in file order, also from inside a non-default stream"""
    import threading
    from computervision_codes_amd import extract
    _, cfg = load_golden("cnn_resnet18_odd")
    m = _cnn_model(cfg, torch.bfloat16)
    host = synth.synthetic_frames(21, 64, 64, seed=5)
    for side in (False, True):
        asked, tids = [], set()

        def load(s, e):
            asked.append((s, e)); tids.add(threading.get_ident())
            return (host[s:e].to(cuda, non_blocking=True).float() * 1.0).to(torch.uint8)   # a few launches on the caller's stream
        ctx = torch.cuda.stream(torch.cuda.Stream()) if side else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            a = extract.extract_video_device(m, 21, load, device_batch=8, prefetch=False)
            serial_tids = set(tids); tids.clear(); first = list(asked); asked.clear()
            b = extract.extract_video_device(m, 21, load, device_batch=8, prefetch=True)
        assert first == asked == [(0, 8), (8, 16), (16, 21)]
        assert serial_tids == {threading.get_ident()} and threading.get_ident() not in tids
        assert np.array_equal(a[0], b[0]) and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
        assert a[0].shape[0] == 21
        asked.clear()
        c = extract.extract_video_device(m, 21, load, device_batch=4, load_batch=10, prefetch=2)      # loads of two passes (8 frames), two ahead
        assert sorted(asked) == [(0, 8), (8, 16), (16, 21)]
        assert np.array_equal(a[0], c[0]) and all(np.array_equal(x, y) for x, y in zip(a[1], c[1]))


def test_spatial_cnn_bench_configuration_properties(cuda):
    """BASELINE configs[1] at bench.py's full size (ResNet-50, bf16, 1336 frames of 224x224 per step: the 8-wave 256x256 /
    256x128 tiles): size-independent properties instead of an oracle run -- a frame's feature does not depend on the batch it
    rides in (bit-exact against batches of 8, which take other tile instantiations), the step is deterministic, and the bf16
    features of sampled frames agree with the fp32 parity path."""
    _, cfg = load_golden("cnn_resnet50_224")
    m16 = _cnn_model(cfg, torch.bfloat16)
    n = 1336
    frames = synth.synthetic_frames(16, 224, 224, seed=5).to(cuda).repeat(n // 16 + 1, 1, 1, 1)[:n].contiguous()
    frames[100:108] = synth.synthetic_frames(8, 224, 224, seed=6).to(cuda)          # break the periodicity in two places
    frames[n - 8:] = synth.synthetic_frames(8, 224, 224, seed=7).to(cuda)
    (_, _), (_, _), (_, _), (feat, livt) = m16.extract_u8(frames)
    (_, _), (_, _), (_, _), (feat_b, livt_b) = m16.extract_u8(frames)
    assert torch.equal(feat, feat_b) and torch.equal(livt, livt_b)
    for s0 in (0, 100, n - 8):
        (_, _), (_, _), (_, _), (f8, l8) = m16.extract_u8(frames[s0:s0 + 8].contiguous())
        assert torch.equal(feat[s0:s0 + 8], f8) and torch.equal(livt[s0:s0 + 8], l8), s0
    assert torch.equal(feat[0], feat[16]) and not torch.equal(feat[100], feat[116])  # same frame -> same feature, different -> different
    m32 = _cnn_model(cfg, torch.float32)
    (_, _), (_, _), (_, _), (f32_, l32) = m32.extract_u8(frames[n - 8:].contiguous())
    rng = l32.abs().max().item()
    assert (livt[n - 8:].float() - l32).abs().max().item() < 5e-2 * rng
    assert (feat[n - 8:].float() - f32_).abs().max().item() < 5e-2 * f32_.abs().max().item()


def test_spatial_cnn_bench_step_2672_frames_one_stream(cuda):
    """The step `bench.py` quotes the headline on, as bench.py builds it: ONE stream of 2672 uint8 frames of 224x224 (`bench.parse` default
    --batch, `bench.device_frames`, seed 1234, the same weights), ResNet-50 bf16.  At this size the stem / layer1 maps are 4.29 GB
    (2672 * 56 * 56 * 256 * 2 B = 4 290 248 704 B, 4.7 MB under 2^32) -- the fused launches' 32-bit offsets are exercised where they are tightest.
    * deterministic: two passes give identical bytes;
    * a frame's outputs do not depend on the batch it rides in: bit-exact against batches of 8 (other tile instantiations) at the start,
      across the 2 GiB and 4 GiB byte marks of the layer1 maps, and at the end;
    * the fused launches (stem + pool, layer1 Bottlenecks incl. the one carrying layer2.0's conv1, layer2 conv2 + conv3, the chained
      conv3 -> next conv1 of layers 2 / 3) are bit-identical to one launch per conv AT THIS SIZE (feature + all four logit sets, all 2672 frames);
    * sampled frames agree with the fp32 parity mode (5e-2 of range, argmax / top-5 equal on >= 99 % / 98 %)."""
    import bench
    from computervision_codes_amd import ops
    import sys
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        ba = bench.parse()
    finally:
        sys.argv = argv
    n = ba.batch * max(1, ba.streams)                # the frames of one step on one GPU, as bench.main computes them
    assert (n, ba.streams, ba.network, ba.height, ba.width, ba.dtype) == (2672, 1, "resnet50", 224, 224, "bf16")
    _, cfg = load_golden("cnn_resnet50_224")
    cfg = dict(cfg, seed=1234)                       # bench.py's weights
    m16 = _cnn_model(cfg, torch.bfloat16)
    frames = bench.device_frames(n, 224, 224, 1234, cuda, nbase=64)
    assert frames.shape == (n, 224, 224, 3) and n * 56 * 56 * 256 * 2 < 2 ** 32 < (n + 3) * 56 * 56 * 256 * 2
    calls = {}
    names = ("bottleneck_fused", "bottleneck_fused_next", "stem_maxpool", "chain_gemm", "conv3x3_expand")
    origs = {k: getattr(ops, k) for k in names}
    for k in names:
        setattr(ops, k, (lambda k_: lambda *a_, **kw_: (calls.__setitem__(k_, calls.get(k_, 0) + 1), origs[k_](*a_, **kw_))[1])(k))
    try:
        out = m16.extract_u8(frames)
    finally:
        for k in names:
            setattr(ops, k, origs[k])
    # what ran is the bench's launch set: stem+pool, 2 + 1 fused layer1 blocks, 7 chained pairs (layers 2 and 3)
    assert calls == {"stem_maxpool": 1, "bottleneck_fused": 2, "bottleneck_fused_next": 1, "chain_gemm": 7}, calls
    feat, logits = out[3][0], [out[i][1] for i in range(4)]
    assert torch.isfinite(feat).all() and all(torch.isfinite(l).all() for l in logits)
    out_b = m16.extract_u8(frames)
    assert torch.equal(feat, out_b[3][0]) and all(torch.equal(l, out_b[i][1]) for i, l in enumerate(logits))
    del out_b
    # frames whose layer1 rows straddle the 2 GiB / 4 GiB byte offsets: frame f starts at f * 1 605 632 B
    per_frame = 56 * 56 * 256 * 2
    marks = sorted({0, 2 ** 31 // per_frame - 4, 1336 - 4, 2 ** 32 // per_frame - 8, n - 8})
    for s0 in marks:
        o8 = m16.extract_u8(frames[s0:s0 + 8].contiguous())
        assert torch.equal(feat[s0:s0 + 8], o8[3][0]), s0
        assert all(torch.equal(logits[i][s0:s0 + 8], o8[i][1]) for i in range(4)), s0
    assert not torch.equal(feat[0], feat[64]) and not torch.equal(feat[0], feat[n - 1])     # (no two frames are equal)
    # every fused launch against one launch per conv, at this size
    saved = (m16.fuse_stem_pool, m16.fuse_expand, m16.fuse_next_block, m16.fuse_bottleneck, m16.chain_layers)
    try:
        m16.fuse_stem_pool = m16.fuse_expand = m16.fuse_next_block = m16.fuse_bottleneck = False
        m16.chain_layers = ()
        calls.clear()
        for k in names:
            setattr(ops, k, (lambda k_: lambda *a_, **kw_: (calls.__setitem__(k_, calls.get(k_, 0) + 1), origs[k_](*a_, **kw_))[1])(k))
        ref = m16.extract_u8(frames)
        assert calls == {}, calls
        assert torch.equal(feat, ref[3][0]) and all(torch.equal(l, ref[i][1]) for i, l in enumerate(logits))
        del ref
        # round 2's grouping (conv2 + conv3 of layer2's identity blocks in one launch) as well: it is what MT4_CHAIN=0 / 3 runs
        m16.fuse_stem_pool, m16.fuse_expand, m16.fuse_next_block, m16.fuse_bottleneck = saved[:4]
        calls.clear()
        ref = m16.extract_u8(frames)
        assert calls.get("conv3x3_expand") == 3 and "chain_gemm" not in calls, calls
        assert torch.equal(feat, ref[3][0]) and all(torch.equal(l, ref[i][1]) for i, l in enumerate(logits))
        del ref
    finally:
        for k in names:
            setattr(ops, k, origs[k])
        m16.fuse_stem_pool, m16.fuse_expand, m16.fuse_next_block, m16.fuse_bottleneck, m16.chain_layers = saved
    torch.cuda.empty_cache()
    # sampled frames against the fp32 parity mode (itself pinned to the reference goldens at 1e-3)
    m32 = _cnn_model(cfg, torch.float32)
    pick = torch.tensor(sorted({(i * 167 + 5) % n for i in range(64)} | {0, n - 1}), device=cuda)
    o32 = m32.extract_u8(frames[pick].contiguous())
    f32_ = o32[3][0]
    assert (feat[pick].float() - f32_).abs().max().item() < 5e-2 * f32_.abs().max().item()
    for gi in range(4):
        l32 = o32[gi][1]
        assert (logits[gi][pick].float() - l32).abs().max().item() < 5e-2 * l32.abs().max().item(), gi
        am, t5 = _agreement(logits[gi][pick], l32.cpu().numpy())
        assert am >= 0.99 and t5 >= 0.98, (gi, am, t5)


# ------------------------------------------------------------------------------------------ Swin + Q2L, MS-TCT
Q2L = ["q2l_swinT_224_i", "q2l_swinB_224_v", "q2l_swinB_384_t", "q2l_swinL_384_i"]   # (the last one: the SHIPPED teacher, Scripts/train_fold1.sh:5-12)
MSTCT = ["mstct_tiny", "mstct_full_i", "mstct_full_ivt_ragged", "mstct_D1536_i"]      # (the last one: --input_dim 1536, Scripts/train_fold1.sh:16)


def _q2l_model(cfg, dtype):
    from computervision_codes_amd.spatial_transformer import build_q2l
    args = types.SimpleNamespace(backbone=cfg["backbone"], img_size=cfg["img"], hidden_dim=cfg["hidden"], loss_type=cfg["loss_type"])
    m = build_q2l(args, dtype=dtype).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.q2l_param_shapes(cfg["backbone"], cfg["img"], cfg["hidden"], cfg["loss_type"]), seed=cfg["seed"]))
    return m


@pytest.mark.parametrize("name", Q2L)
def test_q2l_fp32_vs_reference_golden(cuda, name):
    z, cfg = load_golden(name)
    m = _q2l_model(cfg, torch.float32)
    frames = synth.synthetic_frames(cfg["B"], cfg["img"], cfg["img"], seed=cfg["seed"])
    src = m.forward_features(synth.normalize_frames(frames).to(cuda))
    b = cfg["B"]
    # the reference hands [B,C,h,h] to the decoder: same memory order as rows -> compare the strided sample
    flat = src.view(b, -1, src.shape[-1]).permute(0, 2, 1).contiguous().flatten().cpu()
    assert _maxerr(flat[:: max(1, flat.numel() // 8192)], z["src_sample"]) < 1e-3
    out = m(synth.normalize_frames(frames).to(cuda))
    gi = {"i": 0, "v": 1, "t": 2}[cfg["loss_type"]]
    y, feat = out[gi][1], out[3][0]
    assert tuple(y.shape) == z["logits"].shape
    assert _maxerr(y, z["logits"]) < 1e-3 and _maxerr(feat, z["feat"]) < 1e-3, (_maxerr(y, z["logits"]), _maxerr(feat, z["feat"]))
    assert torch.equal(y.cpu().argmax(1), torch.from_numpy(z["logits"]).argmax(1))
    out2 = m(frames.to(cuda))   # uint8 path: normalisation fused into patch extraction
    assert _maxerr(out2[gi][1], z["logits"]) < 1e-3


@pytest.mark.parametrize("name", ["q2l_swinT_224_all", "q2l_swinB_384_all"])
def test_q2l_loss_type_all_with_kd_vs_reference_golden(cuda, name):
    """four decoders over the shared transformer + the always-on KD mixing (`Spatial_transformer/network.py:98-124`); the Swin-B/384 case
    is BASELINE configs[2] as a composite (Swin-B + the K = 100 triplet head beside the three component heads)"""
    z, cfg = load_golden(name)
    m = _q2l_model(cfg, torch.float32)
    frames = synth.synthetic_frames(cfg["B"], cfg["img"], cfg["img"], seed=cfg["seed"])
    tf = [synth.synthetic_features(cfg["B"], 512, seed=cfg["seed"] + k)[0].to(cuda) for k in (1, 2, 3)]
    (kd_i, yi), (kd_v, yv), (kd_t, yt), (feat, yivt) = m(synth.normalize_frames(frames).to(cuda), *tf)
    for got, key in ((yi, "logit_i"), (yv, "logit_v"), (yt, "logit_t"), (yivt, "logit_ivt"), (feat, "feat"), (kd_i, "kd_i"), (kd_v, "kd_v"),
                     (kd_t, "kd_t")):
        assert tuple(got.shape) == z[key].shape and _maxerr(got, z[key]) < 1e-3, (key, _maxerr(got, z[key]))
    for got, key in ((yi, "logit_i"), (yv, "logit_v"), (yt, "logit_t"), (yivt, "logit_ivt")):
        assert torch.equal(got.float().cpu().argmax(1), torch.from_numpy(z[key]).argmax(1)) and _same_topk(got, z[key], 5), key
    with pytest.raises(TypeError):
        m(frames.to(cuda))


def test_q2l_all_batched_encoder_equals_per_task_decoders(cuda):
    """`decode_all` (the shared encoder layer once over the four tasks' tokens stacked along the batch axis) against four `decode` calls:
    identical bits, fp32 and bf16"""
    _, cfg = load_golden("q2l_swinB_384_all")
    for dt in (torch.float32, torch.bfloat16):
        m = _q2l_model(cfg, dt)
        frames = synth.synthetic_frames(3, cfg["img"], cfg["img"], seed=5).to(cuda)
        tf = [synth.synthetic_features(3, 512, seed=6 + k)[0].to(cuda) for k in (1, 2, 3)]
        assert m.batch_decoders
        a = m(frames, *tf)
        m.batch_decoders = False
        b = m(frames, *tf)
        for x, y in zip(a, b):
            assert torch.equal(x[1], y[1]) and (torch.equal(x[0], y[0]) if torch.is_tensor(x[0]) else x[0] == y[0])


def test_q2l_swinB_384_bench_batch_rows_equal_small_batch_rows(cuda):
    """configs[2] as `bench.py` runs it (Swin-B/384, four decoders, bf16, bench.SWIN_BATCH frames per forward: whole rounds of 256 x 256 tiles):
    a frame's logits and features do not depend on what shares its forward -- frames sampled across the big batch equal the same frames run three at a
    time (those small batches are what the golden / agreement tests above pin to the reference)"""
    import bench
    _, cfg = load_golden("q2l_swinB_384_all")
    m = _q2l_model(cfg, torch.bfloat16)
    n = bench.SWIN_BATCH[384]
    base = synth.synthetic_frames(16, cfg["img"], cfg["img"], seed=11).to(cuda)
    frames = base.repeat((n + 15) // 16, 1, 1, 1)[:n].contiguous()
    frames[n - 1] = synth.synthetic_frames(1, cfg["img"], cfg["img"], seed=12).to(cuda)[0]
    tf = [synth.synthetic_features(n, 512, seed=13 + k)[0].to(cuda) for k in (1, 2, 3)]
    big = m(frames, *tf)
    torch.cuda.synchronize()
    for idx in ([0, 1, 2], [111, 112, 113], [n - 3, n - 2, n - 1]):
        small = m(frames[idx].contiguous(), *[t[idx].contiguous() for t in tf])
        for a, b in zip(big, small):
            assert torch.equal(a[1][idx], b[1]), idx
        assert torch.equal(big[3][0][idx], small[3][0])
    assert all(torch.isfinite(o[1]).all() for o in big)


def test_q2l_swinB_384_all_bf16_vs_reference_golden(cuda):
    """BASELINE configs[2] in the mode `bench.py` quotes it in (Swin-B/384, `loss_type all`, bf16) against the REFERENCE golden: DECLARED logit
    error <= 4 % of each head's logit range, feature error <= 2 % of its range, per-head argmax and top-5 sets equal on the golden frame (measured
    equal); and on 24 more frames against the fp32 parity mode (itself pinned to the reference at 1e-3): per-frame logit error <= 2 % (median) /
    3.5 % (worst frame) of the head's range, argmax agreement >= 90 %, top-5 set agreement >= 85 % per head (measured 95.8-100 % / 91.7-100 %,
    profiles/r03_bf16_agreement_probe.txt; head i has a median top-1 margin of 0.08 on these weights against a bf16 error of 0.02).
    The bf16 error is a distribution, not a constant: over the 24 frames head i (the narrowest logit range, 0.77) measures 0.8 ... 2.3 % per frame
    (median 1.3 %), the other heads 0.3 ... 1.1 %; the golden frame itself read 1.7 % in round 3 and 3.1 % in round 4 after one-ulp changes in the
    attention core (rounding pattern, not accuracy: the kernel's error against float64 is unchanged,
    test_window_attention_from_relative_table_equals_expanded_tables) -- hence 4 % for a single frame and the per-frame statistics beside it."""
    z, cfg = load_golden("q2l_swinB_384_all")
    m16, m32 = _q2l_model(cfg, torch.bfloat16), _q2l_model(cfg, torch.float32)
    frames = synth.synthetic_frames(cfg["B"], cfg["img"], cfg["img"], seed=cfg["seed"]).to(cuda)
    tf = [synth.synthetic_features(cfg["B"], 512, seed=cfg["seed"] + k)[0].to(cuda) for k in (1, 2, 3)]
    out = m16(frames, *tf)
    for gi, key in enumerate(("logit_i", "logit_v", "logit_t", "logit_ivt")):
        assert _maxerr(out[gi][1], z[key]) <= 4e-2 * float(np.abs(z[key]).max()), key
        assert _agreement(out[gi][1], z[key]) == (1.0, 1.0), (key, _agreement(out[gi][1], z[key]))
    assert _maxerr(out[3][0], z["feat"]) <= 2e-2 * float(np.abs(z["feat"]).max())
    n = 24
    fr = synth.synthetic_frames(n, cfg["img"], cfg["img"], seed=77).to(cuda)
    tfn = [synth.synthetic_features(n, 512, seed=80 + k)[0].to(cuda) for k in (1, 2, 3)]
    o16 = m16(fr, *tfn)
    for gi in range(4):
        l32 = torch.cat([m32(fr[s:s + 8], *[t[s:s + 8] for t in tfn])[gi][1] for s in range(0, n, 8)])
        am, t5 = _agreement(o16[gi][1], l32.float().cpu().numpy())
        assert am >= 0.90 and t5 >= 0.85, (gi, am, t5)
        per = (o16[gi][1].float().cpu() - l32.float().cpu()).abs().max(1).values / l32.float().abs().max().item()
        assert per.median().item() <= 2e-2 and per.max().item() <= 3.5e-2, (gi, per.median().item(), per.max().item())


@pytest.mark.parametrize("name", ["q2l_swinT_224_i", "q2l_swinB_384_t", "q2l_swinL_384_i"])
def test_q2l_bf16_mode(cuda, name):
    z, cfg = load_golden(name)
    m = _q2l_model(cfg, torch.bfloat16)
    out = m(synth.synthetic_frames(cfg["B"], cfg["img"], cfg["img"], seed=cfg["seed"]).to(cuda))
    gi = {"i": 0, "v": 1, "t": 2}[cfg["loss_type"]]
    rng = float(np.abs(z["logits"]).max())
    assert _maxerr(out[gi][1], z["logits"]) < 8e-2 * rng, (_maxerr(out[gi][1], z["logits"]), rng)


@pytest.mark.parametrize("name", MSTCT)
def test_mstct_fp32_vs_reference_golden(cuda, name):
    from computervision_codes_amd.temporal_mstct import VideoNas
    z, cfg = load_golden(name)
    args = types.SimpleNamespace(loss_type=cfg["loss_type"])
    m = VideoNas(args, list(cfg["inter"]), 2, 8, 8, cfg["D"], cfg["final"]).eval()
    m.load_state_dict(synth.fill_from_shapes(shapes.mstct_shapes(cfg["D"], cfg["inter"], 2, 8, cfg["final"], cfg["loss_type"]), seed=cfg["seed"]))
    x = torch.cat([synth.synthetic_features(cfg["T"], cfg["D"], seed=cfg["seed"] + b) for b in range(cfg["B"])], 0)
    out = m(x.permute(0, 2, 1).to(cuda))          # reference call convention [B,D,T]
    gi = {"i": 0, "v": 1, "t": 2, "ivt": 3}[cfg["loss_type"]]
    y, concat = out[gi][0], out[3][1]
    assert tuple(y.shape) == z["logits"].shape and tuple(concat.shape) == (cfg["B"], 4 * cfg["final"], cfg["T"])
    assert _maxerr(y, z["logits"]) < 1e-3, _maxerr(y, z["logits"])
    assert torch.equal(y.cpu().argmax(-1), torch.from_numpy(z["logits"]).argmax(-1))
    flat = concat.contiguous().flatten().cpu()
    assert _maxerr(flat[:: max(1, flat.numel() // 8192)], z["concat_sample"]) < 1e-3
    out2 = m.forward_btd(x.to(cuda))              # frame-major entry (feature-file layout): same numbers
    assert torch.equal(out2[gi][0], y)
    m.fold_layernorm = False                      # norm1 / norm2 as launches of their own (what longer windows and bf16 run): same result to rounding
    y3 = m.forward_btd(x.to(cuda))[gi][0]
    assert _maxerr(y3, z["logits"]) < 1e-3 and _maxerr(y3, y.cpu().numpy()) < 2e-4, _maxerr(y3, y.cpu().numpy())


def test_spatial_cnn_multi_stream_extract_is_byte_identical(cuda):
    """`extract_u8(frames, streams=2)`: parts of the batch on their own HIP streams -- same bytes as the single-stream call"""
    _, cfg = load_golden("cnn_resnet50_224")
    m = _cnn_model(cfg, torch.bfloat16)
    frames = synth.synthetic_frames(12, 96, 128, seed=9).to(cuda)
    a = m.extract_u8(frames)
    for ns in (2, 3):
        b = m.extract_u8(frames, streams=ns)
        assert torch.equal(a[3][0], b[3][0]) and all(torch.equal(x[1], y[1]) for x, y in zip(a, b))


def test_stem_patch_kernel_bit_identical(cuda):
    """the stem patch kernel (tile 33: the tile's span of the space-to-depth frame and the whole weight matrix staged once, no barrier in
    the K loop) walks K like the generic kernel on the same frame: identical bytes; frames too small for it are refused"""
    from computervision_codes_amd import ops
    from computervision_codes_amd.synth import IMAGENET_MEAN, IMAGENET_STD
    _, cfg = load_golden("cnn_resnet50_224")
    m = _cnn_model(cfg, torch.bfloat16)
    for (b, h, w) in ((5, 224, 224), (3, 64, 96), (2, 256, 448), (7, 32, 34), (1, 32, 32), (12, 224, 224)):   # (12 frames: 588 tiles)
        fr = synth.synthetic_frames(b, h, w, seed=h + w).to(cuda)
        xs = ops.preprocess_u8_s2d(fr, IMAGENET_MEAN, IMAGENET_STD)
        kw = dict(kh=4, kw=1, relu=True, run_pixels=4, out_hw=(h // 2, w // 2))
        ref = ops.conv_nhwc(xs, m._p["stem_s2d"], m._p["stem"][1], tile=20, **kw)
        got = ops.conv_nhwc(xs, m._p["stem_s2d"], m._p["stem"][1], tile=33, **kw)
        auto = ops.conv_nhwc(xs, m._p["stem_s2d"], m._p["stem"][1], **kw)
        assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (b, h, w)
        assert torch.equal(ref.view(torch.int16), auto.view(torch.int16)), (b, h, w)
    xs = ops.preprocess_u8_s2d(synth.synthetic_frames(2, 16, 16, seed=1).to(cuda), IMAGENET_MEAN, IMAGENET_STD)   # 8x8 outputs per image
    with pytest.raises(RuntimeError):
        ops.conv_nhwc(xs, m._p["stem_s2d"], m._p["stem"][1], kh=4, kw=1, relu=True, run_pixels=4, out_hw=(8, 8), tile=33)


def test_space_to_depth_stem_equals_padded_stem(cuda):
    """bf16 throughput path: the stem on the space-to-depth frame (4x1 kernel over 128-byte runs, LDS-DMA staging) against the stem on
    the padded pixel-pair frame (register staging): same products, another summation order -> equal to bf16 rounding of the output"""
    from computervision_codes_amd import ops
    from computervision_codes_amd.synth import IMAGENET_MEAN, IMAGENET_STD
    _, cfg = load_golden("cnn_resnet50_224")
    m = _cnn_model(cfg, torch.bfloat16)
    for (h, w) in ((224, 224), (64, 96), (256, 448)):
        fr = synth.synthetic_frames(3, h, w, seed=h).to(cuda)
        xs = ops.preprocess_u8_s2d(fr, IMAGENET_MEAN, IMAGENET_STD)
        y_new = ops.conv_nhwc(xs, m._p["stem_s2d"], m._p["stem"][1], kh=4, kw=1, relu=True, run_pixels=4, out_hw=(h // 2, w // 2))
        xp = ops.preprocess_u8(fr, IMAGENET_MEAN, IMAGENET_STD, torch.bfloat16)
        b, hp, wp_, _ = xp.shape
        y_old = torch.empty_like(y_new)
        ops.conv_nhwc(xp.view(b, hp, wp_ // 2, 8), m._p["stem"][0], m._p["stem"][1], kh=7, kw=4, stride=(2, 1), relu=True, out=y_old)
        d = (y_new.float() - y_old.float()).abs().max().item()
        assert d <= 2 ** -7 * max(1.0, y_old.float().abs().max().item()), (h, w, d)
