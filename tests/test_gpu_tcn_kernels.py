"""GPU: the temporal-head latency path (`mt4_tcn_conv`, `mt4_tcn_dilated_residual_layer`, `mt4_tcn_stage`; csrc/tcn_kernels.hip) against
torch's CPU conv1d -- every comb geometry (dilation below / at / above the tile span and the video length, ragged T, several videos per
call, ragged channel counts), residual + ReLU epilogue, fp32 (exact MFMA chain: tight tolerance) and bf16."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_btc, w_oik, bias, taps, d, res, relu):
    y = F.conv1d(x_btc.permute(0, 2, 1), w_oik, bias, stride=1, padding=d * (taps - 1) // 2, dilation=d).permute(0, 2, 1)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


CASES = [  # (B, T, Cin, Cout, taps, dilation)
    (1, 256, 512, 512, 3, 1), (1, 256, 512, 512, 3, 2), (1, 256, 512, 512, 3, 4), (1, 256, 512, 512, 3, 8), (1, 256, 512, 512, 3, 16),
    (1, 256, 512, 512, 3, 32), (1, 256, 512, 512, 3, 64), (1, 256, 512, 512, 3, 128), (1, 256, 512, 512, 3, 256), (1, 256, 512, 512, 3, 1024),
    (1, 256, 512, 512, 1, 1), (1, 256, 2048, 512, 1, 1), (1, 256, 512, 131, 1, 1), (1, 256, 512, 100, 1, 1),
    (1, 77, 64, 64, 3, 16), (1, 77, 64, 64, 3, 64), (1, 77, 64, 64, 3, 128), (1, 1, 64, 48, 3, 1), (1, 5, 32, 16, 3, 2), (1, 33, 96, 20, 3, 3),
    (2, 120, 64, 64, 3, 8), (3, 50, 128, 131, 3, 5), (2, 300, 64, 32, 3, 64), (1, 1500, 64, 64, 3, 512), (1, 1000, 64, 64, 3, 4), (2, 40, 32, 7, 1, 1),
    # wide 1-tap layers (MS-TCT's nn.Linear GEMMs), whole and ragged channel counts
    (1, 256, 864, 6912, 1, 1), (1, 256, 256, 3100, 1, 1), (1, 256, 576, 2592, 1, 1), (1, 256, 128, 1508, 1, 1), (2, 100, 64, 3075, 1, 1), (1, 250, 6912, 864, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tcn_conv_vs_torch(cuda, dtype):
    from computervision_codes_amd import ops
    rng = np.random.default_rng(77 if dtype == torch.float32 else 78)
    for ci, (b, t, cin, cout, taps, d) in enumerate(CASES):
        if dtype == torch.bfloat16 and (cin * 2) % 128:
            continue
        x = torch.from_numpy(rng.standard_normal((b, t, cin)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((cout, cin, taps)) / np.sqrt(cin * taps)).astype(np.float32))
        bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
        use_res, relu = bool(ci % 2), bool((ci // 2) % 2)
        res = torch.from_numpy(rng.standard_normal((b, t, cout)).astype(np.float32)) if use_res else None
        if dtype == torch.bfloat16:
            x, w = x.to(dtype).float(), w.to(dtype).float()
            res = res.to(dtype).float() if res is not None else None
        ref = _ref(x, w, bias, taps, d, res, relu)
        wp = ops.pack_conv_weight(w.unsqueeze(2).to(cuda), None, dtype)
        y = ops.tcn_conv(x.to(dtype).to(cuda), wp, bias.to(cuda), taps=taps, dilation=d, residual=res.to(dtype).to(cuda) if use_res else None,
                         relu=relu)
        err = (y.float().cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err < (2e-5 if dtype == torch.float32 else 1.2e-2), ((b, t, cin, cout, taps, d), err)
        if dtype == torch.bfloat16:   # fp32 output of bf16 operands (the heads): only the operand rounding remains
            y32 = ops.tcn_conv(x.to(dtype).to(cuda), wp, bias.to(cuda), taps=taps, dilation=d, relu=relu, out_dtype=torch.float32)
            ref32 = _ref(x, w, bias, taps, d, None, relu)
            assert (y32.cpu() - ref32).abs().max().item() / max(1.0, ref32.abs().max().item()) < 2e-5


def test_tcn_conv_matches_generic_kernel_and_is_batch_independent(cuda):
    """same operands through mt4_conv_nhwc (other K order: equal up to fp32 reassociation); and a video's rows are bit-identical whether it
    runs alone or beside another video"""
    from computervision_codes_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 200, 512, generator=g).to(cuda)
    w = (torch.randn(512, 512, 3, generator=g) / 40).to(cuda)
    bias = torch.randn(512, generator=g).to(cuda)
    wp = ops.pack_conv_weight(w.unsqueeze(2), None, torch.float32)
    for d in (1, 4, 32, 128, 512):
        y = ops.tcn_conv(x, wp, bias, taps=3, dilation=d, relu=True)
        yg = ops.conv_nhwc(x.view(2, 1, 200, 512), wp, bias, kh=1, kw=3, pad=(0, d), dil=(1, d), relu=True).view(2, 200, 512)
        assert (y - yg).abs().max().item() < 2e-5 * max(1.0, yg.abs().max().item())
        y0 = ops.tcn_conv(x[:1].contiguous(), wp, bias, taps=3, dilation=d, relu=True)
        assert torch.equal(y0[0], y[0])


def test_tcn_stage_and_layer_vs_torch(cuda):
    from computervision_codes_amd import ops
    g = torch.Generator().manual_seed(9)
    C_, T_, L = 64, 150, 9
    x = torch.randn(1, T_, C_, generator=g)
    wd = [torch.randn(C_, C_, 3, generator=g) / 14 for _ in range(L)]
    bd = [torch.randn(C_, generator=g) * 0.1 for _ in range(L)]
    w1 = [torch.randn(C_, C_, 1, generator=g) / 8 for _ in range(L)]
    b1 = [torch.randn(C_, generator=g) * 0.1 for _ in range(L)]
    ref = x
    for i in range(L):
        h = _ref(ref, wd[i], bd[i], 3, 2 ** i, None, True)
        ref = _ref(h, w1[i], b1[i], 1, 1, ref, False)
    pk = lambda ws: [ops.pack_conv_weight(w.unsqueeze(2).to(cuda), None, torch.float32) for w in ws]
    st = ops.TcnStage(pk(wd), [b.to(cuda) for b in bd], pk(w1), [b.to(cuda) for b in b1])
    xd = x.to(cuda)
    out = ops.tcn_stage(st, xd)
    assert torch.equal(xd.cpu(), x)        # the stage input is never written
    assert (out.cpu() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    # one layer through its own entry point
    y1 = ops.tcn_layer(xd, st._keep[0][3], st._keep[1][3], st._keep[2][3], st._keep[3][3], 8)
    r1 = _ref(_ref(x, wd[3], bd[3], 3, 8, None, True), w1[3], b1[3], 1, 1, x, False)
    assert (y1.cpu() - r1).abs().max().item() < 2e-5 * max(1.0, r1.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fpn_topdown(cuda, dtype):
    from computervision_codes_amd import ops
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(3, 1000, generator=g).to(dtype)
    lev = torch.randn(4, 1000, generator=g).to(dtype)
    ref = lev.clone()
    for l in (2, 1, 0):
        ref[l] = (lat[l].float() + ref[l + 1].float()).to(dtype)
    d = lev.to(cuda)
    ops.fpn_topdown(lat.to(cuda), d)
    assert torch.equal(d.cpu(), ref)


def test_tcn_conv_rejects_unsupported_geometry(cuda):
    from computervision_codes_amd import _lib, ops
    x = torch.zeros(1, 8, 24, device=cuda)   # 96-byte rows: not whole K-steps
    wp = ops.pack_conv_weight(torch.zeros(16, 24, 1, 3, device=cuda), None, torch.float32)
    with pytest.raises(_lib.Mt4Error):
        ops.tcn_conv(x, wp, None, taps=3, dilation=1)
    assert not ops.tcn_supported(24, torch.float32) and ops.tcn_supported(512, torch.bfloat16)


@pytest.mark.parametrize("b,t,d", [(3, 256, 1), (2, 100, 4), (5, 64, 64), (1, 77, 256), (4, 130, 16), (2, 2000, 512)])
def test_tcn_layer_fused_is_bit_identical_to_the_two_launches(cuda, b, t, d):
    """`mt4_tcn_layer_fused_bf16` (a DilatedResidualLayer in one launch, hidden map in LDS; `Temporal_tenco/network.py:186-198`) against the dilated
    conv + ReLU and the 1 x 1 conv + residual through `mt4_conv_nhwc`: same bits (same K walk, bias-initialised accumulators, h rounded to bf16 where
    the stand-alone launch stores it, fp32 residual add); ragged last tiles, dilations beyond the video (outer taps read padding only), several
    videos: a video's rows do not depend on what rides along"""
    from computervision_codes_amd import ops
    g = torch.Generator().manual_seed(b * 1000 + t + d)
    bf = torch.bfloat16
    x = torch.randn((b, t, 512), generator=g).to(bf).to(cuda)
    w1 = (torch.randn((512, 512, 3), generator=g) * 0.03).to(cuda)
    w2 = (torch.randn((512, 512, 1), generator=g) * 0.05).to(cuda)
    b1, b2 = (torch.randn(512, generator=g) * 0.1).to(cuda), (torch.randn(512, generator=g) * 0.1).to(cuda)
    w1p, w2p = ops.pack_conv_weight(w1.unsqueeze(2), None, bf), ops.pack_conv_weight(w2.unsqueeze(2), None, bf)
    h = ops.conv_nhwc(x.view(b, 1, t, 512), w1p, b1, kh=1, kw=3, pad=(0, d), dil=(1, d), relu=True, tile=1)
    ref = ops.conv_nhwc(h, w2p, b2, kh=1, kw=1, residual=x.view(b, 1, t, 512), tile=1).view(b, t, 512)
    got = ops.tcn_layer_fused(x, ops.pack_fragments(w1p), b1, ops.pack_fragments(w2p), b2, d)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (got.float() - ref.float()).abs().max().item()
    one = ops.tcn_layer_fused(x[b - 1:].contiguous(), ops.pack_fragments(w1p), b1, ops.pack_fragments(w2p), b2, d)
    assert torch.equal(one[0].view(torch.int16), got[b - 1].view(torch.int16))
    # against fp32 torch on the same bf16 operands (h rounded to bf16 in between): the arithmetic itself
    xf = x.float().permute(0, 2, 1)
    hf = torch.relu(torch.nn.functional.conv1d(xf, w1.to(bf).float(), b1, padding=d, dilation=d)).to(bf).float()
    yf = (xf + torch.nn.functional.conv1d(hf, w2.to(bf).float(), b2)).permute(0, 2, 1)
    assert (got.float() - yf).abs().max().item() <= 2 ** -7 * max(1.0, yf.abs().max().item())


def test_tenco_throughput_mode_uses_the_fused_layer_and_matches_the_two_launch_model(cuda):
    """`temporal_tenco.VideoNas(dtype=bf16)` on 48 videos of 128 frames (96 tiles of 64 frames: the lower end of the window the one-launch layer is used in): every DilatedResidualLayer is one launch;
    logits and features equal the model with the gate closed, bit for bit"""
    import types
    from computervision_codes_amd import ops, shapes, synth
    from computervision_codes_amd.temporal_tenco import VideoNas
    args = types.SimpleNamespace(fpn=True, output=False, hier=False, mask=True)
    sd = synth.fill_from_shapes(shapes.tenco_shapes(11, 10, 3, 512, 512, 100, fpn=True), seed=5)
    m = VideoNas(args, 11, 10, 3, 512, 512, 100, dtype=torch.bfloat16).eval().load_state_dict(sd)
    x = torch.stack([synth.synthetic_features(128, 512, seed=9 + i)[0] for i in range(6)]).repeat(8, 1, 1).to(cuda)
    calls = []
    orig = ops.tcn_layer_fused
    ops.tcn_layer_fused = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
    try:
        a = m(x, False)
        assert len(calls) == 41
        m.fused_layer_min_tiles = 10 ** 9
        calls.clear()
        bb = m(x, False)
        assert not calls
    finally:
        ops.tcn_layer_fused = orig
    for la, lb in zip(a[:5], bb[:5]):
        for u, v in zip(la, lb):
            assert torch.equal(u, v)
    assert torch.equal(a[0][0][:6], a[0][0][6:12])          # the same six videos: the same rows whatever rides along


@pytest.mark.parametrize("rows,cin,cout", [(256, 256, 768), (256, 384, 1152), (256, 576, 4608), (256, 864, 2592), (200, 864, 6912), (37, 32, 48), (1, 64, 16),
                                           (70, 896, 50)])
def test_linear_ln_vs_float64_layernorm_linear(cuda, rows, cin, cout):
    """`mt4_tcn_linear_stats_f32` -> `mt4_tcn_linear_ln_f32`: x = a . Wp^T + bp + r leaves the LayerNorm partials of its rows, Linear(LayerNorm(x)) is ONE
    launch behind it (MS-TCT's proj -> norm2 -> linear1, linear2 -> norm1 -> q | kv): against float64 on the CPU, as close as LayerNorm + Linear as two
    launches; rows with a mean far from zero (where sum x^2 - mean^2 loses digits) stay within the model's 1e-3"""
    from computervision_codes_amd import ops
    rng = np.random.default_rng(rows + cin + cout)
    r = torch.from_numpy(rng.standard_normal((rows, cin)).astype(np.float32)) * 2.0 + torch.from_numpy(rng.standard_normal((rows, 1)).astype(np.float32)) * 1.5
    a = torch.from_numpy(rng.standard_normal((rows, 64)).astype(np.float32))
    wpr = torch.from_numpy((rng.standard_normal((cin, 64)) / 8.0).astype(np.float32))
    bp = torch.from_numpy(rng.standard_normal(cin).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    g = torch.from_numpy((1.0 + 0.3 * rng.standard_normal(cin)).astype(np.float32))
    be = torch.from_numpy((0.2 * rng.standard_normal(cin)).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((rows, cout)).astype(np.float32))
    wpp = ops.pack_linear_weight(wpr.to(cuda), torch.float32)
    x, st = ops.linear_stats(a.to(cuda), wpp, bp.to(cuda), residual=r.to(cuda))
    with ops.latency_tiles():
        assert torch.equal(x, ops.linear(a.to(cuda), wpp, bp.to(cuda), residual=r.to(cuda)))        # the statistics ride along: same rows
    x64 = x.double().cpu()
    assert tuple(st.shape) == (cin // 16, rows, 2)
    assert (st[..., 0].double().cpu().sum(0) - x64.sum(1)).abs().max().item() < 1e-5 * max(1.0, x64.abs().sum(1).max().item())
    assert (st[..., 1].double().cpu().sum(0) / (x64 * x64).sum(1) - 1).abs().max().item() < 1e-5
    ref = F.linear(F.layer_norm(x64, (cin,), g.double(), be.double(), 1e-5), w.double(), b.double())
    wf, cs, bf = ops.fold_layernorm(w, b, g, be)
    wp = ops.pack_linear_weight(wf.to(cuda), torch.float32)
    with ops.latency_tiles():
        two = ops.linear(ops.layernorm(x, g.to(cuda), be.to(cuda)), ops.pack_linear_weight(w.to(cuda), torch.float32), b.to(cuda))
    st2 = torch.empty((cout // 16, rows, 2), device=cuda) if cout % 16 == 0 else None
    one = ops.linear_ln(x, wp, cs.to(cuda), bf.to(cuda), stats_in=st, stats_out=st2)
    scale = max(1.0, ref.abs().max().item())
    e_two, e_one = (two.double().cpu() - ref).abs().max().item() / scale, (one.double().cpu() - ref).abs().max().item() / scale
    assert e_one < 2e-5 and e_one < 4 * e_two + 2e-6, (e_one, e_two)
    if st2 is not None:
        assert (st2[..., 0].double().cpu().sum(0) - one.double().cpu().sum(1)).abs().max().item() < 1e-5 * max(1.0, one.abs().sum(1).max().item())
    one_r = ops.linear_ln(x, wp, cs.to(cuda), bf.to(cuda), stats_in=st, residual=res.to(cuda), relu=True)
    assert (one_r.double().cpu() - F.relu(ref + res.double())).abs().max().item() / scale < 2e-5
    assert torch.equal(one, ops.linear_ln(x, wp, cs.to(cuda), bf.to(cuda), stats_in=st))                                     # deterministic
    if rows > 40:   # a frame's row does not depend on what shares its launch
        assert torch.equal(one[:33], ops.linear_ln(x[:33].contiguous(), wp, cs.to(cuda), bf.to(cuda), stats_in=st[:, :33].contiguous()))
    xs, sts = ops.linear_stats(a.to(cuda), wpp, bp.to(cuda), residual=(r + 30.0).to(cuda))    # mean 30, sigma 2: the one-pass variance loses ~8 bits
    far = ops.linear_ln(xs, wp, cs.to(cuda), bf.to(cuda), stats_in=sts)
    ref_far = F.linear(F.layer_norm(xs.double().cpu(), (cin,), g.double(), be.double(), 1e-5), w.double(), b.double())
    assert (far.double().cpu() - ref_far).abs().max().item() / scale < 1e-3
    csd, bfd = cs.to(cuda), bf.to(cuda)      # the partial sums are not optional
    assert ops.lib.mt4_tcn_linear_ln_f32(x.data_ptr(), wp.data_ptr(), csd.data_ptr(), bfd.data_ptr(), None, one.data_ptr(), rows, cin, cout, 1e-5, 0, None, None, None) != 0
