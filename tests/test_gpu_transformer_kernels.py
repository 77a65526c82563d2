"""GPU: the transformer-stage C-ABI kernels against torch-CPU fp32 oracle ops on seeded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _tol(dtype, f32=2e-5, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c", [96, 128, 864, 1024, 4096])
def test_layernorm_plain(cuda, c, dtype):
    from computervision_codes_amd import ops
    x = _rand((37, c), 1, 3.0).to(dtype).float()
    g, b = _rand((c,), 2) + 1.5, _rand((c,), 3)
    ref = F.layer_norm(x, (c,), g, b, 1e-5)
    got = ops.layernorm(x.to(cuda, dtype), g.to(cuda), b.to(cuda)).float().cpu()
    assert (got - ref).abs().max().item() < _tol(dtype, 1e-5, 3e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("res,ws,shift", [(24, 12, 6), (24, 12, 0), (14, 7, 3), (12, 12, 0)])
def test_layernorm_window_gather_matches_roll_partition(cuda, res, ws, shift, dtype):
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_transformer import _window_row_map
    from oracle.swin_q2l import window_partition
    b, c = 2, 64
    x = _rand((b, res * res, c), 5).to(dtype).float()
    g, be = _rand((c,), 6) + 1.5, _rand((c,), 7)
    y = F.layer_norm(x, (c,), g, be, 1e-5).view(b, res, res, c)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    ref = window_partition(y, ws).reshape(-1, c)
    rm = _window_row_map(res, ws, shift).to(cuda)
    got = ops.layernorm(x.view(-1, c).to(cuda, dtype), g.to(cuda), be.to(cuda), row_map=rm, group=1, l_out=res * res, l_in=res * res)
    assert (got.float().cpu() - ref).abs().max().item() < _tol(dtype, 1e-5, 3e-2)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_patch_merging_gather(cuda, dtype):
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_transformer import _merge_row_map
    b, res, c = 2, 12, 32
    x = _rand((b, res * res, c), 8).to(dtype).float()
    g, be = _rand((4 * c,), 9) + 1.5, _rand((4 * c,), 10)
    xv = x.view(b, res, res, c)
    cat = torch.cat([xv[:, 0::2, 0::2], xv[:, 1::2, 0::2], xv[:, 0::2, 1::2], xv[:, 1::2, 1::2]], -1).view(b, -1, 4 * c)
    ref = F.layer_norm(cat, (4 * c,), g, be, 1e-5).reshape(-1, 4 * c)
    got = ops.layernorm(x.view(-1, c).to(cuda, dtype), g.to(cuda), be.to(cuda), row_map=_merge_row_map(res).to(cuda), group=4,
                        l_out=(res // 2) ** 2, l_in=res * res, m_out=b * (res // 2) ** 2)
    assert (got.float().cpu() - ref).abs().max().item() < _tol(dtype, 1e-5, 3e-2)


def _attn_ref(q, k, v, scale, bias, mask, nw):
    # q [B,H,Nq,hd], k/v [B,H,Nk,hd]
    s = (q * scale) @ k.transpose(-2, -1)
    if bias is not None:
        s = s + bias.unsqueeze(0)
    if mask is not None:
        b = q.shape[0]
        s = (s.view(b // nw, nw, *s.shape[1:]) + mask.unsqueeze(1).unsqueeze(0)).view(*s.shape)
    return s.softmax(-1) @ v


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [
    dict(B=8, H=4, Nq=144, Nk=144, hd=32, bias=True, nw=4),     # Swin stage (w12) with shift mask
    dict(B=6, H=3, Nq=49, Nk=49, hd=32, bias=True, nw=0),       # w7, no mask
    dict(B=2, H=8, Nq=256, Nk=256, hd=108, bias=False, nw=0),   # MS-TCT stage 4
    dict(B=2, H=8, Nq=101, Nk=101, hd=48, bias=False, nw=0),    # MS-TCT stage 2, ragged T
    dict(B=2, H=8, Nq=40, Nk=40, hd=6, bias=False, nw=0),       # odd head dim -> element-wise staging
    dict(B=3, H=4, Nq=6, Nk=144, hd=256, bias=False, nw=0),     # Q2L decoder cross-attention, multi-chunk keys
    dict(B=2, H=4, Nq=144, Nk=144, hd=192, bias=False, nw=0),   # Q2L encoder (Swin-T hidden 768)
    dict(B=2, H=4, Nq=144, Nk=144, hd=384, bias=False, nw=0),   # Q2L encoder over Swin-L (hidden 1536: the shipped teacher, Scripts/train_fold1.sh:5-12)
    dict(B=3, H=4, Nq=6, Nk=144, hd=384, bias=False, nw=0),     # ... its decoder cross-attention (6 queries: task i)
    dict(B=1, H=2, Nq=20, Nk=33, hd=512, bias=True, nw=0),      # the widest head the core takes
])
def test_attention_core(cuda, cfg, dtype):
    from computervision_codes_amd import ops
    B, H, Nq, Nk, hd = cfg["B"], cfg["H"], cfg["Nq"], cfg["Nk"], cfg["hd"]
    c = H * hd
    # packed projection buffer [B*N, 3C] like the Swin qkv linear when Nq == Nk
    q = _rand((B * Nq, c), 11, 2.0).to(dtype)
    kv = _rand((B * Nk, 2 * c), 12, 2.0).to(dtype)
    bias = _rand((H, Nq, Nk), 13) if cfg["bias"] else None
    mask = None
    if cfg["nw"]:
        mask = torch.where(_rand((cfg["nw"], Nq, Nk), 14) > 0.3, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()
    scale = hd ** -0.5
    qd, kvd = q.to(cuda), kv.to(cuda)
    out = ops.attention(qd, kvd[:, :c], kvd[:, c:], batch=B, heads=H, nq=Nq, nk=Nk, hd=hd, q_stride=c, k_stride=2 * c, v_stride=2 * c,
                        scale=scale, bias=bias.to(cuda) if bias is not None else None, mask=mask.to(cuda) if mask is not None else None)
    qf = q.float().view(B, Nq, H, hd).permute(0, 2, 1, 3)
    kf = kv.float()[:, :c].reshape(B, Nk, H, hd).permute(0, 2, 1, 3)
    vf = kv.float()[:, c:].reshape(B, Nk, H, hd).permute(0, 2, 1, 3)
    ref = _attn_ref(qf, kf, vf, scale, bias, mask, cfg["nw"]).permute(0, 2, 1, 3).reshape(B * Nq, c)
    assert (out.float().cpu() - ref).abs().max().item() < _tol(dtype, 2e-5, 2e-2)


@pytest.mark.parametrize("hd", [256, 384])
@pytest.mark.parametrize("B,H,Nq,Nk", [(3, 4, 144, 144), (2, 4, 100, 144), (5, 4, 6, 144), (2, 2, 15, 49), (1, 4, 70, 160), (2, 1, 10, 7)])
def test_mha_mfma_head_dim_256(cuda, B, H, Nq, Nk, hd):
    """the matrix-unit core behind `mt4_attention` for bf16, head dim 256 (Swin-B, d = 1024) / 384 (Swin-L, d = 1536), no bias / mask (Q2L encoder
    self-attention and decoder cross-attention, `transformer.py:186-189,275-283`): fp32 oracle on the same bf16-rounded q/k/v; ragged query blocks and key tiles"""
    from computervision_codes_amd import ops
    c = H * hd
    q = _rand((B * Nq, c), 81, 1.0).to(torch.bfloat16)
    kv = _rand((B * Nk, 2 * c), 82, 1.0).to(torch.bfloat16)
    scale = hd ** -0.5
    qd, kvd = q.to(cuda), kv.to(cuda)
    out = ops.attention(qd, kvd[:, :c], kvd[:, c:], batch=B, heads=H, nq=Nq, nk=Nk, hd=hd, q_stride=c, k_stride=2 * c, v_stride=2 * c, scale=scale)
    sp = lambda t, n: t.float().reshape(B, n, H, hd).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(q, Nq), sp(kv[:, :c], Nk), sp(kv[:, c:], Nk), scale, None, None, 0).permute(0, 2, 1, 3).reshape(B * Nq, c)
    assert torch.isfinite(out.float()).all()
    assert (out.float().cpu() - ref).abs().max().item() < 2e-2
    # sharper scores (a few keys dominate): the probabilities' bf16 rounding shows most here
    out2 = ops.attention(qd, kvd[:, :c], kvd[:, c:], batch=B, heads=H, nq=Nq, nk=Nk, hd=hd, q_stride=c, k_stride=2 * c, v_stride=2 * c, scale=1.0)
    ref2 = _attn_ref(sp(q, Nq), sp(kv[:, :c], Nk), sp(kv[:, c:], Nk), 1.0, None, None, 0).permute(0, 2, 1, 3).reshape(B * Nq, c)
    assert (out2.float().cpu() - ref2).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,H,Nq,Nk,hd", [(1, 8, 256, 256, 32), (1, 8, 256, 256, 48), (1, 8, 256, 256, 72), (1, 8, 256, 256, 108),
                                           (2, 3, 100, 77, 108), (1, 2, 70, 130, 20), (3, 1, 5, 16, 128), (1, 8, 37, 37, 72),
                                           (16, 8, 64, 200, 48), (4, 8, 250, 256, 108)])      # (the last two: enough 64-query workgroups, one wave per 16 queries;
                                                                                              #  the others: four waves per 16 queries, keys split)
def test_mha_mfma_fp32(cuda, B, H, Nq, Nk, hd):
    """the exact-fp32 matrix-unit core behind `mt4_attention` (fp32, no bias / mask, <= 256 keys, head dim <= 128: MS-TCT's
    Global_Relational_Block, `Temporal_Encoder.py:80-86`, incl. the short last chunk of a video) against the fp32 oracle"""
    from computervision_codes_amd import ops
    c = H * hd
    qkv = _rand((B * max(Nq, Nk), 3 * c), 91, 1.5)
    d = qkv.to(cuda)
    scale = hd ** -0.5
    out = ops.attention(d[:B * Nq, :c], d[:B * Nk, c:2 * c], d[:B * Nk, 2 * c:], batch=B, heads=H, nq=Nq, nk=Nk, hd=hd, q_stride=3 * c, k_stride=3 * c,
                        v_stride=3 * c, scale=scale)
    sp = lambda t, n: t.reshape(B, n, H, hd).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(qkv[:B * Nq, :c], Nq), sp(qkv[:B * Nk, c:2 * c], Nk), sp(qkv[:B * Nk, 2 * c:], Nk), scale, None, None, 0)
    ref = ref.permute(0, 2, 1, 3).reshape(B * Nq, c)
    assert (out.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", DT)
def test_linear_gelu_rowmap_and_column_slices(cuda, dtype):
    from computervision_codes_amd import ops
    m, k, n = 288, 64, 96
    x = _rand((m, k), 21).to(dtype).float()
    w = _rand((n, k), 22, 0.2).to(dtype).float()
    b = _rand((n,), 23)
    wp = ops.pack_linear_weight(w.to(cuda), dtype)
    ref = F.gelu(F.linear(x, w, b))
    got = ops.linear(x.to(cuda, dtype), wp, b.to(cuda), act="gelu")
    assert (got.float().cpu() - ref).abs().max().item() < _tol(dtype, 2e-5, 2e-2)
    # scatter epilogue: out[perm[m]] = res[perm[m]] + x[m] @ W^T, per-"image" map of length 144
    perm = torch.randperm(144, generator=torch.Generator().manual_seed(3)).to(torch.int32)
    res = _rand((m, n), 24).to(dtype).float()
    full = torch.cat([perm.long(), perm.long() + 144])
    ref2 = torch.empty(m, n)
    ref2[full] = res[full] + F.linear(x, w, b)
    got2 = ops.linear(x.to(cuda, dtype), wp, b.to(cuda), residual=res.to(cuda, dtype), out_row_map=perm.to(cuda))
    assert (got2.float().cpu() - ref2).abs().max().item() < _tol(dtype, 2e-5, 3e-2)
    # column-slice output and residual of wider buffers
    wide = torch.zeros((m, 3 * n), dtype=dtype, device=cuda)
    rwide = _rand((m, 2 * n), 25).to(dtype)
    ops.linear(x.to(cuda, dtype), wp, b.to(cuda), residual=rwide.to(cuda)[:, n:], out=wide[:, n:2 * n])
    ref3 = F.linear(x, w, b) + rwide.float()[:, n:]
    assert (wide[:, n:2 * n].float().cpu() - ref3).abs().max().item() < _tol(dtype, 2e-5, 3e-2)
    assert wide[:, :n].abs().max().item() == 0 and wide[:, 2 * n:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DT)
def test_patchify_matches_conv4x4(cuda, dtype):
    from computervision_codes_amd import ops, synth
    fr = synth.synthetic_frames(2, 32, 48, seed=4)
    xn = synth.normalize_frames(fr)
    w = _rand((16, 3, 4, 4), 31, 0.2)
    ref = F.conv2d(xn, w, stride=4).flatten(2).transpose(1, 2).reshape(-1, 16)
    for inp in (xn.to(cuda), fr.to(cuda)):
        rows = ops.patchify(inp, 4, dtype, synth.IMAGENET_MEAN, synth.IMAGENET_STD)
        got = rows.float().cpu() @ w.view(16, 48).t()
        assert (got - ref).abs().max().item() < _tol(dtype, 1e-4, 5e-2)
    if dtype == torch.bfloat16:   # uint8 frames, P = 4, bf16 rows run a kernel of their own (12 bytes per thread, table-driven): the generic kernel's bits
        for (b, h, w_) in ((2, 32, 48), (3, 96, 100), (1, 384, 384)):
            fr = synth.synthetic_frames(b, h, w_, seed=h + w_).to(cuda)
            assert torch.equal(ops.patchify(fr, 4, torch.bfloat16, synth.IMAGENET_MEAN, synth.IMAGENET_STD),
                               ops.patchify(fr, 4, torch.float32, synth.IMAGENET_MEAN, synth.IMAGENET_STD).to(torch.bfloat16))


@pytest.mark.parametrize("dtype", DT)
def test_small_elementwise_pieces(cuda, dtype):
    from computervision_codes_amd import ops
    x = _rand((3 * 10, 64), 41).to(dtype)
    p = _rand((10, 64), 42).to(dtype)
    got = ops.add_rowbcast(x.to(cuda), p.to(cuda)).float().cpu()
    assert (got - (x.float().view(3, 10, 64) + p.float()).view(-1, 64)).abs().max().item() < _tol(dtype, 1e-6, 2e-2)
    hs = _rand((4 * 6, 256), 43).to(dtype)
    W, b = _rand((6, 256), 44), _rand((6,), 45)
    got = ops.groupwise_linear(hs.to(cuda), W.to(cuda), b.to(cuda), 4, 6).cpu()
    ref = (W.unsqueeze(0) * hs.float().view(4, 6, 256)).sum(-1) + b
    assert (got - ref).abs().max().item() < 1e-4
    xd = _rand((2, 33, 64), 46).to(dtype)
    wd, bd = _rand((64, 3), 47), _rand((64,), 48)
    got = ops.dwconv1d_k3(xd.to(cuda), wd.to(cuda), bd.to(cuda), act="gelu").float().cpu()
    ref = F.gelu(F.conv1d(xd.float().transpose(1, 2), wd.unsqueeze(1), bd, padding=1, groups=64)).transpose(1, 2)
    assert (got - ref).abs().max().item() < _tol(dtype, 1e-5, 2e-2)


def test_kd_mix_matches_reference_branch(cuda):
    """reduced form vs the oracle of `Spatial_cnn/network.py:47-71` (itself pinned to the reference in gen_golden)"""
    from computervision_codes_amd import ops, shapes, synth
    from oracle import spatial_cnn as o_cnn
    sd = synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet18"), seed=9)
    s = _rand((5, 512), 51, 3.0)
    tf = [_rand((5, 1536), 52 + i) for i in range(3)]
    ref = o_cnn.kd_branch(sd, s, *tf)
    teas = [F.conv1d(t.unsqueeze(-1), sd[f"{m}.weight"], sd[f"{m}.bias"]).squeeze(-1) for m, t in zip(("mi", "mv", "mt"), tf)]
    mixed = ops.kd_mix(s.to(cuda), *[t.contiguous().to(cuda) for t in teas])
    for o, wname, r in zip(mixed, ("wi", "wv", "wt"), ref):
        got = F.conv1d(o.cpu().unsqueeze(-1), sd[wname + ".weight"], sd[wname + ".bias"]).squeeze(-1)
        assert (got - r).abs().max().item() < 1e-4


@pytest.mark.parametrize("cfg", [dict(B=8, H=4, N=144, nw=4), dict(B=6, H=3, N=49, nw=0), dict(B=4, H=8, N=49, nw=2),
                                 dict(B=2, H=8, N=256, nw=0), dict(B=3, H=2, N=17, nw=0)])
def test_window_attention_mfma_bf16(cuda, cfg):
    """MFMA core (bf16, hd 32) vs the fp32 oracle on the same bf16-rounded q/k/v; also agrees with the generic kernel"""
    from computervision_codes_amd import ops
    B, H, N = cfg["B"], cfg["H"], cfg["N"]
    c = H * 32
    qkv = _rand((B * N, 3 * c), 61, 2.0).to(torch.bfloat16)
    bias = _rand((H, N, N), 62)
    mask = torch.where(_rand((cfg["nw"], N, N), 63) > 0.3, torch.tensor(-100.0), torch.tensor(0.0)).contiguous() if cfg["nw"] else None
    scale = 32 ** -0.5
    d = qkv.to(cuda)
    out = ops.window_attention_bf16(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], batch=B, heads=H, n=N, q_stride=3 * c, k_stride=3 * c, v_stride=3 * c,
                                    scale=scale, bias_padded=ops.pad_attention_bias(bias.to(cuda)),
                                    mask_padded=ops.pad_attention_bias(mask.to(cuda), 0.0) if mask is not None else None)
    f = qkv.float()
    sp = lambda t: t.reshape(B, N, H, 32).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(f[:, :c]), sp(f[:, c:2 * c]), sp(f[:, 2 * c:]), scale, bias, mask, cfg["nw"]).permute(0, 2, 1, 3).reshape(B * N, c)
    assert (out.float().cpu() - ref).abs().max().item() < 3e-2
    gen = ops.attention(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], batch=B, heads=H, nq=N, nk=N, hd=32, q_stride=3 * c, k_stride=3 * c, v_stride=3 * c,
                        scale=scale, bias=bias.to(cuda), mask=mask.to(cuda) if mask is not None else None)
    assert (out.float() - gen.float()).abs().max().item() < 3e-2


@pytest.mark.parametrize("ws,H,res,shift", [(12, 4, 24, 6), (12, 4, 24, 0), (7, 8, 14, 3), (7, 3, 7, 0), (4, 2, 8, 2)])
def test_window_attention_from_relative_table_equals_expanded_tables(cuda, ws, H, res, shift):
    """the table / region-id form of the Swin core (`mt4_window_attention_rel_bf16`) against the same kernel fed with the expanded
    [H,N,N] bias and [nW,N,N] mask the reference materialises (`swin_transformer.py:92-103,129-132,210-229`): bit-identical on unshifted
    blocks.  On shifted blocks the region form adds +100 where the regions are equal (one more MFMA on one-hot region vectors) instead of -100
    where they differ -- the same softmax with every score of a row moved by 100: outputs differ by one bf16 ulp in < 0.5 % of the elements, and
    both forms are equally close to the float64 result"""
    from computervision_codes_amd import ops
    from computervision_codes_amd.spatial_transformer import _rel_pos_index, _shift_mask, _shift_regions
    N, nwin = ws * ws, (res // ws) ** 2
    B = 2 * nwin
    c = H * 32
    qkv = _rand((B * N, 3 * c), 71, 2.0).to(torch.bfloat16).to(cuda)
    table = _rand(((2 * ws - 1) ** 2, H), 72)                                   # the reference's parameter layout [T, heads]
    bias = table[_rel_pos_index(ws).view(-1)].view(N, N, H).permute(2, 0, 1).contiguous()
    mask = _shift_mask(res, ws, shift) if shift else None
    scale = 32 ** -0.5
    kw = dict(batch=B, heads=H, q_stride=3 * c, k_stride=3 * c, v_stride=3 * c, scale=scale)
    ref = ops.window_attention_bf16(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], n=N, bias_padded=ops.pad_attention_bias(bias.to(cuda)),
                                    mask_padded=ops.pad_attention_bias(mask.to(cuda), 0.0) if shift else None, **kw)
    got = ops.window_attention_rel_bf16(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], ws=ws, rel_table=table.t().contiguous().to(cuda),
                                        region=_shift_regions(res, ws, shift).to(torch.int32).to(cuda) if shift else None, **kw)
    if not shift:
        assert torch.equal(got, ref)
        return
    d = (got.float() - ref.float()).abs()
    assert (d > 0).float().mean().item() < 5e-3 and d.max().item() <= 2.0 ** -5 * max(1.0, ref.float().abs().max().item() / 4.0), (d.max().item(), (d > 0).float().mean().item())
    q, k, v = [t.double().cpu().view(B, N, H, 32).permute(0, 2, 1, 3) for t in (qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:])]
    qs = (q.float() * scale).to(torch.bfloat16).double()       # (the kernel rounds the scaled q to bf16)
    sc = qs @ k.transpose(-1, -2) + bias.double()[None] + mask.double().repeat(B // mask.shape[0], 1, 1)[:, None]
    o64 = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, c)
    e_got, e_ref = (got.double().cpu() - o64).abs().max().item(), (ref.double().cpu() - o64).abs().max().item()
    assert e_got <= 1.05 * e_ref + 1e-6, (e_got, e_ref)
