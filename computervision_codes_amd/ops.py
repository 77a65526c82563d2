"""Thin torch-tensor wrappers over the C-ABI (device memory and streams are torch's: plumbing only)."""
from __future__ import annotations

import contextlib
import contextvars
import os
import ctypes as C
import functools
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import MT4_BF16, MT4_F32, ConvDesc, check, lib

_DT = {torch.float32: MT4_F32, torch.bfloat16: MT4_BF16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Mt4Error("mt4 ops need device tensors (no CPU fallback exists)")


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise _lib.Mt4Error(f"unsupported dtype {dtype}")


def packed_k(cin: int, kh: int, kw: int, dtype: torch.dtype) -> int:
    return int(lib.mt4_conv_packed_k(cin, kh, kw, dt_code(dtype)))


def pack_conv_weight(w_oihw: torch.Tensor, scale: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """OIHW float32 (device) -> packed [Cout][Kpad] of `dtype`, optionally scaled per output channel."""
    _need_cuda(w_oihw, scale)
    w = w_oihw.contiguous().float()
    cout, cin, kh, kw = w.shape
    out = torch.empty((cout, packed_k(cin, kh, kw, dtype)), dtype=dtype, device=w.device)
    sc = scale.contiguous().float() if scale is not None else None
    check(lib.mt4_pack_conv_weight(w.data_ptr(), sc.data_ptr() if sc is not None else None, out.data_ptr(),
                                   cout, cin, kh, kw, dt_code(dtype), _stream()), "mt4_pack_conv_weight")
    return out


def pack_stem_weight(w_oihw: torch.Tensor, scale: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    _need_cuda(w_oihw, scale)
    w = w_oihw.contiguous().float()
    assert tuple(w.shape[1:]) == (3, 7, 7)
    cout = w.shape[0]
    out = torch.empty((cout, packed_k(8, 7, 4, dtype)), dtype=dtype, device=w.device)
    sc = scale.contiguous().float() if scale is not None else None
    check(lib.mt4_pack_stem_weight(w.data_ptr(), sc.data_ptr() if sc is not None else None, out.data_ptr(), cout,
                                   dt_code(dtype), _stream()), "mt4_pack_stem_weight")
    return out


def conv_out_size(h: int, k: int, stride: int, pad: int, dil: int) -> int:
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


_TCN_LINEAR = True     # latency contexts: nn.Linear on few rows through the TCN latency kernel
_TCN_LINEAR_MAX_ROWS = 512
_LATENCY_TILES = contextvars.ContextVar("mt4_latency_tiles", default=False)   # per thread / task: forwards may run from several threads


@contextlib.contextmanager
def latency_tiles():
    """Inside this context `tile=0` launches ask for `tile=-1`: automatic choice that may take the K-split tiles (four wave groups per
    workgroup share one output tile's K loop).  For the temporal heads -- one short video, 40+ dependent launches with few tiles and a
    long K each -- that cuts the per-launch critical path; the K summation order then differs from the other tiles' (fp32
    reassociation), so the spatial extractors, whose features are bit-identical whatever batch a frame rides in, do not use it."""
    token = _LATENCY_TILES.set(True)
    try:
        yield
    finally:
        _LATENCY_TILES.reset(token)


def with_latency_tiles(fn):
    """decorator form of `latency_tiles` (the temporal heads' forward / train step)"""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with latency_tiles():
            return fn(*a, **k)
    return wrapped


STAT_REPLICAS = 8                    # MT4_STAT_REPLICAS of include/mt4hip.h


def conv_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, kh: int, kw: int,
              stride: Tuple[int, int] = (1, 1), pad: Tuple[int, int] = (0, 0), dil: Tuple[int, int] = (1, 1),
              residual: Optional[torch.Tensor] = None, relu: bool = False, out_dtype: Optional[torch.dtype] = None,
              out: Optional[torch.Tensor] = None, tile: int = 0, act: Optional[str] = None,
              out_row_map: Optional[torch.Tensor] = None, y_ld: int = 0, res_ld: int = 0, out_hw: Optional[Tuple[int, int]] = None,
              out_rows_per_image: int = 0, run_pixels: int = 1, second=None, stat_sums: Optional[torch.Tensor] = None):
    """y = act(conv(x, w) + bias [+ residual]); x [B,H,W,Cin] contiguous channels-last storage.
    stat_sums: float64 [STAT_REPLICAS, 2, Cout], zeroed by the caller: the launch adds the channel sums / sums of squares of the y it stores
    (`mt4_conv_desc.stat_sums`) -- what `bn_apply_sums_t` turns into the train-mode BatchNorm without a statistics pass over y.
    second = (x2 [B,H2,W2,C2], stride): a second K source behind x, gathered at x2[b, s*ho, s*wo] -- conv3 + the downsample branch of a strided
    Bottleneck in one accumulator chain; `w_packed` = torch.cat([w3_packed, wds_packed], 1), `bias` = b3 + bds (bf16 1x1 launches).
    run_pixels > 1: a tap reads a contiguous run of that many pixels (Cin_eff = run_pixels * x.shape[-1]; kw must be 1, no padding,
    pass out_hw): the space-to-depth stem."""
    _need_cuda(x, w_packed, bias, residual)
    assert x.dim() == 4 and x.is_contiguous()
    b, h, w_, cin = x.shape
    pix = 0
    if run_pixels > 1:
        assert kw == 1 and pad == (0, 0) and out_hw is not None
        pix, cin = cin, cin * run_pixels
    cout = w_packed.shape[0]
    ho = conv_out_size(h, kh, stride[0], pad[0], dil[0])
    wo = conv_out_size(w_, kw, stride[1], pad[1], dil[1])
    if out_hw is not None:      # explicit output grid (taps beyond the image read zeros): data-gradient phases
        ho, wo = out_hw
    od = out_dtype or x.dtype
    if out is None:
        assert y_ld == 0
        out = torch.empty((b, ho, wo, cout), dtype=od, device=x.device)
    elif y_ld == 0 and out_rows_per_image == 0:
        assert out.is_contiguous() and out.numel() == b * ho * wo * cout and out.dtype == od
    elif y_ld == 0:
        assert out.is_contiguous() and out.numel() == b * out_rows_per_image * cout and out.dtype == od
    else:  # `out` is a column-slice view of a [rows, y_ld] buffer: its data_ptr is the slice start
        assert out.dtype == od and out.stride(-1) == 1
    res_f32 = 0
    if residual is not None:
        if residual.dtype == torch.float32 and x.dtype == torch.bfloat16:      # mixed-precision training GEMM: bf16 operands, fp32 output + residual
            assert od == torch.float32
            res_f32 = 1
        else:
            assert residual.dtype == x.dtype
        assert residual.stride(-1) == 1
        if res_ld == 0:
            assert residual.is_contiguous() and residual.numel() == b * (out_rows_per_image or ho * wo) * cout
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout
    x2p, x2h, x2w, x2c, x2s = None, 0, 0, 0, 0
    if second is not None:
        x2, x2s = second
        _need_cuda(x2)
        assert x2.dtype == x.dtype == torch.bfloat16 and x2.is_contiguous() and x2.dim() == 4 and x2.shape[0] == b and kh == kw == 1
        x2p, (_, x2h, x2w, x2c) = x2.data_ptr(), x2.shape
        assert w_packed.dtype == x.dtype and w_packed.shape[1] == packed_k(cin, 1, 1, x.dtype) + packed_k(x2c, 1, 1, x.dtype)
    else:
        assert w_packed.dtype == x.dtype and w_packed.shape[1] == packed_k(cin, kh, kw, x.dtype)
    act_code = {None: 1 if relu else 0, "none": 0, "relu": 1, "gelu": 2, "relu_gate": 3}[act]
    if out_row_map is not None:
        assert out_row_map.dtype == torch.int32 and out_row_map.is_cuda and out_row_map.is_contiguous()
        assert (b * ho * wo) % out_row_map.numel() == 0
    if tile == 0 and _LATENCY_TILES.get():
        tile = -1
    d = ConvDesc(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                 residual.data_ptr() if residual is not None else None, out.data_ptr(),
                 out_row_map.data_ptr() if out_row_map is not None else None,
                 b, h, w_, cin, ho, wo, cout, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1],
                 act_code, dt_code(x.dtype), dt_code(od), tile, out_row_map.numel() if out_row_map is not None else 0, y_ld, res_ld,
                 out_rows_per_image, pix, 0, None, None, None, 0, res_f32, x2p, x2h, x2w, x2c, x2s, 0)
    if stat_sums is not None:
        _need_cuda(stat_sums)
        assert stat_sums.dtype == torch.float64 and stat_sums.is_contiguous() and stat_sums.numel() == STAT_REPLICAS * 2 * cout
        d.stat_sums = stat_sums.data_ptr()
    check(lib.mt4_conv_nhwc(C.byref(d), _stream()), "mt4_conv_nhwc")
    return out


def pack_fragments(w_packed: torch.Tensor) -> torch.Tensor:
    """a packed bf16 matrix [rows, K] in MFMA fragment order (`mt4_pack_fragments_bf16`): what `conv3x3_expand` reads the expansion weights from"""
    _need_cuda(w_packed)
    assert w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous() and w_packed.dim() == 2
    out = torch.empty_like(w_packed)
    check(lib.mt4_pack_fragments_bf16(w_packed.data_ptr(), w_packed.shape[0], w_packed.shape[1], out.data_ptr(), _stream()), "mt4_pack_fragments_bf16")
    return out


def chain_gemm_supported(k1: int, n1: int, n2: int, conv: bool) -> bool:
    """shapes `chain_gemm` runs (`mt4_chain_gemm_bf16`)"""
    return (k1, n1, n2) in ((256, 1024, 256), (128, 512, 128)) or (conv and (k1, n1, n2) == (128, 512, 256))


CHAIN_MIN_TILES = 48    # 128-row tiles below which the chained launch is not used (chain_gemm_pays; profiles/r04_chain_small_batch_ab.txt)


def chain_gemm_pays(rows: int) -> bool:
    """`mt4_chain_gemm_bf16` runs ONE 128-row, 8-wave workgroup per CU and walks the whole K1 -> N1 -> N2 chain serially inside it, with no
    tile choice: below about a round of the 256 CUs (ResNet-50 at batch 1: 7 workgroups in layer2, 2 in layer3) the two `conv_nhwc` / `linear`
    launches, whose small tiles spread the same work over the chip, are faster -- and bit-identical, so callers simply fall back.  The
    threshold is measured (`profiles/r04_chain_small_batch_ab.txt`); latency contexts (`latency_tiles`) never chain."""
    return (rows + 127) // 128 >= CHAIN_MIN_TILES and not _LATENCY_TILES.get()


def chain_gemm(x2d: torch.Tensor, w1_frag: torch.Tensor, b1: torch.Tensor, w2_frag: torch.Tensor, b2: torch.Tensor, *, r1: Optional[torch.Tensor] = None,
               r2: Optional[torch.Tensor] = None, y1: Optional[torch.Tensor] = None, y2: Optional[torch.Tensor] = None):
    """Two dependent 1x1 convolutions / linear layers in one launch (`mt4_chain_gemm_bf16`), bit-identical to the two `conv_nhwc` launches.
    r1 given (Bottleneck form): H = relu(x W1^T + b1 + r1) is stored (returned first: the block output), y2 = relu(H W2^T + b2).
    r2 given (MLP form): y2 = gelu(x W1^T + b1) W2^T + b2 + r2, H never exists.  x2d [M, K1] bf16 (row pitch = its stride);
    w1_frag / w2_frag = `pack_fragments` of the packed [N1, K1] / [N2, N1] matrices.  Returns (y1 or None, y2)."""
    _need_cuda(x2d, w1_frag, b1, w2_frag, b2, r1, r2, y1, y2)
    assert x2d.dim() == 2 and x2d.dtype == torch.bfloat16 and x2d.stride(1) == 1 and (r1 is None) != (r2 is None)
    m, k1 = x2d.shape
    n1, n2 = w1_frag.shape[0], w2_frag.shape[0]
    assert tuple(w1_frag.shape) == (n1, k1) and tuple(w2_frag.shape) == (n2, n1) and w1_frag.dtype == w2_frag.dtype == torch.bfloat16
    assert b1.dtype == b2.dtype == torch.float32 and b1.numel() == n1 and b2.numel() == n2
    conv = r1 is not None
    if conv:
        assert r1.dtype == torch.bfloat16 and r1.is_contiguous() and r1.numel() == m * n1
        y1 = torch.empty((m, n1), dtype=torch.bfloat16, device=x2d.device) if y1 is None else y1
        assert y1.dtype == torch.bfloat16 and y1.is_contiguous() and y1.numel() == m * n1
    else:
        assert r2.dtype == torch.bfloat16 and r2.is_contiguous() and r2.numel() == m * n2 and y1 is None
    y2 = torch.empty((m, n2), dtype=torch.bfloat16, device=x2d.device) if y2 is None else y2
    assert y2.dtype == torch.bfloat16 and y2.is_contiguous() and y2.numel() == m * n2
    check(lib.mt4_chain_gemm_bf16(x2d.data_ptr(), x2d.stride(0), m, k1, w1_frag.data_ptr(), b1.data_ptr(), n1, r1.data_ptr() if conv else None,
                                  y1.data_ptr() if conv else None, 1 if conv else 2, w2_frag.data_ptr(), b2.data_ptr(), n2,
                                  None if conv else r2.data_ptr(), 1 if conv else 0, y2.data_ptr(), _stream()), "mt4_chain_gemm_bf16")
    return y1, y2


def conv3x3_expand(x: torch.Tensor, w2_packed: torch.Tensor, b2: torch.Tensor, w3_frag: torch.Tensor, b3: torch.Tensor, residual: torch.Tensor,
                   out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """conv2 (3x3, 128 -> 128, + bn2 + ReLU) and conv3 (1x1, + bn3 + residual + ReLU) of a stride-1 Bottleneck in one launch
    (`mt4_conv_desc.fuse_expand`): the 128-channel map stays in LDS.  Returns None where the kernel does not run (few tiles): the caller launches
    the two convs -- bit-identical results."""
    _need_cuda(x, w2_packed, b2, w3_frag, b3, residual, out)
    b, h, w_, cin = x.shape
    cout3 = w3_frag.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and cin == 128 and tuple(w2_packed.shape) == (128, packed_k(128, 3, 3, torch.bfloat16))
    assert w3_frag.dtype == torch.bfloat16 and w3_frag.shape[1] == 128 and residual.is_contiguous() and tuple(residual.shape) == (b, h, w_, cout3)
    y = torch.empty((b, h, w_, cout3), dtype=torch.bfloat16, device=x.device) if out is None else out
    assert y.is_contiguous() and tuple(y.shape) == (b, h, w_, cout3) and y.dtype == torch.bfloat16
    d = ConvDesc(x.data_ptr(), w2_packed.data_ptr(), b2.data_ptr(), residual.data_ptr(), y.data_ptr(), None,
                 b, h, w_, cin, h, w_, 128, 3, 3, 1, 1, 1, 1, 1, 1, 1, dt_code(x.dtype), dt_code(x.dtype), 0, 0, 0, 0, 0, 0, cout3,
                 w3_frag.data_ptr(), b3.data_ptr(), y.data_ptr(), 1, 0, None, 0, 0, 0, 0, 1)
    rc = lib.mt4_conv_nhwc(C.byref(d), _stream())
    if rc == _lib.MT4_EUNSUPPORTED:
        return None
    check(rc, "mt4_conv_nhwc")
    return y


def tcn_conv(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, taps: int, dilation: int = 1,
             residual: Optional[torch.Tensor] = None, relu: bool = False, out_dtype: Optional[torch.dtype] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Temporal-head latency path (`mt4_tcn_conv`): x [B,T,Cin] frame-major rows, w packed by `pack_conv_weight` for a 1 x taps kernel;
    y = act(conv1d(x, w, dilation, padding = dilation*(taps-1)/2) + bias [+ residual]) -> [B,T,Cout]"""
    _need_cuda(x, w_packed, bias, residual, out)
    assert x.dim() == 3 and x.is_contiguous()
    b, t, cin = x.shape
    cout = w_packed.shape[0]
    od = out_dtype or x.dtype
    assert w_packed.dtype == x.dtype and w_packed.shape[1] == packed_k(cin, 1, taps, x.dtype)
    if out is None:
        out = torch.empty((b, t, cout), dtype=od, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == b * t * cout and out.dtype == od
    if residual is not None:
        assert residual.dtype == x.dtype and residual.is_contiguous() and residual.numel() == b * t * cout
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout
    d = _lib.TcnDesc(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                     residual.data_ptr() if residual is not None else None, out.data_ptr(), b, t, cin, cout, taps, dilation,
                     1 if relu else 0, dt_code(x.dtype), dt_code(od))
    check(lib.mt4_tcn_conv(C.byref(d), _stream()), "mt4_tcn_conv")
    return out


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """host side of `linear_ln`: nn.Linear (w [N,K], b [N]) behind nn.LayerNorm (gamma, beta [K]) -> (gamma o W, its row sums, W . beta + b), float32
    (sums in float64)"""
    w64, g64 = w.double(), gamma.double()
    wf = w64 * g64[None, :]
    return wf.float(), wf.float().double().sum(1).float().contiguous(), (w64 @ beta.double() + b.double()).float().contiguous()


def linear_ln(x2d: torch.Tensor, w_folded_packed: torch.Tensor, colsum: torch.Tensor, bias_folded: torch.Tensor, *, eps: float = 1e-5,
              stats_in: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False,
              stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Linear(LayerNorm(x)) on the rows of one short window in one launch of the latency kernel (`mt4_tcn_linear_ln_f32`); operands from
    `fold_layernorm` + `pack_linear_weight`.  fp32; rows within `latency_linear_ok`.  stats_in [K / 16, M, 2]: the partial sums the launch that
    produced x left (`linear_stats` / `stats_out` of this op)."""
    _need_cuda(x2d, w_folded_packed, colsum, bias_folded, residual, stats_in, stats_out)
    assert x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.is_contiguous() and w_folded_packed.dtype == torch.float32
    m, k = x2d.shape
    n = w_folded_packed.shape[0]
    assert w_folded_packed.shape[1] == packed_k(k, 1, 1, torch.float32) and tcn_supported(k, torch.float32)
    assert colsum.dtype == torch.float32 and colsum.numel() == n and bias_folded.dtype == torch.float32 and bias_folded.numel() == n
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == m * n
    assert stats_in.dtype == torch.float32 and stats_in.is_contiguous() and tuple(stats_in.shape) == (k // 16, m, 2) and k % 16 == 0
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and tuple(stats_out.shape) == (n // 16, m, 2) and n % 16 == 0
    y = torch.empty((m, n), dtype=torch.float32, device=x2d.device)
    check(lib.mt4_tcn_linear_ln_f32(x2d.data_ptr(), w_folded_packed.data_ptr(), colsum.data_ptr(), bias_folded.data_ptr(),
                                    residual.data_ptr() if residual is not None else None, y.data_ptr(), m, k, n, eps, 1 if relu else 0,
                                    stats_in.data_ptr(), stats_out.data_ptr() if stats_out is not None else None,
                                    _stream()), "mt4_tcn_linear_ln_f32")
    return y


def linear_stats(x2d: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, residual: Optional[torch.Tensor] = None,
                 relu: bool = False):
    """nn.Linear on one short window through the latency kernel, leaving the LayerNorm partials of its output rows (`mt4_tcn_linear_stats_f32`):
    -> (y [M, N], stats [N / 16, M, 2]) for the `linear_ln` launch that normalises y"""
    _need_cuda(x2d, w_packed, bias, residual)
    assert x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.is_contiguous() and w_packed.dtype == torch.float32
    m, k = x2d.shape
    n = w_packed.shape[0]
    assert w_packed.shape[1] == packed_k(k, 1, 1, torch.float32) and tcn_supported(k, torch.float32) and n % 16 == 0
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.numel() == m * n
    y = torch.empty((m, n), dtype=torch.float32, device=x2d.device)
    st = torch.empty((n // 16, m, 2), dtype=torch.float32, device=x2d.device)
    check(lib.mt4_tcn_linear_stats_f32(x2d.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       residual.data_ptr() if residual is not None else None, y.data_ptr(), m, k, n, 1 if relu else 0, st.data_ptr(),
                                       _stream()), "mt4_tcn_linear_stats_f32")
    return y, st


def latency_linear_ok(rows: int, cin: int, dtype: torch.dtype) -> bool:
    """inside a latency context: does a GEMM / 1-D conv over `rows` frames of `cin` channels take the temporal head's latency kernel?"""
    return bool(_LATENCY_TILES.get()) and _TCN_LINEAR and rows <= _TCN_LINEAR_MAX_ROWS and tcn_supported(cin, dtype)


def tcn_supported(cin: int, dtype: torch.dtype) -> bool:
    """geometry contract of the latency path: whole 128-byte K-steps per tap"""
    return (cin * (2 if dtype == torch.bfloat16 else 4)) % 128 == 0


class TcnStage:
    """pointer tables of one stage's layer stack for `mt4_tcn_stage` (built once per model; keeps the tensors alive)"""

    def __init__(self, w_dilated, b_dilated, w_1x1, b_1x1):
        self.n = len(w_dilated)
        self._keep = (list(w_dilated), list(b_dilated), list(w_1x1), list(b_1x1))
        arr = lambda ts: (C.c_void_p * self.n)(*[t.data_ptr() for t in ts])
        self.wd, self.bd, self.w1, self.b1 = arr(w_dilated), arr(b_dilated), arr(w_1x1), arr(b_1x1)
        self.dtype = w_dilated[0].dtype if self.n else torch.float32


def tcn_stage(stage: TcnStage, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """run the stage's DilatedResidualLayers on x [B,T,C]; x is left untouched, the result goes to `out` (or a new tensor)"""
    _need_cuda(x, out)
    assert x.dim() == 3 and x.is_contiguous() and x.dtype == stage.dtype and stage.n >= 1
    b, t, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.shape == x.shape and out.dtype == x.dtype
    scratch = torch.empty((3,) + tuple(x.shape), dtype=x.dtype, device=x.device)   # h, buf_a, buf_b
    check(lib.mt4_tcn_stage(x.data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(), scratch[0].data_ptr(), out.data_ptr(), stage.wd,
                            stage.bd, stage.w1, stage.b1, stage.n, b, t, c, dt_code(x.dtype), _stream()), "mt4_tcn_stage")
    return out


def tcn_layer(x: torch.Tensor, w_dilated: torch.Tensor, b_dilated: torch.Tensor, w_1x1: torch.Tensor, b_1x1: torch.Tensor,
              dilation: int) -> torch.Tensor:
    """one DilatedResidualLayer (`mt4_tcn_dilated_residual_layer`): x [B,T,C] -> x + conv_1x1(relu(conv_dilated(x)))"""
    _need_cuda(x, w_dilated, b_dilated, w_1x1, b_1x1)
    assert x.dim() == 3 and x.is_contiguous()
    b, t, c = x.shape
    h, y = torch.empty_like(x), torch.empty_like(x)
    check(lib.mt4_tcn_dilated_residual_layer(x.data_ptr(), w_dilated.data_ptr(), b_dilated.data_ptr(), w_1x1.data_ptr(), b_1x1.data_ptr(),
                                             h.data_ptr(), y.data_ptr(), b, t, c, dilation, dt_code(x.dtype), _stream()),
          "mt4_tcn_dilated_residual_layer")
    return y


def fpn_topdown(lat: torch.Tensor, levels: torch.Tensor) -> None:
    """levels[l] = lat[l] + levels[l+1], l = nlev-2 .. 0, in place (`mt4_fpn_topdown`)"""
    _need_cuda(lat, levels)
    nlev = levels.shape[0]
    n = levels[0].numel()
    assert lat.is_contiguous() and levels.is_contiguous() and lat.dtype == levels.dtype and lat.shape[0] == nlev - 1 and lat[0].numel() == n
    check(lib.mt4_fpn_topdown(lat.data_ptr(), levels.data_ptr(), nlev, n, dt_code(levels.dtype), _stream()), "mt4_fpn_topdown")


def stem_pad_dims(h: int, w: int) -> Tuple[int, int]:
    return h + 6, (w + 6 + 1) & ~1


def preprocess_u8(frames: torch.Tensor, mean, std, dtype: torch.dtype) -> torch.Tensor:
    """uint8 [B,H,W,3] -> normalised zero-padded [B,H+6,Wp,4] (`mt4_preprocess_u8`)."""
    _need_cuda(frames)
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[-1] == 3
    b, h, w, _ = frames.shape
    hp, wp = stem_pad_dims(h, w)
    out = torch.empty((b, hp, wp, 4), dtype=dtype, device=frames.device)
    check(lib.mt4_preprocess_u8(frames.data_ptr(), out.data_ptr(), b, h, w, _lib._FLOAT3(*mean), _lib._FLOAT3(*std),
                                dt_code(dtype), _stream()), "mt4_preprocess_u8")
    return out


def preprocess_u8_s2d(frames: torch.Tensor, mean, std) -> torch.Tensor:
    """uint8 [B,H,W,3] (H, W even) -> normalised bf16 space-to-depth image of the 3-padded frame: [B,(H+6)/2,(W+6)/2,16] with channel
    (dy*2+dx)*3 + c = pixel (2y+dy-3, 2x+dx-3), channels 12..15 zero (`mt4_preprocess_u8_s2d`): a 7x7/2 kernel row becomes 4 pixels x
    16 channels = one 128-byte run"""
    _need_cuda(frames)
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[-1] == 3
    b, h, w, _ = frames.shape
    assert h % 2 == 0 and w % 2 == 0
    out = torch.empty((b, (h + 6) // 2, (w + 6) // 2, 16), dtype=torch.bfloat16, device=frames.device)
    check(lib.mt4_preprocess_u8_s2d(frames.data_ptr(), out.data_ptr(), b, h, w, _lib._FLOAT3(*mean), _lib._FLOAT3(*std), _stream()),
          "mt4_preprocess_u8_s2d")
    return out


def stem_s2d_weight(w_oihw: torch.Tensor, scale: Optional[torch.Tensor]) -> torch.Tensor:
    """ResNet stem weight [64,3,7,7] -> packed bf16 rows for the space-to-depth stem: a 4x1 kernel over runs of 4 pixels x 16 channels,
    element kw'*16 + (dy*2+dx)*3 + c of kernel row kh' = w[c][2kh'+dy][2kw'+dx] (zero beyond the 7x7 support)"""
    co = w_oihw.shape[0]
    w8 = torch.zeros((co, 3, 8, 8), dtype=torch.float32, device=w_oihw.device)
    w8[:, :, :7, :7] = w_oihw.float()
    w8 = w8.view(co, 3, 4, 2, 4, 2)                                  # [co][c][kh'][dy][kw'][dx]
    run = torch.zeros((co, 4, 4, 16), dtype=torch.float32, device=w_oihw.device)   # [co][kh'][kw'][(dy,dx,c) + 4 zero]
    run[..., :12] = w8.permute(0, 2, 4, 3, 5, 1).reshape(co, 4, 4, 12)
    oihw = run.view(co, 4, 64).permute(0, 2, 1).reshape(co, 64, 4, 1).contiguous()  # "Cin" = 64-element run, KH = 4, KW = 1
    return pack_conv_weight(oihw, scale, torch.bfloat16)


def pad_nchw(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """normalised float32 NCHW [B,3,H,W] -> zero-padded channels-last [B,H+6,Wp,4] (`mt4_pad_nchw_f32`)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.shape[1] == 3
    x = x.contiguous()
    b, _, h, w = x.shape
    hp, wp = stem_pad_dims(h, w)
    out = torch.empty((b, hp, wp, 4), dtype=dtype, device=x.device)
    check(lib.mt4_pad_nchw_f32(x.data_ptr(), out.data_ptr(), b, h, w, dt_code(dtype), _stream()), "mt4_pad_nchw_f32")
    return out


def pil_resize_tables(in_size: int, out_size: int):
    """Pillow's `precompute_coeffs` (bilinear filter, support 1, widened by the scale when shrinking = its antialiasing) and
    `normalize_coeffs_8bpc` (22 fractional bits) in the same float64 operation order -> (bounds int32 [out,2], coeffs int32 [out,ksize])"""
    import math
    import numpy as np
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C (int): truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            v = 1.0 - t if t < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


_RESIZE_TABLES = {}


def resize_bilinear_u8(frames: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """uint8 [B,H,W,C] on the GPU -> uint8 [B,out_h,out_w,C], byte-identical to `PIL.Image.resize((out_w,out_h), BILINEAR)` of
    every frame (horizontal pass, uint8 rounding, vertical pass: `mt4_resize_pass_u8`)."""
    _need_cuda(frames)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.is_contiguous()
    b, h, w, c = frames.shape
    x = frames
    for axis, (n_in, n_out) in enumerate(((w, out_w), (h, out_h))):
        if n_in == n_out:
            continue
        key = (n_in, n_out, frames.device)
        if key not in _RESIZE_TABLES:
            bd, kk = pil_resize_tables(n_in, n_out)
            _RESIZE_TABLES[key] = (torch.from_numpy(bd).to(frames.device), torch.from_numpy(kk).to(frames.device))
        bd, kk = _RESIZE_TABLES[key]
        hin, win = x.shape[1], x.shape[2]
        hout, wout = (hin, n_out) if axis == 0 else (n_out, win)
        y = torch.empty((b, hout, wout, c), dtype=torch.uint8, device=frames.device)
        check(lib.mt4_resize_pass_u8(x.data_ptr(), y.data_ptr(), bd.data_ptr(), kk.data_ptr(), kk.shape[1], b, hin, win, hout, wout, c, axis,
                                     _stream()), "mt4_resize_pass_u8")
        x = y
    return x


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.is_contiguous()
    b, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, ho, wo, c), dtype=x.dtype, device=x.device)
    check(lib.mt4_maxpool3x3s2_nhwc(x.data_ptr(), y.data_ptr(), b, h, w, c, dt_code(x.dtype), _stream()), "mt4_maxpool3x3s2_nhwc")
    return y


def stem_maxpool_supported(h: int, w: int) -> bool:
    """frame sizes `stem_maxpool` takes: even, conv rows (w / 2 pixels) in whole 16-pixel tiles of at most 224 (frames up to 448 wide)"""
    return h % 2 == 0 and w % 32 == 0 and w // 2 <= 224


def stem_maxpool(xs: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """conv1 / bn1 / relu / maxpool of the ResNet stem in one launch (`mt4_stem_maxpool_bf16`): xs = `preprocess_u8_s2d(...)` [B,Hs,Ws,16]
    bf16, w_packed = `stem_s2d_weight(...)` -> [B,Hp,Wp,64] bf16, bit-identical to the stem conv followed by `maxpool3x3s2`"""
    _need_cuda(xs, w_packed, bias)
    assert xs.dtype == torch.bfloat16 and xs.is_contiguous() and xs.dim() == 4 and xs.shape[3] == 16
    assert w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous() and w_packed.shape[1] % 64 == 0 and bias.dtype == torch.float32
    b, hs, ws, _ = xs.shape
    cout, kh = w_packed.shape[0], w_packed.shape[1] // 64
    ho, wo = hs - kh + 1, ws - 3
    y = torch.empty((b, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1, cout), dtype=torch.bfloat16, device=xs.device)
    check(lib.mt4_stem_maxpool_bf16(xs.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), b, hs, ws, cout, kh, _stream()),
          "mt4_stem_maxpool_bf16")
    return y


def global_avgpool(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.is_contiguous()
    b, h, w, c = x.shape
    y = torch.empty((b, c), dtype=torch.float32, device=x.device)
    check(lib.mt4_global_avgpool_nhwc(x.data_ptr(), y.data_ptr(), b, h * w, c, dt_code(x.dtype), _stream()), "mt4_global_avgpool_nhwc")
    return y


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    _need_cuda(x, w, bias)
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    b, k = x.shape
    n = w.shape[0]
    y = torch.empty((b, n), dtype=torch.float32, device=x.device)
    check(lib.mt4_linear_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), b, k, n,
                             _stream()), "mt4_linear_f32")
    return y


def bottleneck_pack(conv1, conv2, conv3, ds=None):
    """the weights of one 64-mid-channel Bottleneck in the fragment order `mt4_bottleneck_fused_bf16` reads (once per model load).  conv1 / conv2 /
    conv3 / ds = (packed bf16 weight with the BatchNorm scale folded in, float32 bias); returns (w_frag, b1, b2, b3, bds | None, cin)"""
    (w1, b1), (w2, b2), (w3, b3) = conv1, conv2, conv3
    wd, bd = ds if ds is not None else (None, None)
    _need_cuda(w1, b1, w2, b2, w3, b3, wd, bd)
    cin = 64 if ds is not None else 256
    assert w1.dtype == w2.dtype == w3.dtype == torch.bfloat16 and tuple(w1.shape) == (64, packed_k(cin, 1, 1, torch.bfloat16)) and \
        tuple(w2.shape) == (64, packed_k(64, 3, 3, torch.bfloat16)) and tuple(w3.shape) == (256, packed_k(64, 1, 1, torch.bfloat16))
    assert wd is None or tuple(wd.shape) == (256, packed_k(cin, 1, 1, torch.bfloat16))
    nbytes = int(lib.mt4_bottleneck_packed_bytes(cin, 1 if ds is not None else 0))
    frag = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    check(lib.mt4_bottleneck_pack_bf16(w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), wd.data_ptr() if wd is not None else None, cin, frag.data_ptr(),
                                       _stream()), "mt4_bottleneck_pack_bf16")
    return frag, b1.contiguous(), b2.contiguous(), b3.contiguous(), (bd.contiguous() if bd is not None else None), cin


def bottleneck_fused(x: torch.Tensor, packed, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one stride-1 ResNet Bottleneck of 64 mid channels in one launch (`mt4_bottleneck_fused_bf16`): x [B,H,W,Cin] bf16 NHWC, packed =
    `bottleneck_pack(...)`"""
    frag, b1, b2, b3, bd, cin = packed
    _need_cuda(x, frag, out)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == cin
    b, h, w, _ = x.shape
    y = torch.empty((b, h, w, 256), dtype=torch.bfloat16, device=x.device) if out is None else out
    assert y.is_contiguous() and tuple(y.shape) == (b, h, w, 256) and y.dtype == torch.bfloat16
    check(lib.mt4_bottleneck_fused_bf16(x.data_ptr(), y.data_ptr(), frag.data_ptr(), b1.data_ptr(), b2.data_ptr(), b3.data_ptr(),
                                        bd.data_ptr() if bd is not None else None, b, h, w, cin, 64, _stream()), "mt4_bottleneck_fused_bf16")
    return y


def bottleneck_pack_next(conv1_next):
    """conv1 (+ folded bn1) of the strided Bottleneck behind layer1, (packed bf16 [128, 256], bias), in the fragment order of
    `bottleneck_fused_next`"""
    w, b = conv1_next
    _need_cuda(w, b)
    assert w.dtype == torch.bfloat16 and tuple(w.shape) == (128, packed_k(256, 1, 1, torch.bfloat16)) and b.numel() == 128
    frag = torch.empty(int(lib.mt4_bottleneck_next_packed_bytes()), dtype=torch.uint8, device=w.device)
    check(lib.mt4_bottleneck_pack_next_bf16(w.data_ptr(), frag.data_ptr(), _stream()), "mt4_bottleneck_pack_next_bf16")
    return frag, b.contiguous()


def bottleneck_fused_next(x: torch.Tensor, packed, packed_next):
    """`bottleneck_fused` for the last identity block of layer1 plus the following block's conv1 + bn1 + relu on the result while it is in LDS
    (`mt4_bottleneck_fused_next_bf16`): returns (y at the even pixels [B,(H+1)//2,(W+1)//2,256], conv1 output [B,H,W,128]), bit-identical to the
    separate launches"""
    frag, b1, b2, b3, bd, cin = packed
    fn, bn = packed_next
    _need_cuda(x, frag, fn)
    assert bd is None and cin == 256 and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 256
    b, h, w, _ = x.shape
    y_even = torch.empty((b, (h + 1) // 2, (w + 1) // 2, 256), dtype=torch.bfloat16, device=x.device)
    t = torch.empty((b, h, w, 128), dtype=torch.bfloat16, device=x.device)
    check(lib.mt4_bottleneck_fused_next_bf16(x.data_ptr(), y_even.data_ptr(), t.data_ptr(), frag.data_ptr(), b1.data_ptr(), b2.data_ptr(), b3.data_ptr(),
                                             fn.data_ptr(), bn.data_ptr(), b, h, w, _stream()), "mt4_bottleneck_fused_next_bf16")
    return y_even, t


# ----------------------------------------------------------------------------------------- transformer-stage pieces
def linear(x2d: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, act: Optional[str] = None,
           residual: Optional[torch.Tensor] = None, out_row_map: Optional[torch.Tensor] = None,
           out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear / 1x1 conv on rows: y[M,N] = act(x[M,K] @ W^T + b [+ residual]) through the conv kernel's GEMM mode.
    `out` / `residual` may be column slices ([:, a:b]) of wider row-major buffers."""
    m, k = x2d.shape
    y_ld = 0 if out is None or out.is_contiguous() else out.stride(0)
    res_ld = 0 if residual is None or residual.is_contiguous() else residual.stride(0)
    n = w_packed.shape[0]
    lat = bool(_LATENCY_TILES.get()) and _TCN_LINEAR and m <= _TCN_LINEAR_MAX_ROWS
    if lat and x2d.dtype == torch.float32 and n >= 8 * k and n >= 3072 and 128 <= m:
        # ... except the widest expansions (MS-TCT's linear1, C -> 8 C from C = 384): thousands of 32 x 16 tiles are many rounds of the chip, the generic
        # kernel's plain 32 x 32 tiles (no K split: same sums whatever rides along) run them 12-20 % faster (384 -> 3072: 13.2 -> 11.6 us, 864 -> 6912: 45.7 -> 36.8)
        y = conv_nhwc(x2d.view(m, 1, 1, k), w_packed, bias, kh=1, kw=1, residual=residual, act=act, out_row_map=out_row_map,
                      out_dtype=out_dtype, out=out, y_ld=y_ld, res_ld=res_ld, tile=6)
        return y if out is not None else y.view(m, -1)
    if (lat and act in (None, "none", "relu") and out_row_map is None and y_ld == 0 and res_ld == 0 and tcn_supported(k, x2d.dtype) and x2d.is_contiguous() and (residual is None or residual.dtype == x2d.dtype)
            and (out_dtype is None or out_dtype == x2d.dtype or out_dtype == torch.float32)):
        # a short sequence's nn.Linear inside a latency context (MS-TCT on one window): the temporal head's latency kernel as a 1-tap conv
        # (32 x 16 tiles, one workgroup per CU, no barrier in the K loop) instead of the K-split tiles of the generic kernel
        y = tcn_conv(x2d.view(1, m, k), w_packed, bias, taps=1, residual=residual, relu=(act == "relu"), out_dtype=out_dtype,
                     out=out.view(1, m, -1) if out is not None else None)
        return out if out is not None else y.view(m, -1)
    y = conv_nhwc(x2d.view(m, 1, 1, k), w_packed, bias, kh=1, kw=1, residual=residual, act=act, out_row_map=out_row_map,
                  out_dtype=out_dtype, out=out, y_ld=y_ld, res_ld=res_ld)
    return y if out is not None else y.view(m, -1)


def pack_linear_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[N,K] (or Conv1d/2d 1x1 [N,K,1(,1)]) float32 -> packed GEMM weight"""
    w = w.reshape(w.shape[0], -1)
    return pack_conv_weight(w[:, :, None, None], None, dtype)


def layernorm(x2d: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, row_map: Optional[torch.Tensor] = None, group: int = 1,
              l_out: int = 0, l_in: int = 0, m_out: Optional[int] = None, eps: float = 1e-5) -> torch.Tensor:
    _need_cuda(x2d, gamma, beta, row_map)
    assert x2d.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    m_in, c = x2d.shape
    m_out = m_in if m_out is None else m_out
    assert gamma.numel() == group * c
    y = torch.empty((m_out, group * c), dtype=x2d.dtype, device=x2d.device)
    check(lib.mt4_layernorm(x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), m_out, c, group,
                            row_map.data_ptr() if row_map is not None else None, l_out, l_in, eps, dt_code(x2d.dtype), _stream()),
          "mt4_layernorm")
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, batch: int, heads: int, nq: int, nk: int, hd: int,
              q_stride: int, k_stride: int, v_stride: int, scale: float, bias: Optional[torch.Tensor] = None,
              mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q/k/v may be views into one packed projection buffer (pointer + row stride addressing)."""
    _need_cuda(q, k, v, bias, mask)
    out = torch.empty((batch * nq, heads * hd), dtype=q.dtype, device=q.device)
    nw = mask.shape[0] if mask is not None else 1
    if bias is not None:
        assert bias.dtype == torch.float32 and tuple(bias.shape) == (heads, nq, nk) and bias.is_contiguous()
    if mask is not None:
        assert mask.dtype == torch.float32 and tuple(mask.shape[1:]) == (nq, nk) and mask.is_contiguous() and batch % nw == 0
    check(lib.mt4_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
                            mask.data_ptr() if mask is not None else None, batch, heads, nq, nk, hd, q_stride, k_stride, v_stride,
                            heads * hd, nw, scale, dt_code(q.dtype), _stream()), "mt4_attention")
    return out


def pad_attention_bias(bias: torch.Tensor, key_pad_value: float = -1e30) -> torch.Tensor:
    """[H|nW, N, N] float32 -> [.., NP, NP] (NP = round_up(N,16)); padded key columns get `key_pad_value`, padded query rows 0"""
    g, n, _ = bias.shape
    np_ = (n + 15) // 16 * 16
    out = torch.zeros((g, np_, np_), dtype=torch.float32, device=bias.device)
    out[:, :, n:] = key_pad_value
    out[:, :n, :n] = bias
    return out.contiguous()


def window_attention_bf16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, batch: int, heads: int, n: int, q_stride: int,
                          k_stride: int, v_stride: int, scale: float, bias_padded: torch.Tensor,
                          mask_padded: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MFMA attention core for head dim 32 (see mt4_window_attention_bf16); bias/mask padded by `pad_attention_bias`."""
    _need_cuda(q, k, v, bias_padded, mask_padded)
    assert q.dtype == torch.bfloat16 and bias_padded.dtype == torch.float32
    np_ = (n + 15) // 16 * 16
    assert tuple(bias_padded.shape) == (heads, np_, np_)
    nw = 1
    if mask_padded is not None:
        assert tuple(mask_padded.shape[1:]) == (np_, np_) and batch % mask_padded.shape[0] == 0
        nw = mask_padded.shape[0]
    out = torch.empty((batch * n, heads * 32), dtype=torch.bfloat16, device=q.device)
    check(lib.mt4_window_attention_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), bias_padded.data_ptr(),
                                        mask_padded.data_ptr() if mask_padded is not None else None, batch, heads, n, q_stride, k_stride,
                                        v_stride, heads * 32, nw, scale, _stream()), "mt4_window_attention_bf16")
    return out


def window_attention_rel_bf16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, batch: int, heads: int, ws: int, q_stride: int,
                              k_stride: int, v_stride: int, scale: float, rel_table: torch.Tensor,
                              region: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MFMA attention core of a Swin block from its relative-position table [heads, (2ws-1)^2] (fp32) and, for a shifted block, the
    region ids [nW, ws*ws] (int32) of the window types -- no [N,N] bias / mask tensors (see mt4_window_attention_rel_bf16)"""
    _need_cuda(q, k, v, rel_table, region)
    n = ws * ws
    assert q.dtype == torch.bfloat16 and rel_table.dtype == torch.float32 and tuple(rel_table.shape) == (heads, (2 * ws - 1) ** 2)
    assert rel_table.is_contiguous()
    nw = 1
    if region is not None:
        assert region.dtype == torch.int32 and region.is_contiguous() and region.shape[1] == n and batch % region.shape[0] == 0
        nw = region.shape[0]
    out = torch.empty((batch * n, heads * 32), dtype=torch.bfloat16, device=q.device)
    check(lib.mt4_window_attention_rel_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), rel_table.data_ptr(),
                                            region.data_ptr() if region is not None else None, ws, batch, heads, q_stride, k_stride, v_stride,
                                            heads * 32, nw, scale, _stream()), "mt4_window_attention_rel_bf16")
    return out


def patchify(img: torch.Tensor, patch: int, dtype: torch.dtype, mean=None, std=None) -> torch.Tensor:
    """float32 NCHW [B,3,H,W] (normalised) or uint8 NHWC [B,H,W,3] -> [B*H/P*W/P, 3*P*P] rows"""
    _need_cuda(img)
    img = img.contiguous()
    from_u8 = img.dtype == torch.uint8
    if from_u8:
        b, h, w, _ = img.shape
    else:
        assert img.dtype == torch.float32
        b, _, h, w = img.shape
    out = torch.empty((b * (h // patch) * (w // patch), 3 * patch * patch), dtype=dtype, device=img.device)
    z = _lib._FLOAT3(0, 0, 0)
    check(lib.mt4_patchify(img.data_ptr(), out.data_ptr(), b, h, w, patch, 1 if from_u8 else 0,
                           _lib._FLOAT3(*mean) if from_u8 else z, _lib._FLOAT3(*std) if from_u8 else z, dt_code(dtype), _stream()),
          "mt4_patchify")
    return out


def add_rowbcast(x2d: torch.Tensor, p2d: torch.Tensor) -> torch.Tensor:
    _need_cuda(x2d, p2d)
    assert x2d.is_contiguous() and p2d.is_contiguous() and x2d.dtype == p2d.dtype and x2d.shape[1] == p2d.shape[1]
    y = torch.empty_like(x2d)
    check(lib.mt4_add_rowbcast(x2d.data_ptr(), p2d.data_ptr(), y.data_ptr(), x2d.shape[0], p2d.shape[0], x2d.shape[1],
                               dt_code(x2d.dtype), _stream()), "mt4_add_rowbcast")
    return y


def groupwise_linear(hs: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], batch: int, k: int) -> torch.Tensor:
    _need_cuda(hs, w, bias)
    assert hs.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    d = hs.shape[-1]
    out = torch.empty((batch, k), dtype=torch.float32, device=hs.device)
    check(lib.mt4_groupwise_linear(hs.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), batch, k, d,
                                   dt_code(hs.dtype), _stream()), "mt4_groupwise_linear")
    return out


def dwconv1d_k3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: str = "none") -> torch.Tensor:
    """x [B,T,C]; w float32 [C,3]"""
    _need_cuda(x, w, bias)
    assert x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    b, t, c = x.shape
    y = torch.empty_like(x)
    check(lib.mt4_dwconv1d_k3(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), b, t, c,
                              {"none": 0, "relu": 1, "gelu": 2}[act], dt_code(x.dtype), _stream()), "mt4_dwconv1d_k3")
    return y


def kd_mix(s: torch.Tensor, ti: torch.Tensor, tv: torch.Tensor, tt: torch.Tensor):
    _need_cuda(s, ti, tv, tt)
    for t in (s, ti, tv, tt):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == s.shape
    outs = [torch.empty_like(s) for _ in range(3)]
    check(lib.mt4_kd_mix(s.data_ptr(), ti.data_ptr(), tv.data_ptr(), tt.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                         outs[2].data_ptr(), s.shape[0], s.shape[1], _stream()), "mt4_kd_mix")
    return tuple(outs)


# ----------------------------------------------------------------------------------------- training pieces (fp32)
def wgrad_conv1d(dy: torch.Tensor, x: torch.Tensor, dw_packed: torch.Tensor, *, batch: int, t: int, taps: int, dil: int, pad: int,
                 accumulate: bool = False, bias_grad: Optional[torch.Tensor] = None) -> None:
    """bias_grad [cout] (optional, zeroed by the caller): += the column sums of dy in the same launch (instead of `colsum`)"""
    _need_cuda(dy, x, dw_packed, bias_grad)
    assert dy.dtype == x.dtype == dw_packed.dtype == torch.float32 and dy.is_contiguous() and x.is_contiguous() and dw_packed.is_contiguous()
    cout, cin = dy.shape[-1], x.shape[-1]
    assert dw_packed.shape == (cout, packed_k(cin, 1, taps, torch.float32)) and dy.numel() == batch * t * cout
    assert bias_grad is None or (bias_grad.dtype == torch.float32 and bias_grad.numel() >= cout)
    check(lib.mt4_wgrad_conv1d_f32(dy.data_ptr(), x.data_ptr(), dw_packed.data_ptr(), batch, t, cout, cin, taps, dil, pad,
                                   1 if accumulate else 0, bias_grad.data_ptr() if bias_grad is not None else None, _stream()), "mt4_wgrad_conv1d_f32")


def colsum(x2d: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> None:
    _need_cuda(x2d, out)
    assert x2d.dtype == out.dtype == torch.float32 and x2d.stride(-1) == 1
    m, c = x2d.shape
    check(lib.mt4_colsum_f32(x2d.data_ptr(), out.data_ptr(), m, out.numel(), x2d.stride(0), 1 if accumulate else 0, _stream()), "mt4_colsum_f32")


def bce_logits(y: torch.Tensor, z: torch.Tensor, col_scale: torch.Tensor, dy: torch.Tensor, col_loss: torch.Tensor) -> None:
    """y [M,N] (row pitch y.stride(0)), z [M,N] dense, dy [M,>=N] (row pitch dy.stride(0)); col_loss accumulates"""
    _need_cuda(y, z, col_scale, dy, col_loss)
    m, n = z.shape
    assert z.is_contiguous() and y.stride(-1) == 1 and dy.stride(-1) == 1
    check(lib.mt4_bce_logits_f32(y.data_ptr(), z.data_ptr(), col_scale.data_ptr(), dy.data_ptr(), col_loss.data_ptr(), m, n, y.stride(0),
                                 dy.stride(0), _stream()), "mt4_bce_logits_f32")


def sgd_step(p: torch.Tensor, g: torch.Tensor, lr: float, weight_decay: float, grad_scale: float = 1.0) -> None:
    _need_cuda(p, g)
    assert p.dtype == g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and p.numel() == g.numel()
    check(lib.mt4_sgd_step_f32(p.data_ptr(), g.data_ptr(), p.numel(), lr, weight_decay, grad_scale, _stream()), "mt4_sgd_step_f32")


def mul_add(a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(a, b, c)
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float32 and a.numel() == b.numel()
    y = torch.empty_like(a) if out is None else out
    check(lib.mt4_mul_add_f32(a.data_ptr(), b.data_ptr(), c.data_ptr() if c is not None else None, y.data_ptr(), a.numel(), _stream()),
          "mt4_mul_add_f32")
    return y


def transpose_pack_conv1d(w_packed: torch.Tensor, cout: int, cin: int, taps: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(w_packed)
    kt = packed_k(cout, 1, taps, torch.float32)
    wt = torch.empty((cin, kt), dtype=torch.float32, device=w_packed.device) if out is None else out
    check(lib.mt4_transpose_pack_conv1d_f32(w_packed.data_ptr(), wt.data_ptr(), cout, cin, taps, _stream()), "mt4_transpose_pack_conv1d_f32")
    return wt


# ----------------------------------------------------------------------------------------- spatial training pieces (fp32)
def bn_stats(x2d, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, sums=None):
    """`sums`: optional zeroed float64 scratch of 2*C (callers with many BatchNorms zero one arena per step)"""
    _need_cuda(x2d)
    m, c = x2d.shape
    if sums is None:
        sums = torch.zeros(2 * c, dtype=torch.float64, device=x2d.device)   # scratch: float64 accumulators
    assert sums.dtype == torch.float64 and sums.numel() == 2 * c
    mean, invstd = torch.empty(c, device=x2d.device), torch.empty(c, device=x2d.device)
    check(lib.mt4_bn_stats_f32(x2d.data_ptr(), sums.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                               running_mean.data_ptr() if running_mean is not None else None,
                               running_var.data_ptr() if running_var is not None else None, m, c, momentum, eps, _stream()), "mt4_bn_stats_f32")
    return mean, invstd


def bn_apply(x2d, mean, invstd, gamma, beta, residual=None, relu=True):
    y = torch.empty_like(x2d)
    m, c = x2d.shape
    check(lib.mt4_bn_apply_f32(x2d.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                               residual.data_ptr() if residual is not None else None, y.data_ptr(), m, c, 1 if relu else 0, _stream()),
          "mt4_bn_apply_f32")
    return y


def bn_backward(dy, y_post, x2d, mean, invstd, gamma, dgamma, dbeta, relu=True, want_dres=False, sums=None, beta=None):
    """with `beta` and relu the ReLU gate is recomputed from x2d (the forward's fp32 expression: the stored output's sign bit for bit) instead
    of read from y_post -- units without a residual input"""
    m, c = x2d.shape
    if sums is None:
        sums = torch.zeros(2 * c, dtype=torch.float64, device=x2d.device)   # scratch: float64 accumulators
    assert sums.dtype == torch.float64 and sums.numel() == 2 * c
    dx = torch.empty_like(x2d)
    dres = torch.empty_like(x2d) if want_dres else None
    code = 0 if not relu else (2 if beta is not None else 1)
    check(lib.mt4_bn_backward_f32(dy.data_ptr(), y_post.data_ptr() if (y_post is not None and code == 1) else None, x2d.data_ptr(), mean.data_ptr(),
                                  invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr() if beta is not None else None, sums.data_ptr(), dx.data_ptr(),
                                  dres.data_ptr() if want_dres else None, dgamma.data_ptr(), dbeta.data_ptr(), m, c, code, _stream()),
          "mt4_bn_backward_f32")
    return dx, dres


def wgrad_conv2d(dy, x, dw_packed, kh, kw, stride, pad, dil=(1, 1), zero=True):
    """dy [B,Ho,Wo,Cout], x [B,H,W,Cin]; dw_packed is zeroed here (zero=False: the caller already did) and filled"""
    _need_cuda(dy, x, dw_packed)
    b, ho, wo, cout = dy.shape
    _, h, w, cin = x.shape
    assert dw_packed.shape == (cout, packed_k(cin, kh, kw, torch.float32)) and dy.is_contiguous() and x.is_contiguous()
    if zero:
        dw_packed.zero_()
    check(lib.mt4_wgrad_conv2d_f32(dy.data_ptr(), x.data_ptr(), dw_packed.data_ptr(), b, h, w, cin, ho, wo, cout, kh, kw, stride[0], stride[1],
                                   pad[0], pad[1], dil[0], dil[1], _stream()), "mt4_wgrad_conv2d_f32")


def maxpool3x3s2_bwd(x, dy):
    b, h, w, c = x.shape
    dx = torch.zeros_like(x)
    check(lib.mt4_maxpool3x3s2_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), b, h, w, c, _stream()), "mt4_maxpool3x3s2_bwd_f32")
    return dx


def avgpool_bwd(dfeat, b, hw, c):
    dx = torch.empty((b, hw, c), dtype=torch.float32, device=dfeat.device)
    check(lib.mt4_avgpool_bwd_f32(dfeat.data_ptr(), dx.data_ptr(), b, hw, c, _stream()), "mt4_avgpool_bwd_f32")
    return dx


def bce_logits_pw(y, z, pos_weight, col_scale, dy, col_loss):
    m, n = z.shape
    check(lib.mt4_bce_logits_pw_f32(y.data_ptr(), z.data_ptr(), pos_weight.data_ptr() if pos_weight is not None else None, col_scale.data_ptr(),
                                    dy.data_ptr(), col_loss.data_ptr(), m, n, y.stride(0), dy.stride(0), _stream()), "mt4_bce_logits_pw_f32")


def distill_kl(y_s, t_pred, dy_s, loss, temp, grad_scale, accumulate=True):
    b, k = t_pred.shape
    check(lib.mt4_distill_kl_f32(y_s.data_ptr(), t_pred.data_ptr(), dy_s.data_ptr(), loss.data_ptr(), b, k, y_s.stride(0), dy_s.stride(0), temp,
                                 grad_scale, 1 if accumulate else 0, _stream()), "mt4_distill_kl_f32")


def mse(a, b, loss, grad_scale):
    da = torch.empty_like(a)
    check(lib.mt4_mse_f32(a.data_ptr(), b.data_ptr(), da.data_ptr(), loss.data_ptr(), a.numel(), grad_scale, _stream()), "mt4_mse_f32")
    return da


def kd_mix_bwd(s, teas, gs):
    b, c = s.shape
    ds = torch.empty_like(s)
    dtau = torch.empty((b, 3), dtype=torch.float32, device=s.device)
    check(lib.mt4_kd_mix_bwd_f32(s.data_ptr(), teas[0].data_ptr(), teas[1].data_ptr(), teas[2].data_ptr(), gs[0].data_ptr(), gs[1].data_ptr(),
                                 gs[2].data_ptr(), ds.data_ptr(), dtau.data_ptr(), b, c, _stream()), "mt4_kd_mix_bwd_f32")
    return ds, dtau


# ----------------------------------------------------------------------------------------- bf16-operand training pieces
BF16 = torch.bfloat16


def avgpool1d_rows(x: torch.Tensor, k: int = 7, stride: int = 3) -> torch.Tensor:
    """nn.AvgPool1d(k, stride) over the time axis of frame-major rows x [B, T, C] (`Temporal_tenco/network.py:147,154-155`)"""
    _need_cuda(x)
    assert x.dim() == 3 and x.is_contiguous()
    b, t, c = x.shape
    y = torch.empty((b, (t - k) // stride + 1, c), dtype=x.dtype, device=x.device)
    check(lib.mt4_avgpool1d_rows(x.data_ptr(), y.data_ptr(), b, t, c, k, stride, dt_code(x.dtype), _stream()), "mt4_avgpool1d_rows")
    return y


def tcn_layer_fused(x: torch.Tensor, w1_frag: torch.Tensor, b1: torch.Tensor, w2_frag: torch.Tensor, b2: torch.Tensor, dilation: int,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One DilatedResidualLayer in one launch (`mt4_tcn_layer_fused_bf16`; `Temporal_tenco/network.py:186-198`): x [B, T, 512] bf16 -> same shape,
    the hidden map stays in LDS; w*_frag = `pack_fragments` of the packed conv weights.  Bit-identical to the two `conv_nhwc` launches."""
    _need_cuda(x, w1_frag, b1, w2_frag, b2, out)
    assert x.dim() == 3 and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[2] == 512
    assert tuple(w1_frag.shape) == (512, 1536) and tuple(w2_frag.shape) == (512, 512) and w1_frag.dtype == w2_frag.dtype == torch.bfloat16
    assert b1.dtype == b2.dtype == torch.float32 and b1.numel() == b2.numel() == 512
    y = torch.empty_like(x) if out is None else out
    assert y.data_ptr() != x.data_ptr() and y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    check(lib.mt4_tcn_layer_fused_bf16(x.data_ptr(), w1_frag.data_ptr(), b1.data_ptr(), w2_frag.data_ptr(), b2.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1],
                                       512, int(dilation), _stream()), "mt4_tcn_layer_fused_bf16")
    return y


def avgpool1d_rows_bwd(dy: torch.Tensor, t_in: int, k: int = 7, stride: int = 3) -> torch.Tensor:
    """adjoint of `avgpool1d_rows`: dy [B, (t_in - k) // stride + 1, C] fp32 -> dx [B, t_in, C]"""
    _need_cuda(dy)
    assert dy.dim() == 3 and dy.is_contiguous() and dy.dtype == torch.float32 and dy.shape[1] == (t_in - k) // stride + 1
    b, _, c = dy.shape
    dx = torch.empty((b, t_in, c), dtype=torch.float32, device=dy.device)
    check(lib.mt4_avgpool1d_rows_bwd_f32(dy.data_ptr(), dx.data_ptr(), b, t_in, c, k, stride, _stream()), "mt4_avgpool1d_rows_bwd_f32")
    return dx


def interp_linear_rows_bwd(dy: torch.Tensor, t_in: int) -> torch.Tensor:
    """adjoint of `interp_linear_rows(x [B, t_in, C], t_out)`: dy [B, t_out, C] fp32 -> dx [B, t_in, C]"""
    _need_cuda(dy)
    assert dy.dim() == 3 and dy.is_contiguous() and dy.dtype == torch.float32
    b, t_out, c = dy.shape
    dx = torch.empty((b, t_in, c), dtype=torch.float32, device=dy.device)
    check(lib.mt4_interp_linear_rows_bwd_f32(dy.data_ptr(), dx.data_ptr(), b, t_in, t_out, c, _stream()), "mt4_interp_linear_rows_bwd_f32")
    return dx


def interp_linear_rows(x: torch.Tensor, t_out: int) -> torch.Tensor:
    """F.interpolate(x, size=t_out, mode='linear') over the time axis of frame-major rows x [B, T, C] (`Temporal_tenco/network.py:96`)"""
    _need_cuda(x)
    assert x.dim() == 3 and x.is_contiguous()
    b, t, c = x.shape
    y = torch.empty((b, t_out, c), dtype=x.dtype, device=x.device)
    check(lib.mt4_interp_linear_rows(x.data_ptr(), y.data_ptr(), b, t, t_out, c, dt_code(x.dtype), _stream()), "mt4_interp_linear_rows")
    return y


def bn_stats_t(x2d, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, sums=None):
    """`bn_stats` for a bf16 (or fp32) convolution output"""
    _need_cuda(x2d)
    m, c = x2d.shape
    assert sums is not None and sums.dtype == torch.float64 and sums.numel() == 2 * c and x2d.is_contiguous()
    mean, invstd = torch.empty(c, device=x2d.device), torch.empty(c, device=x2d.device)
    check(lib.mt4_bn_stats_t(x2d.data_ptr(), dt_code(x2d.dtype), sums.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                             running_mean.data_ptr() if running_mean is not None else None,
                             running_var.data_ptr() if running_var is not None else None, m, c, momentum, eps, _stream()), "mt4_bn_stats_t")
    return mean, invstd


def bn_apply_sums(x2d, stat_sums, gamma, beta, residual=None, relu=True, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """fp32 twin of `bn_apply_sums_t`: fp32 convolution output, fp32 activation.  Returns (y, mean, invstd)"""
    _need_cuda(x2d, stat_sums, gamma, beta, residual)
    m, c = x2d.shape
    assert x2d.dtype == torch.float32 and x2d.is_contiguous() and stat_sums.dtype == torch.float64 and stat_sums.numel() == STAT_REPLICAS * 2 * c
    mean = torch.empty(c, dtype=torch.float32, device=x2d.device)
    invstd = torch.empty_like(mean)
    y = torch.empty_like(x2d)
    check(lib.mt4_bn_apply_sums_f32(x2d.data_ptr(), stat_sums.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                    running_mean.data_ptr() if running_mean is not None else None,
                                    running_var.data_ptr() if running_var is not None else None, gamma.data_ptr(), beta.data_ptr(),
                                    residual.data_ptr() if residual is not None else None, y.data_ptr(), m, c, momentum, eps, 1 if relu else 0, _stream()),
          "mt4_bn_apply_sums_f32")
    return y, mean, invstd


def bn_apply_sums_t(x2d, stat_sums, gamma, beta, residual=None, relu=True, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """train-mode BatchNorm of a convolution output from the channel sums the convolution's epilogue left (`conv_nhwc(stat_sums=...)`): ONE launch
    computes mean / invstd (returned for the backward), updates the running statistics and applies y = act((x - mean) * invstd * gamma + beta
    [+ residual]).  Returns (y, mean, invstd)"""
    _need_cuda(x2d, stat_sums, gamma, beta, residual)
    m, c = x2d.shape
    assert x2d.is_contiguous() and stat_sums.dtype == torch.float64 and stat_sums.numel() == STAT_REPLICAS * 2 * c
    mean = torch.empty(c, dtype=torch.float32, device=x2d.device)
    invstd = torch.empty_like(mean)
    y = torch.empty((m, c), dtype=BF16, device=x2d.device)
    check(lib.mt4_bn_apply_sums_t(x2d.data_ptr(), dt_code(x2d.dtype), stat_sums.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                  running_mean.data_ptr() if running_mean is not None else None,
                                  running_var.data_ptr() if running_var is not None else None, gamma.data_ptr(), beta.data_ptr(),
                                  residual.data_ptr() if residual is not None else None, y.data_ptr(), m, c, momentum, eps, 1 if relu else 0, _stream()),
          "mt4_bn_apply_sums_t")
    return y, mean, invstd


def bn_apply_t(x2d, mean, invstd, gamma, beta, residual=None, relu=True):
    """bf16 output; x2d bf16 or fp32, residual bf16"""
    m, c = x2d.shape
    assert x2d.is_contiguous() and (residual is None or (residual.dtype == BF16 and residual.is_contiguous()))
    y = torch.empty((m, c), dtype=BF16, device=x2d.device)
    check(lib.mt4_bn_apply_t(x2d.data_ptr(), dt_code(x2d.dtype), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                             residual.data_ptr() if residual is not None else None, y.data_ptr(), m, c, 1 if relu else 0, _stream()), "mt4_bn_apply_t")
    return y


def bn_backward_t(dy, y_post, x2d, mean, invstd, gamma, dgamma, dbeta, relu=True, want_dres=False, sums=None, beta=None):
    """dy / y_post bf16; x2d (the convolution output) bf16 or fp32 -> dx of that type; dres bf16.  With `beta` and relu the ReLU gate is recomputed
    from x2d instead of read from y_post (units without a residual input)"""
    m, c = x2d.shape
    assert sums is not None and sums.dtype == torch.float64 and sums.numel() == 2 * c
    assert dy.dtype == BF16 and dy.is_contiguous() and x2d.is_contiguous() and (y_post is None or (y_post.dtype == BF16 and y_post.is_contiguous()))
    dx = torch.empty_like(x2d)
    dres = torch.empty((m, c), dtype=BF16, device=x2d.device) if want_dres else None
    code = 0 if not relu else (2 if beta is not None else 1)
    check(lib.mt4_bn_backward_t(dy.data_ptr(), y_post.data_ptr() if (y_post is not None and code == 1) else None, x2d.data_ptr(), dt_code(x2d.dtype),
                                mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr() if beta is not None else None, sums.data_ptr(),
                                dx.data_ptr(), dres.data_ptr() if want_dres else None, dgamma.data_ptr(), dbeta.data_ptr(), m, c, code, _stream()),
          "mt4_bn_backward_t")
    return dx, dres


def wgrad_conv2d_bf16(dy, x, dw_packed, k, stride):
    """dw_packed (fp32 packed, ADDED to) from dy [B,Ho,Wo,Cout] and x [B,H,W,Cin], both bf16"""
    _need_cuda(dy, x, dw_packed)
    b, ho, wo, cout = dy.shape
    _, h, w, cin = x.shape
    assert dy.dtype == x.dtype == BF16 and dy.is_contiguous() and x.is_contiguous() and dw_packed.dtype == torch.float32
    assert dw_packed.shape == (cout, packed_k(cin, k, k, torch.float32))
    check(lib.mt4_wgrad_conv2d_bf16(dy.data_ptr(), x.data_ptr(), dw_packed.data_ptr(), b, h, w, cin, ho, wo, cout, k, stride, _stream()),
          "mt4_wgrad_conv2d_bf16")


def maxpool3x3s2_bwd_bf16(x, dy):
    b, h, w, c = x.shape
    assert x.dtype == dy.dtype == BF16 and x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    check(lib.mt4_maxpool3x3s2_bwd_bf16(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), b, h, w, c, _stream()), "mt4_maxpool3x3s2_bwd_bf16")
    return dx


def avgpool_bwd_bf16(dfeat, b, hw, c):
    dx = torch.empty((b, hw, c), dtype=BF16, device=dfeat.device)
    check(lib.mt4_avgpool_bwd_bf16(dfeat.data_ptr(), dx.data_ptr(), b, hw, c, _stream()), "mt4_avgpool_bwd_bf16")
    return dx


REFRESH_TILES_PER_BLOCK = 4          # MT4_REFRESH_TILES_PER_BLOCK of include/mt4hip.h


class RefreshTable:
    """the derived weight matrices of a trainer (bf16 forward copies, transposed data-gradient operators, sub-pixel phase kernels), rebuilt from the
    fp32 master weights by ONE launch (`mt4_refresh_weights`).  `add` allocates a zeroed destination and records how it is filled; `run` launches."""

    def __init__(self, device):
        self.dev, self._entries, self._blocks, self._table = device, [], 0, None

    def add(self, src: torch.Tensor, cout: int, cin: int, dtype: torch.dtype, transposed: bool, tap_map, out_taps_shape=None) -> torch.Tensor:
        """src: packed fp32 master [cout][kpad] with taps of roundup(cin, 4) elements.  Returns dst: [cout][taps x cin] (transposed False) or
        [cin][taps x cout] (transposed True) in the packed layout of `dtype`; dst tap t is the master's tap tap_map[t].  out_taps_shape = (kh, kw) of
        the destination kernel (default (1, len(tap_map)))"""
        from ._lib import RefreshEntry
        tap_map = list(tap_map)
        kh, kw = out_taps_shape or (1, len(tap_map))
        assert kh * kw == len(tap_map) <= 9 and src.dtype == torch.float32 and src.is_contiguous() and src.shape[0] == cout
        rows, cols = (cin, cout) if transposed else (cout, cin)
        dst = torch.zeros((rows, packed_k(cols, kh, kw, dtype)), dtype=dtype, device=self.dev)
        e = RefreshEntry()
        e.src, e.dst, e.block0 = src.data_ptr(), dst.data_ptr(), self._blocks
        e.dst_bf16, e.transposed, e.cout, e.cin, e.ntaps_dst = int(dtype == torch.bfloat16), int(transposed), cout, cin, len(tap_map)
        e.tapw_src, e.kpad_src = (cin + 3) // 4 * 4, src.shape[1]
        e.tapw_dst, e.kpad_dst = ((cols + 7) // 8 * 8 if dtype == torch.bfloat16 else (cols + 3) // 4 * 4), dst.shape[1]
        for i, t in enumerate(tap_map):
            e.tap_map[i] = t
        self._entries.append(e)
        self._keep = getattr(self, "_keep", []) + [src, dst]
        tiles = len(tap_map) * ((cout + 31) // 32) * ((cin + 31) // 32)
        self._blocks += (tiles + REFRESH_TILES_PER_BLOCK - 1) // REFRESH_TILES_PER_BLOCK
        self._table = None
        return dst

    def run(self) -> None:
        if not self._entries:
            return
        if self._table is None:
            raw = b"".join(bytes(e) for e in self._entries)
            self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        check(lib.mt4_refresh_weights(self._table.data_ptr(), len(self._entries), self._blocks, _stream()), "mt4_refresh_weights")


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 copy (round to nearest even): the operand of a mixed-precision GEMM"""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib.mt4_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mt4_cast_f32_bf16")
    return y


def repack_weight_bf16(w_f32_packed, cout, cin, kh, kw, out=None):
    """packed fp32 weights -> packed bf16 weights of the same geometry (into `out` when given: captured graphs keep the address)"""
    _need_cuda(w_f32_packed, out)
    assert w_f32_packed.dtype == torch.float32 and w_f32_packed.is_contiguous() and tuple(w_f32_packed.shape) == (cout, packed_k(cin, kh, kw, torch.float32))
    o = torch.empty((cout, packed_k(cin, kh, kw, BF16)), dtype=BF16, device=w_f32_packed.device) if out is None else out
    check(lib.mt4_repack_weight_bf16(w_f32_packed.data_ptr(), o.data_ptr(), cout, cin, kh, kw, _stream()), "mt4_repack_weight_bf16")
    return o


# ----------------------------------------------------------------------------------------- MS-TCT training pieces (fp32)
def bgemm(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, *, m: int, n: int, k: int, nb0: int, nb1: int, a_strides, b_strides, c_strides,
          alpha: float = 1.0, accumulate: bool = False) -> torch.Tensor:
    """strided batched GEMM `mt4_bgemm_f32`: C[b1][b0] = alpha * A.B (+C); strides {b0, b1, rows, cols} in elements; a / b / c are the tensors
    whose data_ptr is the origin (storage offsets of views are honoured)"""
    _need_cuda(a, b, c)
    assert a.dtype == b.dtype == c.dtype == torch.float32
    i64x4 = C.c_int64 * 4
    check(lib.mt4_bgemm_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, nb0, nb1, i64x4(*a_strides), i64x4(*b_strides), i64x4(*c_strides),
                            alpha, 1 if accumulate else 0, _stream()), "mt4_bgemm_f32")
    return c


def softmax_rows_(s: torch.Tensor, scale: float) -> torch.Tensor:
    """in place: rows of the last dimension -> softmax(scale * row)"""
    _need_cuda(s)
    assert s.dtype == torch.float32 and s.is_contiguous()
    check(lib.mt4_softmax_rows_f32(s.data_ptr(), s.numel() // s.shape[-1], s.shape[-1], scale, _stream()), "mt4_softmax_rows_f32")
    return s


def softmax_bwd_rows_(p: torch.Tensor, dp: torch.Tensor, scale: float) -> torch.Tensor:
    """dp <- scale * p .* (dp - rowsum(p .* dp))"""
    _need_cuda(p, dp)
    assert p.dtype == dp.dtype == torch.float32 and p.is_contiguous() and dp.is_contiguous() and p.shape == dp.shape
    check(lib.mt4_softmax_bwd_rows_f32(p.data_ptr(), dp.data_ptr(), p.numel() // p.shape[-1], p.shape[-1], scale, _stream()),
          "mt4_softmax_bwd_rows_f32")
    return dp


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                  dx: Optional[torch.Tensor] = None, accumulate_dx: bool = False, eps: float = 1e-5) -> torch.Tensor:
    """dgamma / dbeta are ADDED to; dx is written (or added to with accumulate_dx)"""
    _need_cuda(dy, x, gamma, dgamma, dbeta, dx)
    m, c = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dy.shape == x.shape and dy.dtype == x.dtype == torch.float32
    if dx is None:
        assert not accumulate_dx
        dx = torch.empty_like(x)
    check(lib.mt4_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), m, c, eps,
                                    1 if accumulate_dx else 0, _stream()), "mt4_layernorm_bwd_f32")
    return dx


def gelu(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty_like(x)
    check(lib.mt4_gelu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mt4_gelu_f32")
    return y


def gelu_bwd(dy: torch.Tensor, x_pre: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(dy, x_pre, out)
    assert dy.is_contiguous() and x_pre.is_contiguous() and dy.numel() == x_pre.numel() and dy.dtype == torch.float32
    dx = torch.empty_like(dy) if out is None else out
    check(lib.mt4_gelu_bwd_f32(dy.data_ptr(), x_pre.data_ptr(), dx.data_ptr(), dy.numel(), _stream()), "mt4_gelu_bwd_f32")
    return dx


def dwconv1d_k3_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, db: torch.Tensor) -> torch.Tensor:
    """dy, x [B,T,C]; w [C,3]; dw [C,3] / db [C] are ADDED to; returns dx"""
    _need_cuda(dy, x, w, dw, db)
    b, t, c = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dy.shape == x.shape and w.is_contiguous() and dw.is_contiguous()
    dx = torch.empty_like(x)
    check(lib.mt4_dwconv1d_k3_bwd_f32(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), b, t, c, _stream()),
          "mt4_dwconv1d_k3_bwd_f32")
    return dx


def axpby_(x: torch.Tensor, y: torch.Tensor, a: float = 1.0, b: float = 1.0) -> torch.Tensor:
    """y <- a * x + b * y"""
    _need_cuda(x, y)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel() and x.dtype == y.dtype == torch.float32
    check(lib.mt4_axpby_f32(x.data_ptr(), y.data_ptr(), x.numel(), a, b, _stream()), "mt4_axpby_f32")
    return y


def dropout_mask(shape, seed: int, stream_id: int, p: float = 0.5, device="cuda") -> torch.Tensor:
    """keep mask of nn.Dropout(p) drawn on the device from the counter generator of `synth.uniform01(seed, stream_id, n)`: 1/(1-p) where
    u >= p, else 0 (`mt4_dropout_mask_f32`)"""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib.mt4_dropout_mask_f32(out.data_ptr(), out.numel(), seed, stream_id, p, _stream()), "mt4_dropout_mask_f32")
    return out


# ----------------------------------------------------------------------------------------- Swin / Q2L training pieces (fp32)
def gather_rows(x: torch.Tensor, row_map: torch.Tensor, *, l_out: int, l_in: int, group: int = 1, m_out: Optional[int] = None) -> torch.Tensor:
    """y[m][g*C:(g+1)*C] = x[(m // l_out) * l_in + map[(m % l_out) * group + g]]  (`mt4_gather_rows_f32`)"""
    _need_cuda(x, row_map)
    assert x.is_contiguous() and x.dtype == torch.float32 and row_map.dtype == torch.int32 and row_map.numel() == l_out * group
    m_in, c = x.shape
    m_out = m_in if m_out is None else m_out
    y = torch.empty((m_out, group * c), dtype=torch.float32, device=x.device)
    check(lib.mt4_gather_rows_f32(x.data_ptr(), row_map.data_ptr(), y.data_ptr(), m_out, c, group, l_out, l_in, 0, _stream()), "mt4_gather_rows_f32")
    return y


def scatter_rows(y: torch.Tensor, row_map: torch.Tensor, *, l_out: int, l_in: int, group: int = 1, m_in: Optional[int] = None) -> torch.Tensor:
    """the inverse of `gather_rows` (the maps are bijections): x[(m // l_out) * l_in + map[...]] = y[m][g*C:(g+1)*C]"""
    _need_cuda(y, row_map)
    assert y.is_contiguous() and y.dtype == torch.float32 and row_map.dtype == torch.int32
    m_out, gc = y.shape
    c = gc // group
    m_in = m_out * group if m_in is None else m_in
    x = torch.empty((m_in, c), dtype=torch.float32, device=y.device)
    check(lib.mt4_gather_rows_f32(x.data_ptr(), row_map.data_ptr(), y.data_ptr(), m_out, c, group, l_out, l_in, 1, _stream()), "mt4_gather_rows_f32")
    return x


def add_bias_mask_(s: torch.Tensor, bias: torch.Tensor, mask: Optional[torch.Tensor], index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """s [nWin, H, N, N] += bias (+ mask [nW, N, N] of window nWin % nW); bias dense [H, N, N], or with `index` [N*N] (int32) the
    relative-position table [(2ws-1)^2, H] read through it"""
    _need_cuda(s, bias, mask, index)
    nwin, h, n, _ = s.shape
    assert s.is_contiguous() and bias.is_contiguous() and (mask is None or mask.is_contiguous())
    assert (tuple(bias.shape) == (h, n, n)) if index is None else (bias.shape[1] == h and index.dtype == torch.int32 and index.numel() == n * n)
    check(lib.mt4_add_bias_mask_f32(s.data_ptr(), bias.data_ptr(), index.data_ptr() if index is not None else None,
                                    mask.data_ptr() if mask is not None else None, nwin, h, n, mask.shape[0] if mask is not None else 1, _stream()),
          "mt4_add_bias_mask_f32")
    return s


def relpos_table_grad(ds: torch.Tensor, index: torch.Tensor, dtable: torch.Tensor) -> None:
    """dtable [(2ws-1)^2, H] += sum over windows of ds [nWin, H, N, N] scattered through index [N*N] (int32)"""
    _need_cuda(ds, index, dtable)
    nwin, h, n, _ = ds.shape
    assert ds.is_contiguous() and index.dtype == torch.int32 and index.numel() == n * n and dtable.is_contiguous() and dtable.shape[1] == h
    check(lib.mt4_relpos_table_grad_f32(ds.data_ptr(), index.data_ptr(), dtable.data_ptr(), nwin, h, n, _stream()), "mt4_relpos_table_grad_f32")


def rowscale_add(x: torch.Tensor, scale: torch.Tensor, r: Optional[torch.Tensor], rows_per_scale: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m] = scale[m // rows_per_scale] * x[m] (+ r[m])"""
    _need_cuda(x, scale, r, out)
    m, c = x.shape
    assert x.is_contiguous() and scale.is_contiguous() and scale.numel() * rows_per_scale == m and (r is None or r.is_contiguous())
    y = torch.empty_like(x) if out is None else out
    check(lib.mt4_rowscale_add_f32(x.data_ptr(), scale.data_ptr(), r.data_ptr() if r is not None else None, y.data_ptr(), m, c, rows_per_scale, _stream()),
          "mt4_rowscale_add_f32")
    return y


def groupwise_linear_bwd(dy: torch.Tensor, hs: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, db: torch.Tensor) -> torch.Tensor:
    """dy [B,K]; hs [B*K, D]; w [K,D]; dw / db ADDED to; returns dhs [B*K, D]"""
    _need_cuda(dy, hs, w, dw, db)
    b, k = dy.shape
    d = w.shape[1]
    assert dy.is_contiguous() and hs.is_contiguous() and w.is_contiguous() and hs.numel() == b * k * d
    dhs = torch.empty_like(hs)
    check(lib.mt4_groupwise_linear_bwd_f32(dy.data_ptr(), hs.data_ptr(), w.data_ptr(), dhs.data_ptr(), dw.data_ptr(), db.data_ptr(), b, k, d, _stream()),
          "mt4_groupwise_linear_bwd_f32")
    return dhs


def sum_over_batch(x: torch.Tensor, out: torch.Tensor, batch: int, accumulate: bool = False) -> torch.Tensor:
    """out [L, C] (+)= sum_b x[b*L:(b+1)*L]"""
    _need_cuda(x, out)
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == batch * out.numel()
    check(lib.mt4_sum_over_batch_f32(x.data_ptr(), out.data_ptr(), batch, out.numel(), 1 if accumulate else 0, _stream()), "mt4_sum_over_batch_f32")
    return out
