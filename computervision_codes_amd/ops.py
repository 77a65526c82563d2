"""Thin torch-tensor wrappers over the C-ABI (device memory and streams are torch's: plumbing only)."""
from __future__ import annotations

import ctypes as C
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


def conv_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, kh: int, kw: int,
              stride: Tuple[int, int] = (1, 1), pad: Tuple[int, int] = (0, 0), dil: Tuple[int, int] = (1, 1),
              residual: Optional[torch.Tensor] = None, relu: bool = False, out_dtype: Optional[torch.dtype] = None,
              out: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """y = act(conv(x, w) + bias [+ residual]); x [B,H,W,Cin] contiguous channels-last storage."""
    _need_cuda(x, w_packed, bias, residual)
    assert x.dim() == 4 and x.is_contiguous()
    b, h, w_, cin = x.shape
    cout = w_packed.shape[0]
    ho = conv_out_size(h, kh, stride[0], pad[0], dil[0])
    wo = conv_out_size(w_, kw, stride[1], pad[1], dil[1])
    od = out_dtype or x.dtype
    if out is None:
        out = torch.empty((b, ho, wo, cout), dtype=od, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == b * ho * wo * cout and out.dtype == od
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == x.dtype and residual.numel() == out.numel()
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout
    assert w_packed.dtype == x.dtype and w_packed.shape[1] == packed_k(cin, kh, kw, x.dtype)
    d = ConvDesc(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                 residual.data_ptr() if residual is not None else None, out.data_ptr(),
                 b, h, w_, cin, ho, wo, cout, kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1],
                 1 if relu else 0, dt_code(x.dtype), dt_code(od), tile)
    check(lib.mt4_conv_nhwc(C.byref(d), _stream()), "mt4_conv_nhwc")
    return out


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


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.is_contiguous()
    b, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, ho, wo, c), dtype=x.dtype, device=x.device)
    check(lib.mt4_maxpool3x3s2_nhwc(x.data_ptr(), y.data_ptr(), b, h, w, c, dt_code(x.dtype), _stream()), "mt4_maxpool3x3s2_nhwc")
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
