"""ctypes binding of libmt4hip.so (C-ABI declared in include/mt4hip.h).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing the product
fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C computervision_codes_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

# torch owns device memory and streams and ships its own libamdhip64; it must be loaded FIRST so that
# libmt4hip.so binds to the same HIP runtime instance (a second runtime cannot see torch's allocations).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmt4hip.so")

MT4_F32 = 0
MT4_BF16 = 1


class ConvDesc(C.Structure):
    """mirror of `mt4_conv_desc` (include/mt4hip.h)"""
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
        ("out_row_map", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32),
        ("stride_h", C.c_int32), ("stride_w", C.c_int32),
        ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("dil_h", C.c_int32), ("dil_w", C.c_int32),
        ("relu", C.c_int32), ("dtype", C.c_int32), ("out_dtype", C.c_int32), ("tile", C.c_int32),
        ("out_row_map_len", C.c_int32), ("y_ld", C.c_int32), ("res_ld", C.c_int32), ("out_rows_per_image", C.c_int32),
        ("x_pixel_stride", C.c_int32), ("fuse_cout", C.c_int32),
        ("fuse_w", C.c_void_p), ("fuse_bias", C.c_void_p), ("fuse_y", C.c_void_p), ("fuse_relu", C.c_int32), ("residual_float", C.c_int32),
        ("x2", C.c_void_p), ("x2_H", C.c_int32), ("x2_W", C.c_int32), ("x2_C", C.c_int32), ("x2_stride", C.c_int32), ("fuse_expand", C.c_int32),
        ("stat_sums", C.c_void_p),
    ]


class RefreshEntry(C.Structure):
    """mt4_refresh_entry (include/mt4hip.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("block0", C.c_int64), ("dst_bf16", C.c_int32), ("transposed", C.c_int32), ("cout", C.c_int32),
                ("cin", C.c_int32), ("ntaps_dst", C.c_int32), ("tapw_src", C.c_int32), ("kpad_src", C.c_int32), ("tapw_dst", C.c_int32),
                ("kpad_dst", C.c_int32), ("tap_map", C.c_int32 * 9)]


class TcnDesc(C.Structure):
    """mirror of `mt4_tcn_desc` (include/mt4hip.h)"""
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
        ("B", C.c_int32), ("T", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("taps", C.c_int32), ("dilation", C.c_int32), ("relu", C.c_int32), ("dtype", C.c_int32), ("out_dtype", C.c_int32),
    ]


_vp, _i32 = C.c_void_p, C.c_int32
_FLOAT3 = C.c_float * 3

# name -> (restype, argtypes); must list every symbol of include/mt4hip.h
SIGNATURES = {
    "mt4_abi_version": (C.c_int, []),
    "mt4_source_digest": (C.c_char_p, []),
    "mt4_strerror": (C.c_char_p, [C.c_int]),
    "mt4_last_hip_error": (C.c_int, []),
    "mt4_conv_nhwc": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "mt4_conv_tile_count": (C.c_int, []),
    "mt4_conv_packed_k": (C.c_int64, [_i32, _i32, _i32, _i32]),
    "mt4_pack_conv_weight": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_pack_stem_weight": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "mt4_preprocess_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _FLOAT3, _FLOAT3, _i32, _vp]),
    "mt4_preprocess_u8_s2d": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _FLOAT3, _FLOAT3, _vp]),
    "mt4_pad_nchw_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_png_inflate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_int64, C.c_int64, _vp, _vp]),
    "mt4_png_unfilter_rgb8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.c_int64, _vp, _vp]),
    "mt4_copy_spans_u8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mt4_png_stat_files": (C.c_int, [C.POINTER(C.c_char_p), _i32, _vp, _i32]),
    "mt4_png_read_files": (C.c_int, [C.POINTER(C.c_char_p), _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32]),
    "mt4_resize_pass_u8": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_maxpool3x3s2_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_stem_maxpool_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_global_avgpool_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_linear_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, C.c_float, _i32, _vp]),
    "mt4_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                C.c_float, _i32, _vp]),
    "mt4_window_attention_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "mt4_window_attention_rel_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "mt4_patchify": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _FLOAT3, _FLOAT3, _i32, _vp]),
    "mt4_add_rowbcast": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _vp]),
    "mt4_groupwise_linear": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_dwconv1d_k3": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_kd_mix": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mt4_wgrad_conv1d_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mt4_colsum_f32": (C.c_int, [_vp, _vp, C.c_int64, _i32, _i32, _i32, _vp]),
    "mt4_bce_logits_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _vp]),
    "mt4_sgd_step_f32": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_float, C.c_float, _vp]),
    "mt4_mul_add_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "mt4_transpose_pack_conv1d_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_bn_stats_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, C.c_float, C.c_float, _vp]),
    "mt4_bn_apply_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _vp]),
    "mt4_bn_backward_f32": (C.c_int, [_vp] * 12 + [C.c_int64, _i32, _i32, _vp]),
    "mt4_wgrad_conv2d_f32": (C.c_int, [_vp, _vp, _vp] + [_i32] * 15 + [_vp]),
    "mt4_maxpool3x3s2_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_avgpool_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_bce_logits_pw_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_distill_kl_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _i32, _vp]),
    "mt4_mse_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_float, _vp]),
    "mt4_bn_stats_t": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, C.c_float, C.c_float, _vp]),
    "mt4_avgpool1d_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_interp_linear_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_avgpool1d_rows_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_interp_linear_rows_bwd_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_bn_apply_sums_f32": (C.c_int, [_vp] * 10 + [C.c_int64, _i32, C.c_float, C.c_float, _i32, _vp]),
    "mt4_bn_apply_sums_t": (C.c_int, [_vp, _i32] + [_vp] * 9 + [C.c_int64, _i32, C.c_float, C.c_float, _i32, _vp]),
    "mt4_bn_apply_t": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _vp]),
    "mt4_bn_backward_t": (C.c_int, [_vp, _vp, _vp, _i32] + [_vp] * 9 + [C.c_int64, _i32, _i32, _vp]),
    "mt4_wgrad_conv2d_bf16": (C.c_int, [_vp, _vp, _vp] + [_i32] * 9 + [_vp]),
    "mt4_maxpool3x3s2_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_avgpool_bwd_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_cast_f32_bf16": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "mt4_refresh_weights": (C.c_int, [_vp, _i32, C.c_int64, _vp]),
    "mt4_repack_weight_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_kd_mix_bwd_f32": (C.c_int, [_vp] * 9 + [_i32, _i32, _vp]),
    "mt4_tcn_conv": (C.c_int, [C.POINTER(TcnDesc), _vp]),
    "mt4_tcn_dilated_residual_layer": (C.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "mt4_tcn_stage": (C.c_int, [_vp] * 5 + [C.POINTER(_vp)] * 4 + [_i32] * 5 + [_vp]),
    "mt4_bottleneck_packed_bytes": (C.c_int64, [_i32, _i32]),
    "mt4_bottleneck_pack_bf16": (C.c_int, [_vp] * 4 + [_i32, _vp, _vp]),
    "mt4_bottleneck_fused_bf16": (C.c_int, [_vp] * 7 + [_i32] * 5 + [_vp]),
    "mt4_pack_fragments_bf16": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "mt4_chain_gemm_bf16": (C.c_int, [_vp, C.c_int64, C.c_int64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "mt4_bottleneck_next_packed_bytes": (C.c_int64, []),
    "mt4_bottleneck_pack_next_bf16": (C.c_int, [_vp, _vp, _vp]),
    "mt4_bottleneck_fused_next_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_fpn_topdown": (C.c_int, [_vp, _vp, _i32, C.c_int64, _i32, _vp]),
    "mt4_tcn_layer_fused_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mt4_tcn_linear_ln_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp, _vp]),
    "mt4_tcn_linear_stats_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mt4_bgemm_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.c_float, _i32, _vp]),
    "mt4_softmax_rows_f32": (C.c_int, [_vp, C.c_int64, _i32, C.c_float, _vp]),
    "mt4_softmax_bwd_rows_f32": (C.c_int, [_vp, _vp, C.c_int64, _i32, C.c_float, _vp]),
    "mt4_layernorm_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i32, C.c_float, _i32, _vp]),
    "mt4_gelu_f32": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "mt4_gelu_bwd_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "mt4_dwconv1d_k3_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mt4_gather_rows_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mt4_add_bias_mask_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _i32, _vp]),
    "mt4_relpos_table_grad_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i32, _i32, _vp]),
    "mt4_rowscale_add_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _i32, _i32, _vp]),
    "mt4_groupwise_linear_bwd_f32": (C.c_int, [_vp] * 6 + [_i32, _i32, _i32, _vp]),
    "mt4_sum_over_batch_f32": (C.c_int, [_vp, _vp, _i32, C.c_int64, _i32, _vp]),
    "mt4_dropout_mask_f32": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_float, _vp]),
    "mt4_axpby_f32": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_float, _vp]),
}


class Mt4Error(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP library is not built. "
            "Run `make -C computervision_codes_amd/csrc` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    return lib


ABI_VERSION = 10        # what mt4_abi_version() of a matching libmt4hip.so returns (csrc/misc_kernels.hip)
lib = _load()
if lib.mt4_abi_version() != ABI_VERSION:      # a stale libmt4hip.so next to newer Python: fail at import, not in the first launch
    raise ImportError(f"libmt4hip.so reports ABI {lib.mt4_abi_version()}, this package binds ABI {ABI_VERSION}: rebuild (make -C computervision_codes_amd/csrc)")




def _check_sources():
    """the loaded binary must be built from the sources beside it: every test, every committed PMC figure and `bench.py` speak about THOSE"""
    from .srcdigest import library_digest
    lib.mt4_source_digest.restype = C.c_char_p
    lib.mt4_source_digest.argtypes = []
    baked, now = lib.mt4_source_digest().decode(), library_digest()
    if baked != now and os.environ.get("MT4_ALLOW_STALE") != "1":
        raise ImportError(f"libmt4hip.so was built from sources {baked}, the sources in {os.path.join(_HERE, 'csrc')} are {now}: rebuild "
                          "(make -C computervision_codes_amd/csrc) or set MT4_ALLOW_STALE=1 to load the stale library anyway")
    return baked


SOURCE_DIGEST = _check_sources()
MT4_EUNSUPPORTED = -4   # include/mt4hip.h


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = lib.mt4_strerror(code).decode()
        raise Mt4Error(f"{what or 'mt4 call'} failed: {msg} (code {code}, hip error {lib.mt4_last_hip_error()})")

