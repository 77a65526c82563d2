/* mt4hip.h -- C-ABI of libmt4hip.so: the MI355X (gfx950) kernels behind the MT4MTL-KD hot path.
 *
 * The reference (CIAM-Group/ComputerVision_Codes, MT4MTLKD/) has no FFI of its own: its hot path is
 * `nn.Module.forward` calling stock torch ops.  Each entry point below replaces the
 * torch op sequence of the reference lines it cites.  All functions
 *   - take raw DEVICE pointers, sizes and a hipStream_t (passed as void*; NULL = default stream),
 *   - only enqueue work on that stream (no allocation, no synchronisation: graph-capturable),
 *   - return 0 on success or a negative MT4_E* code, never throw,
 *   - keep no global state.
 * Activations are channels-last: images NHWC, frame sequences [T][C] (= the on-disk frame-feature
 * layout `float32 [N_frames, D]`, Spatial_cnn/test.py:266-284).  Weights are pre-packed by
 * mt4_pack_* helpers or by the host (documented per function).
 */
#ifndef MT4HIP_H
#define MT4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MT4_OK 0
#define MT4_EINVAL (-1)   /* bad argument (null pointer, non-positive size)            */
#define MT4_EALIGN (-2)   /* pointer / channel count violates the alignment contract   */
#define MT4_ELAUNCH (-3)  /* hipLaunch failed; see mt4_last_hip_error()                */
#define MT4_EUNSUPPORTED (-4)

#define MT4_F32 0
#define MT4_BF16 1

int mt4_abi_version(void);
const char* mt4_strerror(int code);
/* last hipError_t (as int) seen by this thread inside the library, 0 if none */
int mt4_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, channels-last, fused epilogue:
 *     y = act( conv(x, w) + bias [+ residual] )
 * replaces Conv2d+BatchNorm2d(eval)+ReLU[+add] (Spatial_transformer/models/resnet.py:101-121 as
 * reached from Spatial_cnn/network.py:105,117) and Conv1d(+ReLU)(+x) (Temporal_tenco/network.py:
 * 186-198; 1-D = H=1, KH=1).  BatchNorm is folded into w/bias by the host.
 *
 *   x        [B][H][W][Cin]            dtype
 *   w        [Cout][Kpad]              dtype; row = KH*KW taps x CPT chunks of 16 bytes, tap-major,
 *                                      channel-minor, zero padded; CPT = ceil(Cin*esize/16);
 *                                      Kpad (elements) = round_up(KH*KW*CPT, 8) * (16/esize)
 *   bias     [Cout] float32 or NULL
 *   residual [B][Ho][Wo][Cout] dtype or NULL (added before the activation)
 *   y        [B][Ho][Wo][Cout]         out_dtype (MT4_F32 allowed with bf16 inputs)
 * Contract: Cin*esize % 16 == 0; x, w, y, residual 16-byte aligned.
 */
typedef struct mt4_conv_desc {
    const void* x;
    const void* w;
    const float* bias;
    const void* residual;
    void* y;
    int32_t B, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t KH, KW;
    int32_t stride_h, stride_w;
    int32_t pad_h, pad_w;
    int32_t dil_h, dil_w;
    int32_t relu;       /* 0 / 1 */
    int32_t dtype;      /* MT4_F32 / MT4_BF16 : x, w, residual */
    int32_t out_dtype;  /* MT4_F32 / MT4_BF16 : y */
    int32_t tile;       /* 0 = auto; else a tile id from mt4_conv_tile_count() (for tuning/tests) */
} mt4_conv_desc;

int mt4_conv_nhwc(const mt4_conv_desc* d, void* stream);
int mt4_conv_tile_count(void);
/* elements per packed weight row for a given geometry (Kpad above) */
int64_t mt4_conv_packed_k(int32_t Cin, int32_t KH, int32_t KW, int32_t dtype);

/* Pack OIHW float32 weights (optionally scaled per output channel: the folded BN factor) into the
 * [Cout][Kpad] layout above.  w_oihw [Cout][Cin][KH][KW] float32 device; scale [Cout] float32 or NULL. */
int mt4_pack_conv_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t Cin,
                         int32_t KH, int32_t KW, int32_t dtype, void* stream);

/* ResNet stem 7x7/2 pad 3, Cin=3 (resnet.py:145): the frame is stored padded as
 * [B][H+7][W+8][4] (3 rows/cols of zeros before, 4 rows / 5 cols after; 4th channel zero) so that a
 * kernel row is one contiguous 32-element run; the stem then runs through mt4_conv_nhwc on the
 * view [B][H+7][(W+8)/2][8] with KH=7, KW=4, stride (2,1), pad 0.  This packs its weights:
 * w_oihw [64][3][7][7] -> [Cout][Kpad(8,7,4)] with taps (kh, kw/2), channel slot (kw%2)*4 + c. */
int mt4_pack_stem_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t dtype,
                         void* stream);

/* uint8 frames [B][H][W][3] -> normalised, zero-padded [B][H+7][W+8][4] of dtype
 * ((v/255 - mean[c]) / std[c]; ToTensor+Normalize of Spatial_cnn/dataloader.py:153-162). W % 2 == 0. */
int mt4_preprocess_u8(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, const float mean[3],
                      const float std[3], int32_t dtype, void* stream);
/* same from already-normalised float32 NCHW [B][3][H][W] (the reference's module-call boundary) */
int mt4_pad_nchw_f32(const float* x, void* out, int32_t B, int32_t H, int32_t W, int32_t dtype, void* stream);

/* MaxPool2d(3, stride 2, pad 1) channels-last (resnet.py:149). C*esize % 16 == 0. */
int mt4_maxpool3x3s2_nhwc(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                          void* stream);
/* AdaptiveAvgPool2d(1) channels-last -> float32 [B][C] (resnet.py:157; the hooked `final_feature`). */
int mt4_global_avgpool_nhwc(const void* x, float* y, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* stream);

/* y[B][N] = x[B][K] @ w[N][K]^T + bias  (float32; Classifier.fc, Spatial_cnn/network.py:121-129,
 * the four heads concatenated along N). */
int mt4_linear_f32(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t K, int32_t N,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MT4HIP_H */
