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

/* 10 in this revision (5 -> 6: mt4_stem_maxpool_bf16, mt4_bottleneck_fused_next_bf16, mt4_pack_fragments_bf16, the x2 / fuse_expand fields at the
 * end of mt4_conv_desc; 6 -> 7: mt4_copy_spans_u8; 7 -> 8: mt4_chain_gemm_bf16; 8 -> 9: stat_sums at the end of
 * mt4_conv_desc, mt4_bn_apply_sums_t / _f32, mt4_avgpool1d_rows, mt4_interp_linear_rows, `beta` / relu code 2 of mt4_bn_backward_f32, `bias_grad` of mt4_wgrad_conv1d_f32, MT4_REFRESH_TILES_PER_BLOCK tiles per workgroup in mt4_refresh_weights; 9 -> 10: mt4_source_digest, mt4_attention takes head dims up to 512, mt4_png_stat_files / mt4_png_read_files, mt4_avgpool1d_rows_bwd_f32 / mt4_interp_linear_rows_bwd_f32, mt4_tcn_layer_fused_bf16, mt4_tcn_linear_ln_f32, mt4_tcn_linear_stats_f32).  A binding checks it once at load
 * (computervision_codes_amd/_lib.py: ABI_VERSION). */
int mt4_abi_version(void);
/* 16 hex digits: sha256 over every source file the library was built from (every .hip / .h file of csrc/ and this header), baked in at build time.  No
 * reference counterpart (the reference has no native code): it ties a loaded binary to the sources in the tree -- `_lib.py` compares it with
 * `srcdigest.library_digest()` at load and refuses a stale library. */
const char* mt4_source_digest(void);
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
#define MT4_STAT_REPLICAS 8
typedef struct mt4_conv_desc {
    const void* x;
    const void* w;
    const float* bias;
    const void* residual;
    void* y;
    const int32_t* out_row_map; /* optional [out_row_map_len]: result (and residual) row of pixel m is
                                   (m / len) * len + map[m % len] -- Swin window-reverse + roll-back folded into
                                   the projection's epilogue (swin_transformer.py:255-265); NULL = identity */
    int32_t B, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t KH, KW;
    int32_t stride_h, stride_w;
    int32_t pad_h, pad_w;
    int32_t dil_h, dil_w;
    int32_t relu;       /* activation: 0 none, 1 ReLU, 2 GELU (erf form, nn.GELU default), 3 ReLU-backward gate: `residual` is
                           NOT added but gates the result (y = residual > 0 ? conv : 0; the saved forward activation) */
    int32_t dtype;      /* MT4_F32 / MT4_BF16 : x, w, residual */
    int32_t out_dtype;  /* MT4_F32 / MT4_BF16 : y */
    int32_t tile;       /* 0 = auto; -1 = auto preferring latency: may take the K-split tiles 35-38 (groups of waves share a tile's K loop;
                           the K summation order then differs from the other tiles' by fp32 reassociation -- the temporal heads use it);
                           else a tile id 1..mt4_conv_tile_count() (for tuning/tests).  Ids 21-32 (3x3 patch kernel) and 33-34
                           (space-to-depth stem kernels) cover one geometry each: MT4_EUNSUPPORTED for any other */
    int32_t out_row_map_len;
    int32_t y_ld;       /* row pitch of y in elements (0 = Cout): lets a GEMM write a column slice of a wider buffer
                           (Temporal_Mixer's channel concat, TS_Mixer.py:83) */
    int32_t res_ld;     /* row pitch of residual in elements (0 = Cout) */
    int32_t out_rows_per_image; /* with out_row_map: rows per image on the OUTPUT side (0 = out_row_map_len); > len scatters
                                   the result into a larger image (sub-pixel phases of a strided conv's data gradient) */
    int32_t x_pixel_stride;     /* elements between neighbouring pixels of x (0 = Cin).  < Cin: the Cin elements of a tap are a
                                   contiguous run over several pixels of a narrower image (KW = 1, no padding; the caller pads x so
                                   that every run stays inside it): the ResNet stem on the space-to-depth frame, 4 x 16 channels */
    int32_t fuse_cout;          /* with fuse_expand: output channels of the conv3 run behind this launch's own conv (a multiple of 128) */
    /* fuse_w / fuse_bias / fuse_y / fuse_relu: operands of `fuse_expand` (below).  Round 2's other use -- the next Bottleneck's conv1 in the
       epilogue of a 256-channel conv3 (fuse_w without fuse_expand) -- measured slower than the two launches and is gone: that request returns
       MT4_EUNSUPPORTED; two dependent 1x1 convs in one launch are mt4_chain_gemm_bf16. */
    const void* fuse_w;         /* conv3's weights in fragment order (mt4_pack_fragments_bf16) or NULL */
    const float* fuse_bias;     /* [fuse_cout] or NULL */
    void* fuse_y;               /* [B][Ho][Wo][fuse_cout] bf16 */
    int32_t fuse_relu;          /* 0 / 1 */
    int32_t residual_float;       /* 1: dtype MT4_BF16 with out_dtype MT4_F32 only -- the residual (or the ReLU gate of act 3) is float32 like y: the mixed-precision
                                   training GEMMs (bf16 operands, fp32 activations); 0 otherwise */
    /* Optional second K source: the downsample branch of a strided Bottleneck (resnet.py:116-119: identity = bn(conv1x1 stride s (x_block)))
       accumulated in the SAME launch as conv3 + bn3, instead of a launch of its own whose map is written and read back as `residual`:
           y[b][ho][wo] = act( [w | w2] . [ x[b][ho][wo] ; x2[b][s * ho][s * wo] ] + bias ),   bias = b3 + b_downsample
       `w` rows hold both K ranges back to back ([Cout][Cin + x2_C], each a whole number of 128-byte K-steps).  This conv must be a 1x1 /
       stride 1 / pad 0 bf16 -> bf16 launch without residual or out_row_map, tile 0; MT4_EUNSUPPORTED otherwise.  One fp32 accumulator chain:
       the branch is NOT rounded to bf16 before the add (the two-launch form rounds it), so results differ from that form by bf16 rounding of
       the identity, towards the reference's fp32 arithmetic. */
    const void* x2;             /* [B][x2_H][x2_W][x2_C] bf16 or NULL */
    int32_t x2_H, x2_W, x2_C;
    int32_t x2_stride;          /* s */
    int32_t fuse_expand;        /* 1 with fuse_w on a 3x3 / stride 1 / pad 1 bf16 launch with Cin == Cout == 128 (layer2's conv2): the fused following
                                   conv is the Bottleneck's conv3 + bn3 + add + ReLU (resnet.py:112-119).  This launch's own output stays in LDS
                                   (y is not written), `residual` ([B][H][W][fuse_cout], the block input) is added to y2 = fuse_y before fuse_relu,
                                   fuse_cout is a multiple of 128 and fuse_w is in fragment order (mt4_pack_fragments_bf16 of the packed
                                   [fuse_cout][128] matrix).  Runs where the 3x3 patch kernel runs (>= 256 tiles of 256 x 256): MT4_EUNSUPPORTED
                                   otherwise -- launch the two convs then; the results are bit-identical either way */
    /* Statistics of the train-mode BatchNorm that follows this convolution (Spatial_cnn/run.py:145-224 trains resnet.py's conv -> bn pairs),
       taken in the epilogue instead of a pass over y: the launch ADDS, per output channel c,
           stat_sums[r][0][c] += sum of y[m][c],   stat_sums[r][1][c] += sum of y[m][c]^2      (float64; of the STORED, i.e. rounded, values)
       over its pixels m, spread over the MT4_STAT_REPLICAS copies r (by pixel tile) so that the float64 atomics behind it do not queue on
       one address; the statistics are the sums over r (mt4_bn_apply_sums_t).  The caller zeroes the buffer.  Plain launches only (no fuse_*,
       x2, out_row_map, x_pixel_stride, tile -1 / 33 / K-split ids; Cout % 8 == 0, % 4 for fp32 output): MT4_EUNSUPPORTED otherwise.  NULL = off. */
    double* stat_sums;          /* [MT4_STAT_REPLICAS][2][Cout] float64 */
} mt4_conv_desc;

int mt4_conv_nhwc(const mt4_conv_desc* d, void* stream);
int mt4_conv_tile_count(void);

/* One stride-1 ResNet Bottleneck of 64 mid channels (layer1 of ResNet-50) in ONE launch: conv1 1x1 + bn1 + ReLU, conv2 3x3 + bn2 + ReLU,
 * conv3 1x1 + bn3 + residual + ReLU (Spatial_transformer/models/resnet.py:101-121; what Spatial_cnn/network.py:25-31 runs through
 * torchvision's Bottleneck) with the 64-channel intermediates held in LDS: x is read from HBM once (plus the 1-pixel halo ring of each
 * 8 x 14 pixel tile) and y written once, instead of 8.5 GB per 1336 frames through three mt4_conv_nhwc launches (+ the downsample launch).
 *   x [B][H][W][Cin] bf16, y [B][H][W][256] bf16, biases float32 (bds != NULL selects the downsample block).
 *   Cin == 256: identity residual (layer1.1, layer1.2); Cin == 64 with bds: residual = bf16(wds . x + bds) (layer1.0).
 *   w_frag: the block's weights in fragment order, made once per model load by mt4_bottleneck_pack_bf16 from the matrices packed by
 *   mt4_pack_conv_weight with the BatchNorm scale folded in (w1 [64][Cin], w2 [64][9*64], w3 [256][64], wds [256][Cin] or NULL):
 *   [channel tile of 16][K step of 32][lane] x 16 bytes, so that a wave's load of one MFMA fragment is 1 KB contiguous;
 *   mt4_bottleneck_packed_bytes gives the buffer size.
 * Bit-identical to the launch sequence through mt4_conv_nhwc (same K order, bias-initialised fp32 accumulators, bf16 rounding of the
 * intermediates where that path stores them).  mid != 64 or another Cin: MT4_EUNSUPPORTED.  H*W*512 < 2 GiB. */
int64_t mt4_bottleneck_packed_bytes(int32_t Cin, int32_t has_downsample);
int mt4_bottleneck_pack_bf16(const void* w1, const void* w2, const void* w3, const void* wds, int32_t Cin, void* out, void* stream);
int mt4_bottleneck_fused_bf16(const void* x, void* y, const void* w_frag, const float* b1, const float* b2, const float* b3, const float* bds,
                              int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t mid, void* stream);
/* The same launch for an identity block (Cin == 256) that is FOLLOWED by a strided Bottleneck (layer1.2 -> layer2.0): also runs that block's
 * conv1 + bn1 + ReLU (256 -> 128 channels; resnet.py:101-103 of the next block) on the result while the tile is in LDS.
 *   w_next: the next conv1's packed bf16 matrix [128][256] in fragment order (mt4_bottleneck_pack_next_bf16, mt4_bottleneck_next_packed_bytes),
 *   b_next [128] float32.  t_next [B][H][W][128] bf16 = relu(w_next . y + b_next).
 *   y_even [B][(H+1)/2][(W+1)/2][256] bf16 = the block's output y at the even pixels: all the stride-2 downsample branch (its one other
 *   reader) takes -- pass it as mt4_conv_desc.x2 with x2_stride 1.
 * Both outputs are bit-identical to mt4_bottleneck_fused_bf16 followed by mt4_conv_nhwc. */
int64_t mt4_bottleneck_next_packed_bytes(void);
int mt4_bottleneck_pack_next_bf16(const void* w1_next, void* out, void* stream);
int mt4_bottleneck_fused_next_bf16(const void* x, void* y_even, void* t_next, const void* w_frag, const float* b1, const float* b2, const float* b3,
                                   const void* w_next, const float* b_next, int32_t B, int32_t H, int32_t W, void* stream);
/* a packed bf16 matrix [rows][row_elems] in MFMA fragment order: [rows / 16][row_elems / 32][lane (r16, q)] x 16 bytes =
 * w[tile * 16 + r16][step * 32 + 8 q .. + 8), so that a wave's load of one fragment is 1 KB contiguous (rows % 16 == 0, row_elems % 32 == 0) */
int mt4_pack_fragments_bf16(const void* w_packed, int32_t rows, int32_t row_elems, void* out, void* stream);
/* Two dependent 1x1 convolutions / linear layers in one launch, the map between them never read back (bf16 operands, fp32 accumulate, K-chunk
 * accumulation: the N1 channels are produced 128 at a time and each chunk is at once the next GEMM's K range):
 *   H  = act1(x . w1^T + b1 [+ r1])   [M][N1]   (stored to y1 when given)
 *   y2 = act2(H . w2^T + b2 [+ r2])   [M][N2]
 * Bottleneck form -- conv3 + bn3 + add + ReLU of one block and conv1 + bn1 + ReLU of the next (`Spatial_transformer/models/resnet.py:101-121`):
 *   r1 = the block input, y1 = the block output (the next block's residual), act1 = act2 = 1 (ReLU), r2 = NULL.
 * MLP form -- Swin's Mlp + shortcut (`swin_transformer.py:15-31,267-269`): r1 = y1 = NULL, act1 = 2 (GELU, erf), r2 = the shortcut, act2 = 0.
 * x [M][x_ld] bf16 (row pitch x_ld >= K1 elements), w1_frag / w2_frag = mt4_pack_fragments_bf16 of the packed [N1][K1] / [N2][N1] matrices,
 * b1 / b2 fp32.  (K1, N1, N2) = (256, 1024, 256) or (128, 512, 128), Bottleneck form also (128, 512, 256); MT4_EUNSUPPORTED otherwise.
 * Both results are bit-identical to the two stand-alone mt4_conv_nhwc launches. */
int mt4_chain_gemm_bf16(const void* x, int64_t x_ld, int64_t M, int32_t K1, const void* w1_frag, const float* b1, int32_t N1, const void* r1, void* y1,
                        int32_t act1, const void* w2_frag, const float* b2, int32_t N2, const void* r2, int32_t act2, void* y2, void* stream);
/* elements per packed weight row for a given geometry (Kpad above) */
int64_t mt4_conv_packed_k(int32_t Cin, int32_t KH, int32_t KW, int32_t dtype);

/* Pack OIHW float32 weights (optionally scaled per output channel: the folded BN factor) into the
 * [Cout][Kpad] layout above.  w_oihw [Cout][Cin][KH][KW] float32 device; scale [Cout] float32 or NULL. */
int mt4_pack_conv_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t Cin,
                         int32_t KH, int32_t KW, int32_t dtype, void* stream);

/* ResNet stem 7x7/2 pad 3, Cin=3 (resnet.py:145): the frame is stored padded as
 * [B][H+6][Wp][4], Wp = round_up(W+6, 2) (3 rows/cols of zeros around, 4th channel zero) so that a
 * kernel row is one contiguous 32-element run; the stem then runs through mt4_conv_nhwc on the
 * pixel-pair view [B][H+6][Wp/2][8] with KH=7, KW=4, stride (2,1), pad 0.  This packs its weights:
 * w_oihw [64][3][7][7] -> [Cout][Kpad(8,7,4)] with taps (kh, kw/2), channel slot (kw%2)*4 + c. */
int mt4_pack_stem_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t dtype,
                         void* stream);

/* uint8 frames [B][H][W][3] -> normalised, zero-padded [B][H+6][Wp][4] of dtype
 * ((v/255 - mean[c]) / std[c]; ToTensor+Normalize of Spatial_cnn/dataloader.py:153-162). */
int mt4_preprocess_u8(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, const float mean[3],
                      const float std[3], int32_t dtype, void* stream);
/* same from already-normalised float32 NCHW [B][3][H][W] (the reference's module-call boundary) */
int mt4_pad_nchw_f32(const float* x, void* out, int32_t B, int32_t H, int32_t W, int32_t dtype, void* stream);
/* The same frames as a normalised bf16 space-to-depth image of the 3-padded frame, [B][(H+6)/2][(W+6)/2][16] (H, W even): channel
 * (dy*2+dx)*3 + c of pixel (y, x) = frame pixel (2y+dy-3, 2x+dx-3), channels 12..15 zero.  The 7x7/2 stem (resnet.py:145) becomes a
 * 4x1 kernel over contiguous runs of 4 pixels (mt4_conv_desc.x_pixel_stride = 16, Cin = 64) and runs through the LDS-DMA path. */
int mt4_preprocess_u8_s2d(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, const float mean[3], const float std[3],
                          void* stream);

/* PNG decode on the device (the reference decodes with PIL in its DataLoader workers, Spatial_cnn/dataloader.py:257-261).
 * The host only walks the chunk list (IHDR, concatenated IDAT payload); 8-bit RGB, non-interlaced files.
 *   mt4_png_inflate: B DEFLATE streams (zlib payloads, the 2-byte zlib header stripped), stream i = streams[offsets[i] .. + lengths[i])
 *     -> the filtered scanlines at raw + i * raw_stride, exactly raw_len = H * (1 + 3 W) bytes; one thread per frame.
 *     status[i] (device int32) = 0 or: 1 truncated input, 2 bad block type, 3 stored-length mismatch, 4 bad code lengths,
 *     5 invalid symbol / distance, 6 output overflow, 7 output shorter than raw_len.
 *   mt4_png_unfilter_rgb8: PNG filters 0-4 undone (bytes per pixel 3) -> uint8 [B][H][W][3]; status[i] = 8 for an unknown filter type.
 * Both only enqueue; read status after the stream has drained. */
int mt4_png_inflate(const uint8_t* streams, const int64_t* offsets, const int32_t* lengths, uint8_t* raw, int32_t B, int64_t raw_stride,
                    int64_t raw_len, int32_t* status, void* stream);
int mt4_png_unfilter_rgb8(const uint8_t* raw, uint8_t* out, int32_t B, int32_t H, int32_t W, int64_t raw_stride, int32_t* status, void* stream);
/* n byte spans src[src_off[i] .. + len[i]) -> dst[dst_off[i] ..) (device pointers and device offset / length arrays): packs the IDAT payloads
 * of PNG files uploaded whole into the contiguous streams mt4_png_inflate reads (pngdec.decode_files; the reference's PIL reads the chunks on
 * the host, Spatial_cnn/dataloader.py:257-261). */
int mt4_copy_spans_u8(const uint8_t* src, uint8_t* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* len, int32_t n, void* stream);
/* HOST side of the same path, no device work (csrc/png_host.hip): what the reference's three DataLoader worker processes do around PIL's decoder
 * (open / read / chunk walk, Spatial_cnn/dataloader.py:257-261, Spatial_cnn/test.py:240-241) on `threads` native threads, outside the interpreter
 * lock.  mt4_png_stat_files: sizes[i] = bytes of paths[i] (-1: cannot stat).  mt4_png_read_files: file i is read to dst + file_off[i] (sizes[i]
 * bytes; dst is normally page-locked staging memory that is then uploaded whole) and its chunk list walked: status[i] = 0 ok, 1 cannot open /
 * short read, 2 not a PNG / truncated chunk, 3 not 8-bit RGB non-interlaced, 4 no IHDR / IDAT, 5 more than max_spans IDAT chunks, 6 bad zlib
 * header; width / height [n]; nspans[i] and span_off / span_len [n][max_spans] = offset INSIDE the file and length of every non-empty IDAT payload
 * (the 2-byte zlib header is still part of the first one).  Both return MT4_OK when the call ran (per-file results in sizes / status). */
int mt4_png_stat_files(const char* const* paths, int32_t n, int64_t* sizes, int32_t threads);
int mt4_png_read_files(const char* const* paths, const int64_t* sizes, const int64_t* file_off, int32_t n, uint8_t* dst, int32_t* width,
                       int32_t* height, int64_t* span_off, int32_t* span_len, int32_t* nspans, int32_t max_spans, int32_t* status, int32_t threads);

/* One separable pass of Pillow's 8-bit resize (`Image.resize(size, BILINEAR)` = `transforms.Resize((256,448))`,
 * Spatial_cnn/dataloader.py:155-159, Spatial_transformer likewise): out = clip8((2^21 + sum_i in[lo+i] * coeffs[o][i]) >> 22).
 * bounds [n_out][2] = (lo, count), coeffs [n_out][ksize] int32 with 22 fractional bits, both built by the host as Pillow's
 * precompute_coeffs / normalize_coeffs_8bpc do (ops.pil_resize_tables).  axis 0: width ([B][H][Win][C] -> [B][H][Wout][C]),
 * axis 1: height.  Bit-exact with Pillow when the horizontal pass runs first. */
int mt4_resize_pass_u8(const void* in, void* out, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int32_t B, int32_t Hin,
                       int32_t Win, int32_t Hout, int32_t Wout, int32_t C, int32_t axis, void* stream);
/* MaxPool2d(3, stride 2, pad 1) channels-last (resnet.py:149). C*esize % 16 == 0. */
int mt4_maxpool3x3s2_nhwc(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                          void* stream);
/* conv1 / bn1 / relu / maxpool of the ResNet stem (resnet.py:145-149) in ONE launch on the space-to-depth frame: x_s2d [B][Hs][Ws][16]
 * bf16 (mt4_preprocess_u8_s2d), w_packed [Cout = 64][KH * 64] bf16 (the KH x 1 kernel over runs of 4 pixels, BatchNorm scale folded in;
 * mt4_pack_conv_weight), bias float32 -> y [B][Hp][Wp][64] bf16, Ho = Hs - KH + 1, Wo = Ws - 3, Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1.
 * Bit-identical to mt4_conv_nhwc (relu) followed by mt4_maxpool3x3s2_nhwc; the Ho x Wo x 64 map never reaches memory.  Wo a multiple of
 * 16 and <= 224 (frames up to 448 pixels wide: 224 x 224 and the reference's 256 x 448); otherwise MT4_EUNSUPPORTED and the caller runs
 * the two launches. */
int mt4_stem_maxpool_bf16(const void* x_s2d, const void* w_packed, const float* bias, void* y, int32_t B, int32_t Hs, int32_t Ws,
                          int32_t Cout, int32_t KH, void* stream);
/* AdaptiveAvgPool2d(1) channels-last -> float32 [B][C] (resnet.py:157; the hooked `final_feature`). */
int mt4_global_avgpool_nhwc(const void* x, float* y, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* stream);

/* y[B][N] = x[B][K] @ w[N][K]^T + bias  (float32; Classifier.fc, Spatial_cnn/network.py:121-129,
 * the four heads concatenated along N). */
int mt4_linear_f32(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t K, int32_t N,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer-shaped stages (Swin + Query2Label under Spatial_transformer/, MS-TCT under Temporal_mstct/).
 * Every nn.Linear / 1x1 conv of those stages runs through mt4_conv_nhwc (KH=KW=1, act = GELU where the
 * reference applies nn.GELU); the entry points below are the pieces between the GEMMs.
 */

/* LayerNorm over the last dimension with an optional row gather in front:
 *   y[m][0..G*C) = LN( concat_{g<G} x[src(m,g)][0..C) ) * gamma + beta,
 *   src(m,g) = (m / L_out) * L_in + map[(m % L_out) * G + g]      (map NULL: src = m, G must be 1)
 * G=1 + map: norm1 + cyclic shift + window partition of a Swin block (swin_transformer.py:241-252);
 * G=4 + map: PatchMerging's 2x2 gather + norm (swin_transformer.py:320-326); no map: plain nn.LayerNorm.
 * x, y: dtype; gamma, beta: float32 [G*C]. C*esize % 16 == 0. */
int mt4_layernorm(const void* x, const float* gamma, const float* beta, void* y, int32_t M_out, int32_t C, int32_t G,
                  const int32_t* map, int32_t L_out, int32_t L_in, float eps, int32_t dtype, void* stream);

/* Multi-head attention core (no projections):
 *   out[b,i,h,:] = softmax_j( scale * <q[b,i,h,:], k[b,j,h,:]> + bias[h,i,j] + mask[b % nW,i,j] ) . v[b,j,h,:]
 * q/k/v/out row (b*N + i) starts at ptr + row*stride (elements), head h at +h*hd.  bias [H][Nq][Nk] and
 * mask [nW][Nq][Nk] are float32 or NULL.  Replaces WindowAttention's core (swin_transformer.py:120-141, with the
 * relative-position bias gathered to dense form at load time and the -100 shift mask), nn.MultiheadAttention's
 * core in the Q2L transformer (transformer.py:186-189,275-283) and Global_Relational_Block (Temporal_Encoder.py:
 * 80-86).  hd <= 512.  bf16 with hd == 256 or 384 (Q2L over Swin-B / the shipped Swin-L teacher: 1536 / 4 heads), no bias / mask and Nk <= 160 runs on the matrix units (probabilities rounded to bf16
 * before the PV product, like the window kernel); fp32 with hd <= 128, hd % 4 == 0, no bias / mask and Nk <= 256 on the exact-fp32
 * matrix instruction (MS-TCT's global block). */
int mt4_attention(const void* q, const void* k, const void* v, void* out, const float* bias, const float* mask, int32_t B,
                  int32_t H, int32_t Nq, int32_t Nk, int32_t hd, int32_t q_stride, int32_t k_stride, int32_t v_stride,
                  int32_t o_stride, int32_t nW, float scale, int32_t dtype, void* stream);

/* Same core on the matrix units for the Swin shape: bf16, head dim 32, N <= 256 tokens per window (q, k, v from one
 * packed qkv projection), one workgroup per (window, head).  bias_padded [H][NP][NP] / mask_padded [nW][NP][NP] float32
 * with NP = round_up(N,16); padded KEY columns of bias hold -1e30 (they vanish in the softmax), everything else 0. */
int mt4_window_attention_bf16(const void* q, const void* k, const void* v, void* out, const float* bias_padded,
                              const float* mask_padded, int32_t B, int32_t H, int32_t N, int32_t q_stride, int32_t k_stride,
                              int32_t v_stride, int32_t o_stride, int32_t nW, float scale, void* stream);
/* The same core for a Swin block given its relative-position table instead of the expanded bias: rel_table [H][(2ws-1)^2] fp32
 * (`relative_position_bias_table` transposed, swin_transformer.py:81-83), bias(i,j) = table[(yi-yj+ws-1)(2ws-1) + xi-xj+ws-1]
 * (:92-103); region [nW][ws*ws] int32 = the shifted-window region id of every token of each window type (`img_mask` of :210-221
 * after window_partition; ids 0 .. 15 -- the reference uses 0 .. 8), mask = -100 where the ids differ (:222-229), applied as +100 where they are
 * equal (the same softmax); NULL for an unshifted block.  N = ws*ws <= 256. */
int mt4_window_attention_rel_bf16(const void* q, const void* k, const void* v, void* out, const float* rel_table, const int32_t* region,
                                  int32_t ws, int32_t B, int32_t H, int32_t q_stride, int32_t k_stride, int32_t v_stride, int32_t o_stride,
                                  int32_t nW, float scale, void* stream);

/* Non-overlapping PxP patches as GEMM rows: out[(b*H/P + ph)*W/P + pw][c*P*P + kh*P + kw] (PatchEmbed.proj,
 * swin_transformer.py:435,446).  in: normalised float32 NCHW, or (from_u8) uint8 NHWC frames normalised on the fly. */
int mt4_patchify(const void* in, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t from_u8, const float mean[3],
                 const float std[3], int32_t dtype, void* stream);

/* y[m][:] = x[m][:] + p[m % L][:]   (with_pos_embed, transformer.py:176-178: the sine position code / query
 * embedding is the same for every batch element) */
int mt4_add_rowbcast(const void* x, const void* p, void* y, int64_t M, int32_t L, int32_t C, int32_t dtype, void* stream);

/* GroupWiseLinear (Spatial_transformer/network.py:40-45): out[b][k] = sum_d W[k][d]*hs[b][k][d] + bias[k], float32 out */
int mt4_groupwise_linear(const void* hs, const float* w, const float* bias, float* out, int32_t B, int32_t K, int32_t D,
                         int32_t dtype, void* stream);

/* Depthwise Conv1d(k=3, pad=1, groups=C) over time on [B][T][C] + bias + activation (0 none / 1 ReLU / 2 GELU):
 * Local_Relational_Block.TC followed by its GELU (Temporal_Encoder.py:36-40).  w float32 [C][3]. */
int mt4_dwconv1d_k3(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t T, int32_t C, int32_t act,
                    int32_t dtype, void* stream);

/* Train-time teacher mixing of the KD branch (Spatial_cnn/network.py:56-62) in its reduced form, float32:
 * out_n[b][c] = s[b][c] * softmax_n( s[b][c]/sqrt(C) * sum_d tea_n[b][d] ), n in {i,v,t}. */
int mt4_kd_mix(const float* s, const float* tea_i, const float* tea_v, const float* tea_t, float* out_i, float* out_v,
               float* out_t, int32_t B, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step of the temporal head (Temporal_tenco/run.py:181-235: forward, BCE, backward, SGD).  fp32.
 * Data gradients reuse mt4_conv_nhwc with the transposed / tap-reversed weights produced by
 * mt4_transpose_pack_conv1d_f32 (and act 3 for the ReLU gate).
 */

/* Conv1d weight gradient in the packed layout of mt4_conv_nhwc: dw[co][tap*CPT4 + ci] (+)= sum_{b,t} dy[b,t][co] *
 * x[b, t + tap*dil - pad][ci] (zero outside the sequence).  dy [B*T][Cout], x [B*T][Cin]; Cin, Cout % 4 == 0.
 * bias_grad (optional, [Cout]): bias_grad[co] += sum_{b,t} dy[b,t][co] in the same launch (always ADDED, with fp32 atomics: the caller
 * zeroes it, `accumulate` governs dw_packed only) -- what mt4_colsum_f32 would compute in a launch of its own.  Alignment: dy and x 16 bytes
 * (MT4_EALIGN otherwise); dw_packed and bias_grad need float alignment only. */
int mt4_wgrad_conv1d_f32(const float* dy, const float* x, float* dw_packed, int32_t B, int32_t T, int32_t Cout, int32_t Cin,
                         int32_t taps, int32_t dil, int32_t pad, int32_t accumulate, float* bias_grad, void* stream);
/* out[c] (+)= sum_m x[m*ld + c]   (bias gradients).  `out` needs float alignment only (a bias slice of a flat gradient buffer). */
int mt4_colsum_f32(const float* x, float* out, int64_t M, int32_t C, int32_t ld, int32_t accumulate, void* stream);
/* nn.BCEWithLogitsLoss pieces (run.py:196-212, 331): col_loss[n] += sum_m bce(y[m][n], z[m][n]);
 * dy[m*ld_dy + n] = (sigmoid(y) - z) * col_scale[n].  z [M][N] dense. */
int mt4_bce_logits_f32(const float* y, const float* z, const float* col_scale, float* dy, float* col_loss, int64_t M, int32_t N,
                       int32_t ld_y, int32_t ld_dy, void* stream);
/* torch.optim.SGD without momentum (run.py:343): p -= lr * (grad_scale * g + weight_decay * p) */
int mt4_sgd_step_f32(float* p, const float* g, int64_t n, float lr, float weight_decay, float grad_scale, void* stream);
/* y = a * b (+ c): dropout masks (network.py:194-196) forward and backward */
int mt4_mul_add_f32(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream);
/* packed Conv1d weight [Cout][Kpad(Cin,taps)] -> packed weight of its data-gradient conv [Cin][Kpad(Cout,taps)] */
int mt4_transpose_pack_conv1d_f32(const float* w_packed, float* wt_packed, int32_t Cout, int32_t Cin, int32_t taps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step of the spatial stage (Spatial_cnn/run.py:145-224).  fp32, channels-last.
 */
/* BatchNorm2d in training mode: batch mean / biased variance over the M rows of x [M][C] -> mean, invstd; running stats
 * updated with `momentum` and the unbiased variance (may be NULL).  sums_zeroed: 2*C DOUBLES, zero on entry
 * (the per-channel reductions accumulate in float64, like the CPU BatchNorm of torch). */
int mt4_bn_stats_f32(const float* x, double* sums_zeroed, float* mean, float* invstd, float* running_mean, float* running_var, int64_t M,
                     int32_t C, float momentum, float eps, void* stream);
/* y = act((x - mean) * invstd * gamma + beta [+ residual]); relu 0/1 (resnet.py:105-119) */
int mt4_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* residual,
                     float* y, int64_t M, int32_t C, int32_t relu, void* stream);
/* the statistics step of mt4_bn_stats_f32 + mt4_bn_apply_f32 in ONE launch, from the channel sums a convolution's epilogue left
 * (mt4_conv_desc.stat_sums, [MT4_STAT_REPLICAS][2][C]); mean / invstd are written for the backward, the running statistics advance: the fp32 twin
 * of mt4_bn_apply_sums_t */
int mt4_bn_apply_sums_f32(const float* x, const double* stat_sums, float* mean, float* invstd, float* running_mean, float* running_var,
                          const float* gamma, const float* beta, const float* residual, float* y, int64_t M, int32_t C, float momentum, float eps,
                          int32_t relu, void* stream);
/* backward of the above: dy' = gated dy; dx (w.r.t. x), dres = dy' (w.r.t. residual, may be NULL), dgamma, dbeta.  relu: 0 none; 1 the gate is
 * y_post > 0; 2 it is recomputed from x -- (x - mean) * invstd * gamma + beta > 0, the forward's own fp32 expression = the stored y bit for bit
 * (units WITHOUT a residual input): y_post may be NULL and is not read.  beta: needed for relu 2 only.  sums_zeroed: 2*C float64. */
int mt4_bn_backward_f32(const float* dy, const float* y_post, const float* x, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, double* sums_zeroed, float* dx, float* dres, float* dgamma, float* dbeta, int64_t M, int32_t C,
                        int32_t relu, void* stream);
/* Conv2d weight gradient in the packed layout, accumulated with atomics into a zeroed buffer */
int mt4_wgrad_conv2d_f32(const float* dy, const float* x, float* dw_packed_zeroed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Ho,
                         int32_t Wo, int32_t Cout, int32_t KH, int32_t KW, int32_t stride_h, int32_t stride_w, int32_t pad_h, int32_t pad_w,
                         int32_t dil_h, int32_t dil_w, void* stream);
/* MaxPool2d(3,2,1) backward (first maximum in scan order, like torch) into a zeroed dx; AdaptiveAvgPool2d(1) backward */
int mt4_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx_zeroed, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mt4_avgpool_bwd_f32(const float* dfeat, float* dx, int32_t B, int32_t HW, int32_t C, void* stream);
/* losses of run.py:159-192: BCEWithLogitsLoss(pos_weight), DistillKL (T, on sigmoid(teacher)), MSELoss; each writes the
 * gradient w.r.t. its first argument scaled by grad_scale / col_scale and ADDS its value to *loss / col_loss */
int mt4_bce_logits_pw_f32(const float* y, const float* z, const float* pos_weight, const float* col_scale, float* dy, float* col_loss, int32_t M,
                          int32_t N, int32_t ld_y, int32_t ld_dy, void* stream);
int mt4_distill_kl_f32(const float* y_s, const float* t_pred, float* dy_s, float* loss, int32_t B, int32_t K, int32_t ld_y, int32_t ld_dy,
                       float temp, float grad_scale, int32_t accumulate, void* stream);
int mt4_mse_f32(const float* a, const float* b, float* da, float* loss, int64_t n, float grad_scale, void* stream);
/* backward of mt4_kd_mix: ds [B][C] and dtau [B][3] (gradient w.r.t. the per-teacher sums) from g_n = dL/d(out_n) */
int mt4_kd_mix_bwd_f32(const float* s, const float* tea_i, const float* tea_v, const float* tea_t, const float* g_i, const float* g_v,
                       const float* g_t, float* ds, float* dtau, int32_t B, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16-operand training mode of the spatial stage (the same step as above -- Spatial_cnn/run.py:145-224 -- with the convolutions' GEMM operands
 * in bf16): activations and activation gradients are bf16 tensors, sums run in fp32 (MFMA) / fp64 (BatchNorm reductions), master weights,
 * gradients and SGD stay fp32.  Forward and data-gradient convolutions are mt4_conv_nhwc with dtype MT4_BF16 on bf16 copies of the packed
 * weights (mt4_repack_weight_bf16).  `x_dtype` is the type of the convolution output the BatchNorm reads (MT4_BF16; MT4_F32 for the stem, whose
 * 7x7x3 convolution and weight gradient stay fp32).
 */
int mt4_bn_stats_t(const void* x, int32_t x_dtype, double* sums_zeroed, float* mean, float* invstd, float* running_mean, float* running_var,
                   int64_t M, int32_t C, float momentum, float eps, void* stream);
/* y (bf16) = act( (x - mean) * invstd * gamma + beta [+ residual (bf16)] ) */
int mt4_bn_apply_t(const void* x, int32_t x_dtype, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   const void* residual_bf16, void* y_bf16, int64_t M, int32_t C, int32_t relu, void* stream);
/* mt4_bn_stats_t's second step + mt4_bn_apply_t in ONE launch, from the channel sums a convolution's epilogue left (mt4_conv_desc.stat_sums,
 * [MT4_STAT_REPLICAS][2][C]): mean / invstd are computed (float64, the expressions of mt4_bn_stats_t) and written for the backward, the running
 * statistics take nn.BatchNorm2d's momentum update, and y = act((x - mean) * invstd * gamma + beta [+ residual]) as above. */
int mt4_bn_apply_sums_t(const void* x, int32_t x_dtype, const double* stat_sums, float* mean, float* invstd, float* running_mean, float* running_var,
                        const float* gamma, const float* beta, const void* residual_bf16, void* y_bf16, int64_t M, int32_t C, float momentum, float eps,
                        int32_t relu, void* stream);
/* as mt4_bn_backward_f32 with dy / y_post / dres in bf16 and dx in the type of x.  relu: 0 none; 1 the ReLU gate is read from y_post; 2 it is
 * recomputed from x ((x - mean) * invstd * gamma + beta > 0, the forward's own expression; units WITHOUT a residual input only): y_post may be
 * NULL and is not read */
int mt4_bn_backward_t(const void* dy_bf16, const void* y_post_bf16, const void* x, int32_t x_dtype, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, double* sums_zeroed, void* dx, void* dres_bf16, float* dgamma, float* dbeta, int64_t M,
                      int32_t C, int32_t relu, void* stream);
/* Conv2d weight gradient on bf16 MFMA: dw_packed (fp32, the layout of mt4_pack_conv_weight(MT4_F32)) += sum_pixels dy[p][n] * x[in(p, tap)][c].
 * dy [B][Ho][Wo][Cout] bf16, x [B][H][W][Cin] bf16; K = 1 or 3 (square, pad K / 2), stride 1 or 2 (MT4_EUNSUPPORTED otherwise), Cin % 8 == 0 and
 * Cout % 8 == 0 (MT4_EALIGN).  The pixel range is split over workgroups and summed with fp32 atomics (order run-dependent). */
int mt4_wgrad_conv2d_bf16(const void* dy, const void* x, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Ho, int32_t Wo,
                          int32_t Cout, int32_t K, int32_t stride, void* stream);
/* MaxPool2d(3,2,1) backward, bf16, gather form (no atomics; dx need not be zeroed) */
int mt4_maxpool3x3s2_bwd_bf16(const void* x, const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* AdaptiveAvgPool2d(1) backward: dfeat [B][C] fp32 -> dx [B][HW][C] bf16 */
int mt4_avgpool_bwd_bf16(const float* dfeat, void* dx, int32_t B, int32_t HW, int32_t C, void* stream);
/* y (bf16) = x (fp32), round to nearest even; n % 8 == 0, 16-byte aligned: the operand copy of a mixed-precision GEMM */
int mt4_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, void* stream);
/* packed fp32 weight matrix (mt4_pack_conv_weight(MT4_F32) layout, e.g. the master weights or their transposed copies) -> the packed bf16
 * layout of the same geometry */
/* One launch that rebuilds every weight matrix derived from the fp32 master weights (after an optimizer step): entry i fills dst (fp32 or bf16,
 * packed layout: rows x (ntaps_dst taps of tapw_dst elements, zero padded to kpad_dst)) from the packed master matrix src ([cout][kpad_src],
 * taps of tapw_src elements):
 *     transposed == 0:  dst[n][t][c] = src[n][tap_map[t]][c]      (a bf16 copy of a forward weight)
 *     transposed == 1:  dst[c][t][n] = src[n][tap_map[t]][c]      (data-gradient operators: flipped taps, sub-pixel phases of a strided conv)
 * The unit is a 32 (n) x 32 (c) tile of one tap and a workgroup moves MT4_REFRESH_TILES_PER_BLOCK consecutive tiles of one entry: an entry takes
 * ceil(ntaps_dst * ceil(cout / 32) * ceil(cin / 32) / MT4_REFRESH_TILES_PER_BLOCK) workgroups, block0 = its first one (entries ordered by block0).  Only valid elements are written: zero the destinations once at allocation.  The table lives in device memory. */
#define MT4_REFRESH_TILES_PER_BLOCK 4
typedef struct {
    const float* src;
    void* dst;
    int64_t block0;
    int32_t dst_bf16, transposed, cout, cin, ntaps_dst, tapw_src, kpad_src, tapw_dst, kpad_dst;
    int32_t tap_map[9];
} mt4_refresh_entry;
int mt4_refresh_weights(const mt4_refresh_entry* table_dev, int32_t n_entries, int64_t n_blocks, void* stream);
int mt4_repack_weight_bf16(const float* w_f32_packed, void* w_bf16_packed, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Temporal head, latency path: Conv1d (k = 1 or 3, any dilation, stride 1, padding = dilation*(k-1)/2) over frame-major rows,
 * fused epilogue   y = act( conv(x, w) + bias [+ residual] )
 * for ONE short video (or a few): replaces `DilatedResidualLayer` (Temporal_tenco/network.py:186-198: conv_dilated -> ReLU ->
 * conv_1x1 -> +x, Dropout inactive in eval) and the 1x1 projections / heads around it (network.py:21-24,96,113,129).
 * Same arithmetic contract as mt4_conv_nhwc (fp32 = exact-fp32 MFMA chain, bf16 = fp32 accumulate); the K sum is split over the
 * 8 waves of a workgroup by 128-byte channel slice and the partial tiles are added in a fixed order, so results are deterministic
 * and independent of how many videos ride along, but differ from mt4_conv_nhwc's by fp32 reassociation.
 *   x        [B*T][Cin]   dtype        w  [Cout][taps*Cin] packed by mt4_pack_conv_weight(Cout, Cin, 1, taps, dtype)
 *   bias     [Cout] float32 or NULL    residual [B*T][Cout] dtype or NULL     y [B*T][Cout] out_dtype
 * Contract: Cin*esize % 128 == 0 (MT4_EUNSUPPORTED otherwise: use mt4_conv_nhwc), pointers 16-byte aligned, B*T*Cin*esize < 2 GiB.
 * One workgroup per 32 frames x 16 channels: meant for B*T up to a few hundred frames (above that mt4_conv_nhwc's large tiles
 * move fewer operand bytes per FLOP). */
typedef struct mt4_tcn_desc {
    const void* x;
    const void* w;
    const float* bias;
    const void* residual;
    void* y;
    int32_t B, T, Cin, Cout;
    int32_t taps;      /* 1 or 3 */
    int32_t dilation;  /* >= 1 (ignored for taps == 1) */
    int32_t relu;      /* 0 / 1 */
    int32_t dtype;     /* MT4_F32 / MT4_BF16: x, w, residual */
    int32_t out_dtype; /* MT4_F32 / MT4_BF16: y */
} mt4_tcn_desc;
int mt4_tcn_conv(const mt4_tcn_desc* d, void* stream);
/* One DilatedResidualLayer: y = x + conv_1x1(relu(conv_dilated(x))) (network.py:193-198); h [B*T][C] is scratch for the hidden
 * activation; C*esize % 128 == 0.  Two dependent launches of the kernel above. */
int mt4_tcn_dilated_residual_layer(const void* x, const void* w_dilated, const float* b_dilated, const void* w_1x1, const float* b_1x1,
                                   void* h, void* y, int32_t B, int32_t T, int32_t C, int32_t dilation, int32_t dtype, void* stream);
/* A stage's layer stack (BaseCausalTCN.layers / Refinement.layers, network.py:116-118,130-131,147-148,157-158): n_layers >= 1
 * DilatedResidualLayers with dilation 2^i.  Layer 0 reads x (never written); the layers in between ping-pong between the scratch
 * buffers buf_a / buf_b (needed for n_layers > 1 / > 2); the last layer writes y.  All activations [B*T][C]; the four arrays hold
 * n_layers DEVICE pointers each and live in HOST memory. */
int mt4_tcn_stage(const void* x, void* buf_a, void* buf_b, void* h, void* y, const void* const* w_dilated, const float* const* b_dilated,
                  const void* const* w_1x1, const float* const* b_1x1, int32_t n_layers, int32_t B, int32_t T, int32_t C,
                  int32_t dtype, void* stream);
/* One DilatedResidualLayer in ONE launch, bf16, the head's THROUGHPUT mode (several videos per forward; Temporal_tenco/network.py:186-198):
 * y[b][t] = x[b][t] + W2 . relu(W1 * x + b1) + b2, W1 a k = 3 convolution of `dilation` over the frames of video b (frames outside it are zeros),
 * x, y [B][T][512] bf16 (y != x), w1_frag / w2_frag = mt4_pack_fragments_bf16 of the packed [512][3 * 512] / [512][512] matrices, b1 / b2 fp32.
 * The 512-channel hidden map stays in LDS.  Bit-identical to the two mt4_conv_nhwc launches; tiles of 64 frames never cross a video.  C must be 512. */
int mt4_tcn_layer_fused_bf16(const void* x, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2, void* y, int32_t B, int32_t T,
                             int32_t C, int32_t dilation, void* stream);
/* nn.LayerNorm(Cin) followed by nn.Linear(Cin, Cout) on the `rows` frames of one short window in ONE launch of the latency kernel, fp32 -- MS-TCT's
 * `norm1 -> q | kv` and `norm2 -> linear1` (Temporal_mstct/MSTCT/Temporal_Encoder.py:74-86,35-40 with :112-113): with w_folded = gamma o W (packed like
 * any 1-tap weight of mt4_tcn_conv), colsum[n] = sum_k w_folded[n][k] and bias_folded = W . beta + b,
 *     y[t][n] = rstd_t (x_t . w_folded[n] - mean_t colsum[n]) + bias_folded[n]  (+ residual, ReLU)  ==  Linear(LayerNorm(x_t))[n];
 * mean_t / rstd_t (biased variance, eps inside the root): from stats_in [Cin / 16][rows][2] = per 16 channels (sum, sum of squares) of every row of x, as
 * the launch that produced x left them (`stats_out` of this function or of mt4_tcn_linear_stats_f32; Cin % 16 == 0, Cin <= 896), added in a fixed
 * order.  The normalised map is never written.  stats_out (may be NULL; Cout % 16 == 0): the partials of y for the next such launch.  Geometry contract of mt4_tcn_conv. */
int mt4_tcn_linear_ln_f32(const void* x, const void* w_folded, const float* colsum, const float* bias_folded, const void* residual, void* y,
                          int32_t rows, int32_t Cin, int32_t Cout, float eps, int32_t relu, const float* stats_in, float* stats_out, void* stream);
/* nn.Linear on one short window (mt4_tcn_conv, one tap, fp32: the residual-stream producers `proj` / `linear2` of an MS-TCT block,
 * Temporal_Encoder.py:86,40) that also writes stats_out [Cout / 16][rows][2] for the mt4_tcn_linear_ln_f32 launch that normalises its output. */
int mt4_tcn_linear_stats_f32(const void* x, const void* w, const float* bias, const void* residual, void* y, int32_t rows, int32_t Cin, int32_t Cout,
                             int32_t relu, float* stats_out, void* stream);
/* FPN top-down pathway at equal lengths (network.py:93-106; `F.interpolate(x, size=W, mode='linear')` to the same length is the
 * identity): levels[l] = lat[l] + levels[l+1] for l = nlev-2 .. 0, in place.  lat [nlev-1][n], levels [nlev][n] of dtype, n % 4 == 0. */
int mt4_fpn_topdown(const void* lat, void* levels, int32_t nlev, int64_t n, int32_t dtype, void* stream);
/* `--hier True` of Temporal_tenco, both over the time axis of frame-major rows x [B][Tin][C] (fp32 or bf16; C % 4 == 0, bf16 C % 8 == 0):
 * mt4_avgpool1d_rows: nn.AvgPool1d(k, stride) between the refinement stages (network.py:147,154-155: k = 7, stride = 3),
 *   y [B][(Tin - k) / stride + 1][C], sum of the k rows then one division;
 * mt4_interp_linear_rows: F.interpolate(x, size = Tout, mode = 'linear') (align_corners False) of the FPN's `_upsample_add` (network.py:96),
 *   y [B][Tout][C]: source index max(0, (Tin / Tout)(w + 0.5) - 0.5), the two neighbours weighted by its fraction. */
int mt4_avgpool1d_rows(const void* x, void* y, int32_t B, int32_t Tin, int32_t C, int32_t k, int32_t stride, int32_t dtype, void* stream);
int mt4_interp_linear_rows(const void* x, void* y, int32_t B, int32_t Tin, int32_t Tout, int32_t C, int32_t dtype, void* stream);
/* their adjoints for `--hier True` TRAINING (`loss.backward()` through `Refinement.max_pool_1x1` and `FPN._upsample_add`, Temporal_tenco/run.py:213),
 * fp32, deterministic gathers: dx [B][Tin][C] from dy [B][Tout][C] (Tout = (Tin - k) / stride + 1 for the pool; for the interpolation Tin is the
 * length of the FORWARD's input, Tout of its output). */
int mt4_avgpool1d_rows_bwd_f32(const float* dy, float* dx, int32_t B, int32_t Tin, int32_t C, int32_t k, int32_t stride, void* stream);
int mt4_interp_linear_rows_bwd_f32(const float* dy, float* dx, int32_t B, int32_t Tin, int32_t Tout, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pieces of the transformer-shaped temporal teacher MS-TCT (what torch autograd derives inside
 * Temporal_mstct/run.py:147-235 for Temporal_mstct/MSTCT/Temporal_Encoder.py).  float32.
 */
/* Strided batched GEMM: C[b1][b0] (M x N) = alpha * A[b1][b0] (M x K) . B[b1][b0] (K x N)  (+ C when accumulate).
 * *_strides = element strides {batch-inner b0, batch-outer b1, rows, cols}: for A {b0, b1, m, k}, for B {b0, b1, k, n}, for C {b0, b1, m, n}.
 * Transposes and head slices of a packed projection buffer ([B*T][heads*hd]: b0 = head -> hd, b1 = sequence -> T*C) are strides, not
 * copies: q.k^T, P.v and the four products of the attention backward (Temporal_Encoder.py:76-88) all go through it.  nb0*nb1 <= 65535. */
int mt4_bgemm_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int32_t nb0, int32_t nb1,
                  const int64_t a_strides[4], const int64_t b_strides[4], const int64_t c_strides[4], float alpha, int32_t accumulate,
                  void* stream);
/* P = softmax(scale * S) over the last dimension, in place (Temporal_Encoder.py:82-83); cols <= 1024 */
int mt4_softmax_rows_f32(float* S, int64_t rows, int32_t cols, float scale, void* stream);
/* dS = scale * P .* (dP - rowsum(P .* dP)), written over dP */
int mt4_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int32_t cols, float scale, void* stream);
/* nn.LayerNorm backward: dx (written, or added to when accumulate_dx), dgamma / dbeta ADDED to (float atomics; zero or pre-fill them).
 * Statistics are recomputed from x.  C <= 1024. */
int mt4_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, float* dx, float* dgamma, float* dbeta, int64_t M, int32_t C,
                          float eps, int32_t accumulate_dx, void* stream);
/* nn.GELU (erf form) forward on a float32 buffer, y = gelu(x); n % 4 == 0 (the fused epilogues of mt4_conv_nhwc / mt4_dwconv1d_k3 apply the
 * same function; a training step that keeps the pre-activation calls this instead) */
int mt4_gelu_f32(const float* x, float* y, int64_t n, void* stream);
/* nn.GELU (erf form) backward: dx = dy * gelu'(x) with x the pre-activation; n % 4 == 0 */
int mt4_gelu_bwd_f32(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* backward of mt4_dwconv1d_k3 (depthwise Conv1d k3 pad 1, Temporal_Encoder.py:12,38): dx written, dw [C][3] / db [C] ADDED to */
int mt4_dwconv1d_k3_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int32_t B, int32_t T, int32_t C,
                            void* stream);
/* nn.Dropout(p) keep mask (network.py:58,76,108,113 draw theirs from torch's RNG): out[i] = u_i >= p ? 1/(1-p) : 0 with the counter
 * generator of computervision_codes_amd/synth.py (u_i = splitmix64(splitmix64(seed * 0x100000001B3 + stream_id) + i) >> 11 / 2^53), so a
 * draw is reproducible on the host */
int mt4_dropout_mask_f32(float* out, int64_t n, int64_t seed, int64_t stream_id, float p, void* stream);
/* y = a * x + b * y (b == 0: y is not read); n % 4 == 0 */
int mt4_axpby_f32(const float* x, float* y, int64_t n, float a, float b, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pieces of the Swin + Query2Label teacher (what torch autograd derives inside Spatial_transformer/run.py:150-229 for
 * Spatial_transformer/models/swin_transformer.py and models/transformer.py).  float32.  LayerNorm / GELU / softmax / batched-GEMM
 * backward are the entry points above.
 */
/* Row gather by a per-image map: y[m][g*C + c] = x[(m / l_out) * l_in + map[(m % l_out) * group + g]][c]  -- roll + window_partition
 * (swin_transformer.py:241-252; group 1) and PatchMerging's 2x2 gather (:320-324; group 4).  scatter != 0 runs the inverse (the maps
 * are bijections): x[...] = y[m][...], which is both window_reverse + roll back (:255-265) and the gather's backward.  C % 4 == 0. */
int mt4_gather_rows_f32(const float* x, const int32_t* map, float* y, int64_t m_out, int32_t C, int32_t group, int32_t l_out, int32_t l_in,
                        int32_t scatter, void* stream);
/* S[w][h][i][j] += bias (+ mask[w % nW][i][j]); S [n_windows][heads][N][N]; mask [nW][N][N] or NULL (swin_transformer.py:127-136).
 * index NULL: bias is dense [heads][N][N]; else bias is the relative-position table [(2ws-1)^2][heads] read as bias[index[i*N+j]][h]
 * (`relative_position_bias_table[relative_position_index]`, :127-130), index [N*N] int32 */
int mt4_add_bias_mask_f32(float* S, const float* bias, const int32_t* index, const float* mask, int64_t n_windows, int32_t heads, int32_t N,
                          int32_t nW, void* stream);
/* dtable[index[i*N + j]][h] += sum_w dS[w][h][i][j]: gradient of `relative_position_bias_table` [(2ws-1)^2][heads] through the gather of
 * swin_transformer.py:127-130; index [N*N] int32.  Added to (float atomics). */
int mt4_relpos_table_grad_f32(const float* dS, const int32_t* index, float* dtable, int64_t n_windows, int32_t heads, int32_t N, void* stream);
/* y[m][:] = scale[m / rows_per_scale] * x[m][:] (+ r[m][:]): timm DropPath on a residual branch (swin_transformer.py:266,269; scale =
 * keep mask / keep_prob per sample) and its backward.  r may be NULL.  C % 4 == 0. */
int mt4_rowscale_add_f32(const float* x, const float* scale, const float* r, float* y, int64_t M, int32_t C, int32_t rows_per_scale, void* stream);
/* backward of mt4_groupwise_linear (GroupWiseLinear, Spatial_transformer/network.py:40-45): dhs written, dW / db ADDED to */
int mt4_groupwise_linear_bwd_f32(const float* dy, const float* hs, const float* W, float* dhs, float* dW, float* db, int32_t B, int32_t K, int32_t D,
                                 void* stream);
/* out[i] (+)= sum_b x[b*LC + i], i < LC: gradient of a row-broadcast add (query embedding / position code added to every image) */
int mt4_sum_over_batch_f32(const float* x, float* out, int32_t B, int64_t LC, int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MT4HIP_H */
