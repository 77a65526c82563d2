// One ResNet Bottleneck (stride 1, 64 mid channels: layer1 of ResNet-50) in ONE launch, bf16 NHWC, eval-mode folded BatchNorm
// (`Spatial_transformer/models/resnet.py:101-121`):
//
//     t1 = relu(W1 . x + b1)            1x1, CIN -> 64     on the tile + its 1-pixel halo
//     t2 = relu(W2 * t1 + b2)           3x3, 64 -> 64      t1 zero outside the image (conv2's padding)
//     y  = relu(W3 . t2 + b3 + idt)     1x1, 64 -> 256     idt = x (CIN = 256) or bf16(Wds . x + bds) (layer1.0: CIN = 64)
//
// Layer by layer this block moves 8.5 GB per 1336 frames through HBM (x read twice, t1 / t2 written and read, y written) and every one of
// its launches runs at the HBM roofline; fused, t1 and t2 never leave LDS: x is read once (+ the halo ring), y written once.
//
// A workgroup (4 waves) owns an 8 x 14 pixel tile of one frame: 112 output pixels = 7 MFMA pixel tiles, halo 10 x 16 = 160 pixels = 10 pixel
// tiles (one per halo row).  As in igemm_conv.hip the weight tile is the MFMA A operand and the pixel tile the B operand, so an accumulator
// lane owns 4 consecutive channels of one pixel.  Weight fragments come straight from global memory (L2-resident: 136 KB for the block) into
// registers, issued a phase ahead, from a FRAGMENT-ORDERED copy of the packed weights (mt4_bottleneck_pack_bf16: a wave's load of one
// fragment is 1 KB contiguous = 8 full cache lines; read from the row-major packed layout it was 16 half lines, and with 136 KB of weights
// per tile those were most of the kernel's L2 requests); pixel fragments from LDS rows of 64 channels (128 B, 16-byte chunks XOR-swizzled with row & 7).
//   phase 1: x halo tile in K-chunks of 64 channels, global -> registers -> LDS, two chunks in flight; wave w computes channels 16w..16w+15
//            of t1 for all 160 halo pixels; out-of-image halo pixels are written as zeros.
//   phase 2: conv2 from t1 (tap (kh, kw) of pixel (ty, tx) = t1 row (ty + kh) * 16 + tx + kw); wave w: 16 channels x 7 pixel tiles.
//   phase 3: conv3 in two passes of 128 output channels; wave w owns channel tiles w, w + 4, w + 8, w + 12; residual added in the
//            accumulator layout (identity blocks: picked out of the LDS copy of x chunk i during phase 1 -- no second read), result staged in
//            LDS and stored as 16-byte vectors, 256 B per pixel.
//   phase 4 (NEXT, mt4_bottleneck_fused_next_bf16): the 1x1 conv of the block that follows layer1 (256 -> 128 channels) on the tile while both
//            passes are in LDS; the block's own map is then stored at the even pixels only (its remaining reader is the stride-2 branch).
// K order and MFMA chain (bias-initialised fp32 accumulator; K-steps of 64 channels ascending, 3x3: tap-major; two 32-wide MFMAs per step;
// t1 / t2 / the downsample branch rounded to bf16 exactly where the layer-by-layer path stores them) are those of the stand-alone launches:
// the result is bit-identical to conv1 -> conv2 -> conv3 through mt4_conv_nhwc.
#include "mt4_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace {

constexpr int TH = 8, TW = 14, NHALO = 160, NPX = 128;   // NPX: the 8 x 16 grid of phases 2 and 3 (14 of 16 columns are stored)
constexpr int ROW_B = 128;                   // 64 bf16 channels
constexpr int XS_BYTES = NHALO * ROW_B;      // one K-chunk of the halo tile (also the size of t1)
constexpr int T2_BYTES = NPX * ROW_B;
constexpr int T1_BYTES = XS_BYTES + 2 * ROW_B;   // + the two slack rows the junk columns read
constexpr int YS_BYTES = NPX * 256;          // one pass of the output staging: 128 channels per pixel

struct BneckK {
    const char* x;
    char* y;
    const char *w1, *w2, *w3, *wds;             // fragment-ordered weights (mt4_bottleneck_pack_bf16): [channel tile][K step of 32][lane] x 16 B
    const float *b1, *b2, *b3, *bds;
    int B, H, W, tiles_h, tiles_w;
    int nt;                                  // non-temporal output stores
    // NEXT: the 1x1 conv that follows the block (conv1 + bn1 + relu of the strided Bottleneck behind it, 256 -> 128 channels) runs on the tile's
    // result while it is in LDS: y2 [B][H][W][128]; y itself is then stored at the even pixels only, [B][H2][W2][256] (all its other reader,
    // the stride-2 downsample branch, takes)
    char* y2;
    const char* wn;                          // fragment-ordered [8 channel tiles][8 K steps][lane] x 16 B
    const float* bn;
    int H2, W2;
};

__device__ __forceinline__ f32x4 bias4(const float* b, int n) {
    const float4 t = *(const float4*)(b + n);
    return (f32x4){t.x, t.y, t.z, t.w};
}

// ReLU + round of 4 accumulator values -> 4 bf16: rounding commutes with ReLU (sign-preserving), and on the packed pair ReLU is one
// integer max with 0 per dword (negative bf16 = negative int16; -0.0 -> +0) instead of one v_max_f32 per value
__device__ __forceinline__ uint2 relu_pack4(f32x4 v) {
    uint2 o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.x));
    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.y));
    return o;
}

__device__ __forceinline__ f32x4 mfma_bf16(uint4 wfrag, uint4 xfrag, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag), __builtin_bit_cast(bf16x8_t, xfrag), acc, 0, 0, 0);
}

template <int CIN, bool DS, bool NEXT = false>
__global__ __launch_bounds__(256, 2) void bottleneck64_fused_kernel(const BneckK a) {
    static_assert(!NEXT || !DS, "the following conv rides behind an identity block");
    constexpr int KC = CIN / 64;
    constexpr int NXS = KC > 1 ? 2 : 1;
    static_assert(DS ? CIN == 64 : CIN == 256, "layer1.0 (64 channels in, downsample branch) or layer1.1+ (256 in, identity)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS map: xs[NXS] | t2 | t1 + 2 slack rows (| tail up to the staging size when DS).  The output staging aliases xs (identity: both chunks are dead after phase 1) or
    // t1 + tail (downsample: xs holds the block input, the B operand of the downsample GEMM in phase 3)
    char* const xs = smem;
    char* const t2 = smem + NXS * XS_BYTES;
    char* const t1 = t2 + T2_BYTES;
    char* const ys = DS ? t1 : xs;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;

    int bid = blockIdx.x;
    {   // XCD-contiguous tile ranges: neighbouring tiles (shared halo rows) meet in one XCD's L2
        const int nb = gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7;
        const int xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int tpi = a.tiles_h * a.tiles_w;
    const int img = bid / tpi;
    const int trem = bid - img * tpi;
    const int th = trem / a.tiles_w;
    const int h0 = th * TH, w0 = (trem - th * a.tiles_w) * TW;
    const char* const ximg = a.x + (long long)img * a.H * a.W * (CIN * 2);
    char* const yimg = NEXT ? a.y + (long long)img * a.H2 * a.W2 * 512 : a.y + (long long)img * a.H * a.W * 512;

    // ---------------------------------------------------------------- phase 1: t1 = relu(W1 . x + b1) on the halo tile
    // staging: thread (row tid >> 3, chunk tid & 7) moves 16 bytes of halo pixels row, row + 32, ... (5 passes per K-chunk)
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    int xoff[5];          // byte offset of the halo pixel in the image, -1 outside
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int hp = ld_row + 32 * i;
        const int h = h0 - 1 + (hp >> 4), w = w0 - 1 + (hp & 15);
        xoff[i] = ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) ? (h * a.W + w) * (CIN * 2) + ld_chunk * 16 : -1;
    }
    auto load_chunk = [&](int kc, uint4 (&stg)[5]) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            stg[i] = make_uint4(0, 0, 0, 0);
            if (xoff[i] >= 0) stg[i] = *(const uint4*)(ximg + xoff[i] + kc * 128);
        }
    };
    auto store_chunk = [&](int buf, const uint4 (&stg)[5]) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int hp = ld_row + 32 * i;
            *(uint4*)(xs + buf * XS_BYTES + hp * ROW_B + ((ld_chunk ^ (hp & 7)) << 4)) = stg[i];
        }
    };
    uint4 stg0[5], stg1[5];
    load_chunk(0, stg0);
    if constexpr (KC > 1) load_chunk(1, stg1);

    const int ch1 = wave * 16;                       // this wave's 16 channels of t1 / t2
    uint4 wf1[KC * 2];
    {
        const char* wr = a.w1 + (wave * (KC * 2) * 64 + lane) * 16;       // channel tile `wave`, K step s: 64 lanes x 16 B contiguous
#pragma unroll
        for (int s = 0; s < KC * 2; ++s) wf1[s] = *(const uint4*)(wr + s * 1024);
    }
    f32x4 acc1[10];
    {
        const f32x4 b4 = bias4(a.b1, ch1 + q * 4);
#pragma unroll
        for (int i = 0; i < 10; ++i) acc1[i] = b4;
    }
    const int sw0 = (q ^ (r16 & 7)) << 4, sw1 = ((4 + q) ^ (r16 & 7)) << 4;   // fragment chunk of kk = 0 / 1 in rows 16 i + r16
    // identity blocks: the residual of output channels 64 i + 16 wave + 4 q .. + 3 (channel tile 4 i + wave of pass i / 2) IS x chunk i at the
    // tile's own pixels -- picked out of the LDS copy of the chunk while it is there (row (j + 1) * 16 + r16 + 1 of the halo tile) instead of
    // read from memory a second time (8-byte loads in the accumulator layout: 66 M sector requests to the L2 per block against 23 M for all the x
    // chunks -- the request count, not the bytes, was the cost: FETCH_SIZE is the same with and without the re-read)
    uint2 res[DS ? 1 : 4][DS ? 1 : 8];
    const int res_off = (r16 + 1) * ROW_B + (((2 * wave + (q >> 1)) ^ ((r16 + 1) & 7)) << 4) + (q & 1) * 8;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        if (kc & 1) store_chunk(1, stg1); else store_chunk(0, stg0);
        if (kc + 2 < KC) { if (kc & 1) load_chunk(kc + 2, stg1); else load_chunk(kc + 2, stg0); }
        __syncthreads();
        const char* sb = xs + (kc & 1) * XS_BYTES + r16 * ROW_B;
        // all 20 fragments of the chunk go in flight before the first MFMA (left alone, hipcc pairs each read with its MFMA and every MFMA
        // waits out an LDS round trip); the empty asm keeps the reads above it
        uint4 fx[2][10];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 10; ++i) fx[kk][i] = *(const uint4*)(sb + i * 16 * ROW_B + (kk ? sw1 : sw0));
        if constexpr (!DS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) res[kc][j] = *(const uint2*)(xs + (kc & 1) * XS_BYTES + (j + 1) * (16 * ROW_B) + res_off);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 10; ++i) acc1[i] = mfma_bf16(wf1[kc * 2 + kk], fx[kk][i], acc1[i]);
    }
    // weights of conv2 and the residual pixels go in flight now; they are consumed after the t1 hand-off
    uint4 wf2[18];
    {
        const char* wr = a.w2 + (wave * 18 * 64 + lane) * 16;
#pragma unroll
        for (int s = 0; s < 18; ++s) wf2[s] = *(const uint4*)(wr + s * 1024);
    }
    // From here on the tile is an 8 x 16 grid: pixel tile ty = output row ty, lane r16 = column tx (columns 14, 15 are never stored; they
    // read the two slack rows behind t1).  Every LDS row index is then 16 * (row of tiles) + r16 + shift, so the XOR swizzle depends on the
    // lane only and each fragment address is a per-lane constant plus an immediate.
    const int cb1 = (ch1 + q * 4) * 2;                    // byte offset of this lane's 4 channels in a 128-byte row
    {   // t1 rows: relu, zero outside the image, 4 channels = 8 bytes per lane and pixel tile
        const int wr_off = r16 * ROW_B + (((cb1 >> 4) ^ (r16 & 7)) << 4) + (cb1 & 8);
        const int w = w0 - 1 + r16;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int h = h0 - 1 + i;
            const bool in = (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            uint2 o = make_uint2(0, 0);
            if (in) o = relu_pack4(acc1[i]);
            *(uint2*)(t1 + i * (16 * ROW_B) + wr_off) = o;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 2: t2 = relu(W2 * t1 + b2)
    // fragment (halo row hr, column shift kw, kk) serves the output rows hr, hr - 1, hr - 2 (kh = 0, 1, 2): read once, up to three MFMAs.
    // Per accumulator the K order stays (kh, kw) ascending, kk inner.
    f32x4 acc2[8];
    {
        const f32x4 b4 = bias4(a.b2, ch1 + q * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[j] = b4;
    }
    int toff[3][2];       // byte offset of the fragment of column shift kw, half kk in a halo row
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) toff[kw][kk] = (r16 + kw) * ROW_B + (((kk * 4 + q) ^ ((r16 + kw) & 7)) << 4);
#pragma unroll
    for (int hr = 0; hr < 10; ++hr) {
        uint4 fr[3][2];                                    // the six fragments of a halo row in flight together
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fr[kw][kk] = *(const uint4*)(t1 + hr * (16 * ROW_B) + toff[kw][kk]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {                   // per accumulator the K order stays (kh, kw) ascending, kk inner
            const int ty = hr - kh;
            if (ty >= 0 && ty < 8) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) acc2[ty] = mfma_bf16(wf2[(kh * 3 + kw) * 2 + kk], fr[kw][kk], acc2[ty]);
            }
        }
    }
    uint4 wf3[4][2];
    uint4 wfd[DS ? 4 : 1][2];     // downsample block: the branch's fragments of this wave's four channel tiles, in flight with conv3's (loaded where
                                  // they are used, each of the four waited out an L2 round trip behind 16 MFMAs)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const char* wr = a.w3 + ((4 * i + wave) * 2 * 64 + lane) * 16;
        wf3[i][0] = *(const uint4*)wr;
        wf3[i][1] = *(const uint4*)(wr + 1024);
        if constexpr (DS) {
            const char* wd = a.wds + ((4 * i + wave) * 2 * 64 + lane) * 16;
            wfd[i][0] = *(const uint4*)wd;
            wfd[i][1] = *(const uint4*)(wd + 1024);
        }
    }
    {
        const int wr_off = r16 * ROW_B + (((cb1 >> 4) ^ (r16 & 7)) << 4) + (cb1 & 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint2 o = relu_pack4(acc2[j]);
            *(uint2*)(t2 + j * (16 * ROW_B) + wr_off) = o;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 3: y = relu(W3 . t2 + b3 + idt), two passes of 128 channels
    uint4 fx3[8][2];      // the t2 fragments of all 8 pixel tiles: read once, used by this wave's four channel tiles
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        fx3[j][0] = *(const uint4*)(t2 + j * (16 * ROW_B) + r16 * ROW_B + sw0);
        fx3[j][1] = *(const uint4*)(t2 + j * (16 * ROW_B) + r16 * ROW_B + sw1);
    }
    const int st_tx = tid >> 4, st_chunk = tid & 15;       // read-back: column st_tx of every row, 16 chunks of 16 bytes per pixel
    uint4 wfn[NEXT ? 2 : 1][NEXT ? 8 : 1];             // NEXT: this wave's fragments of the following conv (channel tiles 2 wave, 2 wave + 1)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        char* const ysp = NEXT ? xs + pass * YS_BYTES : ys;   // NEXT keeps both passes (the whole 256-channel tile) for phase 4
        if (pass && !NEXT) __syncthreads();              // the previous pass has been read back
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = pass * 2 + ii;
            const int ct = 4 * i + wave;                  // channel tile of y (16 channels)
            f32x4 acc3[8];
            {
                const f32x4 b4 = bias4(a.b3, ct * 16 + q * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc3[j] = b4;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc3[j] = mfma_bf16(wf3[i][kk], fx3[j][kk], acc3[j]);
            if constexpr (DS) {
                const f32x4 bd4 = bias4(a.bds, ct * 16 + q * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {              // the pixel itself in the halo tile: row ty + 1, column shift 1
                    f32x4 accd = bd4;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) accd = mfma_bf16(wfd[i][kk], *(const uint4*)(xs + (j + 1) * (16 * ROW_B) + toff[1][kk]), accd);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc3[j][e] += bf16_to_f32(f32_to_bf16(accd[e]));   // the stand-alone downsample launch stores bf16
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc3[j][0] += __uint_as_float(res[i][j].x << 16);
                    acc3[j][1] += __uint_as_float(res[i][j].x & 0xffff0000u);
                    acc3[j][2] += __uint_as_float(res[i][j].y << 16);
                    acc3[j][3] += __uint_as_float(res[i][j].y & 0xffff0000u);
                }
            }
            const int cbyte = ((ct - 8 * pass) * 16 + q * 4) * 2;     // within the pass's 256-byte pixel row
            const int wr_off = r16 * 256 + (((cbyte >> 4) ^ r16) << 4) + (cbyte & 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint2 o = relu_pack4(acc3[j]);
                *(uint2*)(ysp + j * (16 * 256) + wr_off) = o;
            }
        }
        if constexpr (NEXT) {
            if (pass == 1) {     // in flight across the barrier and the read-back
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) wfn[c][ks] = *(const uint4*)(a.wn + (((2 * wave + c) * 8 + ks) * 64 + lane) * 16);
            }
        }
        __syncthreads();
        {
            const int w = w0 + st_tx;
            const int rd_off = st_tx * 256 + ((st_chunk ^ st_tx) << 4);
            if constexpr (NEXT) {
                if (st_tx < TW && w < a.W && !(st_tx & 1)) {      // even pixels only (h0, w0 are even)
                    const uint4 v0 = *(const uint4*)(ysp + rd_off), v1 = *(const uint4*)(ysp + 2 * (16 * 256) + rd_off);
                    const uint4 v2 = *(const uint4*)(ysp + 4 * (16 * 256) + rd_off), v3 = *(const uint4*)(ysp + 6 * (16 * 256) + rd_off);
                    char* const yp = yimg + (unsigned)(((h0 >> 1) * a.W2 + (w >> 1)) * 512 + pass * 256 + st_chunk * 16);
                    const int rows = a.H - h0;                     // (> 0)
                    *(uint4*)yp = v0;
                    if (rows > 2) *(uint4*)(yp + a.W2 * 512) = v1;
                    if (rows > 4) *(uint4*)(yp + 2 * a.W2 * 512) = v2;
                    if (rows > 6) *(uint4*)(yp + 3 * a.W2 * 512) = v3;
                }
            } else if (st_tx < TW && w < a.W) {
                uint4 v[8];
#pragma unroll
                for (int ty = 0; ty < 8; ++ty) v[ty] = *(const uint4*)(ysp + ty * (16 * 256) + rd_off);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ty = 0; ty < 8; ++ty) {
                    if (h0 + ty < a.H) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        char* yp = yimg + (unsigned)((h0 + ty) * a.W * 512) + (unsigned)(w * 512 + pass * 256 + st_chunk * 16);
                        if (a.nt) __builtin_nontemporal_store((u32x4){v[ty].x, v[ty].y, v[ty].z, v[ty].w}, (u32x4*)yp);
                        else *(uint4*)yp = v[ty];
                    }
                }
            }
        }
    }
    if constexpr (NEXT) {
        // ------------------------------------------------------------ phase 4: y2 = relu(Wn . y + bn), 256 -> 128 channels, on the tile in LDS
        // wave w: channel tiles 2 w, 2 w + 1 for the 8 pixel tiles; K steps of 32 channels ascending = the stand-alone launch's chain
        f32x4 acc4[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 b4 = bias4(a.bn, (2 * wave + c) * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc4[c][j] = b4;
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            uint4 fy[8];
            const char* yb = xs + (ks >> 2) * YS_BYTES + r16 * 256 + ((((ks & 3) * 4 + q) ^ r16) << 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) fy[j] = *(const uint4*)(yb + j * (16 * 256));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc4[c][j] = mfma_bf16(wfn[c][ks], fy[j], acc4[c][j]);
        }
        __syncthreads();                                 // every wave has read the tile (and pass 1 has been read back)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int cbyte = ((2 * wave + c) * 16 + q * 4) * 2;
            const int wr_off = r16 * 256 + (((cbyte >> 4) ^ r16) << 4) + (cbyte & 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) *(uint2*)(xs + j * (16 * 256) + wr_off) = relu_pack4(acc4[c][j]);
        }
        __syncthreads();
        const int w = w0 + st_tx;
        if (st_tx < TW && w < a.W) {
            const int rd_off = st_tx * 256 + ((st_chunk ^ st_tx) << 4);
            char* const y2img = a.y2 + (long long)img * a.H * a.W * 256;
            uint4 v[8];
#pragma unroll
            for (int ty = 0; ty < 8; ++ty) v[ty] = *(const uint4*)(xs + ty * (16 * 256) + rd_off);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ty = 0; ty < 8; ++ty) {
                if (h0 + ty < a.H) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    char* yp = y2img + (unsigned)(((h0 + ty) * a.W + w) * 256 + st_chunk * 16);
                    if (a.nt) __builtin_nontemporal_store((u32x4){v[ty].x, v[ty].y, v[ty].z, v[ty].w}, (u32x4*)yp);
                    else *(uint4*)yp = v[ty];
                }
            }
        }
    }
}

// Forms of this kernel that were built, are bit-identical, and lost (1336 frames per block, identity blocks): a persistent one (one workgroup of
// 8 waves per CU walking tiles, the next tile's whole x halo fetched by LDS-DMA while the current tile computes, conv1 / conv2 weights resident in
// registers) 2.4-2.7 ms -- eight waves meeting at ten barriers per tile leave the CU idle where two independent workgroups fill each other's
// stalls, and at 256 registers every spill reload in front of a DMA or store costs a vmcnt(0); three workgroups per CU (40 KB of LDS: one x chunk
// buffer reused for t2, t1 reused as a 16 KB staging for four passes of 64 channels; 168 registers; 18 barriers per tile) 1.44 ms, exactly the
// two-workgroup time of that moment: the bound was the number of L2 requests (the residual's 8-byte re-read, see phase 1, and the weight
// fragments' 64-byte pieces, see the header), not occupancy.
template <int CIN, bool DS, bool NEXT = false>
int launch(const BneckK& a, hipStream_t stream) {
    constexpr int NXS = CIN > 64 ? 2 : 1;
    constexpr int LDS = NXS * XS_BYTES + T2_BYTES + (DS ? YS_BYTES : T1_BYTES);
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
    static_assert(!NEXT || LDS >= 2 * YS_BYTES, "both output passes stay in LDS");
    auto fn = bottleneck64_fused_kernel<CIN, DS, NEXT>;
    MT4_RAISE_LDS(fn);
    const long long nblk = (long long)a.B * a.tiles_h * a.tiles_w;
    hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(256), LDS, stream, a);
    return mt4_check_launch();
}

}  // namespace

// fragment-ordered copy of the block's packed weights: for each matrix, [channel tile of 16][K step of 32][lane (r16, q)] x 16 bytes =
// W[tile * 16 + r16][step * 32 + q * 8 .. + 8]
__global__ void bneck_pack_kernel(const u16* __restrict__ w, uint4* __restrict__ out, int ctiles, int ksteps, int row_elems) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ctiles * ksteps * 64) return;
    const int lane = i & 63, fs = i >> 6;
    const int step = fs % ksteps, tile = fs / ksteps;
    out[i] = *(const uint4*)(w + (long long)(tile * 16 + (lane & 15)) * row_elems + step * 32 + (lane >> 4) * 8);
}

static inline long long bneck_frag_bytes(int Cin, bool ds) { return ((long long)4 * (Cin / 32) + 4 * 18 + 16 * 2 + (ds ? 16 * (Cin / 32) : 0)) * 1024; }

extern "C" int64_t mt4_bottleneck_packed_bytes(int32_t Cin, int32_t has_downsample) {
    if (!((has_downsample && Cin == 64) || (!has_downsample && Cin == 256))) return MT4_EUNSUPPORTED;
    return bneck_frag_bytes(Cin, has_downsample != 0);
}

extern "C" int mt4_bottleneck_pack_bf16(const void* w1, const void* w2, const void* w3, const void* wds, int32_t Cin, void* out, void* stream) {
    mt4_clear_error();
    if (!w1 || !w2 || !w3 || !out) return MT4_EINVAL;
    if (!((wds && Cin == 64) || (!wds && Cin == 256))) return MT4_EUNSUPPORTED;
    if (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)wds | (uintptr_t)out) & 15) return MT4_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int k1 = Cin / 32;
    char* o = (char*)out;
    const int r1 = (int)mt4_conv_packed_k(Cin, 1, 1, MT4_BF16), r2 = (int)mt4_conv_packed_k(64, 3, 3, MT4_BF16), r3 = (int)mt4_conv_packed_k(64, 1, 1, MT4_BF16);
    auto go = [&](const void* w, int ctiles, int ksteps, int row_elems) {
        const int n = ctiles * ksteps * 64;
        hipLaunchKernelGGL(bneck_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const u16*)w, (uint4*)o, ctiles, ksteps, row_elems);
        o += (long long)n * 16;
    };
    go(w1, 4, k1, r1);
    go(w2, 4, 18, r2);
    go(w3, 16, 2, r3);
    if (wds) go(wds, 16, k1, r1);
    return mt4_check_launch();
}

extern "C" int mt4_bottleneck_fused_bf16(const void* x, void* y, const void* w_frag, const float* b1, const float* b2, const float* b3, const float* bds,
                                         int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t mid, void* stream) {
    mt4_clear_error();
    if (!x || !y || !w_frag || !b1 || !b2 || !b3 || B <= 0 || H <= 0 || W <= 0) return MT4_EINVAL;
    const bool ds = bds != nullptr;
    if (mid != 64 || !((ds && Cin == 64) || (!ds && Cin == 256))) return MT4_EUNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_frag | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)b3 | (uintptr_t)bds) & 15) return MT4_EALIGN;
    if ((long long)H * W * 512 > 0x7fffffffLL) return MT4_EUNSUPPORTED;    // per-image byte offsets are 32-bit
    BneckK a;
    a.x = (const char*)x; a.y = (char*)y;
    const int k1 = Cin / 32;
    a.w1 = (const char*)w_frag;
    a.w2 = a.w1 + (long long)4 * k1 * 1024;
    a.w3 = a.w2 + (long long)4 * 18 * 1024;
    a.wds = ds ? a.w3 + (long long)16 * 2 * 1024 : nullptr;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.bds = bds;
    a.B = B; a.H = H; a.W = W;
    a.tiles_h = cdiv(H, TH); a.tiles_w = cdiv(W, TW);
    a.nt = 1;
    if ((long long)B * a.tiles_h * a.tiles_w > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    return ds ? launch<64, true>(a, (hipStream_t)stream) : launch<256, false>(a, (hipStream_t)stream);
}

// the 1x1 conv behind a fused identity block (conv1 of the strided Bottleneck that follows layer1: 256 -> 128 channels) in fragment order
extern "C" int64_t mt4_bottleneck_next_packed_bytes(void) { return (int64_t)8 * 8 * 1024; }

extern "C" int mt4_bottleneck_pack_next_bf16(const void* w1_next, void* out, void* stream) {
    mt4_clear_error();
    if (!w1_next || !out) return MT4_EINVAL;
    if (((uintptr_t)w1_next | (uintptr_t)out) & 15) return MT4_EALIGN;
    const int n = 8 * 8 * 64;
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u16*)w1_next, (uint4*)out, 8, 8,
                       (int)mt4_conv_packed_k(256, 1, 1, MT4_BF16));
    return mt4_check_launch();
}

// mt4_bottleneck_fused_bf16 for an identity block (Cin = 256) that is followed by a strided Bottleneck: also runs that block's conv1 + bn1 +
// relu (w_next: mt4_bottleneck_pack_next_bf16, 256 -> 128 channels) on the result while it is in LDS.  y_even [B][(H + 1) / 2][(W + 1) / 2][256]
// receives the block's output at the even pixels (what the stride-2 downsample branch reads), t_next [B][H][W][128] the following conv's
// output; both bit-identical to the separate launches.
extern "C" int mt4_bottleneck_fused_next_bf16(const void* x, void* y_even, void* t_next, const void* w_frag, const float* b1, const float* b2,
                                              const float* b3, const void* w_next, const float* b_next, int32_t B, int32_t H, int32_t W, void* stream) {
    mt4_clear_error();
    if (!x || !y_even || !t_next || !w_frag || !b1 || !b2 || !b3 || !w_next || !b_next || B <= 0 || H <= 0 || W <= 0) return MT4_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y_even | (uintptr_t)t_next | (uintptr_t)w_frag | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)b3 | (uintptr_t)w_next |
         (uintptr_t)b_next) & 15)
        return MT4_EALIGN;
    if ((long long)H * W * 512 > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    BneckK a{};
    a.x = (const char*)x; a.y = (char*)y_even; a.y2 = (char*)t_next;
    a.w1 = (const char*)w_frag;
    a.w2 = a.w1 + (long long)4 * 8 * 1024;
    a.w3 = a.w2 + (long long)4 * 18 * 1024;
    a.wds = nullptr;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.bds = nullptr;
    a.wn = (const char*)w_next; a.bn = b_next;
    a.B = B; a.H = H; a.W = W; a.H2 = (H + 1) / 2; a.W2 = (W + 1) / 2;
    a.tiles_h = cdiv(H, TH); a.tiles_w = cdiv(W, TW);
    a.nt = 1;
    if ((long long)B * a.tiles_h * a.tiles_w > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    return launch<256, false, true>(a, (hipStream_t)stream);
}

// a packed bf16 matrix [rows][row_elems] (mt4_pack_conv_weight) in MFMA fragment order: [rows / 16][row_elems / 32][lane] x 16 bytes
// (mt4_conv_desc.fuse_expand reads the expansion weights this way)
extern "C" int mt4_pack_fragments_bf16(const void* w_packed, int32_t rows, int32_t row_elems, void* out, void* stream) {
    mt4_clear_error();
    if (!w_packed || !out || rows <= 0 || row_elems <= 0) return MT4_EINVAL;
    if ((rows % 16) || (row_elems % 32)) return MT4_EUNSUPPORTED;
    if (((uintptr_t)w_packed | (uintptr_t)out) & 15) return MT4_EALIGN;
    const long long n = (long long)(rows / 16) * (row_elems / 32) * 64;
    if (n > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)w_packed, (uint4*)out, rows / 16,
                       row_elems / 32, row_elems);
    return mt4_check_launch();
}
