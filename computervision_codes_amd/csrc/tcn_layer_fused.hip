// One DilatedResidualLayer of the temporal head in ONE launch (`Temporal_tenco/network.py:186-198`): y = x + W2 . relu(W1 * x + b1) + b2 with
// W1 a k = 3 convolution of dilation d over the frames of one video (padding d: frames outside the video are zeros), W2 a 1 x 1 convolution.
// bf16 operands, fp32 accumulation: the THROUGHPUT mode of the head (several videos per forward) -- the latency path of one short video
// keeps its two launches per layer (a layer's 512 hidden channels would have to cross workgroups there).
//
// A workgroup (8 waves, one per CU) owns 64 consecutive frames of one video and ALL 512 channels, so h = relu(...) never leaves LDS:
//   GEMM1  [64 frames] x [512 ch], K = 3 taps x 512: the frame rows of a K-step (tap, 64-channel slice) are staged global -> registers -> LDS
//          one step ahead (two stages, one barrier per step); wave w owns channels 64 w .. 64 w + 63 over all 64 frames; its weight fragments come
//          straight from L2 into registers in fragment order (mt4_pack_fragments_bf16: 1 KB per wave-level load), one step ahead;
//   h      ReLU, rounded to bf16 exactly where the stand-alone launch stores it, written to an LDS tile [8 slices][64 frames][128 B];
//   GEMM2  K = 512 over that tile, same wave -> channel assignment; + b2 + x (the residual, fp32 add) -> bf16, through LDS, whole rows out.
// K is walked in the generic kernel's order (channel slice outer, taps inner, 32-element MFMA steps ascending; accumulators start at the bias):
// bit-identical to the two mt4_conv_nhwc launches.  Tiles never cross a video: a video's rows do not depend on what rides along.
// What bounds it: the 2 MB of weights a tile pulls through L2 -> CU (~29 us at the ~70 GB/s a CU takes in) against 15.6 us of MFMA time.
#include "mt4_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace {

struct TcnLayerK {
    const char* x;      // [B][T][512] bf16
    const char* w1f;    // fragment order of the packed [512][3 * 512] matrix (K = tap * 512 + channel)
    const float* b1;
    const char* w2f;    // fragment order of the packed [512][512] matrix
    const float* b2;
    char* y;            // [B][T][512] bf16
    int B, T, d, tiles_per_video;
};

constexpr int BM = 64;                 // frames per workgroup
constexpr int C = 512;
constexpr int NSL = C / 64;            // 64-channel slices = K-steps per tap
constexpr int STAGE = BM * 128;        // one K-step of frame rows: [64][128 B], 16-byte chunk index XOR (row & 7)
constexpr int HBYTES = NSL * STAGE;    // the h tile
constexpr int OUT_PITCH = C * 2 + 16;  // bf16 output tile row pitch

__global__ __launch_bounds__(512) void tcn_layer_fused_kernel(const TcnLayerK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;                       // 2 stages
    char* const Hs = smem + 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int vid = blockIdx.x / a.tiles_per_video;
    const int t0 = (blockIdx.x - vid * a.tiles_per_video) * BM;
    const char* const xb = a.x + (long long)vid * a.T * (C * 2);
    char* const yb = a.y + (long long)vid * a.T * (C * 2);

    // staging: thread -> (frame row, 16-byte chunk) of a K-step
    const int srow = tid >> 3, sck = tid & 7;
    const int st_off = srow * 128 + ((sck ^ (srow & 7)) << 4);
    auto load_x = [&](int step) __attribute__((always_inline)) -> uint4 {      // step = slice * 3 + tap
        const int sl = step / 3, tap = step - sl * 3;
        const int t = t0 + srow + (tap - 1) * a.d;
        const bool ok = (unsigned)t < (unsigned)a.T;
        const uint4 v = *(const uint4*)(xb + (long long)(ok ? t : 0) * (C * 2) + sl * 128 + sck * 16);
        return ok ? v : make_uint4(0u, 0u, 0u, 0u);
    };
    // weight fragments: channel tile (wave * 4 + n), 32-element K block kq: ((tile * KQ) + kq) * 1024 + lane * 16
    const char* const w1p = a.w1f + (long long)(wave * 4) * (48 * 1024) + lane * 16;
    const char* const w2p = a.w2f + (long long)(wave * 4) * (16 * 1024) + lane * 16;
    struct W8 { uint4 v[8]; };
    auto load_w1 = [&](int step) __attribute__((always_inline)) -> W8 {
        const int sl = step / 3, tap = step - sl * 3;
        const int kq = (tap * NSL + sl) * 2;                 // the packed K index is tap-major
        W8 w;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w.v[n * 2 + kk] = *(const uint4*)(w1p + (long long)n * (48 * 1024) + (kq + kk) * 1024);
        return w;
    };
    auto load_w2 = [&](int sl) __attribute__((always_inline)) -> W8 {
        W8 w;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w.v[n * 2 + kk] = *(const uint4*)(w2p + (long long)n * (16 * 1024) + (sl * 2 + kk) * 1024);
        return w;
    };

    f32x4 acc[4][4];                                          // [channel tile n][frame tile m]: channels 64 wave + 16 n + 4 q + e, frame 16 m + r16
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const float4 b = *(const float4*)(a.b1 + wave * 64 + n * 16 + q * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = (f32x4){b.x, b.y, b.z, b.w};
    }
    const int frag_off = r16 * 128;
    const int sw0 = ((0 * 4 + q) ^ (r16 & 7)) << 4, sw1 = ((1 * 4 + q) ^ (r16 & 7)) << 4;
    auto gemm_step = [&](const char* stage, const W8& w) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 bx[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) bx[m] = *(const uint4*)(stage + frag_off + m * (16 * 128) + (kk ? sw1 : sw0));
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bf16x8_t A = __builtin_bit_cast(bf16x8_t, w.v[n * 2 + kk]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8_t, bx[m]), acc[n][m], 0, 0, 0);
            }
        }
    };

    // ---------------- GEMM1: 24 K-steps, frame rows one step ahead through two LDS stages, weights one step ahead in registers
    constexpr int NST1 = 3 * NSL;
    uint4 xv = load_x(0);
    *(uint4*)(ring + st_off) = xv;
    xv = load_x(1);
    W8 wa = load_w1(0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NST1; ++s) {
        if (s + 1 < NST1) *(uint4*)(ring + ((s + 1) & 1) * STAGE + st_off) = xv;      // (stage (s + 1) & 1 was last read in step s - 1: behind a barrier)
        if (s + 2 < NST1) xv = load_x(s + 2);
        const W8 wb = load_w1(s + 1 < NST1 ? s + 1 : s);                             // (unconditional: keeps the queue counted)
        gemm_step(ring + (s & 1) * STAGE, wa);
        wa = wb;
        __syncthreads();
    }

    // ---------------- h = relu(.) -> bf16 tile; the residual pieces of the output (accumulator layout) go in flight behind it
    {
        const int cb0 = q * 8;                               // byte offset of channels 4 q .. 4 q + 3 inside a 16-channel tile
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int row = m * 16 + r16;
                const int cb = n * 32 + cb0;                 // byte offset inside the wave's 64-channel slice
                char* hp = Hs + wave * STAGE + row * 128 + (((cb >> 4) ^ (row & 7)) << 4) + (cb & 8);
                *(uint2*)hp = make_uint2(pack_bf16x2(fmaxf(acc[n][m][0], 0.f), fmaxf(acc[n][m][1], 0.f)),
                                         pack_bf16x2(fmaxf(acc[n][m][2], 0.f), fmaxf(acc[n][m][3], 0.f)));
            }
    }
    uint2 rv[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int t = min(t0 + m * 16 + r16, a.T - 1);
            rv[n][m] = *(const uint2*)(xb + (long long)t * (C * 2) + (wave * 64 + n * 16 + q * 4) * 2);
        }
    wa = load_w2(0);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const float4 b = *(const float4*)(a.b2 + wave * 64 + n * 16 + q * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = (f32x4){b.x, b.y, b.z, b.w};
    }
    __syncthreads();   // the h tile is complete

    // ---------------- GEMM2: K = 512 over the h tile (no barrier inside)
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const W8 wb = load_w2(sl + 1 < NSL ? sl + 1 : sl);
        gemm_step(Hs + sl * STAGE, wa);
        wa = wb;
    }
    __syncthreads();   // every wave is done with the h tile: LDS becomes the output tile

    // ---------------- + x (fp32 add), bf16, [64 frames][512 ch] through LDS, whole rows to memory
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint2 r = rv[n][m];
            const float v0 = acc[n][m][0] + __uint_as_float(r.x << 16), v1 = acc[n][m][1] + __uint_as_float(r.x & 0xffff0000u);
            const float v2 = acc[n][m][2] + __uint_as_float(r.y << 16), v3 = acc[n][m][3] + __uint_as_float(r.y & 0xffff0000u);
            *(uint2*)(smem + (m * 16 + r16) * OUT_PITCH + (wave * 64 + n * 16 + q * 4) * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (BM * (C / 8)) / 512; ++i) {
        const int e = tid + i * 512;
        const int row = e >> 6, ck = e & 63;
        if (t0 + row < a.T) *(uint4*)(yb + (long long)(t0 + row) * (C * 2) + ck * 16) = *(const uint4*)(smem + row * OUT_PITCH + ck * 16);
    }
}

}  // namespace

extern "C" int mt4_tcn_layer_fused_bf16(const void* x, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2, void* y, int32_t B,
                                        int32_t T, int32_t C_, int32_t dilation, void* stream) {
    mt4_clear_error();
    if (!x || !w1_frag || !b1 || !w2_frag || !b2 || !y || B <= 0 || T <= 0 || dilation <= 0) return MT4_EINVAL;
    if (C_ != C) return MT4_EUNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w1_frag | (uintptr_t)w2_frag | (uintptr_t)y | (uintptr_t)b1 | (uintptr_t)b2) & 15) return MT4_EALIGN;
    if (x == y) return MT4_EINVAL;                                   // (a tile reads frames d away that another tile writes)
    TcnLayerK k;
    k.x = (const char*)x; k.w1f = (const char*)w1_frag; k.b1 = b1; k.w2f = (const char*)w2_frag; k.b2 = b2; k.y = (char*)y;
    k.B = B; k.T = T; k.d = dilation; k.tiles_per_video = (T + BM - 1) / BM;
    const long long grid = (long long)B * k.tiles_per_video;
    if (grid > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    const int lds = 2 * STAGE + HBYTES;                              // 80 KB (>= the 64 x 1040-byte output tile)
    static_assert(2 * STAGE + HBYTES >= BM * OUT_PITCH, "the output tile overlays ring + h");
    auto fn = tcn_layer_fused_kernel;
    MT4_RAISE_LDS(fn);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, k);
    return mt4_check_launch();
}
