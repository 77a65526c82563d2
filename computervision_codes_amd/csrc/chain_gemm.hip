// Two dependent 1x1 convolutions / linear layers in ONE launch, the map between them never in memory (bf16 operands, fp32 accumulate):
//
//   H  = act1(X . W1^T + b1 [+ R1])          [M][N1]      optionally stored (Y1)
//   Y2 = act2(H . W2^T + b2 [+ R2])          [M][N2]
//
// * ResNet-50 layer2 / layer3 (`Spatial_transformer/models/resnet.py:101-121`): conv3 + bn3 + add + ReLU of one Bottleneck (X = its conv2
//   output, R1 = the block input, Y1 = the block output, which the NEXT block still needs as its residual) followed by conv1 + bn1 + ReLU
//   of the next Bottleneck -- the N1-channel map is written once and never read back (layer3: 536 MB per pair and 1336 frames).
// * Swin MLP (`swin_transformer.py:15-31,267-269`): fc1 + GELU + fc2 + shortcut; the 4C-wide hidden map (1.2 GB per stage-0 block at batch 128)
//   never exists.
//
// A workgroup (8 waves, one per CU) owns 128 rows.  N1 is walked in chunks of 128 channels: GEMM1 of the chunk (K = K1, X tile resident in
// LDS), its epilogue in place on the chunk's LDS image (the residual rows arrive there by coalesced 16-byte loads one chunk ahead), then
// that image is the B operand of GEMM2's K-steps [128 c, 128 c + 128) into the running [N2][128 rows] accumulator ("K-chunk accumulation":
// unlike round 2's `fuse_w`, no tile ever has to hold all N1 channels).  Wave w owns 16 of the chunk's 128 channels in GEMM1 and N2 / 8
// output channels in GEMM2, over all 128 rows: every weight fragment is used by exactly one wave, so weights go straight from L2 into
// registers in fragment order (`mt4_pack_fragments_bf16`: 1 KB per wave-level load) -- no LDS copy, no barrier inside a GEMM -- prefetched one
// phase ahead.  Two barriers per chunk.
//
// Bit-identity with the two stand-alone launches of `mt4_conv_nhwc`: same MFMA (16x16x32 bf16, weights = A operand), accumulators start at
// the bias, K ascends in the same 32-element steps, the intermediate is rounded to bf16 exactly where the stand-alone launch stores it,
// residuals are added to the fp32 accumulator before the activation.
#include "mt4_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace {

struct ChainK {
    const char* x;            // [M][x_ld] bf16
    long long x_ld_bytes;
    const char* w1f;          // fragment order [N1 / 16][K1 / 32][64 lanes] x 16 B
    const float* b1;
    const char* r1;           // [M][N1] bf16 or NULL
    char* y1;                 // [M][N1] bf16 or NULL
    const char* w2f;          // fragment order [N2 / 16][N1 / 32][64 lanes] x 16 B
    const float* b2;
    const char* r2;           // [M][N2] bf16 or NULL
    char* y2;                 // [M][N2] bf16
    int M, K1, N1, N2, act1, act2;
};

constexpr int BM = 128;              // rows per workgroup
constexpr int CH = 128;              // channels of N1 per chunk = 2 K-steps of GEMM2
constexpr int PLANE = BM * 128;      // one 64-channel plane of a row tile: [row][8 x 16 B], chunk index XOR (row & 7)

__device__ __forceinline__ void act4(float (&v)[4], int a) {
    if (a == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (a == 2) {
        gelu_erf_n(v);
    }
}

// NT2: N2 = 128 * NT2 (16-channel tiles per wave in GEMM2); NKS1 = K1 / 64; CONV: the Bottleneck form (r1 and y1 given, act1 = ReLU), else the MLP
// form (no r1 / y1, act1 = GELU).  Everything a load is conditional on is a template parameter: a load behind a runtime condition makes hipcc
// branch around it and drain the queue (cdna_hip_programming.md, projection-GEMM trap (c))
// NCHUNK = N1 / 128: the chunk loop is fully unrolled -- as a loop, hipcc waits vmcnt(0) at its header for the fragments requested an iteration
// ago, which also drains the residual rows requested for the NEXT chunk and the stores of the previous one: every chunk then began with a full
// memory round trip on all eight waves (first version: 65 us per 128-row tile against 14 us of MFMA time).  Straight-line code gets counted waits.
template <int NT2, int NKS1, bool CONV, int NCHUNK>
__global__ __launch_bounds__(512, 2) void chain_gemm_kernel(const ChainK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM;
    constexpr int nks1 = NKS1;                    // 64-channel planes of X = K-steps of GEMM1
    constexpr int ACT1 = CONV ? 1 : 2;
    constexpr int nchunk = NCHUNK;
    const int mlast = a.M - 1;                    // rows past the end load the last row (never stored)
    char* const Xs = smem;
    char* const Hb = smem + nks1 * PLANE;         // two chunk images of 2 planes each
    float* const B1s = (float*)(Hb + 4 * PLANE);  // b1, all N1 values (a load per chunk in front of its first MFMA would drain the queue)
    const long long n1b = (long long)a.N1 * 2, n2b = (long long)a.N2 * 2;

    // ---- the X tile: 16-byte pieces, coalesced (8 lanes = one 128-byte row segment), swizzle on the LDS side
    {
        uint4 xv[NKS1 * 2];
#pragma unroll
        for (int i = 0; i < NKS1 * 2; ++i) {
            const int pid = tid + i * 512;
            const int plane = pid >> 10, row = (pid >> 3) & 127, ck = pid & 7;
            xv[i] = *(const uint4*)(a.x + (long long)min(m0 + row, mlast) * a.x_ld_bytes + plane * 128 + ck * 16);
        }
#pragma unroll
        for (int i = 0; i < NKS1 * 2; ++i) {
            const int pid = tid + i * 512;
            const int plane = pid >> 10, row = (pid >> 3) & 127, ck = pid & 7;
            *(uint4*)(Xs + plane * PLANE + row * 128 + ((ck ^ (row & 7)) << 4)) = xv[i];
        }
    }
    // residual pieces of a chunk: thread t holds pieces t, t + 512, t + 1024, t + 1536 of the [2 planes][128 rows][8 chunks] image (by value:
    // an array captured by reference in these helpers stayed in scratch memory)
    struct R4 { uint4 v0, v1, v2, v3; };
    R4 rr;
    auto load_r = [&](int c) __attribute__((always_inline)) -> R4 {
        R4 r;
        const char* base = a.r1 + (long long)(c * CH) * 2;
        auto piece = [&](int i) __attribute__((always_inline)) -> uint4 {
            const int pid = tid + i * 512;
            const int plane = pid >> 10, row = (pid >> 3) & 127, ck = pid & 7;
            return *(const uint4*)(base + (long long)min(m0 + row, mlast) * n1b + plane * 128 + ck * 16);
        };
        r.v0 = piece(0); r.v1 = piece(1); r.v2 = piece(2); r.v3 = piece(3);
        return r;
    };
    auto put_r = [&](int c, const R4& r) __attribute__((always_inline)) {
        char* hb = Hb + (c & 1) * (2 * PLANE);
        auto dst = [&](int i) __attribute__((always_inline)) -> uint4* {
            const int pid = tid + i * 512;
            const int plane = pid >> 10, row = (pid >> 3) & 127, ck = pid & 7;
            return (uint4*)(hb + plane * PLANE + row * 128 + ((ck ^ (row & 7)) << 4));
        };
        *dst(0) = r.v0; *dst(1) = r.v1; *dst(2) = r.v2; *dst(3) = r.v3;
    };
    for (int i = tid; i < nchunk * (CH / 4); i += 512) *(float4*)(B1s + i * 4) = *(const float4*)(a.b1 + i * 4);
    if constexpr (CONV) {
        put_r(0, load_r(0));
        rr = load_r(min(1, nchunk - 1));
    }

    // ---- weight fragments: wave w, chunk c -> GEMM1 channel tile c * 8 + w; GEMM2 channel tiles w * NT2 + t, K-steps 4 c .. 4 c + 3
    const int f1_stride = (a.K1 >> 5) * 1024;                 // bytes per channel tile of w1f
    const int f2_stride = (a.N1 >> 5) * 1024;
    const char* const w1p = a.w1f + (long long)wave * f1_stride + lane * 16;
    const char* const w2p = a.w2f + (long long)(wave * NT2) * f2_stride + lane * 16;
    uint4 fa1[2 * NKS1];                                       // GEMM1 fragments of the chunk about to run
    auto load_w1 = [&](int c) __attribute__((always_inline)) {
        const char* p = w1p + (long long)c * 8 * f1_stride;
#pragma unroll
        for (int s = 0; s < 2 * NKS1; ++s) fa1[s] = *(const uint4*)(p + s * 1024);
    };
    load_w1(0);

    f32x4 acc2[NT2][8];
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const float4 b = *(const float4*)(a.b2 + (wave * NT2 + t) * 16 + q * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[t][j] = (f32x4){b.x, b.y, b.z, b.w};
    }
    __syncthreads();   // X tile (and R(0)) in LDS

    // B-operand fragment of row tile j, K-half kk of a plane: rows j * 16 + r16, 16-byte chunk kk * 4 + q
    const int frag_off = r16 * 128;
    const int sw0 = ((0 * 4 + q) ^ (r16 & 7)) << 4, sw1 = ((1 * 4 + q) ^ (r16 & 7)) << 4;   // (j * 16 + r16) & 7 == r16 & 7

#pragma unroll
    for (int c = 0; c < nchunk; ++c) {
        char* const hb = Hb + (c & 1) * (2 * PLANE);
        // ---------------- GEMM1: acc1[channel 16 (c * 8 + wave) + 4 q + e][row j * 16 + r16]
        f32x4 acc1[8];
        {
            const float4 b = *(const float4*)(B1s + (c * 8 + wave) * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc1[j] = (f32x4){b.x, b.y, b.z, b.w};
        }
        uint4 fa2[NT2][4];
#pragma unroll
        for (int ks = 0; ks < NKS1; ++ks) {
            const char* xp = Xs + ks * PLANE + frag_off;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t A = __builtin_bit_cast(bf16x8_t, fa1[ks * 2 + kk]);
#pragma unroll
                for (int jh = 0; jh < 8; jh += 4) {       // row-tile fragments four at a time (all eight: 16 more live registers at the kernel's peak)
                    uint4 bx[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bx[j] = *(const uint4*)(xp + (jh + j) * (16 * 128) + (kk ? sw1 : sw0));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc1[jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8_t, bx[j]), acc1[jh + j], 0, 0, 0);
                }
                if (ks == 0 && kk == 0) {
                    // GEMM2's fragments of this chunk: requested behind the first use of fa1 (in front of it the wait for fa1 -- requested a
                    // phase ago -- would drain these too), used after epilogue 1
#pragma unroll
                    for (int t = 0; t < NT2; ++t)
#pragma unroll
                        for (int s = 0; s < 4; ++s) fa2[t][s] = *(const uint4*)(w2p + (long long)t * f2_stride + (c * 4 + s) * 1024);
                }
            }
        }
        load_w1(min(c + 1, nchunk - 1));             // next chunk's GEMM1 fragments: in flight across epilogue 1 and GEMM2 (the last chunk
                                                     // re-requests its own: an unconditional load keeps the queue counted, not drained)
        __syncthreads();                             // Ba: R(c) is in hb (all threads); every wave is done with GEMM2(c - 1) on the other image
        // ---------------- epilogue 1, in place: this wave's 16 channels of all 128 rows
        {
            const int cl = wave * 16 + q * 4;        // channel within the chunk
            const int plane = cl >> 6, cb = (cl & 63) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = j * 16 + r16;
                char* hp = hb + plane * PLANE + row * 128 + (((cb >> 4) ^ (row & 7)) << 4) + (cb & 8);
                float v[4] = {acc1[j][0], acc1[j][1], acc1[j][2], acc1[j][3]};
                if constexpr (CONV) {
                    const uint2 r = *(const uint2*)hp;
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                }
                act4(v, ACT1);
                *(uint2*)hp = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        if constexpr (CONV) {
            put_r(c + 1, rr);                        // the other image is free since Ba (past the last chunk: a spare copy nobody reads)
            rr = load_r(min(c + 2, nchunk - 1));
        }
        __syncthreads();                             // Bb: H(c) complete
        // ---------------- the intermediate map to memory (conv case): whole 128-byte row segments per 8 lanes
        if constexpr (CONV) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pid = tid + i * 512;
                const int plane = pid >> 10, row = (pid >> 3) & 127, ck = pid & 7;
                if (m0 + row < a.M)
                    *(uint4*)(a.y1 + (long long)(m0 + row) * n1b + (c * CH + plane * 64) * 2 + ck * 16) =
                        *(const uint4*)(hb + plane * PLANE + row * 128 + ((ck ^ (row & 7)) << 4));
            }
        }
        // ---------------- GEMM2, K-steps 4 c .. 4 c + 3 (two planes x two halves)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const char* hp = hb + p * PLANE + frag_off;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 bx[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bx[j] = *(const uint4*)(hp + j * (16 * 128) + (kk ? sw1 : sw0));
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    const bf16x8_t A = __builtin_bit_cast(bf16x8_t, fa2[t][p * 2 + kk]);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc2[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(bf16x8_t, bx[j]), acc2[t][j], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();   // every wave is done reading the last chunk image: LDS becomes the output tile

    // ---- epilogue 2: the shortcut (MLP form) in the accumulator layout -- 8-byte loads, all requested before the first use; 128 x N2 values
    // against the 128 x N1 of the chunk residuals -- activation, bf16 tile [128 rows][N2] in LDS (row pitch N2 * 2 + 16 bytes), then whole rows
    // to memory
    constexpr int ACT2 = CONV ? 1 : 0;
    const int pitch = a.N2 * 2 + 16;
    uint2 r2v[NT2][8];
    if constexpr (!CONV) {
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r2v[t][j] = *(const uint2*)(a.r2 + (long long)min(m0 + j * 16 + r16, mlast) * n2b + ((wave * NT2 + t) * 16 + q * 4) * 2);
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int n = (wave * NT2 + t) * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = j * 16 + r16;
            float v[4] = {acc2[t][j][0], acc2[t][j][1], acc2[t][j][2], acc2[t][j][3]};
            if constexpr (!CONV) {
                const uint2 r = r2v[t][j];
                v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
            }
            act4(v, ACT2);
            *(uint2*)(smem + row * pitch + n * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
    __syncthreads();
    const int vpr = a.N2 >> 3;                       // 16-byte vectors per row
    for (int i = tid; i < BM * vpr; i += 512) {
        const int row = i / vpr, ck = i - row * vpr;
        if (m0 + row < a.M) *(uint4*)(a.y2 + (long long)(m0 + row) * n2b + ck * 16) = *(const uint4*)(smem + row * pitch + ck * 16);
    }
}

}  // namespace

// H = act1(x . w1^T + b1 [+ r1]) (stored to y1 when given), y2 = act2(H . w2^T + b2 [+ r2]); act: 0 none, 1 ReLU, 2 GELU(erf).
// x [M][x_ld] bf16 (x_ld >= K1 elements); w1_frag / w2_frag = mt4_pack_fragments_bf16 of the packed [N1][K1] / [N2][N1] matrices.
extern "C" int mt4_chain_gemm_bf16(const void* x, int64_t x_ld, int64_t M, int32_t K1, const void* w1_frag, const float* b1, int32_t N1, const void* r1,
                                   void* y1, int32_t act1, const void* w2_frag, const float* b2, int32_t N2, const void* r2, int32_t act2, void* y2,
                                   void* stream) {
    mt4_clear_error();
    if (!x || !w1_frag || !b1 || !w2_frag || !b2 || !y2 || M <= 0 || x_ld < K1) return MT4_EINVAL;
    if (K1 <= 0 || K1 > 256 || (K1 & 63) || N1 < 2 * CH || (N1 % CH) || (N2 != 128 && N2 != 256)) return MT4_EUNSUPPORTED;
    if (act1 < 0 || act1 > 2 || act2 < 0 || act2 > 2) return MT4_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w1_frag | (uintptr_t)w2_frag | (uintptr_t)r1 | (uintptr_t)y1 | (uintptr_t)r2 | (uintptr_t)y2 | (uintptr_t)b1 | (uintptr_t)b2) & 15) ||
        (x_ld & 7))
        return MT4_EALIGN;
    if (M > (int64_t)0x7fffffff - BM) return MT4_EUNSUPPORTED;
    ChainK k;
    k.x = (const char*)x; k.x_ld_bytes = x_ld * 2; k.w1f = (const char*)w1_frag; k.b1 = b1; k.r1 = (const char*)r1; k.y1 = (char*)y1;
    k.w2f = (const char*)w2_frag; k.b2 = b2; k.r2 = (const char*)r2; k.y2 = (char*)y2;
    k.M = (int)M; k.K1 = K1; k.N1 = N1; k.N2 = N2; k.act1 = act1; k.act2 = act2;
    const int xs = (K1 / 64) * PLANE;
    const int out = BM * (N2 * 2 + 16);
    const int lds = xs + 4 * PLANE + N1 * 4 > out ? xs + 4 * PLANE + N1 * 4 : out;
    const int grid = (int)((M + BM - 1) / BM);
    const bool conv = r1 != nullptr;
    // Bottleneck form: r1 + y1, ReLU after both convs, no r2; MLP form: GELU between, shortcut r2, no activation behind
    if (conv != (y1 != nullptr) || act1 != (conv ? 1 : 2) || act2 != (conv ? 1 : 0) || conv == (r2 != nullptr)) return MT4_EUNSUPPORTED;
    if (K1 != 128 && K1 != 256) return MT4_EUNSUPPORTED;
#define MT4_CHAIN_LAUNCH(NT2_, NKS1_, CONV_, NCH_)                                                 \
    do {                                                                                       \
        auto fn = chain_gemm_kernel<NT2_, NKS1_, CONV_, NCH_>;                                 \
        MT4_RAISE_LDS(fn);                                                                     \
        hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, (hipStream_t)stream, k);            \
    } while (0)
    const int nch = N1 / CH;
    if (nch != 4 && nch != 8) return MT4_EUNSUPPORTED;
    if (conv) {
        if (N2 == 256 && K1 == 256 && nch == 8) MT4_CHAIN_LAUNCH(2, 4, true, 8);          // ResNet-50 layer3: 256 -> 1024 -> 256
        else if (N2 == 256 && K1 == 128 && nch == 4) MT4_CHAIN_LAUNCH(2, 2, true, 4);     // layer2.3 -> layer3.0: 128 -> 512 -> 256
        else if (N2 == 128 && K1 == 128 && nch == 4) MT4_CHAIN_LAUNCH(1, 2, true, 4);     // layer2: 128 -> 512 -> 128
        else return MT4_EUNSUPPORTED;
    } else {
        if (N2 == 256 && K1 == 256 && nch == 8) MT4_CHAIN_LAUNCH(2, 4, false, 8);         // Swin-B stage 1: C = 256
        else if (N2 == 128 && K1 == 128 && nch == 4) MT4_CHAIN_LAUNCH(1, 2, false, 4);    // Swin-B stage 0: C = 128
        else return MT4_EUNSUPPORTED;
    }
#undef MT4_CHAIN_LAUNCH
    return mt4_check_launch();
}
