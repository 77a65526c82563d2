// Temporal-head convolutions for ONE short video on gfx950 (MI355X): Conv1d k in {1, 3}, any dilation, stride 1, 'same' padding,
// frame-major rows [T][C] -- the DilatedResidualLayer of `Temporal_tenco/network.py:186-198` and the 1x1 projections / heads around it
// (`network.py:21-24,96,113,129`).
//
// Why a second kernel beside igemm_conv_kernel: a layer over a 256-frame video is a [256 x 1536] x [1536 x 512] GEMM.  Whatever the
// tiling, a workgroup must pull (rows + channels of its tile) x K operand bytes through L2 -> CU at ~40-70 GB/s per CU, and with
// 32 x 32 tiles on 128 workgroups that is 384 KB per workgroup (7-9 us per launch, 84 dependent launches per video).  This kernel
// cuts the bytes a CU takes in and takes every barrier out of the K loop:
//   * 32 frames x 16 channels per workgroup -> 256 workgroups for T = 256, one per CU; the n-tile picks the XCD (blockIdx & 7), so an
//     XCD's L2 holds 1/8 of the layer's weights and the frame tiles sharing a weight slice hit it there.
//   * comb tiles: the 32 frames of a tile are {t0 + a + j*d : a < A, j < J}, A*J = 32.  Tap k of output slot i = j*A + a is input slot
//     i + k*A of the (J+2)*A input rows {t0 + a + (j'-1)*d}: the three taps read ONE staged copy of the rows at a uniform shift, so
//     the activation bytes are (J+2)/J x the tile instead of 3x (for T = 256 and d >= 8 the combs span the video: no halo at all).
//   * the K loop is split over the 8 waves by 128-byte channel slice; a wave stages ITS slices' rows by LDS-DMA into a private ring
//     (no other wave reads them: the wave's own counted vmcnt is the only synchronisation) and loads ITS weight fragments straight
//     from global memory into registers (each weight element is used by exactly one lane: LDS would only add a copy).
//   * the 8 partial tiles meet in LDS once, are added in the fixed order wave 0..7 (deterministic) and two waves run the fused
//     epilogue (bias, residual, ReLU).
// Arithmetic: fp32 = v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), bf16 = v_mfma_f32_16x16x32_bf16 with fp32 accumulate.
#include "mt4_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

namespace {

struct TcnK {
    const char* x;
    const char* w;
    const float* bias;
    const char* res;
    char* y;
    int T, B, Cin, Cout;
    int tap0;                 // first tap of the packed weight row this launch uses (centre-only launches of a 3-tap weight: 1)
    int d;                    // dilation
    int A, logA, tps, S, TT;  // comb: A residues x J = 32/A chain positions; tps = d/A tiles per super-block of S = J*d frames; TT tiles per sequence
    int slots, np;            // input slots (J + taps - 1) * A and LDS-DMA pieces ceil(slots / 8)
    int SPT;                  // 128-byte K-steps per tap
    int w_row_bytes;
    int n_tiles, nnx;         // channel tiles of 16; channel-tile rounds per XCD slot, ceil(n_tiles / 8)
    unsigned tps_magic, nnx_magic;   // floor(2^32 / v) + 1: q = mulhi(n, magic) is n / v for n, v < 2^16
    int relu;
    unsigned x_bytes;
    // LN (mt4_tcn_linear_ln_f32): LayerNorm over the Cin channels of every frame folded into the GEMM -- w holds gamma o W, bias W . beta + b,
    // ln_colsum[n] = sum_k (gamma o W)[n][k]; the frame's statistics arrive as partial sums written by the launch that produced x (its `stat_out`):
    // stat_in [stat_parts][B * T] x (sum, sum of squares) over 16 channels each, added here in a fixed order.
    // stat_out (any 1-tap fp32 launch, Cout % 16 == 0): this launch's partials of ITS output
    const float* ln_colsum;
    float ln_eps, ln_inv_k;
    const float2* stat_in;
    float2* stat_out;
    int stat_parts;
};

__device__ __forceinline__ v4u tcn_make_srd(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    v4u r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)u);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// one LDS-DMA piece: lane l fetches 16 B at buffer offset voff (range-checked: out-of-range lanes write ZEROS) + soff (wave-uniform)
// into LDS [lds_addr + 16 l).  M0 is compiler-reserved: saved, set and restored inside the statement.
__device__ __forceinline__ void tcn_dma16(v4u srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void tcn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kTcnBM = 32, kTcnBN = 16, kTcnWaves = 8, kTcnRing = 2, kTcnMaxPieces = 8;

constexpr int kTcnMaxStatParts = 56;     // channels / 16 of the widest folded LayerNorm (MS-TCT: 864 / 16 = 54)

// LN: LayerNorm of the input rows folded in, their statistics from `stat_in`
template <typename T, int TAPS, bool OUT_F32, bool LN = false>
__global__ __launch_bounds__(kTcnWaves * 64) void tcn_conv_kernel(const TcnK a) {
    static_assert(!LN || (sizeof(T) == 4 && TAPS == 1), "the LayerNorm fold is the fp32 nn.Linear form");
    constexpr int ES = (int)sizeof(T);
    constexpr int NW = kTcnWaves;
    constexpr int HP = (TAPS - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // every kernel argument is requested NOW (one scalar-memory round trip): left to itself hipcc fetches them in 3-4 dependent rounds of
    // s_load + s_waitcnt, each a trip to L2 on the cold scalar cache of a fresh launch (~0.2 us apiece on an 8 us kernel)
    asm volatile("" ::"s"(a.x), "s"(a.w), "s"(a.bias), "s"(a.res), "s"(a.y), "s"(a.T), "s"(a.B), "s"(a.Cin), "s"(a.Cout), "s"(a.tap0), "s"(a.d),
                 "s"(a.A), "s"(a.logA), "s"(a.tps));
    asm volatile("" ::"s"(a.S), "s"(a.TT), "s"(a.slots), "s"(a.np), "s"(a.SPT), "s"(a.w_row_bytes), "s"(a.n_tiles), "s"(a.nnx), "s"(a.relu),
                 "s"(a.x_bytes), "s"(a.tps_magic), "s"(a.nnx_magic));
    if constexpr (LN) asm volatile("" ::"s"(a.ln_colsum), "s"(a.ln_eps), "s"(a.ln_inv_k), "s"(a.stat_in), "s"(a.stat_parts));
    if constexpr (sizeof(T) == 4 && TAPS == 1) asm volatile("" ::"s"(a.stat_out));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;

    // grid (8, TT, B * nnx): linear block id = x + 8 (y + TT z), and blocks b and b + 8 share an XCD (observed round-robin dispatch; speed
    // only), so the channel tile follows x: all frame tiles of a channel tile read their weight slice from ONE L2, and an XCD's L2 holds
    // 1/8 of the layer's weights
    const int lt = blockIdx.y;                                   // frame tile within its sequence
    const int seq = a.nnx == 1 ? (int)blockIdx.z : (int)__umulhi(blockIdx.z, a.nnx_magic);   // (magic 2^32 + 1 does not fit for a divisor of 1)
    const int n_tile = ((int)blockIdx.z - seq * a.nnx) * 8 + (int)blockIdx.x;
    if (n_tile >= a.n_tiles) return;
    const int n0 = n_tile * kTcnBN;
    const int sup = a.tps == 1 ? lt : (int)__umulhi((unsigned)lt, a.tps_magic), rr = lt - sup * a.tps;
    const int t0 = sup * a.S + rr * a.A;
    const int amask = a.A - 1;

    const int np = a.np;                        // wave-uniform (kernel argument)
    const int region = np * 1024;               // bytes of one staged channel slice
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned my_lds = lds_base + wave * (kTcnRing * region);
    char* const my_smem = smem + wave * (kTcnRing * region);

    // source offset of every LDS position this lane fills: piece p, position (row s = 8p + lane/8, chunk c' = lane & 7) holds chunk
    // c' ^ (s & 7) of input slot s (XOR swizzle on the source side: the DMA destination is lane-linear)
    const v4u rsx = tcn_make_srd(a.x, a.x_bytes);
    constexpr unsigned OOB = 0x80000000u;
    unsigned voff[kTcnMaxPieces];
    const int row_bytes = a.Cin * ES;
    const int seq_row0 = seq * a.T;
#pragma unroll
    for (int p = 0; p < kTcnMaxPieces; ++p) {
        const int s = p * 8 + (lane >> 3);
        const int t = t0 + (s & amask) + ((s >> a.logA) - HP) * a.d;
        const unsigned off = (unsigned)(seq_row0 + t) * (unsigned)row_bytes + (unsigned)(((lane & 7) ^ (s & 7)) << 4);
        voff[p] = (s < a.slots && (unsigned)t < (unsigned)a.T) ? off : OOB;   // (a select, not a branch)
    }
    auto issue_x = [&](int cs, int ring) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(cs * 128);
        const unsigned dst = my_lds + ring * region;
#pragma unroll
        for (int p = 0; p < 4; ++p) tcn_dma16(rsx, voff[p], soff, dst + p * 1024);     // np >= 4
        if (np > 4) {
            tcn_dma16(rsx, voff[4], soff, dst + 4 * 1024);
            if (np > 5) {
                tcn_dma16(rsx, voff[5], soff, dst + 5 * 1024);
                if (np > 6) {
                    tcn_dma16(rsx, voff[6], soff, dst + 6 * 1024);
                    tcn_dma16(rsx, voff[7], soff, dst + 7 * 1024);                     // np == 7 stages one spare piece of zeros
                }
            }
        }
    };

    // weight fragments: lane (r16, q) of the A operand holds 16 bytes of row n0 + r16 at K offset q*16 of each 64-byte half-step
    const int wn = min(n0 + r16, a.Cout - 1);   // rows >= Cout feed accumulator rows that are never stored
    const char* const wrow = a.w + (long long)wn * a.w_row_bytes + q * 16;
    struct WFrag { uint4 v[TAPS][2]; };
    auto load_w = [&](int cs, WFrag& f) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                f.v[tap][kk] = *(const uint4*)(wrow + ((a.tap0 + tap) * a.SPT + cs) * 128 + kk * 64);
    };

    // frame fragments: tap k of output slot i reads input slot i + k*A
    int xaddr[TAPS][2];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int s = mt * 16 + r16 + tap * a.A;
            xaddr[tap][mt] = s * 128 + ((q ^ (s & 7)) << 4);
        }

    // epilogue operands of waves 0 / 1 (frame tile mt = wave): bias and residual are fetched NOW, so their latency hides behind the K loop
    const int ei = (wave & 1) * 16 + r16;                // output slot
    const int et = t0 + (ei & amask) + (ei >> a.logA) * a.d;
    const long long erow = (long long)seq * a.T + et;
    const int en = n0 + q * 4;
    const bool evec = (a.Cout & 3) == 0;                 // then en + 3 < Cout and rows are 16-byte (fp32) / 8-byte (bf16) aligned
    const bool eok = wave < 2 && et < a.T && en < a.Cout;
    // (raw loaded values only: nothing consumes them before the epilogue, so no wait lands in front of the K loop)
    float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pcs = make_float4(0.f, 0.f, 0.f, 0.f);        // LN: column sums of the folded weight
    uint4 pr = make_uint4(0u, 0u, 0u, 0u);               // residual: 4 fp32 or (in .x/.y) 4 bf16
    constexpr int NSP = kTcnMaxStatParts / 4;
    float2 pst[LN ? NSP : 1];                            // LN: the producer's partials q, q + 4, ... of this lane's frame
    if constexpr (LN) {
        const long long nrows = (long long)a.B * a.T;
#pragma unroll
        for (int j = 0; j < NSP; ++j) {
            pst[j] = make_float2(0.f, 0.f);
            if (wave < 2 && et < a.T && q + 4 * j < a.stat_parts) pst[j] = a.stat_in[(q + 4 * j) * nrows + erow];
        }
    }
    if (eok) {
        if constexpr (LN) {
            if (evec) pcs = *(const float4*)(a.ln_colsum + en);
            else {
                pcs.x = a.ln_colsum[en];
                if (en + 1 < a.Cout) pcs.y = a.ln_colsum[en + 1];
                if (en + 2 < a.Cout) pcs.z = a.ln_colsum[en + 2];
                if (en + 3 < a.Cout) pcs.w = a.ln_colsum[en + 3];
            }
        }
        if (a.bias) {
            if (evec) pb = *(const float4*)(a.bias + en);
            else {
                pb.x = a.bias[en];
                if (en + 1 < a.Cout) pb.y = a.bias[en + 1];
                if (en + 2 < a.Cout) pb.z = a.bias[en + 2];
                if (en + 3 < a.Cout) pb.w = a.bias[en + 3];
            }
        }
        if (a.res) {
            const T* rp = (const T*)a.res + erow * a.Cout + en;
            if (evec) {
                if constexpr (sizeof(T) == 4) pr = *(const uint4*)rp;
                else { const uint2 r2 = *(const uint2*)rp; pr.x = r2.x; pr.y = r2.y; }
            } else {
                unsigned e4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (en + e < a.Cout) e4[e] = sizeof(T) == 4 ? ((const unsigned*)rp)[e] : (unsigned)((const u16*)rp)[e];
                if constexpr (sizeof(T) == 4) pr = make_uint4(e4[0], e4[1], e4[2], e4[3]);
                else { pr.x = e4[0] | (e4[1] << 16); pr.y = e4[2] | (e4[3] << 16); }
            }
        }
    }

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto compute = [&](const char* sb, const WFrag& f) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 fw = f.v[tap][kk];
                uint4 fx[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fx[mt] = *(const uint4*)(sb + (xaddr[tap][mt] ^ (kk << 6)));
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw), __builtin_bit_cast(bf16x8_t, fx[mt]),
                                                                          acc[mt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw.x), __uint_as_float(fx[mt].x), acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw.y), __uint_as_float(fx[mt].y), acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw.z), __uint_as_float(fx[mt].z), acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw.w), __uint_as_float(fx[mt].w), acc[mt], 0, 0, 0);
                }
            }
    };
    // wait until the OLDER of two slices in flight has landed: everything issued for the younger one may stay outstanding
    auto wait_older = [&](bool other_in_flight) {
        if (!other_in_flight) { tcn_wait_vm<0>(); return; }
        switch (np) {
            case 4: tcn_wait_vm<4 + 2 * TAPS>(); break;
            case 5: tcn_wait_vm<5 + 2 * TAPS>(); break;
            case 6: tcn_wait_vm<6 + 2 * TAPS>(); break;
            default: tcn_wait_vm<8 + 2 * TAPS>(); break;
        }
    };

    // this wave's channel slices: wave, wave + 8, ...; two in flight (ring slot 0 / 1, fragment sets w0 / w1)
    const int SPT = a.SPT;
    int c0 = wave, c1 = wave + NW;
    bool v0 = c0 < SPT, v1 = c1 < SPT;
    WFrag w0, w1;
    if (v0) { issue_x(c0, 0); load_w(c0, w0); }
    if (v1) { issue_x(c1, 1); load_w(c1, w1); }
    while (v0) {
        wait_older(v1);
        compute(my_smem, w0);
        c0 += 2 * NW;
        v0 = c0 < SPT;
        if (v0) { issue_x(c0, 0); load_w(c0, w0); }   // (the reads of slot 0 have returned: their MFMAs are issued)
        if (!v1) break;
        wait_older(v0);
        compute(my_smem + region, w1);
        c1 += 2 * NW;
        v1 = c1 < SPT;
        if (v1) { issue_x(c1, 1); load_w(c1, w1); }
    }

    // ---- partial tiles -> LDS (accumulator layout), fixed-order sum, fused epilogue by waves 0 / 1 (frame tile mt = wave)
    f32x4* const red = (f32x4*)(smem + NW * kTcnRing * region);
    red[(wave * 2 + 0) * 64 + lane] = acc[0];
    red[(wave * 2 + 1) * 64 + lane] = acc[1];
    __syncthreads();
    if (wave >= 2) return;
    f32x4 sum = red[wave * 64 + lane];
#pragma unroll
    for (int v = 1; v < NW; ++v) sum += red[(v * 2 + wave) * 64 + lane];
    if constexpr (LN) {                                   // y = rstd (x . W' - mean colsum) + (W . beta + b): LayerNorm(x) . W^T + b with the channel scale in W'
        float s1 = pst[0].x, s2 = pst[0].y;
#pragma unroll
        for (int j = 1; j < NSP; ++j) { s1 += pst[j].x; s2 += pst[j].y; }      // (absent parts are zeros)
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);      // parts q, q + 4, ... sit in lanes r16 + 16 q: every lane ends with the same total
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const float mean = s1 * a.ln_inv_k;
        const float var = fmaxf(fmaf(-mean, mean, s2 * a.ln_inv_k), 0.f);
        const float rstd = 1.0f / sqrtf(var + a.ln_eps);
        sum[0] = rstd * fmaf(-mean, pcs.x, sum[0]);
        sum[1] = rstd * fmaf(-mean, pcs.y, sum[1]);
        sum[2] = rstd * fmaf(-mean, pcs.z, sum[2]);
        sum[3] = rstd * fmaf(-mean, pcs.w, sum[3]);
    }
    if (!eok) return;
    sum += (f32x4){pb.x, pb.y, pb.z, pb.w};
    if constexpr (sizeof(T) == 4) sum += (f32x4){__uint_as_float(pr.x), __uint_as_float(pr.y), __uint_as_float(pr.z), __uint_as_float(pr.w)};
    else sum += (f32x4){bf16_to_f32((u16)(pr.x & 0xffff)), bf16_to_f32((u16)(pr.x >> 16)), bf16_to_f32((u16)(pr.y & 0xffff)), bf16_to_f32((u16)(pr.y >> 16))};
    const long long row = erow;
    const int n = en;
    const bool vec = evec;
    if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] = fmaxf(sum[e], 0.f);
    }
    if constexpr (sizeof(T) == 4 && TAPS == 1) {
        if (a.stat_out) {       // LayerNorm partials of the stored row over this tile's 16 channels (Cout % 16 == 0: every lane of the frame is here)
            float v1 = (sum[0] + sum[1]) + (sum[2] + sum[3]);
            float v2 = fmaf(sum[3], sum[3], fmaf(sum[2], sum[2], fmaf(sum[1], sum[1], sum[0] * sum[0])));
            v1 += __shfl_xor(v1, 16); v1 += __shfl_xor(v1, 32);
            v2 += __shfl_xor(v2, 16); v2 += __shfl_xor(v2, 32);
            if (q == 0) a.stat_out[(long long)n_tile * ((long long)a.B * a.T) + row] = make_float2(v1, v2);
        }
    }
    if constexpr (OUT_F32) {
        float* yp = (float*)a.y + row * a.Cout + n;
        if (vec) *(float4*)yp = make_float4(sum[0], sum[1], sum[2], sum[3]);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < a.Cout) yp[e] = sum[e];
        }
    } else {
        u16* yp = (u16*)a.y + row * a.Cout + n;
        if (vec) *(uint2*)yp = make_uint2(pack_bf16x2(sum[0], sum[1]), pack_bf16x2(sum[2], sum[3]));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < a.Cout) yp[e] = f32_to_bf16(sum[e]);
        }
    }
}

// comb geometry for (T, d): A residues x J = 32 / A chain positions per tile, A | d, A <= 16.  Fewest tiles wins, then fewest staged rows.
struct Comb { int A, logA, tps, S, TT, slots; };
Comb choose_comb(int T, int d, int taps) {
    Comb best{};
    long long best_cost = -1;
    for (int logA = 0; logA <= 4; ++logA) {
        const int A = 1 << logA;
        if (taps == 3 && (A > d || d % A)) continue;
        if (taps == 1 && A > 1) break;     // one tap: 32 consecutive frames
        const int J = kTcnBM / A;
        const int dd = taps == 1 ? 1 : d;
        const long long S = (long long)J * dd;
        const int tps = dd / A;
        const long long TT = (T + S - 1) / S * tps;
        const int slots = (J + taps - 1) * A;
        const long long cost = TT * 1024 + slots;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = Comb{A, logA, tps, (int)(S > 0x3fffffff ? 0x3fffffff : S), (int)TT, slots};
        }
    }
    return best;
}

template <typename T, int TAPS, bool OUT_F32, bool LN = false>
int launch_tcn(const TcnK& k, hipStream_t s) {
    const int lds = kTcnWaves * kTcnRing * k.np * 1024 + kTcnWaves * 2 * 64 * 16;
    if (lds > 160 * 1024) return MT4_EUNSUPPORTED;
    auto fn = tcn_conv_kernel<T, TAPS, OUT_F32, LN>;
    if (lds > 65536) MT4_RAISE_LDS(fn);
    hipLaunchKernelGGL(fn, dim3(8, k.TT, k.B * k.nnx), dim3(kTcnWaves * 64), lds, s, k);
    return mt4_check_launch();
}

int tcn_conv_impl(const mt4_tcn_desc* d, hipStream_t s, const float* ln_colsum = nullptr, float ln_eps = 0.f, const float* stat_in = nullptr,
                  float* stat_out = nullptr) {
    if (!d || !d->x || !d->w || !d->y) return MT4_EINVAL;
    if (ln_colsum && (d->dtype != MT4_F32 || d->taps != 1 || !d->bias || !(ln_eps > 0.f) || ((uintptr_t)ln_colsum & 15) || !stat_in || d->Cin % 16 ||
                      d->Cin / 16 > kTcnMaxStatParts || ((uintptr_t)stat_in & 7))) return MT4_EINVAL;
    if (stat_out && (d->dtype != MT4_F32 || d->taps != 1 || d->Cout % 16 || ((uintptr_t)stat_out & 7))) return MT4_EINVAL;
    if (d->B <= 0 || d->T <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->dilation <= 0 || (d->taps != 1 && d->taps != 3)) return MT4_EINVAL;
    if (d->dtype != MT4_F32 && d->dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (d->out_dtype != MT4_F32 && d->out_dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (d->dtype == MT4_F32 && d->out_dtype != MT4_F32) return MT4_EUNSUPPORTED;
    if (d->relu < 0 || d->relu > 1) return MT4_EINVAL;
    const int es = d->dtype == MT4_BF16 ? 2 : 4;
    if ((d->Cin * es) % 128 != 0) return MT4_EUNSUPPORTED;        // whole 128-byte K-steps per tap (every TCN width is)
    if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y | (uintptr_t)d->residual | (uintptr_t)d->bias) & 15) return MT4_EALIGN;
    const long long xb = (long long)d->B * d->T * d->Cin * es;
    if (xb >= 0x7fffffffLL || (long long)d->B * d->T > 0x3fffffffLL) return MT4_EUNSUPPORTED;
    TcnK k{};
    k.x = (const char*)d->x; k.w = (const char*)d->w; k.bias = d->bias; k.res = (const char*)d->residual; k.y = (char*)d->y;
    k.T = d->T; k.B = d->B; k.Cin = d->Cin; k.Cout = d->Cout; k.relu = d->relu;
    k.SPT = d->Cin * es / 128;
    k.w_row_bytes = d->taps * k.SPT * 128;
    k.x_bytes = (unsigned)xb;
    // a dilation >= T puts both outer taps into the padding for every frame: only the centre tap contributes (exactly the same sum)
    int taps = d->taps, dil = d->dilation;
    k.tap0 = 0;
    if (taps == 3 && dil >= d->T) { taps = 1; k.tap0 = 1; }
    if (taps == 1) dil = 1;
    const Comb c = choose_comb(d->T, dil, taps);
    k.d = dil; k.A = c.A; k.logA = c.logA; k.tps = c.tps; k.S = c.S; k.TT = c.TT; k.slots = c.slots;
    k.np = cdiv(c.slots, 8);
    if (k.np > kTcnMaxPieces) return MT4_EUNSUPPORTED;
    k.n_tiles = cdiv(d->Cout, kTcnBN);
    k.nnx = cdiv(k.n_tiles, 8);
    if (c.TT > 65535 || (long long)d->B * k.nnx > 65535 || c.tps > 65535) return MT4_EUNSUPPORTED;   // grid limits; exact magic division
    k.tps_magic = (unsigned)(0x100000000ULL / (unsigned)c.tps) + 1u;
    k.nnx_magic = (unsigned)(0x100000000ULL / (unsigned)k.nnx) + 1u;
    k.stat_out = (float2*)stat_out;
    if (ln_colsum) {
        k.ln_colsum = ln_colsum; k.ln_eps = ln_eps; k.ln_inv_k = 1.0f / (float)d->Cin;
        k.stat_in = (const float2*)stat_in; k.stat_parts = d->Cin / 16;
        return launch_tcn<float, 1, true, true>(k, s);
    }
    if (d->dtype == MT4_F32) return taps == 3 ? launch_tcn<float, 3, true>(k, s) : launch_tcn<float, 1, true>(k, s);
    if (d->out_dtype == MT4_F32) return taps == 3 ? launch_tcn<u16, 3, true>(k, s) : launch_tcn<u16, 1, true>(k, s);
    return taps == 3 ? launch_tcn<u16, 3, false>(k, s) : launch_tcn<u16, 1, false>(k, s);
}

}  // namespace

extern "C" int mt4_tcn_conv(const mt4_tcn_desc* d, void* stream) {
    mt4_clear_error();
    return tcn_conv_impl(d, (hipStream_t)stream);
}

// nn.LayerNorm(Cin) + nn.Linear(Cin, Cout) on the rows of one short window in ONE launch (MS-TCT's norm1 -> q | kv and norm2 -> linear1,
// `Temporal_mstct/MSTCT/Temporal_Encoder.py`): w = gamma o W packed as for mt4_tcn_conv, colsum[n] = sum_k w[n][k], bias = W . beta + b.
// stats_in: the partial sums the launch that produced x left (its stats_out).
extern "C" int mt4_tcn_linear_ln_f32(const void* x, const void* w_folded, const float* colsum, const float* bias_folded, const void* residual, void* y,
                                     int32_t rows, int32_t Cin, int32_t Cout, float eps, int32_t relu, const float* stats_in, float* stats_out,
                                     void* stream) {
    mt4_clear_error();
    if (!colsum || !stats_in) return MT4_EINVAL;
    mt4_tcn_desc d{x, w_folded, bias_folded, residual, y, 1, rows, Cin, Cout, 1, 1, relu, MT4_F32, MT4_F32};
    return tcn_conv_impl(&d, (hipStream_t)stream, colsum, eps, stats_in, stats_out);
}

// nn.Linear on the rows of one short window (mt4_tcn_conv with one tap, fp32) that also leaves the LayerNorm partials of its OUTPUT rows for the
// mt4_tcn_linear_ln_f32 launch behind it: stats_out [Cout / 16][rows] x (sum, sum of squares over 16 channels), Cout % 16 == 0.
extern "C" int mt4_tcn_linear_stats_f32(const void* x, const void* w, const float* bias, const void* residual, void* y, int32_t rows, int32_t Cin,
                                        int32_t Cout, int32_t relu, float* stats_out, void* stream) {
    mt4_clear_error();
    if (!stats_out) return MT4_EINVAL;
    mt4_tcn_desc d{x, w, bias, residual, y, 1, rows, Cin, Cout, 1, 1, relu, MT4_F32, MT4_F32};
    return tcn_conv_impl(&d, (hipStream_t)stream, nullptr, 0.f, nullptr, stats_out);
}

// One DilatedResidualLayer (Temporal_tenco/network.py:186-198, eval): y = x + conv_1x1(relu(conv_dilated(x))), two dependent launches;
// h [B*T][C] is the caller's scratch for the hidden activation.
extern "C" int mt4_tcn_dilated_residual_layer(const void* x, const void* w_dilated, const float* b_dilated, const void* w_1x1, const float* b_1x1,
                                              void* h, void* y, int32_t B, int32_t T, int32_t C, int32_t dilation, int32_t dtype, void* stream) {
    mt4_clear_error();
    mt4_tcn_desc d1{x, w_dilated, b_dilated, nullptr, h, B, T, C, C, 3, dilation, 1, dtype, dtype};
    int rc = tcn_conv_impl(&d1, (hipStream_t)stream);
    if (rc != MT4_OK) return rc;
    mt4_tcn_desc d2{h, w_1x1, b_1x1, x, y, B, T, C, C, 1, 1, 0, dtype, dtype};
    return tcn_conv_impl(&d2, (hipStream_t)stream);
}

// A whole stage (BaseCausalTCN / Refinement layer stack, network.py:116-135,147-162): n_layers >= 1 DilatedResidualLayers with
// dilation 2^i.  Layer 0 reads x (never written), the layers in between ping-pong between buf_a and buf_b, the last one writes y.
// w / b: arrays of n_layers device pointers (host memory).
extern "C" int mt4_tcn_stage(const void* x, void* buf_a, void* buf_b, void* h, void* y, const void* const* w_dilated,
                             const float* const* b_dilated, const void* const* w_1x1, const float* const* b_1x1, int32_t n_layers, int32_t B,
                             int32_t T, int32_t C, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (n_layers < 1 || n_layers > 30 || !x || !h || !y || !w_dilated || !b_dilated || !w_1x1 || !b_1x1) return MT4_EINVAL;
    if (n_layers > 1 && !buf_a) return MT4_EINVAL;
    if (n_layers > 2 && !buf_b) return MT4_EINVAL;
    const void* cur = x;
    for (int i = 0; i < n_layers; ++i) {
        void* nxt = i == n_layers - 1 ? y : (i & 1) ? buf_b : buf_a;
        mt4_tcn_desc d1{cur, w_dilated[i], b_dilated[i], nullptr, h, B, T, C, C, 3, 1 << i, 1, dtype, dtype};
        int rc = tcn_conv_impl(&d1, (hipStream_t)stream);
        if (rc != MT4_OK) return rc;
        mt4_tcn_desc d2{h, w_1x1[i], b_1x1[i], cur, nxt, B, T, C, C, 1, 1, 0, dtype, dtype};
        rc = tcn_conv_impl(&d2, (hipStream_t)stream);
        if (rc != MT4_OK) return rc;
        cur = nxt;
    }
    return MT4_OK;
}

// FPN top-down pathway (Temporal_tenco/network.py:93-106 with equal lengths: F.interpolate(x, size=W) is the identity):
//   level[l] = lat[l] + level[l + 1]   for l = nlev-2 .. 0, in place on `levels` [nlev][n]; lat [nlev-1][n]
template <typename T>
__global__ void fpn_topdown_kernel(const T* __restrict__ lat, T* __restrict__ levels, int nlev, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float acc[4];
    if constexpr (sizeof(T) == 4) { const float4 v = *(const float4*)(levels + (nlev - 1) * n + i); acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w; }
    else { const uint2 v = *(const uint2*)(levels + (nlev - 1) * n + i); acc[0] = bf16_to_f32((u16)(v.x & 0xffff)); acc[1] = bf16_to_f32((u16)(v.x >> 16)); acc[2] = bf16_to_f32((u16)(v.y & 0xffff)); acc[3] = bf16_to_f32((u16)(v.y >> 16)); }
    for (int l = nlev - 2; l >= 0; --l) {
        if constexpr (sizeof(T) == 4) {
            const float4 v = *(const float4*)(lat + l * n + i);
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            *(float4*)(levels + l * n + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            const uint2 v = *(const uint2*)(lat + l * n + i);
            acc[0] += bf16_to_f32((u16)(v.x & 0xffff)); acc[1] += bf16_to_f32((u16)(v.x >> 16));
            acc[2] += bf16_to_f32((u16)(v.y & 0xffff)); acc[3] += bf16_to_f32((u16)(v.y >> 16));
            const uint2 o = make_uint2(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]));
            *(uint2*)(levels + l * n + i) = o;
            // the next level up adds to the ROUNDED value, as separate launches would
            acc[0] = bf16_to_f32((u16)(o.x & 0xffff)); acc[1] = bf16_to_f32((u16)(o.x >> 16)); acc[2] = bf16_to_f32((u16)(o.y & 0xffff)); acc[3] = bf16_to_f32((u16)(o.y >> 16));
        }
    }
}

extern "C" int mt4_fpn_topdown(const void* lat, void* levels, int32_t nlev, int64_t n, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!lat || !levels || nlev < 2 || n <= 0 || (n & 3)) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (((uintptr_t)lat | (uintptr_t)levels) & 15) return MT4_EALIGN;
    const int grid = (int)((n / 4 + 255) / 256);
    if (dtype == MT4_F32) hipLaunchKernelGGL(fpn_topdown_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)lat, (float*)levels, nlev, (long long)n);
    else hipLaunchKernelGGL(fpn_topdown_kernel<u16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u16*)lat, (u16*)levels, nlev, (long long)n);
    return mt4_check_launch();
}
