// Kernels around the GEMMs of the transformer-shaped stages (Swin + Query2Label, MS-TCT): LayerNorm with
// row gather, multi-head attention core, patch extraction, small element-wise pieces.  All HBM-bound or tiny.
#include "mt4_common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------ vector helpers
template <typename T> struct Vec;  // one 16-byte global vector <-> floats
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<u16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const u16* p, float* v) {
        const uint4 t = *(const uint4*)p;
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = bf16_to_f32((u16)(u[e] & 0xffff)); v[2 * e + 1] = bf16_to_f32((u16)(u[e] >> 16)); }
    }
    static __device__ __forceinline__ void store(u16* p, const float* v) {
        *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
};
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<u16>(const u16* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<u16>(u16* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ LayerNorm (+ row gather)
// y[m][0..G*C) = LN( concat_g x[src(m,g)][0..C) ), src(m,g) = (m / L_out) * L_in + map[(m % L_out) * G + g]  (map NULL:
// identity, G = 1).  One wave per output row, values kept in registers (two-pass mean / variance in fp32).
template <typename T, int MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int M_out, int C, int G,
                                                        const int* __restrict__ map, int L_out, int L_in, float eps) {
    constexpr int V = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M_out) return;
    const int N = G * C;
    const int nch = N / V;
    const int b = map ? m / L_out : 0, l = map ? m - b * L_out : 0;
    float v[MAXCH][V];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int j = lane + 64 * i;
        if (j < nch) {
            const int e = j * V;
            const int g = e / C;
            const long long src = map ? (long long)b * L_in + map[l * G + g] : m;
            Vec<T>::load(x + src * C + (e - g * C), v[i]);
#pragma unroll
            for (int k = 0; k < V; ++k) sum += v[i][k];
        }
    }
    const float mean = wave_sum(sum) / (float)N;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int k = 0; k < V; ++k) { const float d = v[i][k] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)N + eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int j = lane + 64 * i;
        if (j < nch) {
            const int e = j * V;
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = (v[i][k] - mean) * rstd * gamma[e + k] + beta[e + k];
            Vec<T>::store(y + (long long)m * N + e, o);
        }
    }
}

// Narrow rows (G*C/V <= 32 chunks: Swin stage 1-2, C = 128 / 256 in bf16): LPR lanes per row, 64/LPR rows per wave, so that every
// lane of the wave carries a 16-byte chunk (one row per wave left 48 / 32 of the 64 lanes idle and moved 256 B per wave-load).
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ y, int M_out, int C, int G,
                                                               const int* __restrict__ map, int L_out, int L_in, float eps) {
    constexpr int V = Vec<T>::N;
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int m = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int j = lane % LPR;
    const int N = G * C;
    const int nch = N / V;
    const bool ok = m < M_out && j < nch;
    float v[V];
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = 0.f;
    const int e = j * V;
    if (ok) {
        const int g = e / C;
        long long src = m;
        if (map) {
            const int b = m / L_out, l = m - b * L_out;
            src = (long long)b * L_in + map[l * G + g];
        }
        Vec<T>::load(x + src * C + (e - g * C), v);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) sum += v[k];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)N;
    float sq = 0.f;
    if (j < nch) {
#pragma unroll
        for (int k = 0; k < V; ++k) { const float d = v[k] - mean; sq += d * d; }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)N + eps);
    if (ok) {
        float o[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (v[k] - mean) * rstd * gamma[e + k] + beta[e + k];
        Vec<T>::store(y + (long long)m * N + e, o);
    }
}

extern "C" int mt4_layernorm(const void* x, const float* gamma, const float* beta, void* y, int32_t M_out, int32_t C, int32_t G,
                             const int32_t* map, int32_t L_out, int32_t L_in, float eps, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !gamma || !beta || !y || M_out <= 0 || C <= 0 || G <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (map && (L_out <= 0 || L_in <= 0 || M_out % L_out != 0)) return MT4_EINVAL;
    if (!map && G != 1) return MT4_EINVAL;
    const int V = dtype == MT4_BF16 ? 8 : 4;
    if (C % V != 0 || (((uintptr_t)x | (uintptr_t)y) & 15)) return MT4_EALIGN;
    const int nch = G * C / V;
    if (nch > 64 * 16) return MT4_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(M_out, 4)), block(256);
#define LN_LAUNCH(TT, MC) hipLaunchKernelGGL((layernorm_kernel<TT, MC>), grid, block, 0, s, (const TT*)x, gamma, beta, (TT*)y, M_out, C, G, map, L_out, L_in, eps)
    if (nch <= 32) {
        const int lpr = nch <= 16 ? 16 : 32;
        const dim3 g2(cdiv(M_out, 4 * (64 / lpr)));
#define LN_NARROW(TT, LL) hipLaunchKernelGGL((layernorm_narrow_kernel<TT, LL>), g2, block, 0, s, (const TT*)x, gamma, beta, (TT*)y, M_out, C, G, map, L_out, L_in, eps)
        if (dtype == MT4_BF16) { if (lpr == 16) LN_NARROW(u16, 16); else LN_NARROW(u16, 32); }
        else { if (lpr == 16) LN_NARROW(float, 16); else LN_NARROW(float, 32); }
#undef LN_NARROW
        return mt4_check_launch();
    }
    const int need = cdiv(nch, 64);
    if (dtype == MT4_BF16) {
        if (need <= 2) LN_LAUNCH(u16, 2); else if (need <= 4) LN_LAUNCH(u16, 4); else if (need <= 8) LN_LAUNCH(u16, 8); else LN_LAUNCH(u16, 16);
    } else {
        if (need <= 2) LN_LAUNCH(float, 2); else if (need <= 4) LN_LAUNCH(float, 4); else if (need <= 8) LN_LAUNCH(float, 8); else LN_LAUNCH(float, 16);
    }
#undef LN_LAUNCH
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ attention core
// out[b,i,h,:] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] + bias[h,i,j] + mask[b % nW,i,j] ) v[b,j,h,:]
// One query per LPQ lanes (each lane owns DPL head dims), keys/values staged in LDS as fp32 in chunks of KC keys,
// online softmax in fp32.  Covers Swin windows (hd 32, N 49/144, bias + shift mask), nn.MultiheadAttention of the
// Q2L transformer (hd 256) and the MS-TCT global block (hd 32..108, T 256).
template <typename T, int DPL, int LPQ>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ out, const float* __restrict__ bias,
                                                        const float* __restrict__ mask, int Nq, int Nk, int hd, int q_stride,
                                                        int k_stride, int v_stride, int o_stride, int nW, float scale, int KC, int vec_ok) {
    constexpr int SL = DPL + 4;        // slice stride in LDS (floats): staggers the LPQ slices over banks
    constexpr int ROW = LPQ * SL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;
    float* Vs = lds + (size_t)KC * ROW;
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x;
    const int qpb = blockDim.x / LPQ;
    const int qi = blockIdx.x * qpb + tid / LPQ;
    const int sl = tid % LPQ;
    const bool q_ok = qi < Nq;
    const int d0 = sl * DPL;

    float qr[DPL], o[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
        o[d] = 0.f;
        qr[d] = (q_ok && d0 + d < hd) ? ld1<T>(q + ((long long)b * Nq + qi) * q_stride + h * hd + d0 + d) * scale : 0.f;
    }
    float mrun = -INFINITY, lrun = 0.f;
    const float* brow = (bias && q_ok) ? bias + ((long long)h * Nq + qi) * Nk : nullptr;
    const float* mrow = (mask && q_ok) ? mask + ((long long)(b % nW) * Nq + qi) * Nk : nullptr;

    const int hd4 = hd >> 2;  // hd % 4 == 0 (checked on the host)
    for (int j0 = 0; j0 < Nk; j0 += KC) {
        const int kc = min(KC, Nk - j0);
        __syncthreads();  // previous chunk fully consumed
        if ((hd & 3) == 0 && vec_ok) {
            for (int e = tid; e < kc * hd4; e += blockDim.x) {
                const int j = e / hd4, d = (e - j * hd4) * 4;
                const int s_ = d / DPL, dd = d - s_ * DPL;  // DPL % 4 == 0: a 4-vector never straddles slices
                const long long kr = ((long long)b * Nk + j0 + j);
                float4 kv, vv;
                if constexpr (sizeof(T) == 2) {
                    const uint2 a = *(const uint2*)(k + kr * k_stride + h * hd + d);
                    const uint2 c = *(const uint2*)(v + kr * v_stride + h * hd + d);
                    kv = make_float4(bf16_to_f32((u16)(a.x & 0xffff)), bf16_to_f32((u16)(a.x >> 16)), bf16_to_f32((u16)(a.y & 0xffff)), bf16_to_f32((u16)(a.y >> 16)));
                    vv = make_float4(bf16_to_f32((u16)(c.x & 0xffff)), bf16_to_f32((u16)(c.x >> 16)), bf16_to_f32((u16)(c.y & 0xffff)), bf16_to_f32((u16)(c.y >> 16)));
                } else {
                    kv = *(const float4*)(k + kr * k_stride + h * hd + d);
                    vv = *(const float4*)(v + kr * v_stride + h * hd + d);
                }
                *(float4*)(Ks + j * ROW + s_ * SL + dd) = kv;
                *(float4*)(Vs + j * ROW + s_ * SL + dd) = vv;
            }
        } else {  // odd head dims / unaligned strides: element-wise staging
            for (int e = tid; e < kc * hd; e += blockDim.x) {
                const int j = e / hd, d = e - j * hd;
                const long long kr = ((long long)b * Nk + j0 + j);
                Ks[j * ROW + (d / DPL) * SL + d % DPL] = ld1<T>(k + kr * k_stride + h * hd + d);
                Vs[j * ROW + (d / DPL) * SL + d % DPL] = ld1<T>(v + kr * v_stride + h * hd + d);
            }
        }
        // zero the padded head dims of this thread's slices once per chunk (qr is 0 there, V must not be NaN)
        if (LPQ * DPL > hd) {
            for (int e = tid; e < kc * (LPQ * DPL - hd); e += blockDim.x) {
                const int j = e / (LPQ * DPL - hd), d = hd + (e - j * (LPQ * DPL - hd));
                Ks[j * ROW + (d / DPL) * SL + d % DPL] = 0.f;
                Vs[j * ROW + (d / DPL) * SL + d % DPL] = 0.f;
            }
        }
        __syncthreads();
        for (int j = 0; j < kc; j += 4) {
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
                if (j + u < kc) {
                    const float* kp = Ks + (j + u) * ROW + sl * SL;
#pragma unroll
                    for (int d = 0; d < DPL; d += 4) {
                        const float4 kk = *(const float4*)(kp + d);
                        acc += qr[d] * kk.x + qr[d + 1] * kk.y + qr[d + 2] * kk.z + qr[d + 3] * kk.w;
                    }
                }
                sc[u] = acc;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int x_ = 1; x_ < LPQ; x_ <<= 1) sc[u] += __shfl_xor(sc[u], x_);
                if (j + u < kc) {
                    if (brow) sc[u] += brow[j0 + j + u];
                    if (mrow) sc[u] += mrow[j0 + j + u];
                } else {
                    sc[u] = -INFINITY;
                }
            }
            const float mnew = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), mrun);
            const float corr = __expf(mrun - mnew);  // exp(-inf) = 0 on the first group
            float p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = __expf(sc[u] - mnew);
            lrun = lrun * corr + (p[0] + p[1]) + (p[2] + p[3]);
            mrun = mnew;
#pragma unroll
            for (int d = 0; d < DPL; ++d) o[d] *= corr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j + u < kc) {
                    const float* vp = Vs + (j + u) * ROW + sl * SL;
#pragma unroll
                    for (int d = 0; d < DPL; d += 4) {
                        const float4 vv = *(const float4*)(vp + d);
                        o[d] += p[u] * vv.x; o[d + 1] += p[u] * vv.y; o[d + 2] += p[u] * vv.z; o[d + 3] += p[u] * vv.w;
                    }
                }
            }
        }
    }
    if (q_ok) {
        const float inv = 1.0f / lrun;
        T* op = out + ((long long)b * Nq + qi) * o_stride + h * hd + d0;
#pragma unroll
        for (int d = 0; d < DPL; ++d)
            if (d0 + d < hd) st1<T>(op + d, o[d] * inv);
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wa_max3(float a, float b, float c) {      // (fmaxf chains compile to v_max_f32 + canonicalising copies: 47 instructions for 36 scores)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef short wa_v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 wa_read_tr(const char* p) {     // ds_read_b64_tr_b16 (EXEC must be all ones where it is used)
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((wa_v4s_t __attribute__((address_space(3)))*)(uintptr_t)p));
}

// ---- multi-head attention core on the matrix units, large head dim (Q2L encoder / decoders: 4 heads x 256, 144 keys): bf16, no
// bias / mask, Nk <= 16*NKT keys.  One workgroup = 64 queries of one (batch, head), a wave owns 16 of them.  As in the window kernel,
// S^T = K Q^T leaves a lane with ONE query column (softmax = 2 xor-shuffles) and the probabilities are already the B operand of
// O^T = V^T P^T.  The head dim runs through LDS in chunks of 64: K rows [key][64] for the scores, then V rows [key][64] for the output, read
// back transposed by `ds_read_b64_tr_b16` (the V^T image of rounds 1-3 was written with two-byte stores: 67 % of the kernel's LDS cycles were
// bank conflicts); the wave's Q fragments (16 queries x HD, pre-scaled) stay in registers.
template <int NKT, int HD>
__global__ __launch_bounds__(256) void mha_mfma_kernel(const u16* __restrict__ q, const u16* __restrict__ k, const u16* __restrict__ v,
                                                       u16* __restrict__ out, int Nq, int Nk, int q_stride, int k_stride, int v_stride,
                                                       int o_stride, float scale) {
    constexpr int NKP = NKT * 16;
    constexpr int NKP2 = ((NKT + 1) / 2) * 32;   // keys padded to whole 32-key MFMA blocks
    constexpr int K_PITCH = 160;                 // bytes per K row of a chunk (128 used): conflict-free b128 fragment reads under the REAL lane grouping of
                                                 // ds_read_b128 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md); 144 (rounds 1-3: conflict-free for 16 consecutive
                                                 // lanes) measured SQ_LDS_BANK_CONFLICT = 69 % of this kernel's LDS cycles (profiles/r04_swin_kernel_counters.txt)
    constexpr int V_PITCH = 128;                 // bytes per V row of a chunk: 4 spans of 32 B (16 head dims each); span dt of key k sits at dt ^ ((k >> 1) & 3), so the
                                                 // 8 keys a 32-lane half of a transposed read takes fall on 8 different bank groups
    constexpr int NCH = HD / 64;
    static_assert(HD % 64 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                    // [NKP][K_PITCH]
    char* Vs = smem + NKP * K_PITCH;    // [NKP2][V_PITCH]
    const int h = blockIdx.x, b = blockIdx.y, q0 = blockIdx.z * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, qd = lane >> 4;
    const int query = q0 + wave * 16 + r16;
    const bool q_ok = query < Nq;

    // Q fragments: k-slot block kk (32 dims) of this lane's query, dims 32kk + 8qd .. +7
    bf16x8_t qf[HD / 32];
    {
        const u16* qp = q + ((long long)b * Nq + (q_ok ? query : 0)) * q_stride + h * HD + qd * 8;
#pragma unroll
        for (int kk = 0; kk < HD / 32; ++kk) {
            uint4 t = q_ok ? *(const uint4*)(qp + kk * 32) : make_uint4(0, 0, 0, 0);
            uint32_t* u = (uint32_t*)&t;
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = pack_bf16x2(__uint_as_float(u[i] << 16) * scale, __uint_as_float(u[i] & 0xffff0000u) * scale);
            qf[kk] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NCH; ++c) {
        if (c) __syncthreads();
        {   // K chunk: rows of 64 dims = 8 pieces of 16 B.  All loads of the chunk are issued before the first LDS store (as a load -> store loop
            // the chunk cost one memory round trip per iteration; rows beyond Nk read row Nk - 1 and are zeroed by a select, not a branch)
            constexpr int SI = (NKP * 8 + 255) / 256;
            uint4 kreg[SI];
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                const uint4 t = *(const uint4*)(k + ((long long)b * Nk + min(row, Nk - 1)) * k_stride + h * HD + c * 64 + pc * 8);
                kreg[i] = row < Nk ? t : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                if (e < NKP * 8) *(uint4*)(Ks + row * K_PITCH + pc * 16) = kreg[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Ks + (kt * 16 + r16) * K_PITCH + kk * 64 + qd * 16));
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c * 2 + kk], s[kt], 0, 0, 0);
            }
    }
    // softmax over the keys of this lane's query column: lane holds keys 16kt + 4qd .. +3
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kt * 16 + qd * 4 + e >= Nk) s[kt][e] = -1e30f;
        mx = wa_max3(mx, s[kt][0], s[kt][1]);
        mx = wa_max3(mx, s[kt][2], s[kt][3]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    constexpr float L2E = 1.4426950408889634f;       // p = 2^((s - mx) log2 e): a packed fma + two v_exp_f32 per pair of scores
    const float nmx = -mx * L2E;
    mt4_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const mt4_f32x2 t0 = (mt4_f32x2){s[kt][0], s[kt][1]} * L2E + nmx, t1 = (mt4_f32x2){s[kt][2], s[kt][3]} * L2E + nmx;
        const mt4_f32x2 e0 = {__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y)}, e1 = {__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)};
        sum2 += e0;
        sum2 += e1;
        s[kt] = (f32x4){e0.x, e0.y, e1.x, e1.y};
    }
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    uint4 pf[(NKT + 1) / 2];   // probabilities as the B operand: k-slots 8qd..8qd+7 of block kb = keys 32kb + 4qd .. +3 and 32kb + 16 + 4qd .. +3
#pragma unroll
    for (int kb = 0; kb < (NKT + 1) / 2; ++kb) {
        const int k0 = 2 * kb, k1 = 2 * kb + 1;
        pf[kb].x = pack_bf16x2(s[k0][0], s[k0][1]);
        pf[kb].y = pack_bf16x2(s[k0][2], s[k0][3]);
        if (k1 < NKT) { pf[kb].z = pack_bf16x2(s[k1][0], s[k1][1]); pf[kb].w = pack_bf16x2(s[k1][2], s[k1][3]); }
        else { pf[kb].z = 0; pf[kb].w = 0; }
    }
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();   // previous chunk's (or the K chunk's) readers are done
        {   // V chunk, rows of 64 dims like K (loads first, as for K)
            constexpr int SI = (NKP2 * 8 + 255) / 256;
            uint4 vreg[SI];
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                const uint4 t = *(const uint4*)(v + ((long long)b * Nk + min(row, Nk - 1)) * v_stride + h * HD + c * 64 + pc * 8);
                vreg[i] = row < Nk ? t : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                if (e < NKP2 * 8) *(uint4*)(Vs + row * V_PITCH + (((((pc >> 1) ^ ((row >> 1) & 3)) << 1) | (pc & 1)) << 4)) = vreg[i];
            }
        }
        __syncthreads();
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < (NKT + 1) / 2; ++kb)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // lane 4 q + p of the 16-lane group qd supplies key 32 kb (+ 16) + 4 qd + q, dims 16 dt + 4 p .. + 3 and receives dim 16 dt + r16 of the 4 keys
                const int vkey = qd * 4 + (r16 >> 2), vp = r16 & 3;
                const char* vr = Vs + (kb * 32 + vkey) * V_PITCH + ((((dt ^ ((vkey >> 1) & 3)) << 1) | (vp >> 1)) << 4) + (vp & 1) * 8;
                const uint2 v0 = wa_read_tr(vr);
                const uint2 v1 = wa_read_tr(vr + 16 * V_PITCH);
                const uint4 vf = make_uint4(v0.x, v0.y, v1.x, v1.y);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[kb]), o[dt], 0, 0, 0);
            }
        if (q_ok) {   // lane (r16 = query, qd): dims 64c + 16dt + 4qd .. +3
            u16* op = out + ((long long)b * Nq + query) * o_stride + h * HD + c * 64 + qd * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(op + dt * 16) = make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
        }
    }
}

// ---- the same core in fp32 on the exact-fp32 matrix instruction (16x16x4): MS-TCT's Global_Relational_Block (`Temporal_Encoder.py:80-86`:
// 8 heads x 32..108 dims over T <= 256 frames) in the parity mode.  The contraction index of a product may be permuted freely as long as
// both operands agree, so within a group of 16 dims MFMA j takes dim 4*qd + j from lane group qd: a lane's 4 slots are 4 CONSECUTIVE
// floats = one 16-byte read (K rows, Q registers); likewise the 4 MFMAs of a 16-key tile take key 4*qd + e, which is what a lane of the
// score tile already holds.  Head dims run through LDS in chunks of 32 (K rows, then V^T), zero-padded to whole groups.
template <int NKT>
__global__ __launch_bounds__(256) void mha_mfma_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           float* __restrict__ out, int Nq, int Nk, int hd, int q_stride, int k_stride,
                                                           int v_stride, int o_stride, float scale) {
    constexpr int NKP = NKT * 16;
    constexpr int K_PITCH = 40;          // floats per K row of a chunk (32 used): 160 B = conflict-free b128 reads under the real lane grouping (36: 2-way)
    constexpr int VT_PITCH = NKP + 4;    // floats per V^T row: = 4 mod 64 for NKP = 128, 256
    constexpr int MAXG = 8;              // hd <= 128: at most 8 groups of 16 dims
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = (float*)smem;            // [NKP][K_PITCH]; the V^T chunk [32][VT_PITCH] reuses the space
    float* Vt = (float*)smem;
    const int h = blockIdx.x, b = blockIdx.y, q0 = blockIdx.z * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, qd = lane >> 4;
    const int query = q0 + wave * 16 + r16;
    const bool q_ok = query < Nq;
    const int nch = (hd + 31) / 32;

    float4 qf[MAXG];
    {
        const float* qp = q + ((long long)b * Nq + (q_ok ? query : 0)) * q_stride + h * hd;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int d = g * 16 + qd * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q_ok && d < hd) t = *(const float4*)(qp + d);
            qf[g] = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
        }
    }
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXG / 2; ++c) {
        if (c < nch) {   // (uniform)
            if (c) __syncthreads();
            {   // K chunk: rows of 32 dims = 8 pieces of 4 floats; every load issued before the first LDS store (clamped address + select)
                constexpr int SI = (NKP * 8 + 255) / 256;
                float4 kreg[SI];
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                    const int d = c * 32 + pc * 4;
                    const float4 t = *(const float4*)(k + ((long long)b * Nk + min(row, Nk - 1)) * k_stride + h * hd + min(d, hd - 4));
                    kreg[i] = (row < Nk && d < hd) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                    if (e < NKP * 8) *(float4*)(Ks + row * K_PITCH + pc * 4) = kreg[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 kf = *(const float4*)(Ks + (kt * 16 + r16) * K_PITCH + g * 16 + qd * 4);
                    const float4 qv = qf[c * 2 + g];
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qv.x, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qv.y, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qv.z, s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qv.w, s[kt], 0, 0, 0);
                }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kt * 16 + qd * 4 + e >= Nk) s[kt][e] = -1e30f;
            mx = fmaxf(mx, s[kt][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[kt][e] = __expf(s[kt][e] - mx); sum += s[kt][e]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        {   // V chunk, transposed: Vt[d][key] (loads first, as for K)
            constexpr int SI = (NKP * 8 + 255) / 256;
            float4 vreg[SI];
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                const int d = c * 32 + pc * 4;
                const float4 t = *(const float4*)(v + ((long long)b * Nk + min(row, Nk - 1)) * v_stride + h * hd + min(d, hd - 4));
                vreg[i] = (row < Nk && d < hd) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                if (e < NKP * 8) {
                    Vt[(pc * 4 + 0) * VT_PITCH + row] = vreg[i].x;
                    Vt[(pc * 4 + 1) * VT_PITCH + row] = vreg[i].y;
                    Vt[(pc * 4 + 2) * VT_PITCH + row] = vreg[i].z;
                    Vt[(pc * 4 + 3) * VT_PITCH + row] = vreg[i].w;
                }
            }
        }
        __syncthreads();
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float4 vf = *(const float4*)(Vt + (dt * 16 + r16) * VT_PITCH + kt * 16 + qd * 4);   // keys 16kt + 4qd .. +3 of dim row r16
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, s[kt][0], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, s[kt][1], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, s[kt][2], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, s[kt][3], o[dt], 0, 0, 0);
            }
        if (q_ok) {   // lane (r16 = query, qd): dims 32c + 16dt + 4qd .. +3
            float* op = out + ((long long)b * Nq + query) * o_stride + h * hd;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = c * 32 + dt * 16 + qd * 4;
                if (d < hd) *(float4*)(op + d) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            }
        }
    }
}

// The same core for a launch that cannot fill the chip with one wave per 16 queries (one MS-TCT window: 8 heads x 256 queries = 128 waves,
// each busy 1.7 us per 32-dim chunk with its 128 fp32 matrix instructions): the four waves of a workgroup share 16 queries and split the KEYS
// (key tiles 4w .. 4w+3 of 16), so 4x the waves carry a quarter of the matrix work each.  Row maxima and sums meet in LDS (fixed order
// wave 0..3: deterministic), the partial P.V tiles of a chunk likewise, and 128 threads store the summed chunk.
template <int NKT>
__global__ __launch_bounds__(256) void mha_mfma_f32_ksplit_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                  float* __restrict__ out, int Nq, int Nk, int hd, int q_stride, int k_stride,
                                                                  int v_stride, int o_stride, float scale) {
    constexpr int NKP = NKT * 16, NKW = NKT / 4;
    constexpr int K_PITCH = 40, VT_PITCH = NKP + 4, MAXG = 8;
    constexpr int MAIN = NKP * K_PITCH > 32 * VT_PITCH ? NKP * K_PITCH : 32 * VT_PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = (float*)smem;
    float* Vt = (float*)smem;
    float* red = (float*)smem + MAIN;          // [2][4][16]: row maxima, row sums per wave
    float* part = red + 128;                   // [4][16][32]: a chunk's partial output tiles per wave
    const int h = blockIdx.x, b = blockIdx.y, q0 = blockIdx.z * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, qd = lane >> 4;
    const int query = q0 + r16;
    const bool q_ok = query < Nq;
    const int nch = (hd + 31) / 32;

    float4 qf[MAXG];
    {
        const float* qp = q + ((long long)b * Nq + (q_ok ? query : 0)) * q_stride + h * hd;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int d = g * 16 + qd * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q_ok && d < hd) t = *(const float4*)(qp + d);
            qf[g] = make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale);
        }
    }
    f32x4 s[NKW];
#pragma unroll
    for (int j = 0; j < NKW; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXG / 2; ++c) {
        if (c < nch) {   // (uniform)
            if (c) __syncthreads();
            {
                constexpr int SI = (NKP * 8 + 255) / 256;
                float4 kreg[SI];
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                    const int d = c * 32 + pc * 4;
                    const float4 t = *(const float4*)(k + ((long long)b * Nk + min(row, Nk - 1)) * k_stride + h * hd + min(d, hd - 4));
                    kreg[i] = (row < Nk && d < hd) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                    if (e < NKP * 8) *(float4*)(Ks + row * K_PITCH + pc * 4) = kreg[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NKW; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 kf = *(const float4*)(Ks + ((wave * NKW + j) * 16 + r16) * K_PITCH + g * 16 + qd * 4);
                    const float4 qv = qf[c * 2 + g];
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qv.x, s[j], 0, 0, 0);
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qv.y, s[j], 0, 0, 0);
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qv.z, s[j], 0, 0, 0);
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qv.w, s[j], 0, 0, 0);
                }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if ((wave * NKW + j) * 16 + qd * 4 + e >= Nk) s[j][e] = -1e30f;
            mx = fmaxf(mx, s[j][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (qd == 0) red[wave * 16 + r16] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[r16], red[16 + r16]), fmaxf(red[32 + r16], red[48 + r16]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[j][e] = __expf(s[j][e] - mx); sum += s[j][e]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (qd == 0) red[64 + wave * 16 + r16] = sum;
    __syncthreads();
    const float inv = 1.0f / (((red[64 + r16] + red[80 + r16]) + red[96 + r16]) + red[112 + r16]);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();      // the K chunk / the previous V chunk and its partial tiles are done with
        {
            constexpr int SI = (NKP * 8 + 255) / 256;
            float4 vreg[SI];
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                const int d = c * 32 + pc * 4;
                const float4 t = *(const float4*)(v + ((long long)b * Nk + min(row, Nk - 1)) * v_stride + h * hd + min(d, hd - 4));
                vreg[i] = (row < Nk && d < hd) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < SI; ++i) {
                const int e = tid + i * 256, row = e >> 3, pc = e & 7;
                if (e < NKP * 8) {
                    Vt[(pc * 4 + 0) * VT_PITCH + row] = vreg[i].x;
                    Vt[(pc * 4 + 1) * VT_PITCH + row] = vreg[i].y;
                    Vt[(pc * 4 + 2) * VT_PITCH + row] = vreg[i].z;
                    Vt[(pc * 4 + 3) * VT_PITCH + row] = vreg[i].w;
                }
            }
        }
        __syncthreads();
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < NKW; ++j)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float4 vf = *(const float4*)(Vt + (dt * 16 + r16) * VT_PITCH + (wave * NKW + j) * 16 + qd * 4);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, s[j][0], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, s[j][1], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, s[j][2], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, s[j][3], o[dt], 0, 0, 0);
            }
        // lane (r16 = query, qd) holds dims 16dt + 4qd .. +3 of the chunk: part[wave][query][dim]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            *(float4*)(part + (wave * 16 + r16) * 32 + dt * 16 + qd * 4) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        __syncthreads();
        if (tid < 128) {      // thread: query tid / 8, dims 4 (tid % 8) .. +3 of the chunk
            const int qq = tid >> 3, d4 = (tid & 7) * 4;
            const float4 a0 = *(const float4*)(part + (0 * 16 + qq) * 32 + d4), a1 = *(const float4*)(part + (1 * 16 + qq) * 32 + d4);
            const float4 a2 = *(const float4*)(part + (2 * 16 + qq) * 32 + d4), a3 = *(const float4*)(part + (3 * 16 + qq) * 32 + d4);
            const float iv = 1.0f / (((red[64 + qq] + red[80 + qq]) + red[96 + qq]) + red[112 + qq]);
            const int d = c * 32 + d4;
            if (q0 + qq < Nq && d < hd)
                *(float4*)(out + ((long long)b * Nq + q0 + qq) * o_stride + h * hd + d) =
                    make_float4((((a0.x + a1.x) + a2.x) + a3.x) * iv, (((a0.y + a1.y) + a2.y) + a3.y) * iv,
                                (((a0.z + a1.z) + a2.z) + a3.z) * iv, (((a0.w + a1.w) + a2.w) + a3.w) * iv);
        }
    }
    (void)inv;
}

extern "C" int mt4_attention(const void* q, const void* k, const void* v, void* out, const float* bias, const float* mask,
                             int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t hd, int32_t q_stride, int32_t k_stride,
                             int32_t v_stride, int32_t o_stride, int32_t nW, float scale, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || hd <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (mask && nW <= 0) return MT4_EINVAL;
    if (hd > 512 || B > 65535 || H > 65535) return MT4_EUNSUPPORTED;
    const int es = dtype == MT4_BF16 ? 2 : 4;
    // 4-element vector staging needs every (row, head) start 4-element aligned
    const int vec_ok = (hd % 4 == 0) && (k_stride * es) % (4 * es) == 0 && (v_stride * es) % (4 * es) == 0 &&
                       (((uintptr_t)k | (uintptr_t)v) & (4 * es - 1)) == 0;
    hipStream_t s = (hipStream_t)stream;
    // large head dim, bf16, no bias / mask, keys fit one workgroup's score registers: matrix-unit kernel 
    // (head dim 256: Q2L over Swin-B, d = 1024 / 4 heads; 384: over Swin-L, d = 1536 -- the shipped teacher, Scripts/train_fold1.sh:5-12)
    if (dtype == MT4_BF16 && (hd == 256 || hd == 384) && !bias && !mask && Nk <= 160 && (q_stride % 8) == 0 && (k_stride % 8) == 0 && (v_stride % 8) == 0 &&
        (o_stride % 4) == 0 && ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0) && (((uintptr_t)out & 7) == 0) && cdiv(Nq, 64) <= 65535) {
        {
            const int nkt = cdiv(Nk, 16);
            const dim3 grid(H, B, cdiv(Nq, 64)), block(256);
#define MHA_LAUNCH_HD(NKTV, HDV) { constexpr int nkp = NKTV * 16, nkp2 = ((NKTV + 1) / 2) * 32; const size_t lds = (size_t)nkp * 160 + (size_t)nkp2 * 128; \
            hipLaunchKernelGGL((mha_mfma_kernel<NKTV, HDV>), grid, block, lds, s, (const u16*)q, (const u16*)k, (const u16*)v, (u16*)out, Nq, Nk, \
                               q_stride, k_stride, v_stride, o_stride, scale); }
#define MHA_LAUNCH(NKTV) { if (hd == 256) MHA_LAUNCH_HD(NKTV, 256) else MHA_LAUNCH_HD(NKTV, 384) }
            switch (nkt) {
                case 1: MHA_LAUNCH(1) break; case 2: MHA_LAUNCH(2) break; case 3: MHA_LAUNCH(3) break; case 4: MHA_LAUNCH(4) break;
                case 5: MHA_LAUNCH(5) break; case 6: MHA_LAUNCH(6) break; case 7: MHA_LAUNCH(7) break; case 8: MHA_LAUNCH(8) break;
                case 9: MHA_LAUNCH(9) break; default: MHA_LAUNCH(10) break;
            }
#undef MHA_LAUNCH_HD
#undef MHA_LAUNCH
            return mt4_check_launch();
        }
    }
    // fp32 (parity mode), no bias / mask, <= 256 keys, head dim <= 128: exact-fp32 matrix-unit kernel (MS-TCT's global block)
    if (dtype == MT4_F32 && hd <= 128 && (hd % 4) == 0 && !bias && !mask && Nk <= 256 && (q_stride % 4) == 0 && (k_stride % 4) == 0 &&
        (v_stride % 4) == 0 && (o_stride % 4) == 0 && ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0) &&
        cdiv(Nq, 64) <= 65535) {
        const dim3 block(256);
        // too few 64-query workgroups for the chip (one MS-TCT window: 32): four waves per 16 queries, keys split between them
        if ((long long)H * B * cdiv(Nq, 64) < 128 && cdiv(Nq, 16) <= 65535) {
            const dim3 grid(H, B, cdiv(Nq, 16));
            if (Nk <= 128) {
                const size_t lds = sizeof(float) * (size_t)((128 * 40 > 32 * 132 ? 128 * 40 : 32 * 132) + 128 + 4 * 16 * 32);
                hipLaunchKernelGGL((mha_mfma_f32_ksplit_kernel<8>), grid, block, lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out,
                                   Nq, Nk, hd, q_stride, k_stride, v_stride, o_stride, scale);
            } else {
                const size_t lds = sizeof(float) * (size_t)((256 * 40 > 32 * 260 ? 256 * 40 : 32 * 260) + 128 + 4 * 16 * 32);
                hipLaunchKernelGGL((mha_mfma_f32_ksplit_kernel<16>), grid, block, lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out,
                                   Nq, Nk, hd, q_stride, k_stride, v_stride, o_stride, scale);
            }
            return mt4_check_launch();
        }
        const dim3 grid(H, B, cdiv(Nq, 64));
        if (Nk <= 128) {
            const size_t lds = sizeof(float) * (size_t)(128 * 40 > 32 * 132 ? 128 * 40 : 32 * 132);
            hipLaunchKernelGGL((mha_mfma_f32_kernel<8>), grid, block, lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out, Nq, Nk, hd,
                               q_stride, k_stride, v_stride, o_stride, scale);
        } else {
            const size_t lds = sizeof(float) * (size_t)(256 * 40 > 32 * 260 ? 256 * 40 : 32 * 260);
            hipLaunchKernelGGL((mha_mfma_f32_kernel<16>), grid, block, lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out, Nq, Nk, hd,
                               q_stride, k_stride, v_stride, o_stride, scale);
        }
        return mt4_check_launch();
    }
    // (DPL, LPQ) with DPL*LPQ >= hd, fewest lanes per query first
    static const int cfgs[][2] = {{32, 1}, {24, 2}, {32, 2}, {20, 4}, {28, 4}, {32, 4}, {32, 8},
                                  {4, 8}, {8, 8}, {12, 8}, {16, 8}, {24, 8},   // 8 lanes per query, few dims per lane
                                  {48, 8}, {64, 8}};                           // head dims 257 .. 512 (Q2L over Swin-L: 1536 / 4 heads = 384)
    int ci = -1;
    for (int i = 0; i < 7; ++i)
        if (cfgs[i][0] * cfgs[i][1] >= hd) { ci = i; break; }
    if (ci < 0)
        for (int i = 12; i < 14; ++i)
            if (cfgs[i][0] * cfgs[i][1] >= hd) { ci = i; break; }
    if (ci < 0) return MT4_EUNSUPPORTED;
    int threads = ((Nq * cfgs[ci][1] + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    // a launch that cannot fill the chip (one short sequence: MS-TCT's global block has T = 256 queries x 8 heads) takes 8 lanes per
    // query instead: 8x less serial work per lane, 8x more workgroups (workgroup size: 64/128/256 threads within 2 %, A/B)
    if (hd <= 64 && (long long)cdiv(Nq, threads / cfgs[ci][1]) * H * B < 128) {   // (measured: 103 -> 72 us at hd 32, 94 -> 88 at hd 48; slower for hd >= 72)
        for (int i = 7; i < 12; ++i)
            if (cfgs[i][0] * 8 >= hd) { ci = i; threads = 256; break; }
    }
    const int DPL = cfgs[ci][0], LPQ = cfgs[ci][1];
    const int row = LPQ * (DPL + 4);
    int KC = 8192 / row;             // 2 * KC * row * 4 bytes <= 64 KiB
    KC = (KC / 4) * 4;
    if (KC > ((Nk + 3) / 4) * 4) KC = ((Nk + 3) / 4) * 4;
    const size_t lds = (size_t)2 * KC * row * sizeof(float);
    const int qpb = threads / LPQ;
    const dim3 grid(cdiv(Nq, qpb), H, B), block(threads);
#define ATT_LAUNCH(TT, D, L) hipLaunchKernelGGL((attention_kernel<TT, D, L>), grid, block, lds, s, (const TT*)q, (const TT*)k, (const TT*)v, (TT*)out, bias, mask, Nq, Nk, hd, q_stride, k_stride, v_stride, o_stride, nW, scale, KC, vec_ok)
#define ATT_DISPATCH(TT) switch (ci) { case 0: ATT_LAUNCH(TT, 32, 1); break; case 1: ATT_LAUNCH(TT, 24, 2); break; case 2: ATT_LAUNCH(TT, 32, 2); break; \
        case 3: ATT_LAUNCH(TT, 20, 4); break; case 4: ATT_LAUNCH(TT, 28, 4); break; case 5: ATT_LAUNCH(TT, 32, 4); break; case 6: ATT_LAUNCH(TT, 32, 8); break; \
        case 7: ATT_LAUNCH(TT, 4, 8); break; case 8: ATT_LAUNCH(TT, 8, 8); break; case 9: ATT_LAUNCH(TT, 12, 8); break; case 10: ATT_LAUNCH(TT, 16, 8); break; \
        case 11: ATT_LAUNCH(TT, 24, 8); break; case 12: ATT_LAUNCH(TT, 48, 8); break; default: ATT_LAUNCH(TT, 64, 8); }
    if (dtype == MT4_BF16) { ATT_DISPATCH(u16) } else { ATT_DISPATCH(float) }
#undef ATT_DISPATCH
#undef ATT_LAUNCH
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ patch extraction
// out[(b*Ph + ph)*Pw + pw][c*P*P + kh*P + kw] = img(b, c, P*ph + kh, P*pw + kw): the Conv2d(3, E, P, stride P) of
// PatchEmbed becomes a GEMM with K = 3*P*P in the weight's own (c, kh, kw) order.
template <typename T, bool FROM_U8>
__global__ void patchify_kernel(const void* __restrict__ in, T* __restrict__ out, int B, int H, int W, int P, float m0, float m1,
                                float m2, float s0, float s1, float s2) {
    const int Ph = H / P, Pw = W / P, K = 3 * P * P;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * Ph * Pw * K) return;
    const int kk = (int)(idx % K);
    long long t = idx / K;
    const int pw = (int)(t % Pw); t /= Pw;
    const int ph = (int)(t % Ph);
    const int b = (int)(t / Ph);
    const int c = kk / (P * P), r = kk - c * P * P, kh = r / P, kw = r - kh * P;
    const int y = ph * P + kh, x = pw * P + kw;
    float v;
    if (FROM_U8) {
        const uint8_t px = ((const uint8_t*)in)[(((long long)b * H + y) * W + x) * 3 + c];
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        v = ((float)px / 255.0f - mean) / sd;
    } else {
        v = ((const float*)in)[(((long long)b * 3 + c) * H + y) * W + x];
    }
    st1<T>(out + idx, v);
}

// The form every Swin extraction runs (uint8 NHWC frames, P = 4, bf16 rows): a thread owns the 4 pixels x 3 channels = 12 consecutive bytes of one
// frame row inside one patch and writes three 8-byte runs of the patch's row; ToTensor + Normalize come out of a 3 x 256-entry table built per
// workgroup with the generic kernel's own expression (same bits).  The one-thread-per-element kernel above spends ~100 integer instructions per
// output value on its index arithmetic: 236 us for 128 frames of 384 x 384 against the ~40 us its 170 MB take.
__global__ __launch_bounds__(256) void patchify_u8_p4_bf16_kernel(const uint8_t* __restrict__ in, u16* __restrict__ out, int B, int H, int W, float m0,
                                                                  float m1, float m2, float s0, float s1, float s2) {
    __shared__ u16 lut[3][256];
    for (int e = threadIdx.x; e < 768; e += 256) {
        const int c = e >> 8, px = e & 255;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        lut[c][px] = f32_to_bf16(((float)px / 255.0f - mean) / sd);
    }
    __syncthreads();
    const int Wq = W >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * Wq) return;
    const int xq = (int)(idx % Wq);
    const long long t = idx / Wq;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const uint32_t* src = (const uint32_t*)(in + ((t * W) + xq * 4) * 3);      // 12 bytes, 4-byte aligned (W % 4 == 0)
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    uint8_t px[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { px[i] = (w0 >> (8 * i)) & 255; px[4 + i] = (w1 >> (8 * i)) & 255; px[8 + i] = (w2 >> (8 * i)) & 255; }
    const int ph = y >> 2, kh = y & 3;
    u16* dst = out + (((long long)b * (H >> 2) + ph) * Wq + xq) * 48 + kh * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t lo = (uint32_t)lut[c][px[c]] | ((uint32_t)lut[c][px[3 + c]] << 16);
        const uint32_t hi = (uint32_t)lut[c][px[6 + c]] | ((uint32_t)lut[c][px[9 + c]] << 16);
        *(uint2*)(dst + c * 16) = make_uint2(lo, hi);
    }
}

extern "C" int mt4_patchify(const void* in, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t from_u8,
                            const float mean[3], const float std[3], int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || P <= 0 || H % P || W % P) return MT4_EINVAL;
    if (from_u8 && (!mean || !std)) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const long long total = (long long)B * (H / P) * (W / P) * 3 * P * P;
    const int grid = (int)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    const float m0 = from_u8 ? mean[0] : 0, m1 = from_u8 ? mean[1] : 0, m2 = from_u8 ? mean[2] : 0;
    const float s0 = from_u8 ? std[0] : 1, s1 = from_u8 ? std[1] : 1, s2 = from_u8 ? std[2] : 1;
    if (dtype == MT4_BF16 && from_u8 && P == 4 && (((uintptr_t)in & 3) == 0) && (((uintptr_t)out & 7) == 0)) {
        const long long threads = (long long)B * H * (W / 4);
        hipLaunchKernelGGL(patchify_u8_p4_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const uint8_t*)in, (u16*)out, B, H, W, m0, m1, m2,
                           s0, s1, s2);
        return mt4_check_launch();
    }
    if (dtype == MT4_BF16) {
        if (from_u8) hipLaunchKernelGGL((patchify_kernel<u16, true>), dim3(grid), dim3(256), 0, s, in, (u16*)out, B, H, W, P, m0, m1, m2, s0, s1, s2);
        else hipLaunchKernelGGL((patchify_kernel<u16, false>), dim3(grid), dim3(256), 0, s, in, (u16*)out, B, H, W, P, m0, m1, m2, s0, s1, s2);
    } else {
        if (from_u8) hipLaunchKernelGGL((patchify_kernel<float, true>), dim3(grid), dim3(256), 0, s, in, (float*)out, B, H, W, P, m0, m1, m2, s0, s1, s2);
        else hipLaunchKernelGGL((patchify_kernel<float, false>), dim3(grid), dim3(256), 0, s, in, (float*)out, B, H, W, P, m0, m1, m2, s0, s1, s2);
    }
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ y = x + p[row % L]
template <typename T>
__global__ void add_rowbcast_kernel(const T* __restrict__ x, const T* __restrict__ p, T* __restrict__ y, long long M, int L, int C) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * cv) return;
    const long long m = idx / cv;
    const int e = (int)(idx - m * cv) * V;
    float a[V], b[V];
    Vec<T>::load(x + m * C + e, a);
    Vec<T>::load(p + (m % L) * C + e, b);
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] += b[k];
    Vec<T>::store(y + m * C + e, a);
}

extern "C" int mt4_add_rowbcast(const void* x, const void* p, void* y, int64_t M, int32_t L, int32_t C, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !p || !y || M <= 0 || L <= 0 || C <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int V = dtype == MT4_BF16 ? 8 : 4;
    if (C % V || (((uintptr_t)x | (uintptr_t)p | (uintptr_t)y) & 15)) return MT4_EALIGN;
    const long long total = M * (C / V);
    const int grid = (int)((total + 255) / 256);
    if (dtype == MT4_BF16) hipLaunchKernelGGL(add_rowbcast_kernel<u16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u16*)x, (const u16*)p, (u16*)y, (long long)M, L, C);
    else hipLaunchKernelGGL(add_rowbcast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)p, (float*)y, (long long)M, L, C);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ GroupWiseLinear
// out[b][k] = sum_d W[k][d] * hs[b][k][d] + bias[k]   (one wave per (b,k))
template <typename T>
__global__ void groupwise_linear_kernel(const T* __restrict__ hs, const float* __restrict__ w, const float* __restrict__ bias,
                                        float* __restrict__ out, int BK, int K, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= BK) return;
    const int kcls = r % K;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += ld1<T>(hs + (long long)r * D + d) * w[(long long)kcls * D + d];
    s = wave_sum(s);
    if (lane == 0) out[r] = s + (bias ? bias[kcls] : 0.f);
}

extern "C" int mt4_groupwise_linear(const void* hs, const float* w, const float* bias, float* out, int32_t B, int32_t K, int32_t D,
                                    int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!hs || !w || !out || B <= 0 || K <= 0 || D <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int grid = cdiv(B * K, 4);
    if (dtype == MT4_BF16) hipLaunchKernelGGL(groupwise_linear_kernel<u16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u16*)hs, w, bias, out, B * K, K, D);
    else hipLaunchKernelGGL(groupwise_linear_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)hs, w, bias, out, B * K, K, D);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ depthwise conv1d k3 (+GELU)
// y[b][t][c] = act( w[c][0] x[b][t-1][c] + w[c][1] x[b][t][c] + w[c][2] x[b][t+1][c] + bias[c] ), zero padded in t
template <typename T>
__global__ void dwconv1d_k3_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   T* __restrict__ y, int B, int Tn, int C, int act) {
    constexpr int V = Vec<T>::N;
    const int cv = C / V;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * Tn * cv) return;
    const int e = (int)(idx % cv) * V;
    const long long bt = idx / cv;
    const int t = (int)(bt % Tn);
    float xm[V], x0[V], xp[V], o[V];
    Vec<T>::load(x + bt * C + e, x0);
    if (t > 0) Vec<T>::load(x + (bt - 1) * C + e, xm);
    else {
#pragma unroll
        for (int k = 0; k < V; ++k) xm[k] = 0.f;
    }
    if (t + 1 < Tn) Vec<T>::load(x + (bt + 1) * C + e, xp);
    else {
#pragma unroll
        for (int k = 0; k < V; ++k) xp[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float* wc = w + (long long)(e + k) * 3;
        float r = wc[0] * xm[k] + wc[1] * x0[k] + wc[2] * xp[k] + (bias ? bias[e + k] : 0.f);
        if (act == 1) r = fmaxf(r, 0.f);
        else if (act == 2) r = gelu_erf(r);
        o[k] = r;
    }
    Vec<T>::store(y + bt * C + e, o);
}

extern "C" int mt4_dwconv1d_k3(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t T, int32_t C, int32_t act,
                               int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !w || !y || B <= 0 || T <= 0 || C <= 0 || act < 0 || act > 2) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int V = dtype == MT4_BF16 ? 8 : 4;
    if (C % V || (((uintptr_t)x | (uintptr_t)y) & 15)) return MT4_EALIGN;
    const long long total = (long long)B * T * (C / V);
    const int grid = (int)((total + 255) / 256);
    if (dtype == MT4_BF16) hipLaunchKernelGGL(dwconv1d_k3_kernel<u16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u16*)x, w, bias, (u16*)y, B, T, C, act);
    else hipLaunchKernelGGL(dwconv1d_k3_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (float*)y, B, T, C, act);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ KD teacher mixing
// Spatial_cnn/network.py:56-62 in reduced form: the reference's [B,C,C] stack contracts to
//   attn[b,c,n] = softmax_n( s[b,c] / sqrt(C) * sum_d tea_n[b,d] ),   out_n[b,c] = s[b,c] * attn[b,c,n]
__global__ void kd_mix_kernel(const float* __restrict__ s, const float* __restrict__ t0, const float* __restrict__ t1,
                              const float* __restrict__ t2, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2, int C) {
    __shared__ float red[3][4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        a0 += t0[(long long)b * C + c]; a1 += t1[(long long)b * C + c]; a2 += t2[(long long)b * C + c];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; red[2][wave] = a2; }
    __syncthreads();
    const float ts0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float ts1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float ts2 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const float inv = rsqrtf((float)C);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sv = s[(long long)b * C + c];
        const float z = sv * inv;
        const float l0 = z * ts0, l1 = z * ts1, l2 = z * ts2;
        const float mx = fmaxf(l0, fmaxf(l1, l2));
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(l2 - mx);
        const float r = 1.0f / (e0 + e1 + e2);
        o0[(long long)b * C + c] = sv * e0 * r;
        o1[(long long)b * C + c] = sv * e1 * r;
        o2[(long long)b * C + c] = sv * e2 * r;
    }
}

extern "C" int mt4_kd_mix(const float* s, const float* tea_i, const float* tea_v, const float* tea_t, float* out_i, float* out_v,
                          float* out_t, int32_t B, int32_t C, void* stream) {
    mt4_clear_error();
    if (!s || !tea_i || !tea_v || !tea_t || !out_i || !out_v || !out_t || B <= 0 || C <= 0) return MT4_EINVAL;
    hipLaunchKernelGGL(kd_mix_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, s, tea_i, tea_v, tea_t, out_i, out_v, out_t, C);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ MFMA window attention (bf16, hd = 32)
// One workgroup (4 waves) per (window, head); N <= 256 tokens, head dim 32 -- every Swin stage (heads of 32 dims,
// N = 144 or 49) and MS-TCT stage 1.  Q (pre-scaled), K and V live in LDS as [token][32] rows of 64 bytes (16-byte slots XOR-swizzled);
// the V^T fragments of the second product come out of the row-major V image by `ds_read_b64_tr_b16` (a transposed [32][token] copy cost
// 24 two-byte LDS stores per thread).  Per 16-query tile a wave computes S^T = K Q^T + bias with ONE
// v_mfma_f32_16x16x32_bf16 per 16-key tile (contraction over the 32 head dims), so a lane owns one query column and
// 4 consecutive keys per tile: the softmax row lives in 4 lanes (2 xor-shuffles), and the probabilities are already in
// B-operand layout for O^T = V^T P^T (k-slot (q,e) of a 32-key block = key 16*kt + 4q + e for e < 4, the next tile's
// for e >= 4; the V^T fragment is read in the same order), so P never touches LDS.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));


template <int NT, int NW = 4>  // key/query tiles of 16; waves per workgroup
// (at 8 - 9 tiles -- Swin's 12 x 12 windows -- the kernel is held to 168 registers = three waves per SIMD: 10 spilled registers, +3 % unmasked and
//  +11 % on shifted blocks, same-box; four waves per SIMD spill 50 and lose 60 %.  9 tiles run in workgroups of THREE waves: over 4 waves
//  three of them idle for a third of the loop -- 0.207 -> 0.167 ms on stage 3 of Swin-B/384 at batch 128, 0.451 -> 0.335 on a shifted block)
__global__ __launch_bounds__(NW * 64, (NT >= 8 && NT <= 9) ? 3 : 1) void window_attention_mfma_kernel(const u16* __restrict__ q, const u16* __restrict__ k,
                                                                    const u16* __restrict__ v, u16* __restrict__ out,
                                                                    const float* __restrict__ bias, const float* __restrict__ mask, int N,
                                                                    int q_stride, int k_stride, int v_stride, int o_stride, int nW,
                                                                    float scale, const float* __restrict__ rel_table,
                                                                    const int* __restrict__ region, int ws) {
    // rel_table != NULL: the [N][N] bias is never materialised -- bias(i,j) = table[h][(yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1)]
    // (`swin_transformer.py:92-103,129-132`) is gathered from the head's (2ws-1)^2 table in LDS, and the shifted-window mask is
    // (region[w][i] != region[w][j]) ? -100 : 0 (`:222-229`) from the per-token region ids of the window type.  With the
    // expanded tables the kernel pulled 2 x N*N*4 bytes per (window, head) through L2 inside its softmax loop: 28 % of the
    // time of an unshifted block and 60 % more in a shifted one.
    constexpr int NP = NT * 16;
    constexpr int NP2 = ((NT + 1) / 2) * 32;  // keys padded to whole 32-key MFMA blocks
    constexpr int QK_PITCH = 64;              // bytes per Q/K row, no padding: the 16-byte slot of head dims 8 q .. 8 q + 7 is XORed with (row >> 1) & 3.
                                              // A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md,
                                              // LDS table): under THAT grouping this layout is conflict-free, while the 80-byte pitch of rounds 1-3 (conflict-free
                                              // for 16 consecutive lanes) was 2-way conflicted on every Q / K fragment read
    // V rows of 64 bytes like K; keys NP .. NP2 - 1 (the padding half of the last 32-key block) are zero rows.  The 32-byte half of head dims
    // 16 dt .. 16 dt + 15 is swapped on keys 4 .. 7 (mod 8): the two 4-key blocks a 32-lane half of a transposed read takes then sit on different banks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = smem + NP * QK_PITCH;
    char* Vs = smem + 2 * NP * QK_PITCH;
    constexpr int V_BYTES = NP2 * QK_PITCH;
    float* Tb = (float*)(smem + 2 * NP * QK_PITCH + V_BYTES);   // rel mode: [2T] table + -1e30 pad area, then key index, region id per token
    const int b = blockIdx.y, h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nth = NW * 64, nw = NW;          // 4 waves, or 3 (9 tiles)
    const int r16 = lane & 15, qd = lane >> 4;

    // ---- stage Q (scaled), K rows and V^T.  Every global load of the workgroup is issued before the first LDS store: left as a loop of
    // load -> store -> load ..., the three (four, with the bias table) round trips to memory ran one after the other and were half of a
    // workgroup's lifetime
    constexpr int SI = (NP * 4 + nth - 1) / nth;
    uint4 qv[SI], kv[SI], vv[SI];
    const u16* const qb = q + (long long)b * N * q_stride + h * 32;
    const u16* const kb = k + (long long)b * N * k_stride + h * 32;
    const u16* const vb = v + (long long)b * N * v_stride + h * 32;
#pragma unroll
    for (int i = 0; i < SI; ++i) {
        const int e = tid + i * nth;
        const int row = e >> 2, pc = e & 3;
        qv[i] = make_uint4(0, 0, 0, 0); kv[i] = make_uint4(0, 0, 0, 0); vv[i] = make_uint4(0, 0, 0, 0);
        if (row < N) {      // (workgroup-uniform bases + 32-bit lane offsets: the 64-bit address arithmetic per load was a tenth of the kernel's VALU work)
            qv[i] = *(const uint4*)(qb + (unsigned)(row * q_stride + pc * 8));
            kv[i] = *(const uint4*)(kb + (unsigned)(row * k_stride + pc * 8));
            vv[i] = *(const uint4*)(vb + (unsigned)(row * v_stride + pc * 8));
        }
    }
    // the window type's region ids (shifted blocks) travel with the first round of loads too: fetched where they are used, after the LDS stores,
    // they were one more round trip to memory on every workgroup's critical path (+1.2 us per workgroup: a shifted block cost 30-40 % more than
    // an unshifted one although only the windows of the last row / column hold more than one region)
    static_assert(NP <= nth, "one token per thread");
    const int* const rg = (rel_table && region) ? region + (long long)(b % nW) * N : nullptr;
    int rid_own = 0, rid_first = 0;
    if (rg) { rid_first = rg[0]; rid_own = tid < N ? rg[tid] : rid_first; }
    constexpr int TI = 6;                        // bias-table entries per thread held in registers ((2 ws - 1)^2 + 4 <= TI * nth: ws <= 16)
    float tbv[TI];
    const int T0 = (2 * ws - 1) * (2 * ws - 1);
    const bool tb_regs = rel_table && 2 * T0 + 4 <= TI * nth;
    if (tb_regs) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int e = tid + i * nth;
            tbv[i] = e < T0 ? rel_table[(long long)h * T0 + e] : -1e30f;
        }
    }
#pragma unroll
    for (int i = 0; i < SI; ++i) {
        const int e = tid + i * nth;
        if (e >= NP * 4) continue;
        const int row = e >> 2, pc = e & 3;
        if (row < N) {
            uint32_t* qu = (uint32_t*)&qv[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qu[j] = pack_bf16x2(__uint_as_float(qu[j] << 16) * scale, __uint_as_float(qu[j] & 0xffff0000u) * scale);
        }
        *(uint4*)(Qs + row * QK_PITCH + ((pc ^ ((row >> 1) & 3)) << 4)) = qv[i];
        *(uint4*)(Ks + row * QK_PITCH + ((pc ^ ((row >> 1) & 3)) << 4)) = kv[i];
        *(uint4*)(Vs + row * QK_PITCH + ((pc ^ (((row >> 2) & 1) << 1)) << 4)) = vv[i];
    }
    const int T = (2 * ws - 1) * (2 * ws - 1);
    int* Kidx = (int*)(Tb + 2 * T + 4);
    // shifted blocks: the mask -100 [region(i) != region(j)] (`swin_transformer.py:222-229`) enters as +100 [region(i) == region(j)] -- the same softmax,
    // every score of a row moved by the same 100 -- and that is a dot product of one-hot region vectors scaled by 10: one more MFMA per key tile
    // on a second operand image Oh [token][32 bf16] (dims 16 .. 31 zeros; rows laid out like Q / K) instead of a compare, a select and an add per
    // score (12 VALU per key tile: windows holding more than one region ran 1.3-1.8 x the time of the others)
    char* const Oh = smem + ((2 * NP * QK_PITCH + V_BYTES + (2 * T + 4) * 4 + NP * 4 + 15) & ~15);     // (an offset from `smem`: the compiler keeps it an LDS address)
    int mixed = 0;   // shifted block: does this window type hold more than one region?  (only the last row / column of windows do)
    if (rel_table) {
        if (tb_regs) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int e = tid + i * nth;
                if (e < 2 * T + 4) Tb[2 * T + 3 - e] = tbv[i];      // the table is kept REVERSED in LDS (see the score loop)
            }
        } else {
            for (int e = tid; e < 2 * T + 4; e += nth) Tb[2 * T + 3 - e] = e < T ? rel_table[(long long)h * T + e] : -1e30f;
        }
        if (tid < NP) {
            const int e = tid;
            const int yj = (int)(((unsigned)e * (65536u / (unsigned)ws + 1u)) >> 16), xj = e - yj * ws;      // e / ws for e < 256, ws <= 16 (exact)
            Kidx[e] = e < N ? yj * (2 * ws - 1) + xj : -T - 3;                // padded keys index the -1e30 area
          if (rg) {
            uint4 lo, hi;          // 10.0 (bf16 0x4120) at halfword rid_own of 16
            const unsigned hv = (rid_own & 1) ? 0x41200000u : 0x4120u;
            const int hw = rid_own >> 1;
            lo.x = hw == 0 ? hv : 0u; lo.y = hw == 1 ? hv : 0u; lo.z = hw == 2 ? hv : 0u; lo.w = hw == 3 ? hv : 0u;
            hi.x = hw == 4 ? hv : 0u; hi.y = hw == 5 ? hv : 0u; hi.z = hw == 6 ? hv : 0u; hi.w = hw == 7 ? hv : 0u;
            const int osw = (e >> 1) & 3;
            *(uint4*)(Oh + e * QK_PITCH + ((0 ^ osw) << 4)) = lo;
            *(uint4*)(Oh + e * QK_PITCH + ((1 ^ osw) << 4)) = hi;
            *(uint4*)(Oh + e * QK_PITCH + ((2 ^ osw) << 4)) = make_uint4(0, 0, 0, 0);
            *(uint4*)(Oh + e * QK_PITCH + ((3 ^ osw) << 4)) = make_uint4(0, 0, 0, 0);
            if (rid_own != rid_first) mixed = 1;
          }
        }
        mixed = __syncthreads_or(mixed);
    }
    if (NP2 > NP) {  // zero rows for the keys of the padding half-block (their probabilities are zeros, but 0 x stale LDS could be NaN)
        for (int e = tid; e < (NP2 - NP) * 4; e += nth) *(uint4*)(Vs + NP * QK_PITCH + e * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // what a lane needs of the key side does not depend on the query tile: the table index of its 4 keys per key tile (ws % 4 == 0: 4 consecutive
    // keys share an image row, their table entries are 4 consecutive floats)
    const int sw = (qd ^ ((r16 >> 1) & 3)) << 4;          // ((16 t + r16) >> 1) & 3 == (r16 >> 1) & 3
    const bool row4 = rel_table && (ws & 3) == 0;
    int kidx[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) kidx[kt] = row4 ? Kidx[kt * 16 + qd * 4] : 0;

#pragma unroll 1
    for (int qt = wave; qt < NT; qt += nw) {
        const bf16x8_t qf = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Qs + (qt * 16 + r16) * QK_PITCH + sw));
        const int query = qt * 16 + r16;
        // S^T = K Q^T with the bias (+ mask) as the accumulator's initial value: a lane owns query `query` and keys 16 kt + 4 qd .. + 3 of every key tile.
        // Bias (+ mask) rows are padded to [NP][NP] on the host: padded keys hold -1e30
        f32x4 s[NT];
        const float* brow = bias ? bias + ((long long)h * NP + query) * NP + qd * 4 : nullptr;
        const float* mrow = mask ? mask + ((long long)(b % nW) * NP + query) * NP + qd * 4 : nullptr;
        const int qq = query < N ? query : 0;
        const int yi = (int)(((unsigned)qq * (65536u / (unsigned)ws + 1u)) >> 16), xi = qq - yi * ws;      // (exact for qq < 256, ws <= 16)
        const int qbase = (yi + ws - 1) * (2 * ws - 1) + xi + ws - 1;
        // table entry qbase - k sits at Tb[2T + 3 - qbase + k]: the 4 consecutive keys of a lane (entries qbase - k0, ... - 3) are 4 ASCENDING floats that land
        // in the accumulator's register order -- read in the table's own order they arrived reversed and cost a v_mov each (36 per query tile)
        const float* const tbq = Tb + (2 * T + 3 - qbase);
        // straight-line copies of the key loop per bias form (0: expanded [N][N] rows from memory, 1: table, 4 keys per image row, 2: table, any window
        // size), with and without the region product: left as run-time branches inside the loop they were re-evaluated for every key tile and kept
        // the compiler from issuing the nine tiles' LDS reads ahead of their MFMAs
        auto scores = [&](auto form, auto with_regions) {
            constexpr int FORM = decltype(form)::value;
            constexpr bool REG = decltype(with_regions)::value;
            bf16x8_t qo;
            if constexpr (REG) qo = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Oh + qq * QK_PITCH + ((qd ^ ((qq >> 1) & 3)) << 4)));
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 b4;
                if constexpr (FORM == 1) {
                    const float* tp = tbq + kidx[kt];
                    b4 = (f32x4){tp[0], tp[1], tp[2], tp[3]};
                } else if constexpr (FORM == 2) {
                    const int4 kk = *(const int4*)(Kidx + kt * 16 + qd * 4);
                    b4 = (f32x4){tbq[kk.x], tbq[kk.y], tbq[kk.z], tbq[kk.w]};
                } else {
                    const float4 bb = *(const float4*)(brow + kt * 16);
                    b4 = (f32x4){bb.x, bb.y, bb.z, bb.w};
                    if (mrow) {
                        const float4 mm = *(const float4*)(mrow + kt * 16);
                        b4 += (f32x4){mm.x, mm.y, mm.z, mm.w};
                    }
                }
                const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Ks + (kt * 16 + r16) * QK_PITCH + sw));
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, b4, 0, 0, 0);
                if constexpr (REG) {
                    const bf16x8_t ko = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Oh + (kt * 16 + r16) * QK_PITCH + sw));
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ko, qo, s[kt], 0, 0, 0);
                }
            }
        };
        using std::integral_constant;
        if (!rel_table) scores(integral_constant<int, 0>{}, std::false_type{});
        else if (row4) {
            if (mixed) scores(integral_constant<int, 1>{}, std::true_type{});
            else scores(integral_constant<int, 1>{}, std::false_type{});
        } else {
            if (mixed) scores(integral_constant<int, 2>{}, std::true_type{});
            else scores(integral_constant<int, 2>{}, std::false_type{});
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { mx = wa_max3(mx, s[kt][0], s[kt][1]); mx = wa_max3(mx, s[kt][2], s[kt][3]); }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // p = 2^((s - mx) log2 e): one packed fma + two v_exp_f32 per pair of scores, the row sum in packed adds
        constexpr float L2E = 1.4426950408889634f;
        const float nmx = -mx * L2E;
        mt4_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const mt4_f32x2 t0 = (mt4_f32x2){s[kt][0], s[kt][1]} * L2E + nmx, t1 = (mt4_f32x2){s[kt][2], s[kt][3]} * L2E + nmx;
            const mt4_f32x2 e0 = {__builtin_amdgcn_exp2f(t0.x), __builtin_amdgcn_exp2f(t0.y)}, e1 = {__builtin_amdgcn_exp2f(t1.x), __builtin_amdgcn_exp2f(t1.y)};
            sum2 += e0;
            sum2 += e1;
            s[kt] = (f32x4){e0.x, e0.y, e1.x, e1.y};
        }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        // O^T = V^T P^T: lane 4 q + p of the 16-lane group qd supplies the 8 bytes of key 32 kb (+ 16) + 4 qd + q, head dims 16 dt + 4 p .. + 3, and
        // receives head dim 16 dt + r16 of the block's 4 keys -- the A-operand half the lane needs
        const int vkey = qd * 4 + (r16 >> 2), vp = r16 & 3;
#pragma unroll
        for (int kb = 0; kb < (NT + 1) / 2; ++kb) {
            const int k0 = 2 * kb, k1 = 2 * kb + 1;
            uint4 pf;
            pf.x = pack_bf16x2(s[k0][0], s[k0][1]);
            pf.y = pack_bf16x2(s[k0][2], s[k0][3]);
            if (k1 < NT) { pf.z = pack_bf16x2(s[k1][0], s[k1][1]); pf.w = pack_bf16x2(s[k1][2], s[k1][3]); }
            else { pf.z = 0; pf.w = 0; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // (keys 32 kb + vkey and + 16 share (key >> 2) & 1: one swizzle term)
                const char* vr = Vs + (kb * 32 + vkey) * QK_PITCH + ((((dt << 1) | (vp >> 1)) ^ (((vkey >> 2) & 1) << 1)) << 4) + (vp & 1) * 8;
                const uint2 v0 = wa_read_tr(vr);                      // keys 32 kb + 4 qd .. + 3
                const uint2 v1 = wa_read_tr(vr + 16 * QK_PITCH);      // keys 32 kb + 16 + 4 qd .. + 3
                const uint4 vf = make_uint4(v0.x, v0.y, v1.x, v1.y);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf), o[dt], 0, 0, 0);
            }
        }
        if (query < N) {
            u16* op = out + ((long long)b * N + query) * o_stride + h * 32 + qd * 4;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *(uint2*)(op + dt * 16) = make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
        }
    }
}

static int window_attention_launch(const void* q, const void* k, const void* v, void* out, const float* bias_padded,
                                   const float* mask_padded, const float* rel_table, const int32_t* region, int32_t ws, int32_t B, int32_t H,
                                   int32_t N, int32_t q_stride, int32_t k_stride, int32_t v_stride, int32_t o_stride, int32_t nW, float scale,
                                   void* stream) {
    mt4_clear_error();
    if (!q || !k || !v || !out || (!bias_padded && !rel_table) || B <= 0 || H <= 0 || N <= 0 || N > 256) return MT4_EINVAL;
    if ((mask_padded || region) && nW <= 0) return MT4_EINVAL;
    if (rel_table && (ws <= 0 || ws * ws != N)) return MT4_EINVAL;
    if (B > 65535) return MT4_EUNSUPPORTED;
    if ((q_stride | k_stride | v_stride) % 8 || o_stride % 4 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || ((uintptr_t)out & 7))
        return MT4_EALIGN;
    const int NT = (N + 15) / 16;
    const int NP = NT * 16, NP2 = ((NT + 1) / 2) * 32;
    size_t lds = (size_t)2 * NP * 64 + (size_t)NP2 * 64;
    if (rel_table) lds += ((size_t)2 * (2 * ws - 1) * (2 * ws - 1) + 4) * 4 + (size_t)NP * 4 + 16 + (size_t)NP * 64;
    if (!rel_table) ws = 1;
    const dim3 grid(H, B), block(256);
    hipStream_t s = (hipStream_t)stream;
#define WA(NTV) hipLaunchKernelGGL((window_attention_mfma_kernel<NTV>), grid, block, lds, s, (const u16*)q, (const u16*)k, (const u16*)v, (u16*)out, bias_padded, mask_padded, N, q_stride, k_stride, v_stride, o_stride, nW, scale, rel_table, region, ws)
    switch (NT) {
        case 1: WA(1); break; case 2: WA(2); break; case 3: WA(3); break; case 4: WA(4); break;
        case 5: WA(5); break; case 6: WA(6); break; case 7: WA(7); break; case 8: WA(8); break;
        case 9:       // three waves: 9 query tiles over 4 waves leave three of them idle for a third of the loop
            hipLaunchKernelGGL((window_attention_mfma_kernel<9, 3>), grid, dim3(192), lds, s, (const u16*)q, (const u16*)k, (const u16*)v, (u16*)out, bias_padded,
                               mask_padded, N, q_stride, k_stride, v_stride, o_stride, nW, scale, rel_table, region, ws);
            break;
        case 10: WA(10); break; case 11: WA(11); break; case 12: WA(12); break;
        case 13: WA(13); break; case 14: WA(14); break; case 15: WA(15); break; default: WA(16); break;
    }
#undef WA
    return mt4_check_launch();
}

extern "C" int mt4_window_attention_bf16(const void* q, const void* k, const void* v, void* out, const float* bias_padded,
                                         const float* mask_padded, int32_t B, int32_t H, int32_t N, int32_t q_stride, int32_t k_stride,
                                         int32_t v_stride, int32_t o_stride, int32_t nW, float scale, void* stream) {
    return window_attention_launch(q, k, v, out, bias_padded, mask_padded, nullptr, nullptr, 0, B, H, N, q_stride, k_stride, v_stride, o_stride,
                                   nW, scale, stream);
}

extern "C" int mt4_window_attention_rel_bf16(const void* q, const void* k, const void* v, void* out, const float* rel_table,
                                             const int32_t* region, int32_t ws, int32_t B, int32_t H, int32_t q_stride, int32_t k_stride,
                                             int32_t v_stride, int32_t o_stride, int32_t nW, float scale, void* stream) {
    return window_attention_launch(q, k, v, out, nullptr, nullptr, rel_table, region, ws, B, H, ws * ws, q_stride, k_stride, v_stride, o_stride,
                                   nW, scale, stream);
}
