// bf16-operand training mode of the spatial stage (Spatial_cnn/run.py:145-224 with the convolutions' GEMM operands in bf16): activations and
// activation gradients are stored in bf16, every sum runs in fp32 (MFMA accumulators) or fp64 (BatchNorm reductions), the master weights, their
// gradients and the optimizer stay fp32.  Forward and data-gradient convolutions are mt4_conv_nhwc in bf16; this file holds what that mode adds:
//   * the weight gradient on bf16 MFMA (`v_mfma_f32_16x16x32_bf16`, operands transposed on the way out of LDS by `ds_read_b64_tr_b16`),
//   * train-mode BatchNorm forward / backward with bf16 tensors (the stem keeps an fp32 convolution output: its 7x7x3 geometry runs in fp32),
//   * pooling backward with bf16 gradients, and the fp32 -> bf16 copy of a packed weight matrix.
#include "mt4_common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace {

// ------------------------------------------------------------------------------------------------ element access
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<u16>(const u16* p) {
    const uint2 v = *(const uint2*)p;
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void st4<u16>(u16* p, float4 v) { *(uint2*)p = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }

// ------------------------------------------------------------------------------------------------ BatchNorm2d (training), typed tensors
// same arithmetic as train2d_kernels.hip: float64 per-channel reductions, a thread owns 4 consecutive channels.  The reductions run 1024-thread
// blocks (64 channels x 64 row phases) and at most ~512 of them: every block ends in 128 fp64 atomics on its slab's 128 addresses, and with 2048
// blocks of 256 threads on a 64-channel tensor those 2048-deep same-address chains (~25 ns a link at the L2) took longer than the stream --
// 52-58 us for 59 MB where the apply kernel moves twice the bytes in 20-29 us (profiles/r03_bn_training_kernels.txt)
__device__ __forceinline__ void bn_block_reduce(double (&acc)[8], double* __restrict__ sums, int C, int c0) {
    __shared__ double red[16][8][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                            // the 4 row phases of a wave
        acc[j] += __shfl_xor(acc[j], 16);
        acc[j] += __shfl_xor(acc[j], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[w][j][lane] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int g = threadIdx.x & 15, j = threadIdx.x >> 4;
        double t = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) t += red[rr][j][g];
        const int c = c0 + g * 4 + (j & 3);
        if (c < C) atomicAdd(sums + (j >> 2) * C + c, t);
    }
}

template <typename TZ>
__global__ __launch_bounds__(1024) void bn_stats_t_kernel(const TZ* __restrict__ x, double* __restrict__ sums, long long M, int C) {
    const int c0 = blockIdx.x * 64, c = c0 + (threadIdx.x & 15) * 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        // 4 rows in flight per thread: with one 8-byte load outstanding per lane the 32 waves of a CU hold 16 KB in flight, a latency bound of
        // ~2.6 TB/s (measured); rows are still added in ascending order, so the sums keep their bits
        const long long st = (long long)gridDim.y * 64;
        long long m = (long long)blockIdx.y * 64 + (threadIdx.x >> 4);
        auto add = [&](const float4 v) {
            acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
            acc[4] += (double)v.x * v.x; acc[5] += (double)v.y * v.y; acc[6] += (double)v.z * v.z; acc[7] += (double)v.w * v.w;
        };
        for (; m + 3 * st < M; m += 4 * st) {
            const float4 v0 = ld4<TZ>(x + m * C + c), v1 = ld4<TZ>(x + (m + st) * C + c), v2 = ld4<TZ>(x + (m + 2 * st) * C + c),
                         v3 = ld4<TZ>(x + (m + 3 * st) * C + c);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; m < M; m += st) add(ld4<TZ>(x + m * C + c));
    }
    bn_block_reduce(acc, sums, C, c0);
}

int bn_reduce_slabs(long long M, int C) {                                      // 64-row slabs of the 1024-thread reductions
    long long gy = (M + 63) / 64;
    const long long cap = (512 + cdiv(C, 64) - 1) / cdiv(C, 64);
    if (gy > cap) gy = cap;
    return gy < 1 ? 1 : (int)gy;
}

int bn_row_slabs(long long M, int C) {
    long long gy = (M + 63) / 64;
    const long long cap = (2048 + cdiv(C, 64) - 1) / cdiv(C, 64);
    if (gy > cap) gy = cap;
    return gy < 1 ? 1 : (int)gy;
}

__global__ void bn_finalize_t_kernel(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                     float* __restrict__ run_var, long long M, int C, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = sums[c] / (double)M;
    double var = sums[C + c] / (double)M - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    }
}

// y (bf16) = act( (x - mean) * invstd * gamma + beta [+ residual (bf16)] ).  Column slabs like the reductions: a thread owns 4 consecutive
// channels and walks rows, its per-channel parameters in registers -- one element group per thread re-loaded 64 bytes of parameters for every
// 16 bytes of tensor traffic and ran at 3.8-4.1 TB/s where the bare stream measures 5-6 (tools/bn_stream_micro.hip).  Same expressions: same bits.
// FIN: mean / invstd come from the channel sums a convolution's epilogue left (mt4_conv_desc.stat_sums): the first 64 threads fold the replicas and
// evaluate bn_finalize_t_kernel's expressions for the workgroup's 64 channels (float64: done by every thread it would cost more than the stream),
// the row-slab 0 workgroups also write mean / invstd for the backward and update the running statistics
template <typename TZ, bool FIN>
__global__ __launch_bounds__(256) void bn_apply_t_kernel(const TZ* __restrict__ x, float* __restrict__ mean, float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, const u16* __restrict__ res,
                                                         u16* __restrict__ y, long long M, int C, int relu, const double* __restrict__ sums,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var, float momentum, float eps) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
    float4 mu, is;
    if constexpr (FIN) {
        __shared__ __attribute__((aligned(16))) float s_mu[64], s_is[64];
        const int ch = blockIdx.x * 64 + threadIdx.x;
        if (threadIdx.x < 64 && ch < C) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < MT4_STAT_REPLICAS; ++r) { s1 += sums[(long long)(2 * r) * C + ch]; s2 += sums[(long long)(2 * r + 1) * C + ch]; }
            const double m_ = s1 / (double)M;
            double var = s2 / (double)M - m_ * m_;
            var = var > 0.0 ? var : 0.0;
            const float mf = (float)m_, isf = (float)(1.0 / sqrt(var + (double)eps));
            s_mu[threadIdx.x] = mf;
            s_is[threadIdx.x] = isf;
            if (blockIdx.y == 0) {
                mean[ch] = mf;
                invstd[ch] = isf;
                if (run_mean) {
                    run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * mf;
                    run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
                }
            }
        }
        __syncthreads();
        if (c >= C) return;
        mu = *(const float4*)(s_mu + (threadIdx.x & 15) * 4);
        is = *(const float4*)(s_is + (threadIdx.x & 15) * 4);
    } else {
        if (c >= C) return;
        mu = *(const float4*)(mean + c);
        is = *(const float4*)(invstd + c);
    }
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    const long long st = (long long)gridDim.y * 16;
    long long m = (long long)blockIdx.y * 16 + (threadIdx.x >> 4);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](long long i, const float4 xv, const float4 r) {
        float4 o = make_float4((xv.x - mu.x) * is.x * g.x + b.x, (xv.y - mu.y) * is.y * g.y + b.y, (xv.z - mu.z) * is.z * g.z + b.z,
                               (xv.w - mu.w) * is.w * g.w + b.w);
        if (res) { o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4<u16>(y + i, o);
    };
    for (; m + 3 * st < M; m += 4 * st) {
        const long long i0 = m * C + c, i1 = (m + st) * C + c, i2 = (m + 2 * st) * C + c, i3 = (m + 3 * st) * C + c;
        const float4 x0 = ld4<TZ>(x + i0), x1 = ld4<TZ>(x + i1), x2 = ld4<TZ>(x + i2), x3 = ld4<TZ>(x + i3);
        float4 r0 = zero, r1 = zero, r2 = zero, r3 = zero;
        if (res) { r0 = ld4<u16>(res + i0); r1 = ld4<u16>(res + i1); r2 = ld4<u16>(res + i2); r3 = ld4<u16>(res + i3); }
        put(i0, x0, r0); put(i1, x1, r1); put(i2, x2, r2); put(i3, x3, r3);
    }
    for (; m < M; m += st) {
        const long long i0 = m * C + c;
        put(i0, ld4<TZ>(x + i0), res ? ld4<u16>(res + i0) : zero);
    }
}

// relu: 0 none; 1 the ReLU gate is read from the stored output y; 2 it is recomputed from x -- (x - mean) * invstd * gamma + beta > 0, the
// forward's own fp32 expression -- for units without a residual input: y need not be read (2 of the 6 / 8 bytes per element these HBM-bound
// kernels move)
template <typename TZ>
__global__ __launch_bounds__(1024) void bn_bwd_reduce_t_kernel(const u16* __restrict__ dy, const u16* __restrict__ y, const TZ* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ sums,
                                                              long long M, int C, int relu) {
    const int c0 = blockIdx.x * 64, c = c0 + (threadIdx.x & 15) * 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
        float4 ga = make_float4(0, 0, 0, 0), be = ga;
        if (relu == 2) { ga = *(const float4*)(gamma + c); be = *(const float4*)(beta + c); }
        const long long st = (long long)gridDim.y * 64;
        long long m = (long long)blockIdx.y * 64 + (threadIdx.x >> 4);
        auto add = [&](float4 g, const float4 xv, const float4 yv) {
            if (relu == 2) {
                if (!((xv.x - mu.x) * is.x * ga.x + be.x > 0.f)) g.x = 0.f;
                if (!((xv.y - mu.y) * is.y * ga.y + be.y > 0.f)) g.y = 0.f;
                if (!((xv.z - mu.z) * is.z * ga.z + be.z > 0.f)) g.z = 0.f;
                if (!((xv.w - mu.w) * is.w * ga.w + be.w > 0.f)) g.w = 0.f;
            } else if (relu) {
                if (!(yv.x > 0.f)) g.x = 0.f;
                if (!(yv.y > 0.f)) g.y = 0.f;
                if (!(yv.z > 0.f)) g.z = 0.f;
                if (!(yv.w > 0.f)) g.w = 0.f;
            }
            acc[0] += (double)g.x; acc[1] += (double)g.y; acc[2] += (double)g.z; acc[3] += (double)g.w;
            acc[4] += (double)g.x * (double)((xv.x - mu.x) * is.x); acc[5] += (double)g.y * (double)((xv.y - mu.y) * is.y);
            acc[6] += (double)g.z * (double)((xv.z - mu.z) * is.z); acc[7] += (double)g.w * (double)((xv.w - mu.w) * is.w);
        };
        const bool rd_y = relu == 1;
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
        // 2 rows x up to 3 tensors in flight per thread (see bn_stats_t_kernel); rows are added in ascending order
        for (; m + st < M; m += 2 * st) {
            const long long i0 = m * C + c, i1 = (m + st) * C + c;
            const float4 g0 = ld4<u16>(dy + i0), g1 = ld4<u16>(dy + i1);
            const float4 x0 = ld4<TZ>(x + i0), x1 = ld4<TZ>(x + i1);
            float4 y0 = one, y1 = one;
            if (rd_y) { y0 = ld4<u16>(y + i0); y1 = ld4<u16>(y + i1); }
            add(g0, x0, y0); add(g1, x1, y1);
        }
        for (; m < M; m += st) {
            const long long i0 = m * C + c;
            add(ld4<u16>(dy + i0), ld4<TZ>(x + i0), rd_y ? ld4<u16>(y + i0) : one);
        }
    }
    bn_block_reduce(acc, sums, C, c0);
}

// dx has the type of x (the convolution output: bf16, or fp32 for the stem); dres = the gated gradient, bf16.  Column slabs, parameters and the two
// per-channel means of the reduction in registers (see bn_apply_t_kernel)
template <typename TZ>
__global__ __launch_bounds__(256) void bn_bwd_apply_t_kernel(const u16* __restrict__ dy, const u16* __restrict__ y, const TZ* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const double* __restrict__ sums, TZ* __restrict__ dx, u16* __restrict__ dres,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, long long M, int C, int relu) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
    if (c >= C) return;
    const double invM = 1.0 / (double)M;
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c), ga = *(const float4*)(gamma + c);
    float4 bg = make_float4(0, 0, 0, 0);
    if (relu == 2) bg = *(const float4*)(beta + c);
    const float4 m1 = make_float4((float)(sums[c] * invM), (float)(sums[c + 1] * invM), (float)(sums[c + 2] * invM), (float)(sums[c + 3] * invM));
    const float4 m2 = make_float4((float)(sums[C + c] * invM), (float)(sums[C + c + 1] * invM), (float)(sums[C + c + 2] * invM),
                                  (float)(sums[C + c + 3] * invM));
    if (blockIdx.y == 0 && (threadIdx.x >> 4) == 0) {
        *(float4*)(dbeta + c) = make_float4((float)sums[c], (float)sums[c + 1], (float)sums[c + 2], (float)sums[c + 3]);
        *(float4*)(dgamma + c) = make_float4((float)sums[C + c], (float)sums[C + c + 1], (float)sums[C + c + 2], (float)sums[C + c + 3]);
    }
    const long long st = (long long)gridDim.y * 16;
    long long m = (long long)blockIdx.y * 16 + (threadIdx.x >> 4);
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    const bool rd_y = relu == 1;
    auto put = [&](long long i, float4 g, const float4 xv, const float4 yv) {
        if (relu == 2) {
            if (!((xv.x - mu.x) * is.x * ga.x + bg.x > 0.f)) g.x = 0.f;
            if (!((xv.y - mu.y) * is.y * ga.y + bg.y > 0.f)) g.y = 0.f;
            if (!((xv.z - mu.z) * is.z * ga.z + bg.z > 0.f)) g.z = 0.f;
            if (!((xv.w - mu.w) * is.w * ga.w + bg.w > 0.f)) g.w = 0.f;
        } else if (relu) {
            if (!(yv.x > 0.f)) g.x = 0.f;
            if (!(yv.y > 0.f)) g.y = 0.f;
            if (!(yv.z > 0.f)) g.z = 0.f;
            if (!(yv.w > 0.f)) g.w = 0.f;
        }
        float4 o;
        o.x = ga.x * is.x * (g.x - m1.x - (xv.x - mu.x) * is.x * m2.x);
        o.y = ga.y * is.y * (g.y - m1.y - (xv.y - mu.y) * is.y * m2.y);
        o.z = ga.z * is.z * (g.z - m1.z - (xv.z - mu.z) * is.z * m2.z);
        o.w = ga.w * is.w * (g.w - m1.w - (xv.w - mu.w) * is.w * m2.w);
        st4<TZ>(dx + i, o);
        if (dres) st4<u16>(dres + i, g);
    };
    for (; m + st < M; m += 2 * st) {
        const long long i0 = m * C + c, i1 = (m + st) * C + c;
        const float4 g0 = ld4<u16>(dy + i0), g1 = ld4<u16>(dy + i1);
        const float4 x0 = ld4<TZ>(x + i0), x1 = ld4<TZ>(x + i1);
        float4 y0 = one, y1 = one;
        if (rd_y) { y0 = ld4<u16>(y + i0); y1 = ld4<u16>(y + i1); }
        put(i0, g0, x0, y0); put(i1, g1, x1, y1);
    }
    for (; m < M; m += st) {
        const long long i0 = m * C + c;
        put(i0, ld4<u16>(dy + i0), ld4<TZ>(x + i0), rd_y ? ld4<u16>(y + i0) : one);
    }
}

// ------------------------------------------------------------------------------------------------ Conv2d weight gradient, bf16 operands
//   dW[n][tap][c] += sum over output pixels p of dy[p][n] * x[in(p, tap)][c]        n = output channel, c = input channel
// A GEMM whose reduction index (the pixel) is the SLOW index of both operands in memory ([pixel][channel] rows).  The MFMA wants 8 consecutive
// reduction elements per lane for both, i.e. both operands transposed: each is staged [pixel][64 channels] (128-byte rows, as it arrives from
// HBM) and read with ds_read_b64_tr_b16 -- per 16-lane group a block of 4 pixels x 16 channels, delivered channel-major: lane i of the group
// gets channel i of the 4 pixels, exactly the A (dy^T) and B (x) fragment halves of v_mfma_f32_16x16x32_bf16.
// A workgroup (4 waves) owns a 64 x 64 (n, c) tile for ONE kernel row kh (its K taps) and walks spatial tiles of TH x 16 output pixels of the
// frames (its share of them); per tile it stages the dy tile and the TH input rows that kernel row reads (16 S + K - 1 pixels each, zeros
// outside the image), then per 32 pixels (two tile rows) runs K x 4 MFMAs per wave: wave (wm, wn) holds dW[32 n][kh][K][32 c] in registers
// across all its tiles and adds it to the fp32 gradient buffer with atomics at the end (the split of the pixel range over workgroups makes the sum order
// run-dependent, like mt4_wgrad_conv2d_f32's).
// LDS rows are 128 B; the 16-byte chunk index is XOR-ed with 2 * (bit 1 | bit 3 << 1) of the row index: the 8 (pixel, group) blocks a 32-lane
// half of a transposed read touches (pixels R .. R + 3 and R + 8 .. R + 11) then fall on 8 different 32-byte bank spans for any R.  The halo
// tile's row pitch is a multiple of 16 pixels, so the swizzle of a fragment address depends on the lane and the column shift kw only: six
// per-lane offsets, everything else is an immediate.
constexpr int WG_ROWB = 128;

__device__ __forceinline__ int wg_swz(int row, int chunk) { return row * WG_ROWB + ((chunk ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)) << 4); }

// one MFMA fragment = two transposed reads (4 + 4 reduction elements per lane), through the compiler's builtin so that it schedules the reads
// against the MFMAs and counts lgkmcnt itself
typedef short v4s_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 ds_read_tr_frag(const char* p_lo, const char* p_hi) {
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(uintptr_t)p_lo);
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(uintptr_t)p_hi);
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

struct WgK {
    const u16* dy;
    const u16* x;
    float* dw;
    int B, H, W, Cin, Ho, Wo, Cout;
    int kpad, tapw;           // fp32 packed row length, elements per tap
    int tiles_h, tiles_w, ntiles, nsplit, cin_tiles;
};

template <int K, int S, int TH, int BMT, int BNT, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void wgrad_bf16_kernel(const WgK a) {
    constexpr bool PREFETCH = true;
    constexpr int PAD = K / 2;
    constexpr int IW = S * 15 + K;                              // input pixels per tile row (with the column halo)
    constexpr int PITCH = (IW + 15) / 16 * 16;                  // pixels per LDS row of the input tile
    constexpr int DY_BYTES = TH * 16 * WG_ROWB;                 // one 64-channel plane of the dy tile
    constexpr int X_BYTES = TH * PITCH * WG_ROWB;               // one 64-channel plane of the input tile
    constexpr int MT = 2 * BMT, NT = 2 * BNT;                   // MFMA tiles per wave: the workgroup tile is (64 BMT) x (64 BNT), waves 2 x 2
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sdy = smem;
    char* const sx = smem + BMT * DY_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    // workgroup = ((64 BMT) x (64 BNT) (n, c) tile, kernel row kh, share of the spatial tiles): a kernel row's K taps = K x MT x NT accumulators
    int bid = blockIdx.x;
    const int split = bid % a.nsplit;
    bid /= a.nsplit;
    const int kh = bid % K;
    const int pair = bid / K;
    const int n0 = (pair / a.cin_tiles) * (64 * BMT), c0 = (pair % a.cin_tiles) * (64 * BNT);

    f32x4 acc[K][MT][NT];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[t][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets.  A block's row for this lane: output pixel (tile row 2 s + (g >> 1), column 8 (g & 1) + 4 h + q); it supplies
    // the 8 bytes of channels 4 p .. 4 p + 3 of the fragment's 16 channels: plane ch / 64, chunk (ch % 64) / 8 + (p >> 1), half p & 1
    int yo[2][MT], xo[K][2][NT];          // [half h][channel tile of the wave]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = 8 * (g & 1) + 4 * h + q;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int ch = wm * (32 * BMT) + mi * 16;
            yo[h][mi] = (ch >> 6) * DY_BYTES + wg_swz((g >> 1) * 16 + col, (ch & 63) / 8 + (p >> 1)) + (p & 1) * 8;
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int ch = wn * (32 * BNT) + ni * 16;
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
                xo[kw][h][ni] = (ch >> 6) * X_BYTES + wg_swz(S * col + kw, (ch & 63) / 8 + (p >> 1)) + (p & 1) * 8 + (g >> 1) * (PITCH * WG_ROWB);
        }
    }

    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    // staging through registers, one tile ahead: the next tile's global loads are in flight while this tile's MFMAs run
    constexpr int NY = TH * 16 / 32, NX = TH * PITCH / 32;      // row passes of 32 rows (256 threads x 16 bytes = 32 rows of 128 B)
    uint4 ry[NY][BMT], rx[NX][BNT];
    auto load_tile = [&](int t) {
        const int tpi = a.tiles_h * a.tiles_w;
        const int img = t / tpi;
        const int tr = t - img * tpi;
        const int th = tr / a.tiles_w;
        const int ho0 = th * TH, wo0 = (tr - th * a.tiles_w) * 16;
        // dy tile: TH x 16 output pixels x (64 BMT) output channels
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int r = ld_row + 32 * i;
            const int ho = ho0 + (r >> 4), wo = wo0 + (r & 15);
            const bool ok = ho < a.Ho && wo < a.Wo;
            const u16* src = a.dy + (((long long)img * a.Ho + ho) * a.Wo + wo) * a.Cout + n0 + ld_chunk * 8;
#pragma unroll
            for (int pm = 0; pm < BMT; ++pm) {
                ry[i][pm] = make_uint4(0, 0, 0, 0);
                if (ok && n0 + pm * 64 + ld_chunk * 8 < a.Cout) ry[i][pm] = *(const uint4*)(src + pm * 64);      // (channel counts need not fill the tile)
            }
        }
        // input tile for this kernel row: LDS row iy = the input row of output row ho0 + iy, IW pixels (rows of PITCH pixels), zeros outside
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int r = ld_row + 32 * i;
            const int iy = r / PITCH, ix = r - iy * PITCH;
            const int hi = S * (ho0 + iy) - PAD + kh, wi = S * wo0 - PAD + ix;
            const bool ok = ix < IW && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const u16* src = a.x + (((long long)img * a.H + hi) * a.W + wi) * a.Cin + c0 + ld_chunk * 8;
#pragma unroll
            for (int pn = 0; pn < BNT; ++pn) {
                rx[i][pn] = make_uint4(0, 0, 0, 0);
                if (ok && c0 + pn * 64 + ld_chunk * 8 < a.Cin) rx[i][pn] = *(const uint4*)(src + pn * 64);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NY; ++i)
#pragma unroll
            for (int pm = 0; pm < BMT; ++pm) *(uint4*)(sdy + pm * DY_BYTES + wg_swz(ld_row + 32 * i, ld_chunk)) = ry[i][pm];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int pn = 0; pn < BNT; ++pn) *(uint4*)(sx + pn * X_BYTES + wg_swz(ld_row + 32 * i, ld_chunk)) = rx[i][pn];
    };
    if (split < a.ntiles) load_tile(split);
    for (int t = split; t < a.ntiles; t += a.nsplit) {
        __syncthreads();                                   // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (PREFETCH && t + a.nsplit < a.ntiles) load_tile(t + a.nsplit);
#pragma unroll
        for (int s = 0; s < TH / 2; ++s) {
            uint4 fa[MT];
            const char* yb = sdy + s * (32 * WG_ROWB);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) fa[mi] = ds_read_tr_frag(yb + yo[0][mi], yb + yo[1][mi]);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                uint4 fb[NT];
                const char* xb = sx + (2 * s) * (PITCH * WG_ROWB);
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) fb[ni] = ds_read_tr_frag(xb + xo[kw][0][ni], xb + xo[kw][1][ni]);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[kw][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[mi]), __builtin_bit_cast(bf16x8_t, fb[ni]),
                                                                                  acc[kw][mi][ni], 0, 0, 0);
            }
        }
        if (!PREFETCH && t + a.nsplit < a.ntiles) load_tile(t + a.nsplit);      // (no registers to spare: staged after the MFMAs)
    }
    // ---- dW += the wave's block: lane (r16 = c, q4 = lane >> 4) holds output channels 4 q4 + e of input channel r16
    const int r16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wm * (32 * BMT) + mi * 16 + q4 * 4 + e;
                    const int c = c0 + wn * (32 * BNT) + ni * 16 + r16;
                    if (n < a.Cout && c < a.Cin) atomicAdd(a.dw + (long long)n * a.kpad + (kh * K + kw) * a.tapw + c, acc[kw][mi][ni][e]);
                }
}

template <int K, int S, int TH, int BMT, int BNT, int OCC = 2>
int launch_wgrad(WgK a, hipStream_t stream) {
    constexpr int IW = S * 15 + K, PITCH = (IW + 15) / 16 * 16;
    constexpr int LDS = BMT * TH * 16 * WG_ROWB + BNT * TH * PITCH * WG_ROWB;
    static_assert(LDS * OCC <= 160 * 1024, "OCC workgroups per CU");
    a.tiles_h = cdiv(a.Ho, TH);
    a.tiles_w = cdiv(a.Wo, 16);
    a.ntiles = a.B * a.tiles_h * a.tiles_w;
    a.cin_tiles = cdiv(a.Cin, 64 * BNT);
    const int pairs = cdiv(a.Cout, 64 * BMT) * a.cin_tiles * K;
    // workgroups per launch: every one ends with (64 BMT)(64 BNT) K fp32 atomics, so as few as keep the CUs busy (two per CU)
    const int target = 256 * OCC;      // (256 / 512 / 1024 / 2048 workgroups measured: more closing atomics cost more than they hide)
    int nsplit = (target + pairs - 1) / pairs;
    if (nsplit > a.ntiles) nsplit = a.ntiles;
    if (nsplit < 1) nsplit = 1;
    a.nsplit = nsplit;
    auto fn = wgrad_bf16_kernel<K, S, TH, BMT, BNT, OCC>;
    MT4_RAISE_LDS(fn);
    hipLaunchKernelGGL(fn, dim3((unsigned)(pairs * nsplit)), dim3(256), LDS, stream, a);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ pooling backward, packed-weight copy
// MaxPool2d(3,2,1) backward in gather form (no atomics): a thread owns 8 channels of one INPUT pixel and sums the gradients of the at most four
// windows whose first maximum -- (kh, kw) scan order, what torch's backward picks -- it is; one rounding.  16-byte loads throughout.
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(u[e] << 16); f[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}

__global__ void maxpool3x3s2_bwd_bf16_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, u16* __restrict__ dx, int B, int H, int W, int C,
                                             int Ho, int Wo) {
    const int CV = C >> 3;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * H * W * CV) return;
    const int cv = (int)(i % CV);
    long long r = i / CV;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H), b = (int)(r / H);
    float own[8], acc[8];
    unpack8(*(const uint4*)(x + (((long long)b * H + h) * W + w) * C + cv * 8), own);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ho = h >> 1; ho <= ((h + 1) >> 1) && ho < Ho; ++ho)
        for (int wo = w >> 1; wo <= ((w + 1) >> 1) && wo < Wo; ++wo) {
            // this pixel takes the window's gradient in channel e iff no EARLIER position holds a value >= own[e] and no LATER one a value > own[e]
            unsigned win = 0xffu;
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = 2 * ho - 1 + kh;
                if ((unsigned)hh >= (unsigned)H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int ww = 2 * wo - 1 + kw;
                    if ((unsigned)ww >= (unsigned)W || (hh == h && ww == w)) continue;
                    float v[8];
                    unpack8(*(const uint4*)(x + (((long long)b * H + hh) * W + ww) * C + cv * 8), v);
                    const bool before = hh < h || (hh == h && ww < w);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (before ? v[e] >= own[e] : v[e] > own[e]) win &= ~(1u << e);
                }
            }
            float g[8];
            unpack8(*(const uint4*)(dy + (((long long)b * Ho + ho) * Wo + wo) * C + cv * 8), g);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (win & (1u << e)) acc[e] += g[e];
        }
    *(uint4*)(dx + (((long long)b * H + h) * W + w) * C + cv * 8) =
        make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
}

__global__ void avgpool_bwd_bf16_kernel(const float* __restrict__ df, u16* __restrict__ dx, int HW, int C, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long long e = i * 4;
    const int c = (int)(e % C);
    const long long b = e / ((long long)HW * C);
    const float4 v = *(const float4*)(df + b * C + c);
    const float s = 1.0f / (float)HW;
    st4<u16>(dx + e, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const float4 a = *(const float4*)(x + i * 8), b = *(const float4*)(x + i * 8 + 4);
    *(uint4*)(y + i * 8) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

__global__ void repack_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, int taps, int tapw32, long long kpad32, int tapw16, long long kpad16,
                                   long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / kpad16;
    const int k = (int)(idx - n * kpad16);
    const int tap = k / tapw16, c = k - tap * tapw16;
    float v = 0.f;
    if (tap < taps && c < tapw32) v = src[n * kpad32 + tap * tapw32 + c];
    dst[idx] = f32_to_bf16(v);
}

// The same gradient from a 16 x 16 input tile x 64 channels per workgroup: the 19 x 19 input pixels under the tile's 9 x 9 windows go to LDS once,
// a window's winner (first maximum in scan order, the rule above) is found once per window instead of once per pixel under it -- 4 nibble
// compares per pixel then replace the 32 neighbour loads and 256 compares of the per-pixel form, which ran at 0.87 ms for a 0.6 GB exchange
// (ResNet-50, 64 x 256 x 448).  Gradients are added in the same (ho, wo) order: same bits.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_bf16_tile_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, u16* __restrict__ dx, int H,
                                                                        int W, int C, int Ho, int Wo) {
    __shared__ uint4 xs[19 * 19][8];
    __shared__ uint4 dys[81][8];
    __shared__ unsigned win[81][8];
    const int cslabs = C >> 6;
    const int b = blockIdx.z / cslabs, cb = (blockIdx.z - b * cslabs) * 64;
    const int h0 = blockIdx.y * 16, w0 = blockIdx.x * 16;
    const uint4 ninf = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u), zero4 = make_uint4(0, 0, 0, 0);
    for (int t = threadIdx.x; t < 19 * 19 * 8; t += 256) {
        const int pix = t >> 3, g = t & 7, r = pix / 19, cc = pix - r * 19;
        const int hh = h0 - 1 + r, ww = w0 - 1 + cc;
        xs[pix][g] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? *(const uint4*)(x + (((long long)b * H + hh) * W + ww) * C + cb + g * 8) : ninf;
    }
    for (int t = threadIdx.x; t < 81 * 8; t += 256) {
        const int wi = t >> 3, g = t & 7, wr = wi / 9, wc = wi - wr * 9;
        const int ho = (h0 >> 1) + wr, wo = (w0 >> 1) + wc;
        dys[wi][g] = (ho < Ho && wo < Wo) ? *(const uint4*)(dy + (((long long)b * Ho + ho) * Wo + wo) * C + cb + g * 8) : zero4;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 81 * 8; t += 256) {
        const int wi = t >> 3, g = t & 7, wr = wi / 9, wc = wi - wr * 9;
        float best[8];
        unpack8(xs[(2 * wr) * 19 + 2 * wc][g], best);
        unsigned idx = 0;
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            float v[8];
            unpack8(xs[(2 * wr + k / 3) * 19 + 2 * wc + k % 3][g], v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[e] > best[e]) { best[e] = v[e]; idx = (idx & ~(15u << (4 * e))) | ((unsigned)k << (4 * e)); }
        }
        win[wi][g] = idx;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 256 * 8; t += 256) {
        const int p = t >> 3, g = t & 7, h = h0 + (p >> 4), w = w0 + (p & 15);
        if (h >= H || w >= W) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int ho = h >> 1; ho <= ((h + 1) >> 1) && ho < Ho; ++ho)
            for (int wo = w >> 1; wo <= ((w + 1) >> 1) && wo < Wo; ++wo) {
                const int wi = (ho - (h0 >> 1)) * 9 + (wo - (w0 >> 1));
                const unsigned k = (unsigned)((h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1))) * 0x11111111u;
                const unsigned diff = win[wi][g] ^ k;                        // a zero nibble: this pixel is that channel's winner
                float gr[8];
                unpack8(dys[wi][g], gr);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!((diff >> (4 * e)) & 15u)) acc[e] += gr[e];
            }
        *(uint4*)(dx + (((long long)b * H + h) * W + w) * C + cb + g * 8) =
            make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    }
}

// ------------------------------------------------------------------------------------------------ derived weight matrices in one launch
// After every optimizer step the trainer rebuilds, from the fp32 master weights, the matrices its kernels read: bf16 copies for the forward
// convolutions, transposed (tap-flipped) copies for the stride-1 data gradients, the four sub-pixel phase kernels of a stride-2 3x3 data
// gradient -- some 200 small matrices.  One launch walks a table of them (a launch each cost 1.5 ms per step of ~4 us launches).
__global__ __launch_bounds__(256) void refresh_weights_kernel(const mt4_refresh_entry* __restrict__ tab, int n_entries) {
    // a 32 (n) x 32 (c) tile of one tap goes through LDS: reads coalesced along c, writes coalesced along the destination's
    // fast index (n when transposed).  Only valid elements are written: the destinations' padding was zeroed once, at allocation.
    __shared__ float tile[MT4_REFRESH_TILES_PER_BLOCK][32][33];
    int lo = 0, hi = n_entries - 1;                        // the entry of this block: last one whose first block is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mt4_refresh_entry e = tab[lo];
    const int ct = (e.cin + 31) / 32, nt = (e.cout + 31) / 32, ntiles = e.ntaps_dst * nt * ct;
    const int first = (int)((long long)blockIdx.x - e.block0) * MT4_REFRESH_TILES_PER_BLOCK;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // a block moves MT4_REFRESH_TILES_PER_BLOCK consecutive tiles of its entry, all their loads in flight together: one tile per block spent most
    // of its life in the table search and one dependent load -> LDS -> store chain (0.41 ms per step for ~0.2 GB)
#pragma unroll
    for (int q = 0; q < MT4_REFRESH_TILES_PER_BLOCK; ++q) {
        int b = first + q;
        if (b >= ntiles) break;
        const int tc = b % ct;
        b /= ct;
        const int tn = b % nt, tp = b / nt, n0 = tn * 32, c = tc * 32 + tx;
        const long long tap_off = (long long)e.tap_map[tp] * e.tapw_src;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + ty + 8 * i;
            tile[q][ty + 8 * i][tx] = (n < e.cout && c < e.cin) ? e.src[(long long)n * e.kpad_src + tap_off + c] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MT4_REFRESH_TILES_PER_BLOCK; ++q) {
        int b = first + q;
        if (b >= ntiles) break;
        const int tc = b % ct;
        b /= ct;
        const int tn = b % nt, tp = b / nt, n0 = tn * 32, c0 = tc * 32, c = c0 + tx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long o;
            float v;
            bool ok;
            if (e.transposed) {
                const int cc = c0 + ty + 8 * i, nn = n0 + tx;
                ok = cc < e.cin && nn < e.cout;
                o = (long long)cc * e.kpad_dst + tp * e.tapw_dst + nn;
                v = tile[q][tx][ty + 8 * i];
            } else {
                const int nn = n0 + ty + 8 * i;
                ok = nn < e.cout && c < e.cin;
                o = (long long)nn * e.kpad_dst + tp * e.tapw_dst + c;
                v = tile[q][ty + 8 * i][tx];
            }
            if (ok) {
                if (e.dst_bf16) ((u16*)e.dst)[o] = f32_to_bf16(v);
                else ((float*)e.dst)[o] = v;
            }
        }
    }
}

}  // namespace

extern "C" int mt4_refresh_weights(const mt4_refresh_entry* table_dev, int32_t n_entries, int64_t n_blocks, void* stream) {
    mt4_clear_error();
    if (!table_dev || n_entries <= 0 || n_blocks <= 0 || n_blocks > 0x7fffffffLL) return MT4_EINVAL;
    hipLaunchKernelGGL(refresh_weights_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n_entries);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int mt4_bn_stats_t(const void* x, int32_t x_dtype, double* sums_zeroed, float* mean, float* invstd, float* running_mean, float* running_var,
                              int64_t M, int32_t C, float momentum, float eps, void* stream) {
    mt4_clear_error();
    if (!x || !sums_zeroed || !mean || !invstd || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(C, 64), bn_reduce_slabs(M, C));
    if (x_dtype == MT4_BF16) hipLaunchKernelGGL(bn_stats_t_kernel<u16>, grid, dim3(1024), 0, s, (const u16*)x, sums_zeroed, (long long)M, C);
    else if (x_dtype == MT4_F32) hipLaunchKernelGGL(bn_stats_t_kernel<float>, grid, dim3(1024), 0, s, (const float*)x, sums_zeroed, (long long)M, C);
    else return MT4_EUNSUPPORTED;
    hipLaunchKernelGGL(bn_finalize_t_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, sums_zeroed, mean, invstd, running_mean, running_var, (long long)M, C,
                       momentum, eps);
    return mt4_check_launch();
}

extern "C" int mt4_bn_apply_t(const void* x, int32_t x_dtype, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              const void* residual_bf16, void* y_bf16, int64_t M, int32_t C, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!x || !mean || !invstd || !gamma || !beta || !y_bf16 || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    const dim3 grid(cdiv(C, 64), bn_row_slabs(M, C));
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MT4_BF16)
        hipLaunchKernelGGL((bn_apply_t_kernel<u16, false>), grid, dim3(256), 0, s, (const u16*)x, (float*)mean, (float*)invstd, gamma, beta, (const u16*)residual_bf16,
                           (u16*)y_bf16, (long long)M, C, relu, (const double*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f);
    else if (x_dtype == MT4_F32)
        hipLaunchKernelGGL((bn_apply_t_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, (float*)mean, (float*)invstd, gamma, beta,
                           (const u16*)residual_bf16, (u16*)y_bf16, (long long)M, C, relu, (const double*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f);
    else return MT4_EUNSUPPORTED;
    return mt4_check_launch();
}

extern "C" int mt4_bn_apply_sums_t(const void* x, int32_t x_dtype, const double* stat_sums, float* mean, float* invstd, float* running_mean, float* running_var,
                                   const float* gamma, const float* beta, const void* residual_bf16, void* y_bf16, int64_t M, int32_t C, float momentum,
                                   float eps, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!x || !stat_sums || !mean || !invstd || !gamma || !beta || !y_bf16 || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    const dim3 grid(cdiv(C, 64), bn_row_slabs(M, C));
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MT4_BF16)
        hipLaunchKernelGGL((bn_apply_t_kernel<u16, true>), grid, dim3(256), 0, s, (const u16*)x, mean, invstd, gamma, beta, (const u16*)residual_bf16, (u16*)y_bf16,
                           (long long)M, C, relu, stat_sums, running_mean, running_var, momentum, eps);
    else if (x_dtype == MT4_F32)
        hipLaunchKernelGGL((bn_apply_t_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, mean, invstd, gamma, beta, (const u16*)residual_bf16,
                           (u16*)y_bf16, (long long)M, C, relu, stat_sums, running_mean, running_var, momentum, eps);
    else return MT4_EUNSUPPORTED;
    return mt4_check_launch();
}

extern "C" int mt4_bn_backward_t(const void* dy_bf16, const void* y_post_bf16, const void* x, int32_t x_dtype, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, double* sums_zeroed, void* dx, void* dres_bf16, float* dgamma, float* dbeta,
                                 int64_t M, int32_t C, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!dy_bf16 || !x || !mean || !invstd || !gamma || !sums_zeroed || !dx || !dgamma || !dbeta || M <= 0 || C <= 0) return MT4_EINVAL;
    if (relu < 0 || relu > 2 || (relu == 1 && !y_post_bf16) || (relu == 2 && !beta)) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    if (x_dtype != MT4_BF16 && x_dtype != MT4_F32) return MT4_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g1(cdiv(C, 64), bn_reduce_slabs(M, C)), g2(cdiv(C, 64), bn_row_slabs(M, C));
    const u16 *dy = (const u16*)dy_bf16, *yp = (const u16*)y_post_bf16;
    if (x_dtype == MT4_BF16) {
        hipLaunchKernelGGL(bn_bwd_reduce_t_kernel<u16>, g1, dim3(1024), 0, s, dy, yp, (const u16*)x, mean, invstd, gamma, beta, sums_zeroed, (long long)M, C, relu);
        hipLaunchKernelGGL(bn_bwd_apply_t_kernel<u16>, g2, dim3(256), 0, s, dy, yp, (const u16*)x, mean, invstd, gamma, beta, sums_zeroed, (u16*)dx,
                           (u16*)dres_bf16, dgamma, dbeta, (long long)M, C, relu);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_t_kernel<float>, g1, dim3(1024), 0, s, dy, yp, (const float*)x, mean, invstd, gamma, beta, sums_zeroed, (long long)M, C, relu);
        hipLaunchKernelGGL(bn_bwd_apply_t_kernel<float>, g2, dim3(256), 0, s, dy, yp, (const float*)x, mean, invstd, gamma, beta, sums_zeroed, (float*)dx,
                           (u16*)dres_bf16, dgamma, dbeta, (long long)M, C, relu);
    }
    return mt4_check_launch();
}

extern "C" int mt4_wgrad_conv2d_bf16(const void* dy, const void* x, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Ho, int32_t Wo,
                                     int32_t Cout, int32_t K, int32_t stride, void* stream) {
    mt4_clear_error();
    if (!dy || !x || !dw_packed || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0) return MT4_EINVAL;
    if ((K != 1 && K != 3) || (stride != 1 && stride != 2)) return MT4_EUNSUPPORTED;
    if ((Cin % 8) || (Cout % 8)) return MT4_EALIGN;
    if (Ho != (H + 2 * (K / 2) - K) / stride + 1 || Wo != (W + 2 * (K / 2) - K) / stride + 1) return MT4_EINVAL;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw_packed) & 15) return MT4_EALIGN;
    WgK a;
    a.dy = (const u16*)dy; a.x = (const u16*)x; a.dw = dw_packed;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kpad = (int)mt4_conv_packed_k(Cin, K, K, MT4_F32);
    a.tapw = (Cin + 3) / 4 * 4;
    hipStream_t s = (hipStream_t)stream;
    // wider (n, c) tiles where the channel counts allow: twice the FLOPs per staged byte
    const bool m2 = Cout > 64, n2 = Cin > 64;
    if (K == 1 && stride == 1) {
        if (m2 && n2) return launch_wgrad<1, 1, 8, 2, 2>(a, s);
        if (m2) return launch_wgrad<1, 1, 8, 2, 1>(a, s);
        if (n2) return launch_wgrad<1, 1, 8, 1, 2>(a, s);
        return launch_wgrad<1, 1, 8, 1, 1>(a, s);
    }
    if (K == 1) {
        if (m2) return launch_wgrad<1, 2, 8, 2, 1>(a, s);
        return launch_wgrad<1, 2, 8, 1, 1>(a, s);
    }
    // 3x3: 64 x 64 per kernel row (K x 16 accumulator registers per wave leave room for the one-tile-ahead staging registers), tiles of 4 rows
    // and THREE workgroups per CU: a workgroup waits out a load round trip per tile (request behind the barrier, needed at the next one), and
    // with two per CU the matrix cores idled through most of it -- same-box, ResNet-50 b64 shapes: layer2 117 -> 84 us, layer3 115 -> 71,
    // layer4 105 -> 69, the strided ones 142 -> 89 (profiles/r03_wgrad_experiments.txt).  The 1x1 forms spend ~45 % of their time in the
    // closing fp32 atomics (one dword per clock and L2 channel): more workgroups mean more of those, and they stay at two per CU.
    if (stride == 1) return launch_wgrad<3, 1, 4, 1, 1, 3>(a, s);
    return launch_wgrad<3, 2, 4, 1, 1, 3>(a, s);
}

extern "C" int mt4_maxpool3x3s2_bwd_bf16(const void* x, const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    mt4_clear_error();
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MT4_EINVAL;
    if ((C & 7) || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15)) return MT4_EALIGN;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (!(C & 63) && (long long)B * (C >> 6) <= 65535) {
        hipLaunchKernelGGL(maxpool3x3s2_bwd_bf16_tile_kernel, dim3(cdiv(W, 16), cdiv(H, 16), B * (C >> 6)), dim3(256), 0, (hipStream_t)stream, (const u16*)x,
                           (const u16*)dy, (u16*)dx, H, W, C, Ho, Wo);
        return mt4_check_launch();
    }
    const long long n = (long long)B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool3x3s2_bwd_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy,
                       (u16*)dx, B, H, W, C, Ho, Wo);
    return mt4_check_launch();
}

extern "C" int mt4_avgpool_bwd_bf16(const float* dfeat, void* dx, int32_t B, int32_t HW, int32_t C, void* stream) {
    mt4_clear_error();
    if (!dfeat || !dx || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return MT4_EINVAL;
    const long long n4 = (long long)B * HW * C / 4;
    hipLaunchKernelGGL(avgpool_bwd_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dfeat, (u16*)dx, HW, C, n4);
    return mt4_check_launch();
}

extern "C" int mt4_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, void* stream) {
    mt4_clear_error();
    if (!x || !y_bf16 || n <= 0 || (n & 7)) return MT4_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y_bf16) & 15) return MT4_EALIGN;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y_bf16, (long long)(n / 8));
    return mt4_check_launch();
}

extern "C" int mt4_repack_weight_bf16(const float* w_f32_packed, void* w_bf16_packed, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, void* stream) {
    mt4_clear_error();
    if (!w_f32_packed || !w_bf16_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return MT4_EINVAL;
    const long long k32 = mt4_conv_packed_k(Cin, KH, KW, MT4_F32), k16 = mt4_conv_packed_k(Cin, KH, KW, MT4_BF16);
    const long long total = (long long)Cout * k16;
    hipLaunchKernelGGL(repack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_f32_packed, (u16*)w_bf16_packed, KH * KW,
                       (Cin + 3) / 4 * 4, k32, (Cin + 7) / 8 * 8, k16, total);
    return mt4_check_launch();
}
