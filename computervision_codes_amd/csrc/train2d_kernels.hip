// Training-side kernels of the spatial stage (Spatial_cnn/run.py:145-224): train-mode BatchNorm forward/backward,
// Conv2d weight gradient, pooling backward, and the loss pieces (BCE with pos_weight, DistillKL, MSE, KD-mixing backward).
// fp32, channels-last.  Data gradients of the convolutions reuse mt4_conv_nhwc (transposed weights; strided convs as
// 4 sub-pixel phases scattered through the output row map).
#include "mt4_common.h"

// ------------------------------------------------------------------------------------------------ BatchNorm2d (training)
// pass 1: sums[0][c] += sum_m x[m][c], sums[1][c] += sum_m x^2   (sums zeroed by the caller; atomics over row slabs).
// The per-channel reductions of BatchNorm (these and the two of the backward) accumulate in float64: the kernels are
// HBM-bound, CDNA4 runs fp64 VALU at half the fp32 rate, and E[x^2]-E[x]^2 as well as the backward's projections cancel
// badly in fp32 -- torch's CPU BatchNorm (the reference's arithmetic) accumulates in double too.
// Work split: a thread owns 4 consecutive channels (one 16-byte load per row) and walks rows; the reductions run 1024-thread workgroups
// (64 channels x 64 row phases), at most ~512 per launch: every workgroup ends in 128 float64 atomics on its slab's 128 addresses, and 2048
// workgroups of 256 threads on a 64-channel map made those same-address chains (~25 ns a link at the L2) longer than the stream itself
// (profiles/r03_bn_training_kernels.txt; the bf16-tensor twins live in train2d_bf16.hip)
__device__ __forceinline__ void bn_block_reduce(double (&acc)[8], double* __restrict__ sums, int C, int c0) {
    __shared__ double red[16][8][17];               // [wave][value][channel group], padded
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                   // the 4 row phases of a wave
        acc[j] += __shfl_xor(acc[j], 16);
        acc[j] += __shfl_xor(acc[j], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[w][j][lane] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) {                        // 16 channel groups x 8 values
        const int g = threadIdx.x & 15, j = threadIdx.x >> 4;
        double t = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) t += red[rr][j][g];
        const int c = c0 + g * 4 + (j & 3);
        if (c < C) atomicAdd(sums + (j >> 2) * C + c, t);
    }
}

__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, long long M, int C) {
    const int c0 = blockIdx.x * 64, c = c0 + (threadIdx.x & 15) * 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        const long long st = (long long)gridDim.y * 64;
        long long m = (long long)blockIdx.y * 64 + (threadIdx.x >> 4);
        auto add = [&](const float4 v) {
            acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
            acc[4] += (double)v.x * v.x; acc[5] += (double)v.y * v.y; acc[6] += (double)v.z * v.z; acc[7] += (double)v.w * v.w;
        };
        for (; m + st < M; m += 2 * st) {           // two rows in flight, added in ascending order
            const float4 v0 = *(const float4*)(x + m * C + c), v1 = *(const float4*)(x + (m + st) * C + c);
            add(v0); add(v1);
        }
        for (; m < M; m += st) add(*(const float4*)(x + m * C + c));
    }
    bn_block_reduce(acc, sums, C, c0);
}

static inline int bn_reduce_slabs(long long M, int C) {                      // 64-row slabs of the 1024-thread reductions
    long long gy = (M + 63) / 64;
    const long long cap = (512 + cdiv(C, 64) - 1) / cdiv(C, 64);
    if (gy > cap) gy = cap;
    return gy < 1 ? 1 : (int)gy;
}

static inline int bn_row_slabs(long long M, int C) {                         // 16-row slabs of the 256-thread streaming kernels
    long long gy = (M + 63) / 64;
    const long long cap = (2048 + cdiv(C, 64) - 1) / cdiv(C, 64);
    if (gy > cap) gy = cap;
    return gy < 1 ? 1 : (int)gy;
}

// pass 2: batch mean / biased variance -> (mean, invstd); running stats with momentum and the UNBIASED variance (torch)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ run_mean, float* __restrict__ run_var, long long M, int C, float momentum, float eps) {
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

extern "C" int mt4_bn_stats_f32(const float* x, double* sums_zeroed, float* mean, float* invstd, float* running_mean, float* running_var,
                                int64_t M, int32_t C, float momentum, float eps, void* stream) {
    mt4_clear_error();
    if (!x || !sums_zeroed || !mean || !invstd || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(cdiv(C, 64), bn_reduce_slabs(M, C)), dim3(1024), 0, s, x, sums_zeroed, (long long)M, C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, sums_zeroed, mean, invstd, running_mean, running_var,
                       (long long)M, C, momentum, eps);
    return mt4_check_launch();
}

// y = act( (x - mean) * invstd * gamma + beta [+ residual] ): column slabs, the 4 channels' parameters in registers (one element group per
// thread re-loaded 64 bytes of parameters per 16 bytes of tensor traffic); the same expressions, the same bits
// FIN: mean / invstd are evaluated here from the channel sums a convolution's epilogue left (mt4_conv_desc.stat_sums), as bn_apply_t_kernel does
// for the bf16 mode: the first 64 threads fold the replicas (bn_finalize_kernel's float64 expressions), row-slab 0 writes mean / invstd for
// the backward and advances the running statistics
template <bool FIN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ y, long long M, int C, int relu, const double* __restrict__ sums,
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
        *(float4*)(y + i) = o;
    };
    for (; m + st < M; m += 2 * st) {
        const long long i0 = m * C + c, i1 = (m + st) * C + c;
        const float4 x0 = *(const float4*)(x + i0), x1 = *(const float4*)(x + i1);
        float4 r0 = zero, r1 = zero;
        if (res) { r0 = *(const float4*)(res + i0); r1 = *(const float4*)(res + i1); }
        put(i0, x0, r0); put(i1, x1, r1);
    }
    for (; m < M; m += st) {
        const long long i0 = m * C + c;
        put(i0, *(const float4*)(x + i0), res ? *(const float4*)(res + i0) : zero);
    }
}

extern "C" int mt4_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* residual, float* y, int64_t M, int32_t C, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!x || !mean || !invstd || !gamma || !beta || !y || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(cdiv(C, 64), bn_row_slabs(M, C)), dim3(256), 0, (hipStream_t)stream, x, (float*)mean, (float*)invstd,
                       gamma, beta, residual, y, (long long)M, C, relu, (const double*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f);
    return mt4_check_launch();
}

extern "C" int mt4_bn_apply_sums_f32(const float* x, const double* stat_sums, float* mean, float* invstd, float* running_mean, float* running_var,
                                     const float* gamma, const float* beta, const float* residual, float* y, int64_t M, int32_t C, float momentum,
                                     float eps, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!x || !stat_sums || !mean || !invstd || !gamma || !beta || !y || M <= 0 || C <= 0) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(cdiv(C, 64), bn_row_slabs(M, C)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, residual,
                       y, (long long)M, C, relu, stat_sums, running_mean, running_var, momentum, eps);
    return mt4_check_launch();
}

// backward pass 1: with dy' = relu ? (y > 0 ? dy : 0) : dy :  sums[0][c] += sum dy',  sums[1][c] += sum dy' * xhat
__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, double* __restrict__ sums, long long M, int C, int relu) {
    // relu 2: the gate is recomputed from x -- (x - mean) * invstd * gamma + beta > 0, the forward's own fp32 expression, i.e. the stored y
    // bit for bit (units without a residual input): y is not read, a third of this kernel's traffic
    const int c0 = blockIdx.x * 64, c = c0 + (threadIdx.x & 15) * 4;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
        float4 ga = make_float4(0, 0, 0, 0), be = ga;
        if (relu == 2) { ga = *(const float4*)(gamma + c); be = *(const float4*)(beta + c); }
        const long long st = (long long)gridDim.y * 64;
        long long m = (long long)blockIdx.y * 64 + (threadIdx.x >> 4);
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
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
        for (; m + st < M; m += 2 * st) {
            const long long i0 = m * C + c, i1 = (m + st) * C + c;
            const float4 g0 = *(const float4*)(dy + i0), g1 = *(const float4*)(dy + i1);
            const float4 x0 = *(const float4*)(x + i0), x1 = *(const float4*)(x + i1);
            float4 y0 = one, y1 = one;
            if (relu == 1) { y0 = *(const float4*)(y + i0); y1 = *(const float4*)(y + i1); }
            add(g0, x0, y0); add(g1, x1, y1);
        }
        for (; m < M; m += st) {
            const long long i0 = m * C + c;
            add(*(const float4*)(dy + i0), *(const float4*)(x + i0), relu == 1 ? *(const float4*)(y + i0) : one);
        }
    }
    bn_block_reduce(acc, sums, C, c0);
}

// backward pass 2: dx = gamma * invstd * (dy' - sum(dy')/M - xhat * sum(dy' xhat)/M);  dres = dy' (gradient of the residual input);
// dgamma = sums[1], dbeta = sums[0]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const double* __restrict__ sums, float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, long long M, int C, int relu) {
    // column slabs: the parameters and the two per-channel means of the reduction stay in registers
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
        *(float4*)(dx + i) = o;
        if (dres) *(float4*)(dres + i) = g;
    };
    for (; m + st < M; m += 2 * st) {
        const long long i0 = m * C + c, i1 = (m + st) * C + c;
        const float4 g0 = *(const float4*)(dy + i0), g1 = *(const float4*)(dy + i1);
        const float4 x0 = *(const float4*)(x + i0), x1 = *(const float4*)(x + i1);
        float4 y0 = one, y1 = one;
        if (relu == 1) { y0 = *(const float4*)(y + i0); y1 = *(const float4*)(y + i1); }
        put(i0, g0, x0, y0); put(i1, g1, x1, y1);
    }
    for (; m < M; m += st) {
        const long long i0 = m * C + c;
        put(i0, *(const float4*)(dy + i0), *(const float4*)(x + i0), relu == 1 ? *(const float4*)(y + i0) : one);
    }
}

extern "C" int mt4_bn_backward_f32(const float* dy, const float* y_post, const float* x, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, double* sums_zeroed, float* dx, float* dres, float* dgamma, float* dbeta,
                                   int64_t M, int32_t C, int32_t relu, void* stream) {
    mt4_clear_error();
    if (!dy || !x || !mean || !invstd || !gamma || !sums_zeroed || !dx || !dgamma || !dbeta || M <= 0 || C <= 0) return MT4_EINVAL;
    if (relu < 0 || relu > 2 || (relu == 1 && !y_post) || (relu == 2 && !beta)) return MT4_EINVAL;
    if (C % 4) return MT4_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C, 64), bn_reduce_slabs(M, C)), dim3(1024), 0, s, dy, y_post, x, mean, invstd, gamma, beta, sums_zeroed,
                       (long long)M, C, relu);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(C, 64), bn_row_slabs(M, C)), dim3(256), 0, s, dy, y_post, x, mean, invstd, gamma, beta, sums_zeroed, dx,
                       dres, dgamma, dbeta, (long long)M, C, relu);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ Conv2d weight gradient
// dW[co][tap*CPT4 + ci] += sum_{b,ho,wo} dY[b,ho,wo][co] * X[b, ho*sh - ph + kh*dh, wo*sw - pw + kw*dw][ci]
// Same scheme as the conv1d kernel (fp32 MFMA 16x16x4 reading row-major LDS tiles), the pixel range split over gridDim.z
// workgroups that add their partial tile with fp32 atomics (dW zeroed by the caller).
struct Wg2 {
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, ph, pw, dh, dw, Kpad, tapw;
    long long M, rows_per_split;
};

__global__ __launch_bounds__(256) void wgrad_conv2d_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dwp, const Wg2 a) {
    constexpr int KT = 32, LDP = 68;
    __shared__ __attribute__((aligned(16))) float sdy[KT * LDP];
    __shared__ __attribute__((aligned(16))) float sx[KT * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int sr = tid >> 4, sc = (tid & 15) * 4;
    const int kcol = k0 + sc;
    const int xtap = kcol / a.tapw, xci = kcol - xtap * a.tapw;
    const bool xcol_ok = xtap < a.KH * a.KW && xci < a.Cin;
    const int kh = xtap / a.KW, kw = xtap - kh * a.KW;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long mb = (long long)blockIdx.z * a.rows_per_split;
    long long me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int HoWo = a.Ho * a.Wo;
    // software pipeline: the global loads of chunk c+1 are in flight while chunk c runs its MFMAs (the loop was one exposed HBM
    // latency per 32 pixels)
    float4 ry[2], rx[2];
    auto fetch = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + sr + 16 * i;
            float4 vy = make_float4(0, 0, 0, 0), vx = make_float4(0, 0, 0, 0);
            if (m < me) {
                if (co0 + sc < a.Cout) vy = *(const float4*)(dy + m * a.Cout + co0 + sc);
                if (xcol_ok) {
                    const int b = (int)(m / HoWo);
                    const int rem = (int)(m - (long long)b * HoWo);
                    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                    const int hi = ho * a.sh - a.ph + kh * a.dh, wi = wo * a.sw - a.pw + kw * a.dw;
                    if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                        vx = *(const float4*)(x + (((long long)b * a.H + hi) * a.W + wi) * a.Cin + xci);
                }
            }
            ry[i] = vy;
            rx[i] = vx;
        }
    };
    fetch(mb);
    for (long long m0 = mb; m0 < me; m0 += KT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = sr + 16 * i;
            *(float4*)(sdy + r * LDP + sc) = ry[i];
            *(float4*)(sx + r * LDP + sc) = rx[i];
        }
        __syncthreads();
        if (m0 + KT < me) fetch(m0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            const int row = kk * 4 + (lane >> 4);
            const float av = sdy[row * LDP + wave * 16 + (lane & 15)];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sx[row * LDP + n * 16 + (lane & 15)], acc[n], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int k = k0 + n * 16 + (lane & 15);
        if (k >= a.Kpad) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + wave * 16 + (lane >> 4) * 4 + e;
            if (co < a.Cout) atomicAdd(dwp + (long long)co * a.Kpad + k, acc[n][e]);
        }
    }
}

// The same product on 128 x 128 (output channel, K column) tiles, waves 2 x 2 with 64 x 64 each: per 4-pixel MFMA step a wave reads 4 + 4
// operand values for 16 MFMAs where the 64 x 64 kernel above reads 1 + 4 for 4 -- that one issues ~1500 instructions per 104 MFMAs and is
// bound by instruction issue, not by its matrix cores (profiles/r03_wgrad_experiments.txt) -- and the pixel coordinates of a thread's rows
// advance incrementally instead of by a 64-bit division per row and chunk.  Same arithmetic: fp32 MFMA 16x16x4 over the pixels in order,
// partial tiles added with fp32 atomics.
template <int BMT, int BNT>      // tile = (64 BMT) output channels x (64 BNT) K columns; waves 2 x 2
__global__ __launch_bounds__(256) void wgrad_conv2d_f32_wide_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    float* __restrict__ dwp, const Wg2 a) {
    constexpr int KT = 32, BM = 64 * BMT, BN = 64 * BNT, LDY = BM + 4, LDX = BN + 4;
    constexpr int MI = 2 * BMT, NJ = 2 * BNT;                   // 16 x 16 accumulator tiles of a wave
    constexpr int YG = BM / 4, XG = BN / 4;                     // float4 column groups per staged row
    constexpr int YP = KT * YG / 256, XP = KT * XG / 256;       // staging passes (rows 256 / groups apart)
    __shared__ __attribute__((aligned(16))) float sdy[KT * LDY];
    __shared__ __attribute__((aligned(16))) float sx[KT * LDX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.x * BN, co0 = blockIdx.y * BM;
    const int yr = tid / YG, yc = (tid % YG) * 4;               // dY: row yr + (256 / YG) i, columns yc ..
    const int xr = tid / XG, xc = (tid % XG) * 4;               // x:  row xr + (256 / XG) i, K columns xc ..
    const int kcol = k0 + xc;
    const int xtap = kcol / a.tapw, xci = kcol - xtap * a.tapw;
    const bool xcol_ok = xtap < a.KH * a.KW && xci < a.Cin;
    const int kh = xtap / a.KW, kw = xtap - kh * a.KW;
    const int dh0 = kh * a.dh - a.ph, dw0 = kw * a.dw - a.pw;
    const bool ycol_ok = co0 + yc < a.Cout;
    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long mb = (long long)blockIdx.z * a.rows_per_split;
    long long me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int HoWo = a.Ho * a.Wo;
    int pb[XP], pho[XP], pwo[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const long long m = mb + xr + (256 / XG) * i;
        pb[i] = (int)(m / HoWo);
        const int rem = (int)(m - (long long)pb[i] * HoWo);
        pho[i] = rem / a.Wo;
        pwo[i] = rem - pho[i] * a.Wo;
    }
    const long long last = me - 1;
    float4 ry[YP], rx[XP];
    auto fetch = [&](long long m0) {      // unconditional loads from clamped addresses, zeroed by selects (no load waits behind a branch)
#pragma unroll
        for (int i = 0; i < YP; ++i) {
            const long long m = m0 + yr + (256 / YG) * i;
            const float4 vy = *(const float4*)(dy + (m < me ? m : last) * a.Cout + (ycol_ok ? co0 + yc : 0));
            ry[i] = (m < me && ycol_ok) ? vy : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const long long m = m0 + xr + (256 / XG) * i;
            const int hi = pho[i] * a.sh + dh0, wi = pwo[i] * a.sw + dw0;
            const bool ok = xcol_ok && m < me && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const int hc = min(max(hi, 0), a.H - 1), wc = min(max(wi, 0), a.W - 1), bc = min(pb[i], a.B - 1);
            const float4 vx = *(const float4*)(x + (((long long)bc * a.H + hc) * a.W + wc) * a.Cin + (xcol_ok ? xci : 0));
            rx[i] = ok ? vx : make_float4(0, 0, 0, 0);
            pwo[i] += KT;                  // the row this slot stages in the next chunk
            while (pwo[i] >= a.Wo) {
                pwo[i] -= a.Wo;
                if (++pho[i] == a.Ho) { pho[i] = 0; ++pb[i]; }
            }
        }
    };
    fetch(mb);
    for (long long m0 = mb; m0 < me; m0 += KT) {
#pragma unroll
        for (int i = 0; i < YP; ++i) *(float4*)(sdy + (yr + (256 / YG) * i) * LDY + yc) = ry[i];
#pragma unroll
        for (int i = 0; i < XP; ++i) *(float4*)(sx + (xr + (256 / XG) * i) * LDX + xc) = rx[i];
        __syncthreads();
        if (m0 + KT < me) fetch(m0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            const int row = kk * 4 + (lane >> 4);
            float av[MI], bv[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = sdy[row * LDY + wm * (32 * BMT) + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = sx[row * LDX + wn * (32 * BNT) + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = k0 + wn * (32 * BNT) + j * 16 + (lane & 15);
            if (k >= a.Kpad) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wm * (32 * BMT) + i * 16 + (lane >> 4) * 4 + e;
                if (co < a.Cout) atomicAdd(dwp + (long long)co * a.Kpad + k, acc[i][j][e]);
            }
        }
}

template <int BMT, int BNT>
static int launch_wgrad32_wide(const float* dy, const float* x, float* dw, Wg2 a, int Cout, hipStream_t s) {
    const int tiles = cdiv(a.Kpad, 64 * BNT) * cdiv(Cout, 64 * BMT);
    long long splits = (768 + tiles - 1) / tiles;
    const long long max_splits = (a.M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.rows_per_split = ((a.M + splits - 1) / splits + 31) / 32 * 32;
    splits = (a.M + a.rows_per_split - 1) / a.rows_per_split;
    hipLaunchKernelGGL((wgrad_conv2d_f32_wide_kernel<BMT, BNT>), dim3(cdiv(a.Kpad, 64 * BNT), cdiv(Cout, 64 * BMT), (unsigned)splits), dim3(256), 0, s, dy, x,
                       dw, a);
    return mt4_check_launch();
}

extern "C" int mt4_wgrad_conv2d_f32(const float* dy, const float* x, float* dw_packed_zeroed, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                    int32_t Ho, int32_t Wo, int32_t Cout, int32_t KH, int32_t KW, int32_t stride_h, int32_t stride_w,
                                    int32_t pad_h, int32_t pad_w, int32_t dil_h, int32_t dil_w, void* stream) {
    mt4_clear_error();
    if (!dy || !x || !dw_packed_zeroed || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || KH <= 0 || KW <= 0)
        return MT4_EINVAL;
    if (Cin % 4 || Cout % 4 || (((uintptr_t)dy | (uintptr_t)x) & 15)) return MT4_EALIGN;
    Wg2 a;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = KH; a.KW = KW; a.sh = stride_h; a.sw = stride_w;
    a.ph = pad_h; a.pw = pad_w; a.dh = dil_h; a.dw = dil_w;
    a.Kpad = (int)mt4_conv_packed_k(Cin, KH, KW, MT4_F32);
    a.tapw = ((Cin * 4 + 15) / 16) * 4;
    a.M = (long long)B * Ho * Wo;
    // wider tiles where the channel counts allow AND the launch still has >= 2 workgroups per CU (each pixel split covers >= 256 pixels: a small
    // batch has few of them -- ResNet-50 at batch 8: 9.6 ms with the 64 x 64 kernel everywhere, 10.2 with the wide tiles);
    // the 64 x 64 kernel below: the narrowest layers, small batches
    {
        hipStream_t s = (hipStream_t)stream;
        const long long max_splits = (a.M + 255) / 256;
        auto fills = [&](int bm, int bn) { return (long long)cdiv(a.Kpad, bn) * cdiv(Cout, bm) * max_splits >= 512; };
        if (Cout > 64 && a.Kpad > 64 && fills(128, 128)) return launch_wgrad32_wide<2, 2>(dy, x, dw_packed_zeroed, a, Cout, s);
        if (a.Kpad >= 128 && fills(64, 128)) return launch_wgrad32_wide<1, 2>(dy, x, dw_packed_zeroed, a, Cout, s);
        if (Cout > 64 && fills(128, 64)) return launch_wgrad32_wide<2, 1>(dy, x, dw_packed_zeroed, a, Cout, s);
    }
    const int tiles = cdiv(a.Kpad, 64) * cdiv(Cout, 64);
    long long splits = (1024 + tiles - 1) / tiles;             // ~4 workgroups per CU in total
    const long long max_splits = (a.M + 255) / 256;             // at least 256 rows per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.rows_per_split = ((a.M + splits - 1) / splits + 31) / 32 * 32;
    splits = (a.M + a.rows_per_split - 1) / a.rows_per_split;
    hipLaunchKernelGGL(wgrad_conv2d_f32_kernel, dim3(cdiv(a.Kpad, 64), cdiv(Cout, 64), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, dy, x,
                       dw_packed_zeroed, a);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ pooling backward
// MaxPool2d(3,2,1): the gradient of a window goes to its FIRST maximum in (kh, kw) scan order (what torch's backward does);
// dx zeroed by the caller; windows overlap -> fp32 atomics.
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W,
                                        int C, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * Ho * Wo * C) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best = -INFINITY;
    long long arg = -1;
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const long long o = (((long long)b * H + hi) * W + wi) * C + c;
            const float v = x[o];
            if (v > best || arg < 0) { if (v > best || arg < 0) { best = v; arg = o; } }
        }
    }
    atomicAdd(dx + arg, dy[idx]);
}

extern "C" int mt4_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx_zeroed, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    mt4_clear_error();
    if (!x || !dy || !dx_zeroed || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MT4_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long n = (long long)B * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx_zeroed, B, H, W, C, Ho,
                       Wo);
    return mt4_check_launch();
}

// AdaptiveAvgPool2d(1) backward: dx[b][p][c] = dfeat[b][c] / HW
__global__ void avgpool_bwd_kernel(const float* __restrict__ df, float* __restrict__ dx, int HW, int C, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long b = i / ((long long)HW * C);
    dx[i] = df[b * C + c] / (float)HW;
}

extern "C" int mt4_avgpool_bwd_f32(const float* dfeat, float* dx, int32_t B, int32_t HW, int32_t C, void* stream) {
    mt4_clear_error();
    if (!dfeat || !dx || B <= 0 || HW <= 0 || C <= 0) return MT4_EINVAL;
    const long long n = (long long)B * HW * C;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dfeat, dx, HW, C, n);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ losses
// BCEWithLogitsLoss(pos_weight) (run.py:322-324): l = (1-z) y + (1 + (pw-1) z) (log1p(exp(-|y|)) + max(-y, 0));
// dy = ((1-z) - lw + lw sigmoid(y)) * col_scale.  pos_weight NULL = 1.
// 256 threads = 64 columns x 4 row groups over a slab of 64 rows; the column sums of a slab meet in LDS and leave as ONE atomic per
// column (a step of the MS-TCT teacher has M = 31 x 256 rows: a thread per column looping over all rows took 4 ms there).
__global__ __launch_bounds__(256) void bce_pw_kernel(const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ pw,
                                                     const float* __restrict__ col_scale, float* __restrict__ dy, float* __restrict__ col_loss, int M,
                                                     int N, int ld_y, int ld_dy) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m0 = blockIdx.y * 64, m1 = min(M, m0 + 64);
    float ls = 0.f;
    if (n < N) {
        const float p = pw ? pw[n] : 1.f, sc = col_scale[n];
        for (int m = m0 + rg; m < m1; m += 4) {
            const float yv = y[(long long)m * ld_y + n], zv = z[(long long)m * N + n];
            const float lw = 1.f + (p - 1.f) * zv;
            ls += (1.f - zv) * yv + lw * (log1pf(expf(-fabsf(yv))) + fmaxf(-yv, 0.f));
            dy[(long long)m * ld_dy + n] = ((1.f - zv) - lw + lw / (1.f + expf(-yv))) * sc;
        }
    }
    part[rg][c] = ls;
    __syncthreads();
    if (rg == 0 && n < N) atomicAdd(col_loss + n, (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}

extern "C" int mt4_bce_logits_pw_f32(const float* y, const float* z, const float* pos_weight, const float* col_scale, float* dy, float* col_loss,
                                     int32_t M, int32_t N, int32_t ld_y, int32_t ld_dy, void* stream) {
    mt4_clear_error();
    if (!y || !z || !col_scale || !dy || !col_loss || M <= 0 || N <= 0 || ld_y < N || ld_dy < N) return MT4_EINVAL;
    if (cdiv(M, 64) > 65535) return MT4_EUNSUPPORTED;
    hipLaunchKernelGGL(bce_pw_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream, y, z, pos_weight, col_scale, dy, col_loss, M, N,
                       ld_y, ld_dy);
    return mt4_check_launch();
}

// DistillKL (run.py:284-295): loss += T^2/B * sum_k p_t (log p_t - log_softmax(y_s/T)),  p_t = softmax(sigmoid(t_pred)/T);
// dy_s[b][k] (+)= scale * T/B * (softmax(y_s/T) - p_t).   One wave per row, K <= 128 (2 columns per lane).
__global__ void distill_kl_kernel(const float* __restrict__ ys, const float* __restrict__ tp, float* __restrict__ dys, float* __restrict__ loss, int B,
                                  int K, int ld_y, int ld_dy, float temp, float scale, int accumulate) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s[2], t[2];
    float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k = lane + 64 * e;
        s[e] = k < K ? ys[(long long)b * ld_y + k] / temp : -INFINITY;
        t[e] = k < K ? (1.f / (1.f + expf(-tp[(long long)b * K + k]))) / temp : -INFINITY;
        ms = fmaxf(ms, s[e]); mt = fmaxf(mt, t[e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ms = fmaxf(ms, __shfl_xor(ms, o)); mt = fmaxf(mt, __shfl_xor(mt, o)); }
    float zs = 0.f, zt = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) { zs += lane + 64 * e < K ? expf(s[e] - ms) : 0.f; zt += lane + 64 * e < K ? expf(t[e] - mt) : 0.f; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { zs += __shfl_xor(zs, o); zt += __shfl_xor(zt, o); }
    const float lzs = logf(zs), lzt = logf(zt);
    float l = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k = lane + 64 * e;
        if (k < K) {
            const float lps = s[e] - ms - lzs, lpt = t[e] - mt - lzt, pt = expf(lpt);
            l += pt * (lpt - lps);
            const float g = scale * temp / (float)B * (expf(lps) - pt);
            float* d = dys + (long long)b * ld_dy + k;
            *d = accumulate ? *d + g : g;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    if (lane == 0) atomicAdd(loss, l * temp * temp / (float)B);
}

extern "C" int mt4_distill_kl_f32(const float* y_s, const float* t_pred, float* dy_s, float* loss, int32_t B, int32_t K, int32_t ld_y, int32_t ld_dy,
                                  float temp, float grad_scale, int32_t accumulate, void* stream) {
    mt4_clear_error();
    if (!y_s || !t_pred || !dy_s || !loss || B <= 0 || K <= 0 || K > 128 || ld_y < K || ld_dy < K) return MT4_EINVAL;
    hipLaunchKernelGGL(distill_kl_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, y_s, t_pred, dy_s, loss, B, K, ld_y, ld_dy, temp, grad_scale,
                       accumulate);
    return mt4_check_launch();
}

// MSELoss (mean): loss += sum (a-b)^2 / n ; da = scale * 2 (a-b) / n
__global__ void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da, float* __restrict__ loss, long long n,
                           float scale) {
    __shared__ float red[4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.f;
    if (i < n) {
        const float d = a[i] - b[i];
        l = d * d / (float)n;
        da[i] = scale * 2.f * d / (float)n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int mt4_mse_f32(const float* a, const float* b, float* da, float* loss, int64_t n, float grad_scale, void* stream) {
    mt4_clear_error();
    if (!a || !b || !da || !loss || n <= 0) return MT4_EINVAL;
    hipLaunchKernelGGL(mse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, da, loss, (long long)n, grad_scale);
    return mt4_check_launch();
}

// KD mixing backward (forward: mt4_kd_mix).  u_n = s a_n, a = softmax_n(l), l_n = s tau_n / sqrt(C), tau_n = sum_d tea_n[b][d].
// Given g_n = dL/du_n:  ds[b][c] = sum_n g_n a_n + sum_n dl_n tau_n / sqrt(C),  dtau_n[b] = sum_c dl_n s / sqrt(C),
// dl_n = a_n (g_n s - sum_m a_m g_m s).   One workgroup per batch row.
__global__ void kd_mix_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t0, const float* __restrict__ t1,
                                  const float* __restrict__ t2, const float* __restrict__ g0, const float* __restrict__ g1,
                                  const float* __restrict__ g2, float* __restrict__ ds, float* __restrict__ dtau, int C) {
    __shared__ double red[3][4];      // the row reductions in float64 (dl_n is a difference of nearly equal terms)
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { a0 += t0[(long long)b * C + c]; a1 += t1[(long long)b * C + c]; a2 += t2[(long long)b * C + c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; red[2][wave] = a2; }
    __syncthreads();
    const float tau0 = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]), tau1 = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]),
                tau2 = (float)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    __syncthreads();
    const float inv = rsqrtf((float)C);
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const long long i = (long long)b * C + c;
        const float sv = s[i], z = sv * inv;
        const float l0 = z * tau0, l1 = z * tau1, l2 = z * tau2, mx = fmaxf(l0, fmaxf(l1, l2));
        float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(l2 - mx);
        const float r = 1.f / (e0 + e1 + e2);
        e0 *= r; e1 *= r; e2 *= r;
        const float q0 = g0[i] * sv, q1 = g1[i] * sv, q2 = g2[i] * sv;       // dL/da_n
        const float dot = e0 * q0 + e1 * q1 + e2 * q2;
        const float dl0 = e0 * (q0 - dot), dl1 = e1 * (q1 - dot), dl2 = e2 * (q2 - dot);
        ds[i] = g0[i] * e0 + g1[i] * e1 + g2[i] * e2 + (dl0 * tau0 + dl1 * tau1 + dl2 * tau2) * inv;
        d0 += (double)dl0 * z; d1 += (double)dl1 * z; d2 += (double)dl2 * z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
    if (lane == 0) { red[0][wave] = d0; red[1][wave] = d1; red[2][wave] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) dtau[b * 3 + threadIdx.x] = (float)(red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

extern "C" int mt4_kd_mix_bwd_f32(const float* s, const float* tea_i, const float* tea_v, const float* tea_t, const float* g_i, const float* g_v,
                                  const float* g_t, float* ds, float* dtau, int32_t B, int32_t C, void* stream) {
    mt4_clear_error();
    if (!s || !tea_i || !tea_v || !tea_t || !g_i || !g_v || !g_t || !ds || !dtau || B <= 0 || C <= 0) return MT4_EINVAL;
    hipLaunchKernelGGL(kd_mix_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, s, tea_i, tea_v, tea_t, g_i, g_v, g_t, ds, dtau, C);
    return mt4_check_launch();
}
