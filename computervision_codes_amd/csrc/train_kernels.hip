// Training-side kernels for the temporal head (Temporal_tenco/run.py:181-235): weight gradient of a (dilated) Conv1d,
// bias gradient, BCE-with-logits loss + gradient, SGD update, and the small element-wise pieces.  fp32 throughout
// (the reference trains in fp32); the data gradients reuse the forward implicit-GEMM kernel with transposed weights.
#include "mt4_common.h"

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ conv1d weight gradient
// dW[co][tap*Cin + ci] (+)= sum_{b,t} dY[b,t][co] * X[b, t + tap*dil - pad][ci]        (zero outside [0,T))
// The contraction index (time) is the ROW index of both operands, which is exactly the operand shape of the fp32 MFMA
// v_mfma_f32_16x16x4_f32 (lane l holds A[i = l&15][k = l>>4] / B[k = l>>4][j = l&15]): fragments are read straight
// from row-major [time][channel] LDS tiles, no transpose.  One workgroup = a 64(co) x 64(k) tile of dW, looping over
// time in chunks of 32 rows; wave w owns output rows [16w, 16w+16) x 64 columns.
__global__ __launch_bounds__(256) void wgrad_conv1d_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dw, int B, int T, int Cout, int Cin, int taps,
                                                               int dil, int pad, int Kpad, int accumulate, long long rows_per_split,
                                                               float* __restrict__ gb) {
    // gb (optional): the bias gradient gb[co] += sum over rows of dy[.][co], taken by the workgroups of the first K column tile from the dY rows
    // they stage anyway (a launch of colsum_f32_kernel per layer was 139 launches = 0.4 ms of a 7.4 ms whole-video step of the TCN trainer)
    // gridDim.z > 1: the time range is split over workgroups that add their partial tile with fp32 atomics (dW zeroed by the host
    // wrapper unless accumulating) -- a 512 x 1536 weight has only 192 tiles, one per CU with nothing to overlap its load latency
    constexpr int KT = 32, LDP = 68;  // 64 columns + 4 pad floats: the 4 k-rows of a fragment read hit distinct banks
    __shared__ __attribute__((aligned(16))) float sdy[KT * LDP];
    __shared__ __attribute__((aligned(16))) float sx[KT * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    // staging: thread -> (row r = tid/16 + 16*i, 4 consecutive columns c4 = (tid%16)*4)
    const int sr = tid >> 4, sc = (tid & 15) * 4;
    // per staged x column: (tap, ci) of columns sc..sc+3 (a 4-vector never straddles taps when Cin % 4 == 0)
    const int kcol = k0 + sc;
    const int xtap = kcol / Cin, xci = kcol - xtap * Cin;
    const bool xcol_ok = kcol < taps * Cin;
    const int shift = xtap * dil - pad;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long Mall = (long long)B * T;
    const long long mb = (long long)blockIdx.z * rows_per_split;
    const long long M = mb + rows_per_split < Mall ? mb + rows_per_split : Mall;
    float4 ry[2], rx[2];
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);           // this thread's share of the column sums of dY (rows sr + 16 i of every chunk)
    const bool do_gb = gb != nullptr && blockIdx.x == 0;
    auto fetch = [&](long long m0) {   // the next chunk's loads fly while the current chunk runs its MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + sr + 16 * i;
            float4 vy = make_float4(0, 0, 0, 0), vx = make_float4(0, 0, 0, 0);
            if (m < M) {
                if (co0 + sc < Cout) vy = *(const float4*)(dy + m * Cout + co0 + sc);   // Cout % 4 == 0 (host check)
                const int b = (int)(m / T), t = (int)(m - (long long)b * T);
                const int ts = t + shift;
                if (xcol_ok && (unsigned)ts < (unsigned)T) vx = *(const float4*)(x + ((long long)b * T + ts) * Cin + xci);
            }
            ry[i] = vy;
            rx[i] = vx;
        }
    };
    fetch(mb);
    for (long long m0 = mb; m0 < M; m0 += KT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = sr + 16 * i;
            *(float4*)(sdy + r * LDP + sc) = ry[i];
            *(float4*)(sx + r * LDP + sc) = rx[i];
            cs.x += ry[i].x; cs.y += ry[i].y; cs.z += ry[i].z; cs.w += ry[i].w;
        }
        __syncthreads();
        if (m0 + KT < M) fetch(m0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            const int row = kk * 4 + (lane >> 4);
            const float a = sdy[row * LDP + wave * 16 + (lane & 15)];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float bv = sx[row * LDP + n * 16 + (lane & 15)];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[i = co][j = k]: lane holds column j = lane&15, rows 4*(lane>>4) + e
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int k = k0 + n * 16 + (lane & 15);
        if (k >= Kpad) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + wave * 16 + (lane >> 4) * 4 + e;
            if (co >= Cout) continue;
            float* p = dw + (long long)co * Kpad + k;
            if (gridDim.z > 1) atomicAdd(p, acc[n][e]);
            else *p = accumulate ? *p + acc[n][e] : acc[n][e];
        }
    }
    if (do_gb) {   // fold the 16 row lanes of every column group through LDS (the operand tiles are dead), one atomic per column and workgroup
        float* red = sdy;                                   // [16 row lanes][64 columns]
        *(float4*)(red + sr * 64 + sc) = cs;
        __syncthreads();
        if (tid < 64 && co0 + tid < Cout) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * 64 + tid];
            atomicAdd(gb + co0 + tid, t);
        }
    }
}

// zero-fill as a KERNEL node: a hipMemsetAsync captured into one of several consecutive hipGraphs of a stream (graph.SegmentedGraph) left parts of
// its range unwritten on replay (every fourth dword of the bias gradients, a different vector each run; the single-graph capture was fine), so
// the zeroing in front of an atomic accumulation is a launch of our own
__global__ void zero_f32_kernel(float* __restrict__ p, long long n) {
    // any 4-byte aligned p: a scalar head up to the first 16-byte boundary, float4 stores, a scalar tail
    long long head = (long long)(((16 - ((uintptr_t)p & 15)) & 15) >> 2);
    if (head > n) head = n;
    const long long nv = (n - head) >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) ((float4*)(p + head))[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < head) p[i] = 0.f;
    const long long t0 = head + nv * 4;
    if (i < 4 && t0 + i < n) p[t0 + i] = 0.f;
}
static inline void zero_f32(float* p, long long n, hipStream_t s) {
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, p, n);
}

extern "C" int mt4_wgrad_conv1d_f32(const float* dy, const float* x, float* dw_packed, int32_t B, int32_t T, int32_t Cout, int32_t Cin,
                                    int32_t taps, int32_t dil, int32_t pad, int32_t accumulate, float* bias_grad, void* stream) {
    mt4_clear_error();
    if (!dy || !x || !dw_packed || B <= 0 || T <= 0 || Cout <= 0 || Cin <= 0 || taps <= 0 || dil <= 0 || pad < 0) return MT4_EINVAL;
    if (Cin % 4 || Cout % 4 || (((uintptr_t)dy | (uintptr_t)x) & 15)) return MT4_EALIGN;
    const int Kpad = (int)mt4_conv_packed_k(Cin, 1, taps, MT4_F32);
    const long long M = (long long)B * T;
    const long long max_splits = (M + 255) / 256;              // at least 256 rows per split
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cdiv(Kpad, 64) * cdiv(Cout, 64);
    long long splits = (1024 + tiles - 1) / tiles;            // ~4 workgroups per CU in total
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const long long rps = ((M + splits - 1) / splits + 31) / 32 * 32;
    splits = (M + rps - 1) / rps;
    if (splits > 1 && !accumulate) zero_f32(dw_packed, (long long)Cout * Kpad, s);
    const dim3 grid(cdiv(Kpad, 64), cdiv(Cout, 64), (unsigned)splits);
    hipLaunchKernelGGL(wgrad_conv1d_f32_kernel, grid, dim3(256), 0, s, dy, x, dw_packed, B, T, Cout, Cin, taps, dil, pad, Kpad, accumulate, rps, bias_grad);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ column sums (bias gradient)
// out[c] (+)= sum_m x[m][c].  Workgroups = (64-column blocks) x (row slabs), a thread sums a strided set of rows, LDS combine, one
// fp32 atomic per column and slab into `out` (zeroed first unless accumulating).  (One workgroup per 64 columns -- 8 workgroups for
// the TCN's 512 channels -- was 21 % of the training step: 82 launches of ~50 us each.)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long long M, int C, int ld) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (long long m = (long long)blockIdx.y * 4 + w; m < M; m += (long long)gridDim.y * 4) s += x[m * ld + c];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int mt4_colsum_f32(const float* x, float* out, int64_t M, int32_t C, int32_t ld, int32_t accumulate, void* stream) {
    mt4_clear_error();
    if (!x || !out || M <= 0 || C <= 0 || ld < C) return MT4_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) zero_f32(out, C, s);      // (any float-aligned `out`: bias slices of a flat gradient buffer)
    long long slabs = (M + 63) / 64;                     // >= 16 rows per wave and slab
    const long long cap = (1024 + cdiv(C, 64) - 1) / cdiv(C, 64);
    if (slabs > cap) slabs = cap;
    if (slabs < 1) slabs = 1;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(C, 64), (unsigned)slabs), dim3(256), 0, s, x, out, (long long)M, C, ld);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ BCE with logits: loss + gradient
// per element: l = max(y,0) - y z + log1p(exp(-|y|)) ; dy = (sigmoid(y) - z) * col_scale[n]
// col_loss[n] += sum_m l   (the caller weights and normalises per head: mean over T*K_head, run.py:196-212)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ y, const float* __restrict__ z,
                                                         const float* __restrict__ col_scale, float* __restrict__ dy,
                                                         float* __restrict__ col_loss, long long M, int N, int ld_y, int ld_dy) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    __shared__ float red[4][64];
    float ls = 0.f;
    if (n < N) {
        const float sc = col_scale[n];
        for (long long m = (long long)blockIdx.y * 4 + w; m < M; m += (long long)gridDim.y * 4) {
            const float yv = y[m * ld_y + n], zv = z[m * N + n];
            ls += fmaxf(yv, 0.f) - yv * zv + log1pf(expf(-fabsf(yv)));
            dy[m * ld_dy + n] = (1.0f / (1.0f + expf(-yv)) - zv) * sc;
        }
    }
    red[w][threadIdx.x & 63] = ls;
    __syncthreads();
    if (w == 0 && n < N) atomicAdd(col_loss + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int mt4_bce_logits_f32(const float* y, const float* z, const float* col_scale, float* dy, float* col_loss, int64_t M, int32_t N,
                                  int32_t ld_y, int32_t ld_dy, void* stream) {
    mt4_clear_error();
    if (!y || !z || !col_scale || !dy || !col_loss || M <= 0 || N <= 0 || ld_y < N || ld_dy < N) return MT4_EINVAL;
    int gy = (int)((M + 63) / 64);
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(bce_logits_kernel, dim3(cdiv(N, 64), gy), dim3(256), 0, (hipStream_t)stream, y, z, col_scale, dy, col_loss,
                       (long long)M, N, ld_y, ld_dy);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ SGD (no momentum) and element-wise pieces
// p -= lr * (g + wd * p)     (torch.optim.SGD, momentum 0: Temporal_tenco/run.py:343)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long long n, float lr, float wd, float gscale) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 pv = *(float4*)(p + i);
        const float4 gv = *(const float4*)(g + i);
        pv.x -= lr * (gv.x * gscale + wd * pv.x); pv.y -= lr * (gv.y * gscale + wd * pv.y);
        pv.z -= lr * (gv.z * gscale + wd * pv.z); pv.w -= lr * (gv.w * gscale + wd * pv.w);
        *(float4*)(p + i) = pv;
    } else {
        for (long long j = i; j < n; ++j) p[j] -= lr * (g[j] * gscale + wd * p[j]);
    }
}

extern "C" int mt4_sgd_step_f32(float* p, const float* g, int64_t n, float lr, float weight_decay, float grad_scale, void* stream) {
    mt4_clear_error();
    if (!p || !g || n <= 0) return MT4_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g) & 15) return MT4_EALIGN;
    const long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, (long long)n, lr, weight_decay,
                       grad_scale);
    return mt4_check_launch();
}

// y = a * b (+ c)   element-wise fp32: dropout masks forward (z + o*m) and backward (df*m)
__global__ void mul_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ y,
                               long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i] + (c ? c[i] : 0.f);
}

extern "C" int mt4_mul_add_f32(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream) {
    mt4_clear_error();
    if (!a || !b || !y || n <= 0) return MT4_EINVAL;
    hipLaunchKernelGGL(mul_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, y, (long long)n);
    return mt4_check_launch();
}

// packed conv1d weight [Cout][Kpad] (row = taps x CPT chunks) -> the data-gradient operator's packed weight
// [Cin][KpadT] (row = taps x CPT' chunks over Cout, taps reversed): dX = conv(dY, W^T_flipped).  Padding stays zero.
__global__ void transpose_pack_conv1d_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int taps, int Kpad,
                                             int KpadT, int tapw_src, int tapw_dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cin * KpadT) return;
    const int kk = (int)(idx % KpadT);
    const int ci = (int)(idx / KpadT);
    const int tp = kk / tapw_dst, co = kk - tp * tapw_dst;
    float v = 0.f;
    if (tp < taps && co < Cout) v = w[(long long)co * Kpad + (taps - 1 - tp) * tapw_src + ci];
    wt[idx] = v;
}

extern "C" int mt4_transpose_pack_conv1d_f32(const float* w_packed, float* wt_packed, int32_t Cout, int32_t Cin, int32_t taps, void* stream) {
    mt4_clear_error();
    if (!w_packed || !wt_packed || Cout <= 0 || Cin <= 0 || taps <= 0) return MT4_EINVAL;
    const int Kpad = (int)mt4_conv_packed_k(Cin, 1, taps, MT4_F32), KpadT = (int)mt4_conv_packed_k(Cout, 1, taps, MT4_F32);
    const int tapw_src = ((Cin * 4 + 15) / 16) * 4, tapw_dst = ((Cout * 4 + 15) / 16) * 4;   // elements per tap incl. chunk padding
    const long long total = (long long)Cin * KpadT;
    hipLaunchKernelGGL(transpose_pack_conv1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, wt_packed,
                       Cout, Cin, taps, Kpad, KpadT, tapw_src, tapw_dst);
    return mt4_check_launch();
}
