// Small memory-bound kernels around the convolution stack: frame preprocessing, pooling, heads.
#include "mt4_common.h"

thread_local int g_mt4_last_hip_error = 0;

#ifndef MT4_SOURCE_DIGEST
#error "build through csrc/Makefile: it passes -DMT4_SOURCE_DIGEST (srcdigest.library_digest of the sources)"
#endif
extern "C" const char* mt4_source_digest(void) { return MT4_SOURCE_DIGEST; }
extern "C" int mt4_abi_version(void) { return 10; }   // 6: + mt4_stem_maxpool_bf16, mt4_conv_desc.x2 (second K source), mt4_bottleneck_fused_next_bf16; 7: + mt4_copy_spans_u8; 8: + mt4_chain_gemm_bf16; 9: + mt4_conv_desc.stat_sums, mt4_bn_apply_sums_t, mt4_refresh_weights moves MT4_REFRESH_TILES_PER_BLOCK tiles per workgroup, mt4_avgpool1d_rows, mt4_interp_linear_rows; 10: + mt4_source_digest, mt4_attention head dims up to 512
extern "C" int mt4_last_hip_error(void) { return g_mt4_last_hip_error; }
extern "C" const char* mt4_strerror(int code) {
    switch (code) {
        case MT4_OK: return "ok";
        case MT4_EINVAL: return "invalid argument";
        case MT4_EALIGN: return "alignment contract violated";
        case MT4_ELAUNCH: return "kernel launch failed";
        case MT4_EUNSUPPORTED: return "unsupported dtype/shape";
    }
    return "unknown error";
}

// ------------------------------------------------------------------------------------------------ stem input
// out [B][H+6][Wp][4], Wp = round_up(W+6, 2); interior at (+3,+3); channel 3 and borders zero.
template <typename T, bool FROM_U8>
__global__ void stem_input_kernel(const void* __restrict__ in, T* __restrict__ out, int B, int H, int W, int Hp, int Wp,
                                  float m0, float m1, float m2, float s0, float s1, float s2) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one padded pixel per thread
    const long long total = (long long)B * Hp * Wp;
    if (idx >= total) return;
    const int wp = (int)(idx % Wp);
    const long long t = idx / Wp;
    const int hp = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int h = hp - 3, w = wp - 3;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
        if (FROM_U8) {
            const uint8_t* p = (const uint8_t*)in + (((long long)b * H + h) * W + w) * 3;
            // same op order as ToTensor (/255) then Normalize ((x-mean)/std)
            v0 = ((float)p[0] / 255.0f - m0) / s0;
            v1 = ((float)p[1] / 255.0f - m1) / s1;
            v2 = ((float)p[2] / 255.0f - m2) / s2;
        } else {
            const float* p = (const float*)in;
            const long long hw = (long long)H * W;
            const long long o = (long long)b * 3 * hw + (long long)h * W + w;
            v0 = p[o]; v1 = p[o + hw]; v2 = p[o + 2 * hw];
        }
    }
    if constexpr (sizeof(T) == 2) {
        *(uint2*)(out + idx * 4) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, 0.f));
    } else {
        *(float4*)(out + idx * 4) = make_float4(v0, v1, v2, 0.f);
    }
}

static int stem_input_launch(const void* in, void* out, int B, int H, int W, const float* mean, const float* std, int dtype,
                             bool from_u8, hipStream_t s) {
    mt4_clear_error();
    if (!in || !out || B <= 0 || H <= 0 || W <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int Hp = H + 6, Wp = (W + 6 + 1) & ~1;
    const long long total = (long long)B * Hp * Wp;
    const int grid = (int)((total + 255) / 256);
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    const float s0 = std ? std[0] : 1.f, s1 = std ? std[1] : 1.f, s2 = std ? std[2] : 1.f;
    if (dtype == MT4_BF16) {
        if (from_u8) hipLaunchKernelGGL((stem_input_kernel<u16, true>), dim3(grid), dim3(256), 0, s, in, (u16*)out, B, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
        else hipLaunchKernelGGL((stem_input_kernel<u16, false>), dim3(grid), dim3(256), 0, s, in, (u16*)out, B, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
    } else {
        if (from_u8) hipLaunchKernelGGL((stem_input_kernel<float, true>), dim3(grid), dim3(256), 0, s, in, (float*)out, B, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
        else hipLaunchKernelGGL((stem_input_kernel<float, false>), dim3(grid), dim3(256), 0, s, in, (float*)out, B, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
    }
    return mt4_check_launch();
}

extern "C" int mt4_preprocess_u8(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, const float mean[3],
                                 const float std[3], int32_t dtype, void* stream) {
    if (!mean || !std) return MT4_EINVAL;
    return stem_input_launch(frames, out, B, H, W, mean, std, dtype, true, (hipStream_t)stream);
}

extern "C" int mt4_pad_nchw_f32(const float* x, void* out, int32_t B, int32_t H, int32_t W, int32_t dtype, void* stream) {
    return stem_input_launch(x, out, B, H, W, nullptr, nullptr, dtype, false, (hipStream_t)stream);
}

// Space-to-depth form of the same input (bf16, even H and W): out [B][(H+6)/2][(W+6)/2][16], channel (dy*2+dx)*3 + c of pixel (y, x) =
// normalised frame pixel (2y+dy-3, 2x+dx-3), zero outside the frame and in channels 12..15.  The 7x7/2 stem is then a 4x4/1 conv whose
// kernel row is 4 pixels x 16 channels = ONE contiguous 128-byte run: it goes through the LDS-DMA path of mt4_conv_nhwc (x_pixel_stride).
__global__ void stem_input_s2d_kernel(const uint8_t* __restrict__ in, u16* __restrict__ out, int B, int H, int W, int Hs, int Ws, float m0,
                                      float m1, float m2, float s0, float s1, float s2, int nt) {
    // the 3 x 256 possible bf16 values, once per block (two fp32 divisions per value: per element they cost more than the kernel's memory time)
    __shared__ u16 lut[3][256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) {
        const int c = i >> 8;
        const float m = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
        lut[c][i & 255] = f32_to_bf16(((float)(i & 255) / 255.0f - m) / sd);
    }
    __syncthreads();
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one space-to-depth pixel (32 bytes) per thread
    const long long total = (long long)B * Hs * Ws;
    if (idx >= total) return;
    const int xs = (int)(idx % Ws);
    const long long t = idx / Ws;
    const int ys = (int)(t % Hs);
    const int b = (int)(t / Hs);
    // the two pixels of a frame row are 6 contiguous bytes: one unaligned 4-byte + one 2-byte load per row instead of six byte loads (the
    // launch was bound by the number of load instructions: 12 per thread).  Rows / columns outside the frame are clamped and zeroed by a select
    struct __attribute__((packed)) Px2 { uint32_t a; uint16_t b; };
    unsigned raw[12];
    bool ok[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int h = 2 * ys + dy - 3, w0 = 2 * xs - 3;
        const bool hok = (unsigned)h < (unsigned)H;
        ok[dy * 2 + 0] = hok && (unsigned)w0 < (unsigned)W;
        ok[dy * 2 + 1] = hok && (unsigned)(w0 + 1) < (unsigned)W;
        const int hc = min(max(h, 0), H - 1), wc = min(max(w0, 0), W - 2);   // (wc, wc + 1) inside the row; a clamped pair is only read where ok[] is false ... or equals the pair wanted
        const Px2 t = *(const Px2*)(in + (((long long)b * H + hc) * W + wc) * 3);
        const bool shifted = wc != w0;          // left / right border: the pair was moved, its pixels are not the ones asked for
        raw[dy * 6 + 0] = t.a & 0xff; raw[dy * 6 + 1] = (t.a >> 8) & 0xff; raw[dy * 6 + 2] = (t.a >> 16) & 0xff;
        raw[dy * 6 + 3] = t.a >> 24; raw[dy * 6 + 4] = t.b & 0xff; raw[dy * 6 + 5] = t.b >> 8;
        if (shifted) {       // w0 = -3, -1 (both or the first pixel outside) or W - 1 (second outside)
            if (w0 == -1) { raw[dy * 6 + 3] = raw[dy * 6 + 0]; raw[dy * 6 + 4] = raw[dy * 6 + 1]; raw[dy * 6 + 5] = raw[dy * 6 + 2]; }   // pixel 0 of the row is the SECOND of the pair
            else if (w0 == W - 1) { raw[dy * 6 + 0] = raw[dy * 6 + 3]; raw[dy * 6 + 1] = raw[dy * 6 + 4]; raw[dy * 6 + 2] = raw[dy * 6 + 5]; }   // pixel W-1 is the FIRST
        }
    }
    unsigned v[12];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned t = lut[c][raw[px * 3 + c]];
            v[px * 3 + c] = ok[px] ? t : 0u;
        }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* o = (u32x4*)(out + idx * 16);
    const u32x4 lo = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)}, hi = {v[8] | (v[9] << 16), v[10] | (v[11] << 16), 0u, 0u};
    if (nt) { __builtin_nontemporal_store(lo, o); __builtin_nontemporal_store(hi, o + 1); }
    else { o[0] = lo; o[1] = hi; }
}

extern "C" int mt4_preprocess_u8_s2d(const uint8_t* frames, void* out, int32_t B, int32_t H, int32_t W, const float mean[3],
                                     const float std[3], void* stream) {
    mt4_clear_error();
    if (!frames || !out || !mean || !std || B <= 0 || H <= 0 || W <= 0) return MT4_EINVAL;
    if ((H | W) & 1) return MT4_EUNSUPPORTED;
    const int Hs = (H + 6) / 2, Ws = (W + 6) / 2;
    const long long total = (long long)B * Hs * Ws;
    hipLaunchKernelGGL(stem_input_s2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, (u16*)out, B, H,
                       W, Hs, Ws, mean[0], mean[1], mean[2], std[0], std[1], std[2], 1);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ maxpool 3x3/2 pad 1
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int E = 16 / (int)sizeof(T);
    const int CV = C / E;  // 16-byte vectors per pixel
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Ho * Wo * CV;
    if (idx >= total) return;
    const int cv = (int)(idx % CV);
    long long t = idx / CV;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best[E];
#pragma unroll
    for (int e = 0; e < E; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint4 v = *(const uint4*)(x + (((long long)b * H + hi) * W + wi) * C + cv * E);
            if constexpr (sizeof(T) == 2) {
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    best[2 * e] = fmaxf(best[2 * e], bf16_to_f32((u16)(u[e] & 0xffff)));
                    best[2 * e + 1] = fmaxf(best[2 * e + 1], bf16_to_f32((u16)(u[e] >> 16)));
                }
            } else {
                best[0] = fmaxf(best[0], __uint_as_float(v.x)); best[1] = fmaxf(best[1], __uint_as_float(v.y));
                best[2] = fmaxf(best[2], __uint_as_float(v.z)); best[3] = fmaxf(best[3], __uint_as_float(v.w));
            }
        }
    }
    uint4 o;
    if constexpr (sizeof(T) == 2) {
        o = make_uint4(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]), pack_bf16x2(best[4], best[5]),
                       pack_bf16x2(best[6], best[7]));
    } else {
        o = make_uint4(__float_as_uint(best[0]), __float_as_uint(best[1]), __float_as_uint(best[2]), __float_as_uint(best[3]));
    }
    *(uint4*)(y + (((long long)b * Ho + ho) * Wo + wo) * C + cv * E) = o;
}

extern "C" int mt4_maxpool3x3s2_nhwc(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                     void* stream) {
    mt4_clear_error();
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int es = dtype == MT4_BF16 ? 2 : 4;
    if ((C * es) % 16 != 0 || (((uintptr_t)x | (uintptr_t)y) & 15)) return MT4_EALIGN;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C * es / 16);
    const int grid = (int)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MT4_BF16) hipLaunchKernelGGL(maxpool3x3s2_kernel<u16>, dim3(grid), dim3(256), 0, s, (const u16*)x, (u16*)y, B, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C, Ho, Wo);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ global average pool
// one block per (frame, 256-channel slab... ) : thread = one channel, loops over HW (coalesced across threads)
template <typename T>
__global__ void global_avgpool_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const T* p = x + (long long)b * HW * C + c;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) {
        if constexpr (sizeof(T) == 2) s += bf16_to_f32(p[(long long)i * C]);
        else s += p[(long long)i * C];
    }
    y[(long long)b * C + c] = s / (float)HW;
}

// bf16, C % 8 == 0: thread = 8 channels (16-byte loads, 7 pixels in flight), the same i = 0 .. HW-1 summation order per channel
__global__ void global_avgpool_bf16x8_kernel(const uint4* __restrict__ x, float* __restrict__ y, int HW, int CV, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long b = idx / CV;
    const int cv = (int)(idx - b * CV);
    const uint4* p = x + b * HW * CV + cv;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    auto add = [&](const uint4 v) {
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[2 * e] += __uint_as_float(u[e] << 16);
            s[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
        }
    };
    int i = 0;
    for (; i + 7 <= HW; i += 7) {
        uint4 v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = p[(long long)(i + k) * CV];
#pragma unroll
        for (int k = 0; k < 7; ++k) add(v[k]);
    }
    for (; i < HW; ++i) add(p[(long long)i * CV]);
    float4* o = (float4*)(y + idx * 8);
    const float n = (float)HW;
    o[0] = make_float4(s[0] / n, s[1] / n, s[2] / n, s[3] / n);
    o[1] = make_float4(s[4] / n, s[5] / n, s[6] / n, s[7] / n);
}

extern "C" int mt4_global_avgpool_nhwc(const void* x, float* y, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !y || B <= 0 || HW <= 0 || C <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (B > 65535) return MT4_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MT4_BF16 && (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const long long total = (long long)B * (C / 8);
        hipLaunchKernelGGL(global_avgpool_bf16x8_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, (const uint4*)x, y, HW, C / 8, total);
        return mt4_check_launch();
    }
    dim3 grid(cdiv(C, 256), B);
    if (dtype == MT4_BF16) hipLaunchKernelGGL(global_avgpool_kernel<u16>, grid, dim3(256), 0, s, (const u16*)x, y, HW, C);
    else hipLaunchKernelGGL(global_avgpool_kernel<float>, grid, dim3(256), 0, s, (const float*)x, y, HW, C);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ PIL-exact resize pass (uint8)
// One separable pass of Pillow's `ImagingResample` for 8-bit images (`Image.resize(size, BILINEAR)`, which
// `transforms.Resize((256,448))` of Spatial_cnn/dataloader.py:155-159 calls on the decoded PNG): out = clip8((2^21 + sum_i
// in[lo + i] * kk[i]) >> 22) with the integer coefficient table (22 fractional bits) and [lo, n) bounds built by the host exactly
// as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.  axis 0: along width ([B][H][Win][C] -> [B][H][Wout][C]);
// axis 1: along height ([B][Hin][W][C] -> [B][Hout][W][C]).  The horizontal result is rounded to uint8 before the vertical pass,
// as in Pillow, so the two passes compose to the same bytes.
__global__ void resize_pass_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int* __restrict__ bounds,
                                      const int* __restrict__ kk, int ksize, int B, int Hin, int Win, int Hout, int Wout, int C,
                                      int axis) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Hout * Wout * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long p = idx / C;
    const int x = (int)(p % Wout);
    p /= Wout;
    const int y = (int)(p % Hout);
    const int b = (int)(p / Hout);
    const int o = axis == 0 ? x : y;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long long)o * ksize;
    int ss = 1 << 21;
    if (axis == 0) {
        const uint8_t* src = in + (((long long)b * Hin + y) * Win + lo) * C + c;
        for (int i = 0; i < n; ++i) ss += (int)src[(long long)i * C] * k[i];
    } else {
        const uint8_t* src = in + (((long long)b * Hin + lo) * Win + x) * C + c;
        for (int i = 0; i < n; ++i) ss += (int)src[(long long)i * Win * C] * k[i];
    }
    ss >>= 22;
    out[idx] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

extern "C" int mt4_resize_pass_u8(const void* in, void* out, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int32_t B,
                                  int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t C, int32_t axis, void* stream) {
    mt4_clear_error();
    if (!in || !out || !bounds || !coeffs || ksize <= 0 || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0)
        return MT4_EINVAL;
    if ((axis == 0 && Hin != Hout) || (axis == 1 && Win != Wout) || (axis != 0 && axis != 1)) return MT4_EINVAL;
    const long long total = (long long)B * Hout * Wout * C;
    hipLaunchKernelGGL(resize_pass_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)in,
                       (uint8_t*)out, bounds, coeffs, ksize, B, Hin, Win, Hout, Wout, C, axis);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ linear heads (fp32)
// one wave per output element row-dot: block = 4 waves, each wave computes y[b][n] for one n, looping n
__global__ void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ y, int K, int N) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* xr = x + (long long)b * K;
    for (int n = wave; n < N; n += nw) {
        const float* wr = w + (long long)n * K;
        float s = 0.f;
        if ((K & 3) == 0) {
            for (int k = lane * 4; k < K; k += 256) {
                const float4 a = *(const float4*)(xr + k);
                const float4 c = *(const float4*)(wr + k);
                s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
            }
        } else {
            for (int k = lane; k < K; k += 64) s += xr[k] * wr[k];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) y[(long long)b * N + n] = s + (bias ? bias[n] : 0.f);
    }
}

extern "C" int mt4_linear_f32(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t K, int32_t N,
                              void* stream) {
    mt4_clear_error();
    if (!x || !w || !y || B <= 0 || K <= 0 || N <= 0) return MT4_EINVAL;
    if ((K & 3) == 0 && (((uintptr_t)x | (uintptr_t)w) & 15)) return MT4_EALIGN;
    hipLaunchKernelGGL(linear_f32_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, K, N);
    return mt4_check_launch();
}

// ------------------------------------------------------------------------------------------------ `--hier` of Temporal_tenco: pooling between the
// refinement stages (Temporal_tenco/network.py:147,154-155: nn.AvgPool1d(kernel_size=7, stride=3)) and the FPN's linear re-interpolation to the
// lateral's length (network.py:96: F.interpolate(x, size=W, mode='linear'), align_corners False), both over the time axis of frame-major rows
namespace {
template <typename T> __device__ __forceinline__ float4 ldrow4(const T* p);
template <> __device__ __forceinline__ float4 ldrow4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ldrow4<u16>(const u16* p) {
    const uint2 v = *(const uint2*)p;
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void strow4(T* p, float4 v);
template <> __device__ __forceinline__ void strow4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void strow4<u16>(u16* p, float4 v) { *(uint2*)p = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }

template <typename T>
__global__ void avgpool1d_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int Tin, int Tout, int C, int k, int stride, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C >> 2;
    const int c = (int)(i % c4) * 4;
    const long long r = i / c4;
    const int to = (int)(r % Tout);
    const long long b = r / Tout;
    const T* src = x + (b * Tin + (long long)to * stride) * C + c;
    float4 s = ldrow4<T>(src);
    for (int j = 1; j < k; ++j) { const float4 v = ldrow4<T>(src + (long long)j * C); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const float d = (float)k;
    strow4<T>(y + r * C + c, make_float4(s.x / d, s.y / d, s.z / d, s.w / d));
}

template <typename T>
__global__ void interp_linear_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int Tin, int Tout, int C, float scale, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C >> 2;
    const int c = (int)(i % c4) * 4;
    const long long r = i / c4;
    const int w = (int)(r % Tout);
    const long long b = r / Tout;
    float src = scale * ((float)w + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.f - l1;
    const float4 a = ldrow4<T>(x + (b * Tin + i0) * C + c), v = ldrow4<T>(x + (b * Tin + i1) * C + c);
    strow4<T>(y + r * C + c, make_float4(l0 * a.x + l1 * v.x, l0 * a.y + l1 * v.y, l0 * a.z + l1 * v.z, l0 * a.w + l1 * v.w));
}
// adjoints of the two (fp32: the TCN trainer's dtype), both as deterministic gathers over the OUTPUT rows that touch an input row
__global__ void avgpool1d_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Tin, int Tout, int C, int k, int stride, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C >> 2;
    const int c = (int)(i % c4) * 4;
    const long long r = i / c4;
    const int t = (int)(r % Tin);
    const long long b = r / Tin;
    int w0 = t - (k - 1);
    w0 = w0 <= 0 ? 0 : (w0 + stride - 1) / stride;          // first window that reaches t
    int w1 = t / stride;                                    // last window that starts at or before t
    if (w1 > Tout - 1) w1 = Tout - 1;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = w0; w <= w1; ++w) {
        const float4 v = *(const float4*)(dy + (b * Tout + w) * C + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float d = (float)k;
    *(float4*)(dx + r * C + c) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

__global__ void interp_linear_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Tin, int Tout, int C, float scale, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C >> 2;
    const int c = (int)(i % c4) * 4;
    const long long r = i / c4;
    const int t = (int)(r % Tin);
    const long long b = r / Tin;
    // output rows w whose source index max(0, scale (w + 0.5) - 0.5) lies in [t - 1, t + 1): a bracket a row wide on either side, then the
    // forward's own arithmetic decides
    int lo = (int)floorf(((float)t - 0.5f) / scale - 0.5f) - 1, hi = (int)ceilf(((float)t + 1.5f) / scale - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > Tout - 1 ? Tout - 1 : hi;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = lo; w <= hi; ++w) {
        float src = scale * ((float)w + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
        const int i0 = (int)src;
        const int i1 = i0 + (i0 < Tin - 1 ? 1 : 0);
        const float l1 = src - (float)i0, l0 = 1.f - l1;
        const float wt = (i0 == t ? l0 : 0.f) + (i1 == t ? l1 : 0.f);
        if (wt != 0.f) {
            const float4 v = *(const float4*)(dy + (b * Tout + w) * C + c);
            s.x += wt * v.x; s.y += wt * v.y; s.z += wt * v.z; s.w += wt * v.w;
        }
    }
    *(float4*)(dx + r * C + c) = s;
}
}  // namespace

extern "C" int mt4_avgpool1d_rows_bwd_f32(const float* dy, float* dx, int32_t B, int32_t Tin, int32_t C, int32_t k, int32_t stride, void* stream) {
    mt4_clear_error();
    if (!dy || !dx || B <= 0 || Tin <= 0 || C <= 0 || k <= 0 || stride <= 0 || Tin < k) return MT4_EINVAL;
    if ((C & 3) || (((uintptr_t)dy | (uintptr_t)dx) & 15)) return MT4_EALIGN;
    const int Tout = (Tin - k) / stride + 1;
    const long long n4 = (long long)B * Tin * (C >> 2);
    hipLaunchKernelGGL(avgpool1d_rows_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, Tin, Tout, C, k, stride, n4);
    return mt4_check_launch();
}

extern "C" int mt4_interp_linear_rows_bwd_f32(const float* dy, float* dx, int32_t B, int32_t Tin, int32_t Tout, int32_t C, void* stream) {
    mt4_clear_error();
    if (!dy || !dx || B <= 0 || Tin <= 0 || Tout <= 0 || C <= 0) return MT4_EINVAL;
    if ((C & 3) || (((uintptr_t)dy | (uintptr_t)dx) & 15)) return MT4_EALIGN;
    const long long n4 = (long long)B * Tin * (C >> 2);
    hipLaunchKernelGGL(interp_linear_rows_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, Tin, Tout, C,
                       (float)Tin / (float)Tout, n4);
    return mt4_check_launch();
}

extern "C" int mt4_avgpool1d_rows(const void* x, void* y, int32_t B, int32_t Tin, int32_t C, int32_t k, int32_t stride, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !y || B <= 0 || Tin <= 0 || C <= 0 || k <= 0 || stride <= 0 || Tin < k) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if ((C & 3) || (((uintptr_t)x | (uintptr_t)y) & 15) || (dtype == MT4_BF16 && (C & 7))) return MT4_EALIGN;
    const int Tout = (Tin - k) / stride + 1;
    const long long n4 = (long long)B * Tout * (C >> 2);
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (dtype == MT4_F32) hipLaunchKernelGGL(avgpool1d_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, Tin, Tout, C, k, stride, n4);
    else hipLaunchKernelGGL(avgpool1d_rows_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)x, (u16*)y, Tin, Tout, C, k, stride, n4);
    return mt4_check_launch();
}

extern "C" int mt4_interp_linear_rows(const void* x, void* y, int32_t B, int32_t Tin, int32_t Tout, int32_t C, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!x || !y || B <= 0 || Tin <= 0 || Tout <= 0 || C <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if ((C & 3) || (((uintptr_t)x | (uintptr_t)y) & 15) || (dtype == MT4_BF16 && (C & 7))) return MT4_EALIGN;
    const long long n4 = (long long)B * Tout * (C >> 2);
    const dim3 grid((unsigned)((n4 + 255) / 256));
    const float scale = (float)Tin / (float)Tout;
    if (dtype == MT4_F32) hipLaunchKernelGGL(interp_linear_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, Tin, Tout, C, scale, n4);
    else hipLaunchKernelGGL(interp_linear_rows_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)x, (u16*)y, Tin, Tout, C, scale, n4);
    return mt4_check_launch();
}
