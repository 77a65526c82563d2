// Shared helpers for the gfx950 kernels of libmt4hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mt4hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

extern thread_local int g_mt4_last_hip_error;

// hipGetLastError() reports (and clears) the last error of ANY runtime call on this thread, including ones
// the host framework made earlier; clear it before a launch so that the check sees only our launch.
static inline void mt4_clear_error() { (void)hipGetLastError(); }

static inline int mt4_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_mt4_last_hip_error = (int)e;
        return MT4_ELAUNCH;
    }
    return MT4_OK;
}

__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even: a plain cast, which hipcc lowers to the gfx950 hardware convert
// (v_cvt_pk_bf16_f32, NaN stays NaN) -- an integer-arithmetic rounding costs ~6 VALU ops per element and made the
// conv epilogue VALU-issue-bound (SQ_ACTIVE_INST_ANY 41 % of wave cycles on the K=64 layers).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16 f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(u16, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// GELU, erf form (nn.GELU default): x Phi(x), Phi(x) = 0.5 erfc(-x / sqrt 2).  With a = min(|x|, 6) and Q(a) = -log2(erfc(a / sqrt 2)) fitted by
// a (c1 + a (c2 + a (c3 + a (c4 + a c5)))) (weighted least squares on the GELU error over [0, 6]):  t = 2^-Q(a) = erfc(a / sqrt 2),
//     gelu(x) = 0.5 x (1 + sign(x) (1 - t));      measured |gelu - exact| < 9e-7 over [-10, 10] in fp32 arithmetic.
// One transcendental (v_exp_f32) and 9 fma / mul / min / bfi per element, all of which pair up in the packed form below (v_pk_fma_f32 ...):
// 10 issue slots per element against 19 for the Abramowitz-Stegun 7.1.26 form of rounds 1-3 (1 rcp + 1 exp + 11 plain, no packing) and
// ~45 for libm's erff -- the fc1 (+ GELU) epilogues of Swin are VALU-bound on it (73728 x 2048 elements per launch at batch 128).
typedef float mt4_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mt4_f32x2 gelu_erf2(mt4_f32x2 x) {
    const mt4_f32x2 a = {fminf(fabsf(x.x), 6.0f), fminf(fabsf(x.y), 6.0f)};
    mt4_f32x2 p = a * 4.881283152809e-04f + -7.198873334067e-03f;
    p = p * a + 5.214694370026e-02f;
    p = p * a + 4.595955966962e-01f;
    p = p * a + 1.151000605814e+00f;
    const mt4_f32x2 q = p * a;
    const mt4_f32x2 t = {__builtin_amdgcn_exp2f(-q.x), __builtin_amdgcn_exp2f(-q.y)};
    const mt4_f32x2 u = 1.0f - t;
    const mt4_f32x2 hx = x * 0.5f;
    const mt4_f32x2 s = {__builtin_copysignf(u.x, x.x), __builtin_copysignf(u.y, x.y)};
    return hx * s + hx;
}
__device__ __forceinline__ float gelu_erf(float x) {      // (the same arithmetic on one value: identical results)
    const float a = fminf(fabsf(x), 6.0f);
    float p = fmaf(a, 4.881283152809e-04f, -7.198873334067e-03f);
    p = fmaf(p, a, 5.214694370026e-02f);
    p = fmaf(p, a, 4.595955966962e-01f);
    p = fmaf(p, a, 1.151000605814e+00f);
    const float t = __builtin_amdgcn_exp2f(-(p * a));
    const float hx = 0.5f * x;
    return fmaf(hx, __builtin_copysignf(1.0f - t, x), hx);
}
template <int N>
__device__ __forceinline__ void gelu_erf_n(float (&v)[N]) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        const mt4_f32x2 r = gelu_erf2((mt4_f32x2){v[e], v[e + 1]});
        v[e] = r.x; v[e + 1] = r.y;
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// allow > 64 KiB of dynamic LDS for kernel `fn` on the CURRENT device: once per device and call site (the attribute is per device; a
// process-wide flag would leave the second GPU of a process at the 64 KiB default)
#define MT4_RAISE_LDS(fn)                                                                                                   \
    do {                                                                                                                    \
        static bool raised_[64] = {};                                                                                       \
        int dev_ = 0;                                                                                                       \
        (void)hipGetDevice(&dev_);                                                                                          \
        if (dev_ < 0 || dev_ >= 64 || !raised_[dev_]) {                                                                     \
            (void)hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
            if (dev_ >= 0 && dev_ < 64) raised_[dev_] = true;                                                               \
        }                                                                                                                   \
    } while (0)

// output / residual rows are stored / loaded with the non-temporal bit when the launch's output exceeds this many MB (below it the map fits the
// Infinity Cache and stays cacheable for the next layer; each byte is touched once per launch: +2.6 % frames/s on ResNet-50, +4.5 % on Swin-B,
// same-box A/Bs of round 1)
#define MT4_NT_MIN_MB 200
