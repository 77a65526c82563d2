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

// GELU, erf form (nn.GELU default): 0.5 x (1 + erf(x / sqrt(2))) with erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7;
// measured |gelu - exact| < 5e-7 over [-6, 6] in fp32).  libm's erff inlines to ~45 VALU ops per element, which made the
// fc1 (+GELU) epilogues of Swin / MS-TCT VALU-bound; this is 1 rcp + 1 exp + 9 fma/mul.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);   // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(e, x));
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
