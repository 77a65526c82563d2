// Shared helpers for the gfx950 kernels of libmt4hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// round-to-nearest-even f32 -> bf16 (NaN stays NaN: plain cast semantics via the hardware convert)
__device__ __forceinline__ u16 f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// exact GELU (nn.GELU default, approximate='none'): 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
