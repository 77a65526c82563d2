// What the access width costs a BatchNorm-shaped streaming kernel (bf16 [M][C] tensors: three read, two written, a few FMAs per element):
// 8 bytes per lane (4 bf16: the form of csrc/train2d_bf16.hip's kernels) against 16 bytes per lane, one item per thread and grid-stride.
//   hipcc -O3 --offload-arch=gfx950 tools/bn_stream_micro.hip -o /tmp/bn_stream && /tmp/bn_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned short u16;

__device__ __forceinline__ float lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }

template <int NR, int NW>
__global__ void k8(const uint2* __restrict__ a, const uint2* __restrict__ b, const uint2* __restrict__ c, uint2* __restrict__ o1, uint2* __restrict__ o2,
                   const float* __restrict__ coef, long long n, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)((i * 4) % C);
    const float4 k = *(const float4*)(coef + ch);
    uint2 va = a[i], vb = NR > 1 ? b[i] : va, vc = NR > 2 ? c[i] : va;
    float g0 = lo(va.x), g1 = hi(va.x), g2 = lo(va.y), g3 = hi(va.y);
    if (!(lo(vb.x) > 0.f)) g0 = 0.f; if (!(hi(vb.x) > 0.f)) g1 = 0.f; if (!(lo(vb.y) > 0.f)) g2 = 0.f; if (!(hi(vb.y) > 0.f)) g3 = 0.f;
    const uint2 r = make_uint2(pk(k.x * (g0 - lo(vc.x)), k.y * (g1 - hi(vc.x))), pk(k.z * (g2 - lo(vc.y)), k.w * (g3 - hi(vc.y))));
    o1[i] = r;
    if (NW > 1) o2[i] = make_uint2(pk(g0, g1), pk(g2, g3));
}

template <int NR, int NW>
__global__ void k16(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c, uint4* __restrict__ o1, uint4* __restrict__ o2,
                    const float* __restrict__ coef, long long n, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)((i * 8) % C);
    const float4 k0 = *(const float4*)(coef + ch), k1 = *(const float4*)(coef + ch + 4);
    uint4 va = a[i], vb = NR > 1 ? b[i] : va, vc = NR > 2 ? c[i] : va;
    const unsigned A[4] = {va.x, va.y, va.z, va.w}, B[4] = {vb.x, vb.y, vb.z, vb.w}, Cc[4] = {vc.x, vc.y, vc.z, vc.w};
    const float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    unsigned r[4], gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float g0 = lo(A[e]), g1 = hi(A[e]);
        if (!(lo(B[e]) > 0.f)) g0 = 0.f;
        if (!(hi(B[e]) > 0.f)) g1 = 0.f;
        r[e] = pk(kk[2 * e] * (g0 - lo(Cc[e])), kk[2 * e + 1] * (g1 - hi(Cc[e])));
        gq[e] = pk(g0, g1);
    }
    o1[i] = make_uint4(r[0], r[1], r[2], r[3]);
    if (NW > 1) o2[i] = make_uint4(gq[0], gq[1], gq[2], gq[3]);
}

int main() {
    const int C = 256;
    const long long elems = 64LL * 64 * 112 * C;           // a layer1 map of the b64 256 x 448 step
    u16 *buf[5];
    for (auto& p : buf) { CK(hipMalloc(&p, elems * 2)); CK(hipMemset(p, 0x3f, elems * 2)); }
    float* coef; CK(hipMalloc(&coef, C * 4)); CK(hipMemset(coef, 0, C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name, int nr, int nw) {
        float best = 1e9;
        for (int it = 0; it < 8; ++it) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 1 && ms < best) best = ms;
        }
        printf("%-34s %7.3f ms  %6.2f TB/s\n", name, best, (double)elems * 2 * (nr + nw) / best / 1e9);
    };
    const long long n8 = elems / 4, n16 = elems / 8;
    time([&] { hipLaunchKernelGGL((k8<3, 2>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (uint2*)buf[0], (uint2*)buf[1], (uint2*)buf[2], (uint2*)buf[3], (uint2*)buf[4], coef, n8, C); }, "8 B / lane, 3 reads 2 writes", 3, 2);
    time([&] { hipLaunchKernelGGL((k16<3, 2>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)buf[0], (uint4*)buf[1], (uint4*)buf[2], (uint4*)buf[3], (uint4*)buf[4], coef, n16, C); }, "16 B / lane, 3 reads 2 writes", 3, 2);
    time([&] { hipLaunchKernelGGL((k8<2, 1>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (uint2*)buf[0], (uint2*)buf[1], (uint2*)buf[2], (uint2*)buf[3], (uint2*)buf[4], coef, n8, C); }, "8 B / lane, 2 reads 1 write", 2, 1);
    time([&] { hipLaunchKernelGGL((k16<2, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)buf[0], (uint4*)buf[1], (uint4*)buf[2], (uint4*)buf[3], (uint4*)buf[4], coef, n16, C); }, "16 B / lane, 2 reads 1 write", 2, 1);
    time([&] { hipLaunchKernelGGL((k8<1, 1>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (uint2*)buf[0], (uint2*)buf[1], (uint2*)buf[2], (uint2*)buf[3], (uint2*)buf[4], coef, n8, C); }, "8 B / lane, 1 read 1 write", 1, 1);
    time([&] { hipLaunchKernelGGL((k16<1, 1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (uint4*)buf[0], (uint4*)buf[1], (uint4*)buf[2], (uint4*)buf[3], (uint4*)buf[4], coef, n16, C); }, "16 B / lane, 1 read 1 write", 1, 1);
    return 0;
}
