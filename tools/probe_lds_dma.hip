// Probe (GPU box): does an out-of-range `buffer_load_dwordx4 ... lds` lane write ZEROS to LDS, or leave the old bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const char* g, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;          // far out of range
    if ((threadIdx.x & 7) == 2) voff = nbytes - 8;    // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, voff, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    char* g; unsigned* o; unsigned h[256];
    hipMalloc(&g, 4096); hipMalloc(&o, 1024);
    unsigned init[1024]; for (int i = 0; i < 1024; ++i) init[i] = 0x1000 + i;
    hipMemcpy(g, init, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, 1024);   // buffer of 1024 bytes = 64 lanes x 16
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; ++l) printf("lane %2d: %08x %08x %08x %08x\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
