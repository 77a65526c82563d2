// Stand-alone HBM calibration for the read/write mixes of the memory-bound 1x1 layers (GPU box only):
//   hipcc -O3 --offload-arch=gfx950 tools/membw.hip -o /tmp/membw && /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// out[i] (W units) = f(in[i/W... ]) : each thread reads R 16-byte vectors and writes W 16-byte vectors, grid-stride
template <int R, int W>
__global__ void mix_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long n_items) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n_items; i += stride) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 v = in[(long long)r * n_items + i];
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint4 o = acc; o.x += w;
            out[(long long)w * n_items + i] = o;
        }
    }
}

template <int R, int W>
int run(const char* name, long long n_items, int grid) {
    uint4 *in, *out;
    CK(hipMalloc(&in, sizeof(uint4) * n_items * (R ? R : 1)));
    CK(hipMalloc(&out, sizeof(uint4) * n_items * (W ? W : 1)));
    CK(hipMemset(in, 1, sizeof(uint4) * n_items * (R ? R : 1)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<R, W>), dim3(grid), dim3(256), 0, 0, in, out, n_items);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    const double bytes = 16.0 * n_items * (R + W);
    printf("%-28s grid %6d  %8.3f ms  %7.2f TB/s (R %.0f MB, W %.0f MB)\n", name, grid, best, bytes / best / 1e9, 16.0 * n_items * R / 1e6, 16.0 * n_items * W / 1e6);
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}

int main() {
    const long long unit = 103LL * 1000 * 1000 / 16;   // 103 MB units (layer1 64-ch tensor at B=256)
    for (int grid : {2048, 8192, 65536}) {
        run<1, 1>("copy 1:1 (103MB)", unit, grid);
        run<4, 4>("copy 4:4 (411MB)", unit, grid);
        run<1, 4>("ds-like  R1:W4", unit, grid);
        run<5, 4>("conv3-like R5:W4", unit, grid);
        run<4, 1>("conv1-like R4:W1", unit, grid);
        run<4, 0>("read only R4", unit, grid);
        run<0, 4>("write only W4", unit, grid);
    }
    return 0;
}
