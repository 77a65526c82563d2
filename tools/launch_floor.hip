// Launch floor of a chain of dependent kernels on one stream (hipGraph replay): how the fixed cost per launch moves with the launch
// geometry (workgroups, threads, dynamic LDS, kernel-argument bytes).  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args { char pad[136]; int flag; };
extern __shared__ char smem[];
__global__ void empty_kernel(Args a, float* out) {
    if (a.flag == 12345) out[threadIdx.x] = smem[threadIdx.x];   // never true: keeps the arguments and LDS alive
}
// touches global memory: one 16-byte load + store per thread of a resident buffer (a minimal "real" body)
__global__ void touch_kernel(Args a, const float4* in, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = in[i];
    v.x += (float)a.flag;
    out[i] = v;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float* buf;
    CK(hipMalloc(&buf, 64 << 20));
    CK(hipMemset(buf, 0, 64 << 20));
    CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Cfg { int grid, block, lds, touch; };
    const Cfg cfgs[] = {{256, 256, 0, 0}, {256, 512, 0, 0}, {256, 512, 114688, 0}, {256, 512, 147456, 0}, {256, 1024, 0, 0}, {512, 256, 0, 0},
                        {512, 256, 65536, 0}, {128, 512, 0, 0}, {64, 512, 0, 0}, {1024, 256, 0, 0}, {256, 64, 0, 0},
                        {256, 512, 0, 1}, {256, 256, 0, 1}};
    const int N = 40;
    for (const Cfg& c : cfgs) {
        Args a{};
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) {
            if (c.touch) hipLaunchKernelGGL(touch_kernel, dim3(c.grid), dim3(c.block), 0, s, a, (const float4*)buf + (i & 1) * (1 << 20), (float4*)buf + ((i + 1) & 1) * (1 << 20));
            else hipLaunchKernelGGL(empty_kernel, dim3(c.grid), dim3(c.block), c.lds, s, a, buf);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        std::vector<float> ts;
        for (int it = 0; it < 25; ++it) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 5) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("grid %4d block %4d lds %6d %s: %.2f us per launch (graph of %d)\n", c.grid, c.block, c.lds, c.touch ? "touch" : "empty",
               ts[ts.size() / 2] * 1e3 / N, N);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
