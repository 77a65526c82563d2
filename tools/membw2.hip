// Store-pattern calibration: does writing a [M][512 B] tensor as 128-row x 256-B half-row tiles (what a
// 128x128 conv tile's epilogue does for Cout=256 bf16) cost bandwidth versus full-row / linear writes?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each WG writes ROWS rows x SEG bytes at column offset (tile_n * SEG) of ROWB-byte rows
template <int SEG>
__global__ void tile_store(uint4* __restrict__ out, int rows_per_tile, int rowb, int n_tiles_n, int total_tiles, int remap) {
    int bid = blockIdx.x;
    if (remap) {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int tm = bid / n_tiles_n, tn = bid % n_tiles_n;
    constexpr int TPR = SEG / 16;
    const int rr = threadIdx.x / TPR, rc = threadIdx.x % TPR;
    char* base = (char*)out + (long long)tm * rows_per_tile * rowb + tn * SEG;
    for (int r = rr; r < rows_per_tile; r += 256 / TPR) {
        *(uint4*)(base + (long long)r * rowb + rc * 16) = make_uint4(r, rc, tm, tn);
    }
}

int main() {
    const long long M = 802816; const int rowb = 512;
    uint4* out; CK(hipMalloc(&out, M * rowb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto bench = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, (double)M * rowb / best / 1e9);
    };
    for (int remap = 0; remap < 2; ++remap) {
        printf("remap=%d\n", remap);
        bench("128 rows x 256B half rows (2 n-tiles)", [&] { hipLaunchKernelGGL(tile_store<256>, dim3(M / 128 * 2), dim3(256), 0, 0, out, 128, rowb, 2, 0, remap); });
        bench("128 rows x 512B full rows", [&] { hipLaunchKernelGGL(tile_store<512>, dim3(M / 128), dim3(256), 0, 0, out, 128, rowb, 1, 0, remap); });
        bench("64 rows x 512B full rows", [&] { hipLaunchKernelGGL(tile_store<512>, dim3(M / 64), dim3(256), 0, 0, out, 64, rowb, 1, 0, remap); });
        bench("128 rows x 128B quarter rows (4 n-tiles)", [&] { hipLaunchKernelGGL(tile_store<128>, dim3(M / 128 * 4), dim3(256), 0, 0, out, 128, rowb, 4, 0, remap); });
        bench("32 rows x 512B", [&] { hipLaunchKernelGGL(tile_store<512>, dim3(M / 32), dim3(256), 0, 0, out, 32, rowb, 1, 0, remap); });
    }
    return 0;
}
