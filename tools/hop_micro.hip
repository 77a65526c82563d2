// What a serial "hop" costs on a CDNA4 SIMD, and whether a second / third wave on the SIMD can use the time it waits -- the question behind
// the PNG decoder's literal walk (csrc/png_kernels.hip: v_readlane at a running offset, scalar shift + add, exit test, branch).
// Each kernel runs `iters` dependent iterations of one chain per wave; cycles per iteration from s_memtime, for 1, 2, 4 and 8 waves per SIMD
// (workgroups of one wave, grid = 256 CUs x 4 SIMDs x waves).  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/hop_micro.hip -o /tmp/hop_micro && /tmp/hop_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 0: scalar-only chain (shift, add, and, compare, branch)            -- SALU dependent issue
// 1: v_readlane at the running offset + the scalar chain              -- the decoder's hop
// 2: as 1, the value also stored to LDS by all lanes                  -- + one ds_write per hop
// 3: v_readfirstlane of a VALU result that depends on the scalar       -- vector -> scalar -> vector without a lane select
// 4: the chain kept on the vector unit (ds_bpermute gather per hop)    -- LDS crossbar latency, no scalar hand-off
template <int MODE>
__global__ __launch_bounds__(64) void hop_kernel(const int* __restrict__ table, int iters, long long* __restrict__ cycles, int* __restrict__ sink) {
    __shared__ unsigned char buf[128];
    const int lane = threadIdx.x;
    int ev = table[lane];                       // entries: (len << 9) | literal, len 1..11
    unsigned off = 0, acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        unsigned e = (unsigned)__builtin_amdgcn_readfirstlane(ev);
        for (int i = 0; i < iters; ++i) {
            off = (off + (e >> 9)) & 63;
            e = (e * 5 + off) | 512;
            if (e == 0x12345678u) break;
            acc += e;
        }
    } else if (MODE == 1 || MODE == 2) {
        unsigned e = (unsigned)__builtin_amdgcn_readlane(ev, 0);
        for (int i = 0; i < iters; ++i) {
            if (e & 256) break;                                    // (never: the table holds literals only)
            if (MODE == 2) buf[lane == 0 ? (i & 63) : 64 + lane] = (unsigned char)e;
            acc += e;
            off = (off + (e >> 9)) & 63;
            e = (unsigned)__builtin_amdgcn_readlane(ev, off);
        }
    } else if (MODE == 3) {
        unsigned e = (unsigned)__builtin_amdgcn_readfirstlane(ev);
        for (int i = 0; i < iters; ++i) {
            if (e & 256) break;
            acc += e;
            off = (off + (e >> 9)) & 63;
            e = (unsigned)__builtin_amdgcn_readfirstlane((ev + (int)off) & ~256);
        }
    } else {
        int o = 0, e = __builtin_amdgcn_ds_bpermute(0, ev);
        for (int i = 0; i < iters; ++i) {
            acc += (unsigned)e;
            o = (o + (e >> 9)) & 63;
            e = __builtin_amdgcn_ds_bpermute(o << 2, ev);
        }
        off = (unsigned)o;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        cycles[blockIdx.x] = t1 - t0;
        sink[blockIdx.x] = (int)(acc + off) + (MODE == 2 ? buf[3] : 0);
    }
}

int main() {
    int host_table[64];
    for (int i = 0; i < 64; ++i) host_table[i] = ((1 + (i * 7) % 11) << 9) | (i * 3 & 255);
    int *table, *sink;
    long long* cycles;
    const int max_blocks = 256 * 4 * 8;
    CK(hipMalloc(&table, sizeof(host_table)));
    CK(hipMemcpy(table, host_table, sizeof(host_table), hipMemcpyHostToDevice));
    CK(hipMalloc(&sink, max_blocks * sizeof(int)));
    CK(hipMalloc(&cycles, max_blocks * sizeof(long long)));
    const int iters = 200000;
    const char* names[] = {"scalar chain only", "v_readlane hop", "v_readlane hop + ds_write", "v_readfirstlane of a dependent VALU result", "ds_bpermute hop (vector only)"};
    std::vector<long long> h(max_blocks);
    for (int mode = 0; mode < 5; ++mode) {
        for (int waves : {1, 2, 4, 8}) {
            const int blocks = 256 * 4 * waves;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hop_kernel<0><<<blocks, 64>>>(table, iters, cycles, sink); break;
                    case 1: hop_kernel<1><<<blocks, 64>>>(table, iters, cycles, sink); break;
                    case 2: hop_kernel<2><<<blocks, 64>>>(table, iters, cycles, sink); break;
                    case 3: hop_kernel<3><<<blocks, 64>>>(table, iters, cycles, sink); break;
                    default: hop_kernel<4><<<blocks, 64>>>(table, iters, cycles, sink); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), cycles, blocks * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + blocks);
            printf("%-44s %d wave(s) per SIMD: %7.1f cycles per hop inside a wave (median; s_memtime), kernel %.2f ms = %.1f ns per hop per SIMD\n",
                   names[mode], waves, (double)h[blocks / 2] / iters, ms, ms * 1e6 / iters / waves);
        }
    }
    return 0;
}
