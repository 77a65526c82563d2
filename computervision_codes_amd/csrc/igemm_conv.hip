// Implicit-GEMM convolution for gfx950 (MI355X), channels-last, fp32 and bf16 on MFMA.
//
//   D[n][m] = sum_k Wp[n][k] * im2col(x)[m][k]      n = output channel, m = output pixel
//
// Both GEMM operands are K-contiguous: the packed weight row [Kpad] and, per kernel tap, the Cin run of
// one input pixel.  A K-step is 128 bytes of K per row (8 chunks of 16 B = 32 fp32 or 64 bf16), so the
// LDS image, the global->LDS staging and the b128 fragment reads are byte-identical for the two dtypes;
// only the MFMA differs (fp32: 4 x v_mfma_f32_16x16x4_f32 per 16-B fragment pair, exact fp32 FMA chain;
// bf16: 1 x v_mfma_f32_16x16x32_bf16).  The weight tile is the MFMA A operand and the pixel tile the B
// operand, so an accumulator lane owns 4 CONSECUTIVE output channels of one pixel: the fused epilogue
// (bias + residual + ReLU) loads/stores 16-B (fp32) or 8-B (bf16) vectors along the NHWC channel axis.
//
// LDS: [stage][row][8 chunks], chunk index XOR-swizzled with (row & 7): conflict-free for the 8-lane
// ds_write_b128 groups (one 128-B row) and the 16-lane ds_read_b128 groups (16 rows x one chunk).
// Tiles are dealt to XCDs in contiguous ranges (n-tile fastest), so all column tiles of a pixel tile hit
// one XCD's L2 and activations are fetched from HBM once.
#include "mt4_common.h"
#include <cstdlib>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0, num_records = bytes, gfx9 data-format word) from wave-uniform values
__device__ __forceinline__ v4u make_srd(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    v4u r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)u);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// N LDS-DMA pieces in ONE asm statement (one M0 save/restore): piece i goes to LDS [lds_addr + i*STRIDE + lane*16) from
// buffer offset voff[i] (per lane, range-checked: out-of-range lanes write zeros) + soff (wave-uniform, NOT range-checked).
// STRIDE = bytes one staging pass of the whole workgroup covers (waves * 8 rows * 128 B).
template <int N, int STRIDE>
__device__ __forceinline__ void lds_dma16_group(v4u srd, const unsigned (&voff)[N], unsigned soff, unsigned lds_addr) {
    unsigned keep;
    if constexpr (N == 1) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
    } else if constexpr (N == 2) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                     "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(srd), "s"(soff), "s"(lds_addr), "n"(STRIDE) : "memory", "scc");
    } else {
        static_assert(N == 4, "1, 2 or 4 pieces");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
                     "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
                     "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                     "s_add_u32 m0, m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(srd), "s"(soff), "s"(lds_addr),
                       "n"(STRIDE)
                     : "memory", "scc");
    }
}

// one LDS-DMA piece: 64 lanes x 16 B from buffer offset `voff` (per lane) to LDS [lds_addr + lane*16); out-of-range
// lanes write zeros.  M0 (the DMA's LDS base) is compiler-reserved: saved, set and restored inside the statement.
__device__ __forceinline__ void lds_dma16(v4u srd, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(srd), "s"(lds_addr)
        : "memory");
}

struct ConvK {
    const char* x;
    const char* w;
    const float* bias;
    const char* res;
    char* y;
    const int* row_map;  // optional: output/residual row of pixel m = (m / map_len) * map_len + row_map[m % map_len]
    int map_len;
    int map_img;         // output rows per image on the y/residual side (map_len when the map is a permutation)
    int y_ld, res_ld;    // row pitch (elements) of y / residual; Cout when dense
    int res_f32;         // bf16 operands with fp32 output only: the residual is fp32 as well (mixed-precision training GEMMs)
    int B, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, ph, pw, dh, dw, relu;  // relu: 0 none, 1 ReLU, 2 GELU(erf)
    int M, HoWo, CPT, SPT, taps, nsteps, n_tiles, total_tiles;
    long long x_img_bytes;  // H*W*pix_bytes
    int pix_bytes;          // bytes between neighbouring pixels of x: Cin*esize, or less when a K-row spans several pixels (x_pixel_stride)
    unsigned x_bytes, w_bytes;  // buffer sizes for the LDS-DMA range check (patch kernels: x < 2 GiB; weights < 2 GiB)
    long long x_total_bytes;    // B*H*W*pix_bytes (the generic FAST path addresses x relative to a per-tile origin)
    unsigned x_bias;            // (pad_h*W + pad_w)*Cin*esize: how far a padded corner reaches in front of x
    int w_row_bytes;        // nsteps*128
    int nt_epi;             // residual rows are loaded and bf16 output rows stored non-temporally (each is touched once by this launch;
                            // same-box A/B on the ResNet-50 bench: +2.6 % frames/s).  MT4_NO_NT=1 switches it off
    // f_*: the conv3 that the EXPAND form of the 3x3 patch kernel runs behind its own conv (mt4_conv_desc.fuse_expand)
    // DUAL: a second K source behind the first (a 1x1 conv, stride 1): the block input x2 [B][H2][W2][C2] gathered at stride x2_s (the
    // downsample branch of a strided Bottleneck, resnet.py:116-119): y = act([W | W2] . [x ; x2(s*ho, s*wo)] + bias) in ONE accumulator chain
    const char* x2;
    long long x2_img_bytes, x2_total_bytes;
    int x2_W, x2_s, x2_pix_bytes, x2_HoWo, x2_Wo;
    int nsteps1;            // K-steps of the first source
    double* stat_sums;      // mt4_conv_desc.stat_sums: [MT4_STAT_REPLICAS][2][Cout], += sum | sum of squares of the stored values
    const char* f_w;        // packed [f_cout][Cout] bf16 (mt4_pack_conv_weight, 1x1)
    const float* f_bias;
    char* f_y;
    int f_cout, f_relu, f_w_row_bytes;
};

// Per-channel sums of a launch's STORED values for the train-mode BatchNorm behind a convolution (mt4_conv_desc.stat_sums): every thread of the
// staged epilogue keeps the float64 sum and sum of squares of its CH channels over the rows it stores; here the threads that share a channel
// group are folded (lanes of a wave by shuffle, waves through LDS) and the tile's totals are added to one of MT4_STAT_REPLICAS copies of the
// [2][Cout] sums (copy = pixel tile % replicas: the same-address chains of float64 atomics at the L2 stay ~1/8 of the pixel tiles long).
// float64 throughout: the totals equal those of a float64 pass over the stored map to ~1e-16, i.e. the same float32 mean / invstd.
template <int TPR, int NTH, int CH, int BN>
__device__ __forceinline__ void colstat_flush(double (&cs)[CH], double (&cq)[CH], char* smem, double* __restrict__ sums, int n0, int Cout, int tid) {
    static_assert(TPR <= 64 && 64 % TPR == 0, "a wave covers whole rows of the read-back");
    constexpr int NW = NTH / 64;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = TPR; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < CH; ++e) { cs[e] += __shfl_xor(cs[e], o); cq[e] += __shfl_xor(cq[e], o); }
    }
    double* red = (double*)smem;   // [NW][2][BN]
    __syncthreads();               // the staging rows have been read back
    if (lane < TPR) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            red[(wave * 2) * BN + lane * CH + e] = cs[e];
            red[(wave * 2 + 1) * BN + lane * CH + e] = cq[e];
        }
    }
    __syncthreads();
    for (int col = tid; col < 2 * BN; col += NTH) {
        const int h = col / BN, c = col - h * BN;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(w * 2 + h) * BN + c];
        if (n0 + c < Cout) atomicAdd(sums + (long long)h * Cout + n0 + c, t);
    }
}

// KS > 1 (FAST only): the workgroup holds KS groups of WAVES_M x WAVES_N waves; group g runs the K-steps g, g+KS, ... of the SAME output
// tile through its own operand stages and the partial tiles are added in LDS in the fixed order g = 0, 1, ... before the epilogue
// (deterministic, independent of the batch).  For launches with few tiles and a long K (a TCN layer over one short video: 128 workgroups,
// 48 K-steps) the K loop -- one barrier and one DMA round trip per step -- is the launch's critical path; this cuts it KS-fold.
// STATS: the epilogue also adds the channel sums of the stored values to a.stat_sums (colstat_flush); a separate instantiation, so that the
// float64 accumulators cost the inference launches no registers
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool FAST, bool OUT_F32, int KS = 1, bool DUAL = false, bool STATS = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * KS * 64) void igemm_conv_kernel(const ConvK a) {
    static_assert(!DUAL || (FAST && sizeof(T) == 2 && KS == 1), "the second K source exists on the bf16 LDS-DMA path");
    static_assert(!STATS || (KS == 1 && !DUAL && WAVES_M * WAVES_N * BN * 16 <= (BM + BN) * 128), "channel sums: one-group forms whose LDS holds the waves' partial rows");
    constexpr int ES = (int)sizeof(T);
    constexpr int ROWS = BM + BN;
    constexpr int NWG = WAVES_M * WAVES_N;  // waves of one group
    constexpr int NW = NWG * KS;            // 4 waves, or 8 / 16 for the 256-row tiles (1 workgroup per CU)
    constexpr int NTH = NW * 64;
    constexpr int GTH = NWG * 64;
    constexpr int RPP = GTH / 8;            // operand rows one staging pass of a group covers (8 lanes per 128-B row)
    static_assert(KS == 1 || FAST, "K-split groups exist on the LDS-DMA path only");
    constexpr int NLD = ROWS / RPP;
    constexpr int NLD_X = BM / RPP;
    constexpr int NLD_W = BN / RPP;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int STAGE_BYTES = ROWS * 128;
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = KS > 1 ? wave / NWG : 0;         // K-split group of this wave
    const int wave_l = KS > 1 ? wave % NWG : wave;   // wave within its group
    const int tid_l = KS > 1 ? tid % GTH : tid;
    const int wave_m = wave_l / WAVES_N, wave_n = wave_l % WAVES_N;
    const int r16 = lane & 15, q = lane >> 4;

    // ---- block -> tile, XCD-contiguous (bijective for any grid size), channel tile fastest: the column tiles of a
    // pixel tile run at the same time on ONE XCD, so activations are fetched from HBM once and shared through L2.
    // One tile per workgroup: a persistent variant (workgroups walking several tiles, next tile's first K-step and
    // residual rows prefetched across the epilogue) measured SLOWER on every ResNet-50 layer -- its extra live
    // registers cost a wave per SIMD; hardware workgroup turnover already provides that overlap.
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7;
        const int xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int tile_m = bid / a.n_tiles;
    const int tile_n = bid - tile_m * a.n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int ld_row = tid_l >> 3, ld_chunk = tid_l & 7;
    const int st_off = ld_row * 128 + ((ld_chunk ^ (ld_row & 7)) << 4);

    // the accumulators start at the bias of their 4 channels: no bias add in the epilogue (K-split: group 0 only)
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + wave_n * WN + i * 16 + q * 4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias && grp == 0) {
            if ((a.Cout & 3) == 0) {
                if (n < a.Cout) { const float4 t = *(const float4*)(a.bias + n); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < a.Cout) b4[e] = a.bias[n + e];
            }
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = b4;
    }

    const int rd_x = (wave_m * WM + r16) * 128;
    const int rd_w = (BM + wave_n * WN + r16) * 128;
    const int nsteps = a.nsteps;

    auto compute = [&](const char* sb) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw_off = ((kk * 4 + q) ^ (r16 & 7)) << 4;
            uint4 fx[MT], fw[NT];
#pragma unroll
            for (int j = 0; j < MT; ++j) fx[j] = *(const uint4*)(sb + rd_x + j * 16 * 128 + sw_off);
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *(const uint4*)(sb + rd_w + i * 16 * 128 + sw_off);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fx[j]), acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[i].x), __uint_as_float(fx[j].x),
                                                                         acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[i].y), __uint_as_float(fx[j].y),
                                                                         acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[i].z), __uint_as_float(fx[j].z),
                                                                         acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[i].w), __uint_as_float(fx[j].w),
                                                                         acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    if constexpr (FAST) {
        // ================= LDS-DMA staging (`buffer_load_dwordx4 ... lds`) =================
        // No staging registers and no ds_write: each wave-instruction drops 8 rows x 128 B straight into the stage.  The
        // DMA destination is lane-linear, so the XOR swizzle sits on the SOURCE side: the lane at LDS position
        // (row, c') fetches global chunk c' ^ (row & 7).  Out-of-image taps, rows >= M and channels >= Cout use an
        // out-of-range buffer offset: the hardware range check writes ZEROS to LDS for them (probed on gfx950,
        // tools/probe_lds_dma.hip), so padding costs no branch.  Tap validity per staged row is a 64-bit mask built
        // once; per K-step a load costs one add + one select.
        // The DMA is issued from inline asm: through the builtin, hipcc cannot tell the stage being filled from the
        // stage being read and puts `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-step (the load
        // latency then serialises with the MFMAs).  Hidden in asm, the loads stay in flight across the compute phase
        // and are drained by ONE explicit vmcnt(0) in front of the barrier.
        // The x descriptor starts x_bias bytes BEFORE the tensor, x_bias = the farthest a padded corner reaches back:
        // per-lane offsets (pixel (hi0,wi0) + x_bias) are then never negative, and the wave-uniform tap/channel step rides in
        // the instruction's soffset (not range-checked).  Bytes in front of x are never fetched: their taps are invalid.
        // 32-bit buffer offsets are taken from a per-workgroup origin (the image of the tile's first pixel), so the tensor itself may be
        // larger than 2 GiB: only the bytes one tile spans (a few images) must fit the offset range (host-checked)
        const long long org = (long long)(m0 / a.HoWo) * a.x_img_bytes;
        const long long left = a.x_total_bytes - org + a.x_bias;
        const v4u rsx = make_srd(a.x - a.x_bias + org, (unsigned)(left < 0x7fffffffLL ? left : 0x7fffffffLL));
        const v4u rsw = make_srd(a.w, a.w_bytes);
        constexpr unsigned OOB = 0x80000000u;
        const int gch = ld_chunk ^ (ld_row & 7);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave_l);
        const int grp_u = __builtin_amdgcn_readfirstlane(grp);
        char* const smem_g = smem + grp_u * (STAGES * STAGE_BYTES);   // this group's ring of operand stages
        const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_g;
        int xoff[NLD_X];
        // tap validity is separable: byte i of hmask/wmask holds, for staged row i, one bit per kh / kw (KH, KW <= 8)
        unsigned hmask = 0, wmask = 0;
        static_assert(NLD_X <= 4, "one mask byte per staged pixel row");
#pragma unroll
        for (int i = 0; i < NLD_X; ++i) {
            const int m = m0 + ld_row + RPP * i;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            int hi0 = 0, wi0 = 0;
            long long base;
            if (a.HoWo == 1) {   // one output pixel per image (the pure-GEMM remap, or a padded kernel over a 1x1 image)
                hi0 = -a.ph;
                wi0 = -a.pw;
                base = (long long)mm * a.x_img_bytes + (long long)(hi0 * a.W + wi0) * a.pix_bytes;
            } else {
                const int b = mm / a.HoWo;
                const int rem = mm - b * a.HoWo;
                const int ho = rem / a.Wo;
                const int wo = rem - ho * a.Wo;
                hi0 = ho * a.sh - a.ph;
                wi0 = wo * a.sw - a.pw;
                base = (long long)b * a.x_img_bytes + (long long)(hi0 * a.W + wi0) * a.pix_bytes;
            }
            xoff[i] = (int)(base - org) + gch * 16 + (int)a.x_bias;   // >= 0
            if (ok) {
                for (int kh = 0; kh < a.KH; ++kh)
                    if ((unsigned)(hi0 + kh * a.dh) < (unsigned)a.H) hmask |= 1u << (8 * i + kh);
                for (int kw = 0; kw < a.KW; ++kw)
                    if ((unsigned)(wi0 + kw * a.dw) < (unsigned)a.W) wmask |= 1u << (8 * i + kw);
            }
        }
        unsigned woff[NLD_W];
#pragma unroll
        for (int i = 0; i < NLD_W; ++i) {
            const int n = n0 + ld_row + RPP * i;
            woff[i] = n < a.Cout ? (unsigned)(n * a.w_row_bytes + gch * 16) : OOB;
        }
        // DUAL: per staged row the offset of the block-input pixel (s * ho, s * wo) from the image of the tile's first pixel
        unsigned xoff2[DUAL ? NLD_X : 1];
        v4u rsx2 = rsx;
        if constexpr (DUAL) {
            const long long org2 = (long long)(m0 / a.x2_HoWo) * a.x2_img_bytes;
            const long long left2 = a.x2_total_bytes - org2;
            rsx2 = make_srd(a.x2 + org2, (unsigned)(left2 < 0x7fffffffLL ? left2 : 0x7fffffffLL));
#pragma unroll
            for (int i = 0; i < NLD_X; ++i) {
                const int m = m0 + ld_row + RPP * i;
                const int mm = m < a.M ? m : 0;
                const int b = mm / a.x2_HoWo;
                const int rem = mm - b * a.x2_HoWo;
                const int ho = rem / a.x2_Wo;
                const int wo = rem - ho * a.x2_Wo;
                const long long base2 = (long long)b * a.x2_img_bytes + (long long)(ho * a.x2_s * a.x2_W + wo * a.x2_s) * a.x2_pix_bytes;
                xoff2[i] = m < a.M ? (unsigned)((int)(base2 - org2) + gch * 16) : OOB;
            }
        }
        int f_kh = 0, f_kw = 0, f_cs = 0;
        auto issue = [&](int step, int stage) {
            if constexpr (DUAL) {
                if (step >= a.nsteps1) {     // (wave-uniform) K-steps of the second source: weight K-step = step (the first source is a 1x1 conv)
                    const unsigned dst2 = lds_base + stage * STAGE_BYTES + wave_u * 1024;
                    lds_dma16_group<NLD_X, RPP * 128>(rsx2, xoff2, (unsigned)__builtin_amdgcn_readfirstlane((step - a.nsteps1) * 128), dst2);
                    lds_dma16_group<NLD_W, RPP * 128>(rsw, woff, (unsigned)__builtin_amdgcn_readfirstlane(step * 128), dst2 + BM * 128);
                    return;
                }
            }
            const int delta = (f_kh * a.dh * a.W + f_kw * a.dw) * a.pix_bytes + f_cs * 128;   // wave-uniform
            const unsigned bits = (hmask >> f_kh) & (wmask >> f_kw);   // bit 8*i: row i valid for this tap
            // K walk order: taps fastest, the 128-byte channel slice slowest (the packed weights are tap-major, so the weight K-step
            // is tap*SPT + slice).  Consecutive K-steps then read the same cache lines of x shifted by one pixel / one image row
            // and the re-read comes 1-3 steps later instead of SPT*taps later (+1.2 % frames/s, same-box A/B).
            const int wstep = (f_kh * a.KW + f_kw) * a.SPT + f_cs;
#pragma unroll
            for (int adv = 0; adv < KS; ++adv) {   // this group's next K-step is KS steps on
                if (++f_kw == a.KW) {
                    f_kw = 0;
                    if (++f_kh == a.KH) { f_kh = 0; ++f_cs; }
                }
            }
            // (keep the K position in SGPRs: without this hipcc carries it in VGPRs and multiplies with v_mul_lo_u32)
            f_cs = __builtin_amdgcn_readfirstlane(f_cs);
            f_kw = __builtin_amdgcn_readfirstlane(f_kw);
            f_kh = __builtin_amdgcn_readfirstlane(f_kh);
            const unsigned dst = lds_base + stage * STAGE_BYTES + wave_u * 1024;   // wave-uniform LDS byte address
            unsigned vx[NLD_X];
#pragma unroll
            for (int i = 0; i < NLD_X; ++i) vx[i] = ((bits >> (8 * i)) & 1u) ? (unsigned)xoff[i] : OOB;
            lds_dma16_group<NLD_X, RPP * 128>(rsx, vx, (unsigned)__builtin_amdgcn_readfirstlane(delta), dst);
            lds_dma16_group<NLD_W, RPP * 128>(rsw, woff, (unsigned)__builtin_amdgcn_readfirstlane(wstep * 128), dst + BM * 128);
        };
        // Ring of STAGES operand stages, DMA issued STAGES-1 K-steps ahead.  After computing step s the groups of steps
        // s+1 .. s+STAGES-1 are outstanding (fewer at the tail); only the OLDEST must have landed, so the wait leaves the
        // younger groups in flight: counted vmcnt (loads retire in issue order), then the barrier (everyone's pieces
        // landed + all reads of the stage that the next iteration refills are done).
        auto wait_keep = [&](int groups) {   // wait until at most `groups` K-step groups of this wave are outstanding
            if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        };
        static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * NLD <= 63, "wait_keep covers up to 2 groups in flight");
        // K-split: this group walks the steps grp, grp + KS, ...; every group runs the same number of iterations (the barrier is the
        // workgroup's), a group that has run out of steps idles through them
        int my_steps = nsteps, iters = nsteps;
        if constexpr (KS > 1) {
            my_steps = (nsteps - grp_u + KS - 1) / KS;
            iters = (nsteps + KS - 1) / KS;
            for (int adv = 0; adv < grp_u; ++adv) {   // start at K-step grp
                if (++f_kw == a.KW) {
                    f_kw = 0;
                    if (++f_kh == a.KH) { f_kh = 0; ++f_cs; }
                }
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < STAGES - 1; ++s0)
            if (s0 < my_steps) issue(s0, s0);
        wait_keep((my_steps < STAGES - 1 ? my_steps : STAGES - 1) - 1);
        __syncthreads();
        int cur = 0, fill = STAGES - 1;   // stage being computed / stage the next issue refills
        for (int step = 0; step < iters; ++step) {
            if (step + STAGES - 1 < my_steps) issue(step + STAGES - 1, fill);
            if (KS == 1 || step < my_steps) compute(smem_g + cur * STAGE_BYTES);
            const int rem = my_steps - 1 - step;   // K-step groups still outstanding after this one
            wait_keep((rem < STAGES - 1 ? rem : STAGES - 1) - 1);
            __syncthreads();
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            fill = fill + 1 == STAGES ? 0 : fill + 1;
        }
        if constexpr (KS > 1) {
            // partial tiles of groups 1.. -> LDS in accumulator layout; group 0 adds them in order
            f32x4* red = (f32x4*)smem;
            if (grp > 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) red[(((grp - 1) * NWG + wave_l) * (NT * MT) + i * MT + j) * 64 + lane] = acc[i][j];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int g = 1; g < KS; ++g)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j) acc[i][j] += red[(((g - 1) * NWG + wave_l) * (NT * MT) + i * MT + j) * 64 + lane];
            }
            __syncthreads();
        }
    } else {
        // ================= register staging (generic geometry: chunk-granular tap decode, e.g. the stem) =================
        const char* px_base[NLD_X];
        int px_hi0[NLD_X], px_wi0[NLD_X];
        bool px_ok[NLD_X];
#pragma unroll
        for (int i = 0; i < NLD_X; ++i) {
            const int m = m0 + ld_row + RPP * i;
            px_ok[i] = m < a.M;
            const int mm = px_ok[i] ? m : 0;
            const int b = mm / a.HoWo;
            const int rem = mm - b * a.HoWo;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            px_hi0[i] = ho * a.sh - a.ph;
            px_wi0[i] = wo * a.sw - a.pw;
            px_base[i] = a.x + (long long)b * a.x_img_bytes;
        }
        const char* w_ptr[NLD_W];
        bool w_ok[NLD_W];
#pragma unroll
        for (int i = 0; i < NLD_W; ++i) {
            const int n = n0 + ld_row + RPP * i;
            w_ok[i] = n < a.Cout;
            w_ptr[i] = a.w + (long long)(w_ok[i] ? n : 0) * a.w_row_bytes + ld_chunk * 16;
        }
        uint4 stg[NLD];
        auto load_step = [&](int step) {
            const int g = step * 8 + ld_chunk;
            const int tap = g / a.CPT;
            const int cc = g - tap * a.CPT;
            const int kh = tap / a.KW;
            const int kw = tap - kh * a.KW;
            const int choff = cc * 16;
            const bool tap_ok = tap < a.taps;
            const int dhi = kh * a.dh, dwi = kw * a.dw;
#pragma unroll
            for (int i = 0; i < NLD_X; ++i) {
                const int hi = px_hi0[i] + dhi, wi = px_wi0[i] + dwi;
                const bool ok = tap_ok && px_ok[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *(const uint4*)(px_base[i] + (long long)(hi * a.W + wi) * a.pix_bytes + choff);
                stg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < NLD_W; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (w_ok[i]) v = *(const uint4*)(w_ptr[i] + (long long)step * 128);
                stg[NLD_X + i] = v;
            }
        };
        auto store_stage = [&](int stage) {
            char* base = smem + stage * STAGE_BYTES + st_off;
#pragma unroll
            for (int i = 0; i < NLD; ++i) *(uint4*)(base + i * RPP * 128) = stg[i];
        };
        load_step(0);
        store_stage(0);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < nsteps;
            if (more) load_step(step + 1);
            compute(smem + cur * STAGE_BYTES);
            if (more) store_stage(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- fused epilogue: bias + residual + relu
    // Staged path (Cout % CH == 0): the fp32 accumulators (+bias) go through LDS as [pixel][channel] rows so that every
    // lane adds the residual and stores the result as ONE 16-byte vector and a wave-instruction covers whole rows
    // (BN*osize contiguous bytes per pixel) instead of 16 scattered 8/16-byte pieces.
    constexpr int OS = OUT_F32 ? 4 : 2;
    constexpr int CH = 16 / OS;        // channels per 16-byte output vector
    constexpr int ROWB = BN * 4 + 16;  // fp32 row + 16 B pad: conflict-free ds_write_b128 per 8 lanes
    constexpr int PASSES = (BM * ROWB > 4 * STAGE_BYTES) ? 4 : (BM * ROWB > 2 * STAGE_BYTES) ? 2 : 1;   // (BM*ROWB <= 8 stages for every tile)
    static_assert(BM / PASSES * ROWB <= 2 * STAGE_BYTES, "epilogue staging fits the main-loop LDS");
    static_assert(MT % PASSES == 0, "passes split the m-tiles of a wave");
    constexpr int MTP = MT / PASSES;  // m-tiles per wave per pass
    constexpr int WMP = WM / PASSES;  // rows per wave_m group per pass
    constexpr int BMP = BM / PASSES;
    constexpr int TPR = BN / CH;      // threads per row on read-back
    constexpr int RPI = NTH / TPR;    // rows per read-back iteration
    constexpr int ITERS = (BMP + RPI - 1) / RPI;
    constexpr int RB = (sizeof(T) == 2 && CH == 4) ? 8 : 16;  // residual bytes per output vector
    constexpr bool MIXED = sizeof(T) == 2 && OUT_F32;          // bf16 operands, fp32 output: the residual may be fp32 (a.res_f32)
    if ((a.Cout % CH) == 0) {
        // (the K loop ended with a barrier: every wave is done reading the operand stages)
        const int rc = tid % TPR, rr = tid / TPR;
        const int n = n0 + rc * CH;
        double cs[STATS ? CH : 1], cq[STATS ? CH : 1];
        if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < CH; ++e) cs[e] = cq[e] = 0.0;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            // residual rows of this pass go in flight BEFORE the LDS hand-off, all at once: issued inside the read-back
            // loop each load was waited for on the spot (one full HBM latency per row group, serialised)
            long long orow[ITERS];
            uint4 rres[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lrow = rr + it * RPI;
                const int trow = (lrow / WMP) * WM + p * WMP + (lrow % WMP);
                int m = m0 + trow;
                orow[it] = -1;
                rres[it] = make_uint4(0, 0, 0, 0);
                if (lrow < BMP && m < a.M && n < a.Cout) {
                    if (a.row_map) m = (m / a.map_len) * a.map_img + a.row_map[m % a.map_len];
                    orow[it] = m;
                    if (a.res) {
                        const char* rp = a.res + ((long long)m * a.res_ld + n) * ES;
                        if (MIXED && a.res_f32) rres[it] = *(const uint4*)(a.res + ((long long)m * a.res_ld + n) * 4);
                        else if constexpr (RB == 16) {   // read once: non-temporal
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            if (a.nt_epi) { const u32x4 t = __builtin_nontemporal_load((const u32x4*)rp); rres[it] = make_uint4(t.x, t.y, t.z, t.w); }
                            else rres[it] = *(const uint4*)rp;
                        }
                        else { const uint2 r2 = *(const uint2*)rp; rres[it] = make_uint4(r2.x, r2.y, 0, 0); }
                    }
                }
            }
            if (p) __syncthreads();
            if (KS == 1 || grp == 0) {
#pragma unroll
                for (int jj = 0; jj < MTP; ++jj) {
                    const int lrow = wave_m * WMP + jj * 16 + r16;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int nl = wave_n * WN + i * 16 + q * 4;
                        *(f32x4*)(smem + lrow * ROWB + nl * 4) = acc[i][p * MTP + jj];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                if (orow[it] < 0) continue;
                const int lrow = rr + it * RPI;
                float v[CH];
                *(float4*)&v[0] = *(const float4*)(smem + lrow * ROWB + rc * CH * 4);
                if constexpr (CH == 8) *(float4*)&v[4] = *(const float4*)(smem + lrow * ROWB + rc * CH * 4 + 16);
                const long long o = orow[it] * a.y_ld + n;
                if (a.res && a.relu == 3) {   // ReLU backward: pass the gradient where the forward activation was positive
                    const uint4 rv = rres[it];
                    if (MIXED && a.res_f32) {
                        if (!(__uint_as_float(rv.x) > 0.f)) v[0] = 0.f;
                        if (!(__uint_as_float(rv.y) > 0.f)) v[1] = 0.f;
                        if (!(__uint_as_float(rv.z) > 0.f)) v[2] = 0.f;
                        if (!(__uint_as_float(rv.w) > 0.f)) v[3] = 0.f;
                    } else if constexpr (sizeof(T) == 2) {
                        const uint32_t u[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int e = 0; e < CH / 2; ++e) {
                            if (!(__uint_as_float(u[e] << 16) > 0.f)) v[2 * e] = 0.f;
                            if (!(__uint_as_float(u[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
                        }
                    } else {
                        if (!(__uint_as_float(rv.x) > 0.f)) v[0] = 0.f;
                        if (!(__uint_as_float(rv.y) > 0.f)) v[1] = 0.f;
                        if (!(__uint_as_float(rv.z) > 0.f)) v[2] = 0.f;
                        if (!(__uint_as_float(rv.w) > 0.f)) v[3] = 0.f;
                    }
                } else if (a.res) {
                    const uint4 rv = rres[it];
                    if (MIXED && a.res_f32) {
                        v[0] += __uint_as_float(rv.x); v[1] += __uint_as_float(rv.y);
                        v[2] += __uint_as_float(rv.z); v[3] += __uint_as_float(rv.w);
                    } else if constexpr (sizeof(T) == 2) {
                        const uint32_t u[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int e = 0; e < CH / 2; ++e) {
                            v[2 * e] += __uint_as_float(u[e] << 16);
                            v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
                        }
                    } else {
                        v[0] += __uint_as_float(rv.x); v[1] += __uint_as_float(rv.y);
                        v[2] += __uint_as_float(rv.z); v[3] += __uint_as_float(rv.w);
                    }
                }
                if (a.relu == 1) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (a.relu == 2) {
                    gelu_erf_n(v);
                }
                if constexpr (OUT_F32) {
                    *(float4*)(a.y + o * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (STATS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const double dv = (double)v[e]; cs[e] += dv; cq[e] = fma(dv, dv, cq[e]); }
                    }
                } else {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                    if (a.nt_epi) __builtin_nontemporal_store(ov, (u32x4*)(a.y + o * 2));
                    else *(uint4*)(a.y + o * 2) = make_uint4(ov.x, ov.y, ov.z, ov.w);
                    if constexpr (STATS) {   // of the rounded values: the statistics of the tensor BatchNorm reads
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double lo = (double)__uint_as_float(ov[e] << 16), hi = (double)__uint_as_float(ov[e] & 0xffff0000u);
                            cs[2 * e] += lo; cq[2 * e] = fma(lo, lo, cq[2 * e]); cs[2 * e + 1] += hi; cq[2 * e + 1] = fma(hi, hi, cq[2 * e + 1]);
                        }
                    }
                }
            }
        }
        if constexpr (STATS)
            colstat_flush<TPR, NTH, CH, BN>(cs, cq, smem, a.stat_sums + (long long)(tile_m & (MT4_STAT_REPLICAS - 1)) * 2 * a.Cout, n0, a.Cout, tid);
        return;
    }
    // Direct path (ragged Cout, e.g. the 131-wide concatenated heads): per-element guards
    if (KS > 1 && grp != 0) return;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + wave_m * WM + j * 16 + r16;
        if (m >= a.M) continue;
        if (a.row_map) m = (m / a.map_len) * a.map_img + a.row_map[m % a.map_len];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + wave_n * WN + i * 16 + q * 4;
            if (n >= a.Cout) continue;
            const f32x4 v = acc[i][j];
            const long long o = (long long)m * a.y_ld + n;
            const long long ro = (long long)m * a.res_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= a.Cout) break;
                float f = v[e];
                if (a.res) {
                    float rvv;
                    if ((sizeof(T) == 2 && OUT_F32) && a.res_f32) rvv = *(const float*)(a.res + (ro + e) * 4);
                    else if constexpr (sizeof(T) == 2) rvv = bf16_to_f32(*(const u16*)(a.res + (ro + e) * 2));
                    else rvv = *(const float*)(a.res + (ro + e) * 4);
                    if (a.relu == 3) f = rvv > 0.f ? f : 0.f;
                    else f += rvv;
                }
                if (a.relu == 1) f = fmaxf(f, 0.f);
                else if (a.relu == 2) f = gelu_erf(f);
                if constexpr (OUT_F32) *(float*)(a.y + (o + e) * 4) = f;
                else *(u16*)(a.y + (o + e) * 2) = f32_to_bf16(f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 patch kernel
// 3x3 / stride 1 / pad 1 / dense NHWC bf16: the generic kernel above re-fetches the pixel tile for each of the 9 taps (9 x BM rows
// through L2 -> LDS per 64-channel slice), and the K-loop of the N <= 128 layers waits on exactly that fill.  Here the INPUT PATCH
// of the pixel tile is staged once per slice: output pixels are linear in (b, h, w) and so are input pixels (same H x W), so tap
// (kh, kw) of tile row r is patch row r + kh*W + kw of the linear pixel range [m0 - W - 1, m0 + BM + W + 1).  A tap that leaves the
// image (wraps to the neighbouring row / image in linear space) contributes zeros: 9 validity bits per lane and
// m-tile; such a tap reads a row of zeros (a select on the LDS address).  Fill per slice: (BM + 2W + 2) + 9*BN rows instead of 9*(BM + BN).
// Epilogue: bias (in the accumulators) + optional residual + optional ReLU.
// Same K walk (slice outer, taps inner) and the same MFMA chain as the generic kernel: results are bit-identical.
// Weight K-steps (BN rows x 128 B) run through a ring of WS stages, issued WS-1 steps ahead: a K-step of the narrow tiles is shorter
// than the L2 -> LDS latency, so one step of prefetch distance leaves every step waiting for its weights.
// EXPAND (BN == 128 = every channel of the layer): the Bottleneck's conv3 + bn3 + add + ReLU (resnet.py:112-119) follows in the same launch.
// The tile's ReLU'd bf16 result -- what the stand-alone launch stores -- stays in LDS as the pixel operand of a second GEMM over K = 128 against
// the fragment-ordered expansion weights (a.f_w, 128-channel chunks of a.f_cout outputs); each chunk goes through the fp32 staging with the
// block's residual (a.res, pitch f_cout) and is stored to a.f_y.  a.y is not written: the 128-channel map never reaches memory.  Same K order
// and bias-initialised accumulators as the stand-alone 1x1 launch: bit-identical.
// STATS: the epilogue also adds the channel sums of the stored values to a.stat_sums -- a separate instantiation, like the generic kernel's, so
// that the float64 accumulators cost the inference launches no registers.
template <int BM, int BN, int WAVES_M, int WAVES_N, int WS, bool EXPAND = false, bool STATS = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N == 4 ? 2 : 1) void conv3x3_patch_kernel(const ConvK a, const int pra, const int npatch) {
    static_assert(!EXPAND || BN == 128, "the expansion needs a pixel's whole channel vector in the tile");
    static_assert(!(EXPAND && STATS), "the expansion form takes no statistics");
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NTH = NW * 64;
    constexpr int RPP = NTH / 8;
    constexpr int NLD_W = BN >= RPP ? BN / RPP : 1;   // (BN < RPP: only the first BN/8 waves stage weight rows)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int WSTAGE = BN * 128;
    static_assert((BN % RPP == 0 || RPP % BN == 0) && WM % 16 == 0 && WN % 16 == 0 && MT <= 4, "tile shape");
    static_assert((WS >= 2 && WS <= 4) || WS == 9, "weight ring depth (9: all taps of a single-slice layer resident, no barrier in the K loop)");
    constexpr bool RESIDENT = WS == 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int r16 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7;
        const int xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int tile_m = bid / a.n_tiles;
    const int tile_n = bid - tile_m * a.n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ld_row = tid >> 3, ld_chunk = tid & 7;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + wave_n * WN + i * 16 + q * 4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias && n < a.Cout) { const float4 t = *(const float4*)(a.bias + n); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = b4;
    }

    // tap validity of this lane's pixel in each m-tile: bit kh*3+kw
    int vm[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        int m = m0 + wave_m * WM + j * 16 + r16;
        m = m < a.M ? m : a.M - 1;
        const int rem = m % a.HoWo;
        const int ho = rem / a.W;
        const int wo = rem - ho * a.W;
        const int wv = (wo > 0 ? 1 : 0) | 2 | (wo < a.W - 1 ? 4 : 0);
        vm[j] = (ho > 0 ? wv : 0) | (wv << 3) | (ho < a.H - 1 ? wv << 6 : 0);
    }

    const int zr = BM + 2 * a.W + 2;             // patch rows the taps reach; row zr holds zeros (see compute)
    const int shift = (a.W + 1) * a.pix_bytes;   // the patch starts W+1 pixels in front of the tile's first pixel
    const v4u rsx = make_srd(a.x - shift, a.x_bytes + (unsigned)shift);
    const v4u rsw = make_srd(a.w, a.w_bytes);
    constexpr unsigned OOB = 0x80000000u;
    const int gch = ld_chunk ^ (ld_row & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int patch_bytes = pra * 128;
    const unsigned w_base = lds_base + npatch * patch_bytes;
    const int npp = (pra + RPP - 1) / RPP;   // DMA pieces per patch (<= 9: one rides along with each tap's weight step)
    unsigned woff[NLD_W];
#pragma unroll
    for (int i = 0; i < NLD_W; ++i) {
        const int n = n0 + ld_row + RPP * i;
        woff[i] = n < a.Cout ? (unsigned)(n * a.w_row_bytes + gch * 16) : OOB;
    }
    auto issue_patch = [&](int piece, int cs, int buf) {
        if (piece * RPP + wave_u * 8 < pra) {   // wave-uniform: the last piece covers only the allocated rows
            const int p = m0 + piece * RPP + ld_row;   // linear pixel index + (W+1)
            // (rows behind the BM + 2W + 2 the taps reach are fetched out of range = written as ZEROS: the first of them is the row invalid taps read)
            const unsigned v[1] = {(p >= a.W + 1 && piece * RPP + ld_row < zr) ? (unsigned)(p * a.pix_bytes + gch * 16) : OOB};
            lds_dma16_group<1, 0>(rsx, v, (unsigned)__builtin_amdgcn_readfirstlane(cs * 128),
                                  lds_base + buf * patch_bytes + piece * (RPP * 128) + wave_u * 1024);
        }
    };
    auto issue_w = [&](int wstep, int stage) {
        if (BN >= RPP || wave_u * 8 < BN)
        lds_dma16_group<NLD_W, RPP * 128>(rsw, woff, (unsigned)__builtin_amdgcn_readfirstlane(wstep * 128), w_base + stage * WSTAGE + wave_u * 1024);
    };
    const int rd_w = (wave_n * WN + r16) * 128;
    // a tap that leaves the image reads a ROW OF ZEROS instead of the patch: one select on the fragment's LDS address per m-tile and tap.  (Zeroing the
    // fragment itself cost 8 v_and per m-tile and K-step -- with the address arithmetic 2 VALU instructions per MFMA, and on this chip VALU and MFMA
    // issue do not overlap within a SIMD: instruction mix 1344 MFMA x 16 cycles + 2661 VALU x 4 cycles = the wave's lifetime x 1 / 4 waves per SIMD.)
    // The rows: patch rows zr = BM + 2W + 2 and zr + 1, the first ones behind those the taps reach (the host allocates at least two), zero-filled by the
    // staging DMA itself.  (With ONE zero row every second invalid lane read from the other half of the banks than its regular read would: 18.6 % of the
    // kernel's LDS cycles were conflicts against 2.2 % with the masked fragments.)
    auto compute = [&](const char* pb, const char* wb, int toff, int tap) {
        const int prow = wave_m * WM + r16 + toff;
        const int tbit = 1 << tap;
        const char* xr[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) xr[j] = pb + ((vm[j] & tbit) ? prow + j * 16 : zr + ((prow ^ zr) & 1)) * 128;   // (the zero row of the lane's own parity:
                                                                                                                        //  rows are 128 B = half the banks, so the read keeps the bank set of the read it replaces)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int swx = ((kk * 4 + q) ^ (prow & 7)) << 4;     // (any swizzle of the zero row reads zeros)
            const int sww = ((kk * 4 + q) ^ (r16 & 7)) << 4;
            uint4 fx[MT], fw[NT];
#pragma unroll
            for (int j = 0; j < MT; ++j) fx[j] = *(const uint4*)(xr[j] + swx);
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *(const uint4*)(wb + rd_w + i * 16 * 128 + sww);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fx[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    };

    const int SPT = a.SPT;
    const int nsteps = 9 * SPT;
    // K-step s = (slice s / 9, tap s % 9); packed weights are tap-major: weight K-step = tap*SPT + slice
    int i_tap = 0, i_cs = 0, i_stage = 0;   // position of the next weight issue
    auto issue_next_w = [&]() {
        issue_w(i_tap * SPT + i_cs, i_stage);
        if (++i_tap == 9) { i_tap = 0; ++i_cs; }
        i_stage = i_stage + 1 == WS ? 0 : i_stage + 1;
    };
    auto wait_allowed = [&](int n) {   // at most n of this wave's DMA instructions stay in flight (they retire in issue order)
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
    };
    static_assert(RESIDENT || (WS - 2) * (NLD_W + 1) <= 6, "wait_allowed covers up to 6 instructions in flight");
    for (int piece = 0; piece < npp; ++piece) issue_patch(piece, 0, 0);
#pragma unroll
    for (int s0 = 0; s0 < (RESIDENT ? 9 : WS - 1); ++s0)
        if (s0 < nsteps) issue_next_w();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int step = 0, cur = 0;
    int c_prev = 0;   // DMA instructions this wave issued in the previous iteration (WS == 4 keeps two iterations in flight)
#pragma unroll 1
    for (int cs = 0; cs < SPT; ++cs) {
        const char* pb = smem + (cs & (npatch - 1)) * patch_bytes;
        int kh = 0, kw = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++step) {
            // weights of K-step step+WS-1, plus one piece of the next slice's patch (taps 0..npp-1; npp <= 11-WS, so the last piece has
            // left the in-flight window by the end of tap 8)
            int c_now = 0;
            if constexpr (!RESIDENT) {
                if (step + WS - 1 < nsteps) { issue_next_w(); c_now = NLD_W; }
                if (cs + 1 < SPT && tap < npp && tap * RPP + wave_u * 8 < pra) { issue_patch(tap, cs + 1, (cs + 1) & 1); ++c_now; }
            }
            compute(pb, smem + npatch * patch_bytes + cur * WSTAGE, kh * a.W + kw, tap);
            if (++kw == 3) { kw = 0; ++kh; }
            cur = cur + 1 == WS ? 0 : cur + 1;
            if constexpr (!RESIDENT) {
                c_now = __builtin_amdgcn_readfirstlane(c_now);
                wait_allowed(WS == 2 ? 0 : WS == 3 ? c_now : c_now + c_prev);
                c_prev = c_now;
                __syncthreads();
            }
        }
    }
    if constexpr (RESIDENT) __syncthreads();   // (the epilogue reuses the operand LDS)

    // ---- epilogue: (bias in the accumulators) ReLU, fp32 tile -> LDS rows -> 16-byte bf16 vectors
    constexpr int ROWB = BN * 4 + 16;
    constexpr int PASSES = BN >= 256 ? 4 : 2;
    static_assert(MT % PASSES == 0, "passes split the m-tiles of a wave");
    constexpr int MTP = MT / PASSES, WMP = WM / PASSES, BMP = BM / PASSES;
    constexpr int TPR = BN / 8, RPI = NTH / TPR;
    constexpr int ITERS = (BMP + RPI - 1) / RPI;
    const int rc = tid % TPR, rr = tid / TPR;
    const int n = n0 + rc * 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (EXPAND) {
        // t2 tile as the pixel operand: two planes (64 channels = 128 B per pixel each), 16-byte chunks XOR-swizzled with row & 7
        char* const t2p = smem;
        char* const stg = smem + 2 * BM * 128;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = wave_m * WM + j * 16 + r16;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int ch = wave_n * WN + i * 16 + q * 4;
                const int cb = (ch & 63) * 2;
                f32x4 v = acc[i][j];
                if (a.relu == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *(uint2*)(t2p + (ch >> 6) * (BM * 128) + row * 128 + (((cb >> 4) ^ (row & 7)) << 4) + (cb & 8)) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        __syncthreads();
        // per 128-channel output chunk: two staging passes of BM / 2 pixels.  The residual rows of the NEXT pass are requested before this pass's
        // LDS hand-off (issued just in time, every pass waited out an HBM round trip -- 8 per tile), the weight fragments of the next chunk right
        // behind this chunk's MFMAs; the pixel fragments are re-read from LDS per K step (kept in registers they would leave no room for either)
        const int nchunk = a.f_cout >> 7;
        auto load_res = [&](int c, int p, u32x4 (&r)[ITERS]) {
            const int n2 = c * 128 + rc * 8;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lrow = rr + it * RPI;
                const int m = m0 + (lrow / WMP) * WM + p * WMP + (lrow % WMP);
                r[it] = (u32x4){0u, 0u, 0u, 0u};
                if (a.res && lrow < BMP && m < a.M) {
                    const u32x4* rp = (const u32x4*)(a.res + ((long long)m * a.f_cout + n2) * 2);
                    r[it] = a.nt_epi ? __builtin_nontemporal_load(rp) : *rp;
                }
            }
        };
        uint4 fa[NT][4];      // fragment-ordered weights: [channel tile][K step of 32][lane] x 16 B
        auto load_w = [&](int c) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[i][ks] = *(const uint4*)(a.f_w + ((long long)(((c * 8 + wave_n * NT + i) * 4 + ks) * 64 + lane)) * 16);
        };
        static_assert(PASSES == 2, "the residual prefetch alternates between two register sets");
        u32x4 r0[ITERS], r1[ITERS];      // residual rows of pass 0 / pass 1
        load_res(0, 0, r0);
        load_w(0);
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            f32x4 acc2[NT][MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float4 t = *(const float4*)(a.f_bias + c * 128 + wave_n * WN + i * 16 + q * 4);
#pragma unroll
                for (int j = 0; j < MT; ++j) acc2[i][j] = (f32x4){t.x, t.y, t.z, t.w};
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 fb[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int prow = wave_m * WM + j * 16 + r16;
                    fb[j] = *(const uint4*)(t2p + (ks >> 1) * (BM * 128) + prow * 128 + ((((ks & 1) * 4 + q) ^ (prow & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i][ks]), __builtin_bit_cast(bf16x8_t, fb[j]),
                                                                             acc2[i][j], 0, 0, 0);
            }
            if (c + 1 < nchunk) load_w(c + 1);
            const int n2 = c * 128 + rc * 8;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                if (p == 0) load_res(c, 1, r1);
                else if (c + 1 < nchunk) load_res(c + 1, 0, r0);
                if (c || p) __syncthreads();          // the previous pass has been read back
#pragma unroll
                for (int jj = 0; jj < MTP; ++jj) {
                    const int lrow = wave_m * WMP + jj * 16 + r16;
#pragma unroll
                    for (int i = 0; i < NT; ++i) *(f32x4*)(stg + lrow * ROWB + (wave_n * WN + i * 16 + q * 4) * 4) = acc2[i][p * MTP + jj];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lrow = rr + it * RPI;
                    const int m = m0 + (lrow / WMP) * WM + p * WMP + (lrow % WMP);
                    if (lrow >= BMP || m >= a.M) continue;
                    float v[8];
                    *(float4*)&v[0] = *(const float4*)(stg + lrow * ROWB + rc * 32);
                    *(float4*)&v[4] = *(const float4*)(stg + lrow * ROWB + rc * 32 + 16);
                    if (a.res) {
                        const u32x4 rv = p ? r1[it] : r0[it];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(rv[e] << 16);
                            v[2 * e + 1] += __uint_as_float(rv[e] & 0xffff0000u);
                        }
                    }
                    if (a.f_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                    char* yp = a.f_y + ((long long)m * a.f_cout + n2) * 2;
                    if (a.nt_epi) __builtin_nontemporal_store(ov, (u32x4*)yp);
                    else *(uint4*)yp = make_uint4(ov.x, ov.y, ov.z, ov.w);
                }
            }
        }
        return;
    }
    double cs[STATS ? 8 : 1], cq[STATS ? 8 : 1];
#pragma unroll
    for (int e = 0; e < (STATS ? 8 : 1); ++e) cs[e] = cq[e] = 0.0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        // residual rows of the pass (the second conv of a BasicBlock) go in flight before the LDS hand-off, like in the generic kernel
        u32x4 rres[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lrow = rr + it * RPI;
            const int m = m0 + (lrow / WMP) * WM + p * WMP + (lrow % WMP);
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (a.res && lrow < BMP && m < a.M && n < a.Cout) {
                const u32x4* rp = (const u32x4*)(a.res + ((long long)m * a.res_ld + n) * 2);
                rres[it] = a.nt_epi ? __builtin_nontemporal_load(rp) : *rp;
            }
        }
        if (p) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < MTP; ++jj) {
            const int lrow = wave_m * WMP + jj * 16 + r16;
#pragma unroll
            for (int i = 0; i < NT; ++i) *(f32x4*)(smem + lrow * ROWB + (wave_n * WN + i * 16 + q * 4) * 4) = acc[i][p * MTP + jj];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lrow = rr + it * RPI;
            const int m = m0 + (lrow / WMP) * WM + p * WMP + (lrow % WMP);
            if (lrow >= BMP || m >= a.M || n >= a.Cout) continue;
            float v[8];
            *(float4*)&v[0] = *(const float4*)(smem + lrow * ROWB + rc * 32);
            *(float4*)&v[4] = *(const float4*)(smem + lrow * ROWB + rc * 32 + 16);
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(rres[it][e] << 16);
                    v[2 * e + 1] += __uint_as_float(rres[it][e] & 0xffff0000u);
                }
            }
            if (a.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            char* yp = a.y + ((long long)m * a.y_ld + n) * 2;
            if (a.nt_epi) __builtin_nontemporal_store(ov, (u32x4*)yp);
            else *(uint4*)yp = make_uint4(ov.x, ov.y, ov.z, ov.w);
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double lo = (double)__uint_as_float(ov[e] << 16), hi = (double)__uint_as_float(ov[e] & 0xffff0000u);
                    cs[2 * e] += lo; cq[2 * e] = fma(lo, lo, cq[2 * e]); cs[2 * e + 1] += hi; cq[2 * e + 1] = fma(hi, hi, cq[2 * e + 1]);
                }
            }
        }
    }
    if constexpr (STATS) colstat_flush<TPR, NTH, 8, BN>(cs, cq, smem, a.stat_sums + (long long)(tile_m & (MT4_STAT_REPLICAS - 1)) * 2 * a.Cout, n0, a.Cout, tid);
}

// ------------------------------------------------------------------------------------------------ stem patch kernel
// The ResNet stem on the space-to-depth frame (x [B][H][W][16] bf16, kernel KH x 4 pixels, stride 1, valid; x_pixel_stride = 16): through
// the generic kernel a K-step stages 256 overlapping 128-byte runs (pixel p .. p+3) -- 4x the unique bytes -- and the launch runs at the
// L2 -> LDS fill rate.  Here the tile's span of the frame is copied to LDS ONCE as a linear range of 32-byte pixels (p0 = frame index of
// the tile's first output pixel), together with the whole [64][KH*64] weight matrix; every lane keeps the frame index of its output pixel
// relative to p0 and reads its fragments at (base + kh*W + 2*kk + q/2): no barrier inside the K loop.  The pixels lie in LDS as they lie in
// memory, NO swizzle: a `ds_read_b128` is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS
// table), i.e. 8 lanes of one q with the 8 complementary r16 of the next q -- 16 distinct pixels whose pairs (p, p + 8) read opposite 16-byte
// halves: all 64 banks, no conflict.  (Rounds 1-3 swapped the halves on odd groups of 8 pixels, which is right for groups of 16 CONSECUTIVE lanes and
// made every fragment read 2-way conflicted under the real grouping: SQ_LDS_BANK_CONFLICT = 46 % of the LDS cycles of stem_pool_kernel,
// profiles/r04_step_kernel_counters_before.txt.)
// K order = kernel row, then the 128-byte run: the generic kernel's order and MFMA chain -> bit-identical results.
template <int BM, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(6, 6))) void stem_patch_kernel(const ConvK a, const int pra) {
    constexpr int BN = 64;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NTH = NW * 64;
    constexpr int PPP = NTH * 16 / 32;   // pixels one DMA pass of the workgroup covers
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    static_assert(NW == 8 && WM % 16 == 0 && WN % 16 == 0 && MT <= 4, "tile shape (8 waves stage the 64 weight rows in one pass)");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int r16 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7;
        const int xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    // tiles never cross an image (a.n_tiles = tiles per image): the patch then needs no allowance for the KH-1 frame rows between images
    const int img = bid / a.n_tiles;
    const int m0 = img * a.HoWo + (bid - img * a.n_tiles) * BM;
    const int m_end = (img + 1) * a.HoWo;   // first output pixel of the next image

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = wave_n * WN + i * 16 + q * 4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias && n < a.Cout) { const float4 t = *(const float4*)(a.bias + n); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = b4;
    }

    auto frame_index = [&](int m) {   // output pixel -> index of its top-left input pixel in the [B][H][W] frame
        const int b = m / a.HoWo;
        const int rem = m - b * a.HoWo;
        const int ho = rem / a.Wo;
        return (b * a.H + ho) * a.W + (rem - ho * a.Wo);
    };
    const int p0 = __builtin_amdgcn_readfirstlane(frame_index(m0));
    int base[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + wave_m * WM + j * 16 + r16;
        base[j] = frame_index(m < m_end ? m : m_end - 1) - p0 + (q >> 1);
    }

    const v4u rsx = make_srd(a.x, a.x_bytes);
    const v4u rsw = make_srd(a.w, a.w_bytes);
    constexpr unsigned OOB = 0x80000000u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int patch_bytes = pra * 32;
    {   // patch: LDS chunk (pixel t, stored half hs) <- frame pixel p0 + t, half hs ^ (t>>3 & 1)
        const int npp = (pra + PPP - 1) / PPP;
        for (int piece = 0; piece < npp; ++piece) {
            if (piece * PPP + wave_u * 32 < pra) {
                const int t = piece * PPP + (tid >> 1);
                const int h = tid & 1;     // (no swizzle: see the fragment read)
                const unsigned v[1] = {(unsigned)((p0 + t) * 32 + h * 16)};
                lds_dma16_group<1, 0>(rsx, v, 0u, lds_base + piece * (PPP * 32) + wave_u * 1024);
            }
        }
        // weights: KH K-steps of [64 rows][128 B], chunk XOR-swizzled with (row & 7) like the generic operand stages
        const int ld_row = tid >> 3, ld_chunk = tid & 7;
        const int gch = ld_chunk ^ (ld_row & 7);
        const unsigned woff[1] = {ld_row < a.Cout ? (unsigned)(ld_row * a.w_row_bytes + gch * 16) : OOB};
        for (int kh = 0; kh < a.KH; ++kh)
            lds_dma16_group<1, 0>(rsw, woff, (unsigned)__builtin_amdgcn_readfirstlane(kh * 128), lds_base + patch_bytes + kh * (BN * 128) + wave_u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int rd_w = (wave_n * WN + r16) * 128;
#pragma unroll 1
    for (int kh = 0; kh < a.KH; ++kh) {
        const char* wb = smem + patch_bytes + kh * (BN * 128);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sww = ((kk * 4 + q) ^ (r16 & 7)) << 4;
            uint4 fx[MT], fw[NT];
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int pix = base[j] + kh * a.W + kk * 2;
                fx[j] = *(const uint4*)(smem + pix * 32 + ((q & 1) << 4));
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *(const uint4*)(wb + rd_w + i * 16 * 128 + sww);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fx[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // the epilogue reuses the operand LDS

    constexpr int ROWB = BN * 4 + 16;
    constexpr int PASSES = 2;
    static_assert(MT % PASSES == 0, "passes split the m-tiles of a wave");
    constexpr int MTP = MT / PASSES, WMP = WM / PASSES, BMP = BM / PASSES;
    constexpr int TPR = BN / 8, RPI = NTH / TPR;
    constexpr int ITERS = (BMP + RPI - 1) / RPI;
    const int rc = tid % TPR, rr = tid / TPR;
    const int n = rc * 8;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (p) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < MTP; ++jj) {
            const int lrow = wave_m * WMP + jj * 16 + r16;
#pragma unroll
            for (int i = 0; i < NT; ++i) *(f32x4*)(smem + lrow * ROWB + (wave_n * WN + i * 16 + q * 4) * 4) = acc[i][p * MTP + jj];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lrow = rr + it * RPI;
            const int m = m0 + (lrow / WMP) * WM + p * WMP + (lrow % WMP);
            if (lrow >= BMP || m >= m_end || n >= a.Cout) continue;
            float v[8];
            *(float4*)&v[0] = *(const float4*)(smem + lrow * ROWB + rc * 32);
            *(float4*)&v[4] = *(const float4*)(smem + lrow * ROWB + rc * 32 + 16);
            if (a.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ov = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            char* yp = a.y + ((long long)m * a.y_ld + n) * 2;
            if (a.nt_epi) __builtin_nontemporal_store(ov, (u32x4*)yp);
            else *(uint4*)yp = make_uint4(ov.x, ov.y, ov.z, ov.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------ stem + max-pool in one launch
// relu(stem conv) -> MaxPool2d(3, 2, 1) (`Spatial_transformer/models/resnet.py:145-157`, conv1 / bn1 / relu / maxpool) on the space-to-depth
// frame without the 112 x 112 x 64 map ever reaching memory (2.1 GB written and read back per 1336 frames at 224 x 224).  A workgroup owns
// two pooled rows of one frame: conv rows 2 P0 - 1 .. 2 P0 + 3 (five; the first is shared with the tile above and computed by both),
// whose eight space-to-depth rows are staged once together with the weight matrix exactly as in stem_patch_kernel; the five conv rows
// (ReLU, rounded to bf16: what the stand-alone launch stores) meet in LDS over the dead operands and every thread then takes the maximum of
// up to nine 16-byte channel vectors per pooled pixel (values >= 0: an unsigned 16-bit maximum IS the bf16 maximum).  K order and MFMA
// chain are stem_patch_kernel's: bit-identical to stem -> mt4_maxpool3x3s2_nhwc.
struct StemPoolK {
    const char* x;          // [B][Hs][Ws][16] bf16
    const char* w;          // packed [Cout <= 64][KH * 64] bf16
    const float* bias;
    char* y;                // [B][Hp][Wp][64] bf16
    unsigned x_bytes, w_bytes;
    int w_row_bytes, Cout, KH;
    int Hs, Ws, Ho, Wo, Hp, Wp;
    int tiles_per_img;      // pairs of pooled rows x column segments
    int pra;                // staged pixels (multiple of 32)
    // column segments (frames wider than 224 pixels: conv rows of more than 112 pixels): a workgroup computes `cw` conv columns from
    // cs0 = max(2 Q0 - 1, 0) for the wp_seg pooled columns from Q0 = seg * wp_seg; its span is staged as 8 rows of `rs` pixels.
    // One segment: cw = Wo, rs = Ws (the span is one linear range of the frame)
    int segs, cw, rs, wp_seg;
    int ctw;                // conv columns of a segment kept for the pooling (row pitch of the conv tile in LDS): min(cw, 2 wp_seg + 8)
};

template <int MT, int NPASS>
__global__ __launch_bounds__(512, 4) void stem_pool_kernel(const StemPoolK a) {
    constexpr int NT = 2, BN = 64, NTH = 512, PPP = NTH * 16 / 32;
    constexpr int MTP = (MT + NPASS - 1) / NPASS;      // pixel tiles per pass over K.  Two passes (the finished half waits as packed bf16): with all
                                                       // 72 / 80 accumulator registers live the fragments do not fit the 128-register budget of
                                                       // 4 waves per SIMD -- 24 spilled registers per lane and K-step cost the wide form 40 % (1.78 ->
                                                       // 1.26 ms per 584 frames of 256 x 448), far more than a second walk over the LDS operands
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int q8 = nb >> 3, r8 = nb & 7;
        const int xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    }
    const int img = bid / a.tiles_per_img;
    const int trem = bid - img * a.tiles_per_img;
    const int rp = trem / a.segs;
    const int Q0 = (trem - rp * a.segs) * a.wp_seg;     // first pooled column of the segment
    const int cs0 = max(2 * Q0 - 1, 0);                 // first conv column computed
    const int P0 = rp * 2;
    const int c0 = 2 * P0 - 1;                     // first conv row of the tile (-1 for the top tile: computed from zeros / the frame above, never pooled)
    const int tpr = a.cw >> 4;                     // 16-pixel tiles per conv row (segment)
    const int ntile = 5 * tpr;

    f32x4 bias4v[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = wave_n * 32 + i * 16 + q * 4;
        bias4v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias && n < a.Cout) { const float4 t = *(const float4*)(a.bias + n); bias4v[i] = (f32x4){t.x, t.y, t.z, t.w}; }
    }
    const int p0 = __builtin_amdgcn_readfirstlane((img * a.Hs + c0) * a.Ws + cs0);   // frame pixel of LDS pixel 0
    int base[MT], trow[MT], tcol0[MT];             // wave-uniform: LDS pixel of tile j (+ lane_px per lane), its conv row and first column
    const int lane_px = r16 + (q >> 1);
    const int wave_m_u = __builtin_amdgcn_readfirstlane(wave_m);
    const unsigned tpr_magic = 65536u / (unsigned)tpr + 1u;       // t / tpr == (t * magic) >> 16 for t < 5 tpr <= 320, tpr <= 64 (scalar unit: no division there)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int t = min(wave_m_u + 4 * j, ntile - 1);
        const int row = (int)(((unsigned)t * tpr_magic) >> 16);
        trow[j] = row;
        tcol0[j] = (t - row * tpr) * 16;
        base[j] = row * a.rs + tcol0[j];
    }
    const v4u rsx = make_srd(a.x, a.x_bytes);
    const v4u rsw = make_srd(a.w, a.w_bytes);
    constexpr unsigned OOB = 0x80000000u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int patch_bytes = a.pra * 32;
    {
        const int npp = (a.pra + PPP - 1) / PPP;
        for (int piece = 0; piece < npp; ++piece) {
            if (piece * PPP + wave_u * 32 < a.pra) {
                const int t = piece * PPP + (tid >> 1);
                const int h = tid & 1;     // (no swizzle: see the fragment read)
                // (pixels in front of the buffer -- the top tile of frame 0 -- wrap to offsets beyond it: range-checked, zeros)
                unsigned v[1] = {(unsigned)((p0 + t) * 32 + h * 16)};
                if (a.segs > 1) {                  // LDS pixel t = (span row t / rs, column t % rs) of the segment
                    const int row = t / a.rs;
                    const int col = t - row * a.rs;
                    v[0] = cs0 + col < a.Ws ? (unsigned)((p0 + row * a.Ws + col) * 32 + h * 16) : OOB;
                }
                lds_dma16_group<1, 0>(rsx, v, 0u, lds_base + piece * (PPP * 32) + wave_u * 1024);
            }
        }
        const int ld_row = tid >> 3, ld_chunk = tid & 7;
        const int gch = ld_chunk ^ (ld_row & 7);
        const unsigned woff[1] = {ld_row < a.Cout ? (unsigned)(ld_row * a.w_row_bytes + gch * 16) : OOB};
        for (int kh = 0; kh < a.KH; ++kh)
            lds_dma16_group<1, 0>(rsw, woff, (unsigned)__builtin_amdgcn_readfirstlane(kh * 128), lds_base + patch_bytes + kh * (BN * 128) + wave_u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int rd_w = (wave_n * 32 + r16) * 128;
    uint2 res[NT][MT];                              // the conv rows: ReLU, bf16 (what the stand-alone launch stores)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        f32x4 acc[NT][MTP];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int jj = 0; jj < MTP; ++jj) acc[i][jj] = bias4v[i];
#pragma unroll 1
        for (int kh = 0; kh < a.KH; ++kh) {
            const char* wb = smem + patch_bytes + kh * (BN * 128);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sww = ((kk * 4 + q) ^ (r16 & 7)) << 4;
                uint4 fw[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) fw[i] = *(const uint4*)(wb + rd_w + i * 16 * 128 + sww);
#pragma unroll
                for (int j0 = 0; j0 < MTP; j0 += 5) {     // batches of pixel fragments (all of them in flight: over the register budget)
                    uint4 fx[5];
#pragma unroll
                    for (int jj = j0; jj < MTP && jj < j0 + 5; ++jj) {
                        const int j = ps * MTP + jj < MT ? ps * MTP + jj : MT - 1;
                        const int pix = lane_px + (base[j] + kh * a.rs + kk * 2);
                        fx[jj - j0] = *(const uint4*)(smem + pix * 32 + ((q & 1) << 4));
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int jj = j0; jj < MTP && jj < j0 + 5; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fx[jj - j0]),
                                                                                 acc[i][jj], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int jj = 0; jj < MTP; ++jj)
                if (ps * MTP + jj < MT) {      // ReLU on the packed pair: rounding commutes with it, one integer max with 0 per dword (-0.0 -> +0)
                    uint2 o = make_uint2(pack_bf16x2(acc[i][jj][0], acc[i][jj][1]), pack_bf16x2(acc[i][jj][2], acc[i][jj][3]));
                    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.x));
                    asm("v_pk_max_i16 %0, %0, 0" : "+v"(o.y));
                    res[i][ps * MTP + jj] = o;
                }
    }
    __syncthreads();   // the conv rows go over the operands

    // conv tile: pixel (row, column - cs0) at (row * ctw + col) * 128 B, 16-byte chunks XOR-swizzled with (col >> 1) & 7 -- a 128-byte row is half the
    // banks, so columns of one parity share a bank half: the pooling window reads columns 2 apart (one parity per read), and keyed with col & 7
    // those took 4 of the 8 chunk positions (2-way conflicts: 24 % of the kernel's LDS cycles) (columns the pooling does not
    // read are dropped: the wide form stays under 80 KB = two workgroups per CU)
    int wr_lane[NT];      // lane part of the address (a tile's first column is a multiple of 16: (col & 7) == (r16 & 7))
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int cb = (wave_n * 32 + i * 16 + q * 4) * 2;
        wr_lane[i] = r16 * 128 + (((cb >> 4) ^ ((r16 >> 1) & 7)) << 4) + (cb & 8);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        if (wave_m_u + 4 * j < ntile) {
            const int col0 = tcol0[j];
            const int wr_tile = (trow[j] * a.ctw + col0) * 128;             // wave-uniform
            if (col0 + r16 < a.ctw) {
#pragma unroll
                for (int i = 0; i < NT; ++i) *(uint2*)(smem + wr_tile + wr_lane[i]) = res[i][j];
            }
        }
    }
    __syncthreads();

    const int nvec = 2 * a.wp_seg * 8;
    for (int v = tid; v < nvec; v += NTH) {
        const int c8 = v & 7;
        const int pp = v >> 3;
        const int pr = pp >= a.wp_seg ? 1 : 0;
        const int pc = Q0 + pp - pr * a.wp_seg;
        const int P = P0 + pr;
        if (P >= a.Hp || pc >= a.Wp || c8 * 8 >= a.Cout) continue;
        // the 3 x 3 window without branches: a tap outside the map is clamped onto its neighbour inside (the maximum does not mind seeing a value
        // twice; the window's centre is always inside), so every thread runs the same nine 16-byte reads
        int roff[3], coff[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int jr = min(max(2 * pr + k, -c0), a.Ho - 1 - c0);          // conv row c0 + jr in [0, Ho)
            roff[k] = jr * a.ctw * 128;
            const int cr = min(max(2 * pc - 1 + k, 0), a.Wo - 1) - cs0;       // column inside the segment
            coff[k] = cr * 128 + ((c8 ^ ((cr >> 1) & 7)) << 4);
        }
        uint4 best = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const uint4 t = *(const uint4*)(smem + roff[kh] + coff[kw]);
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(best.x) : "v"(t.x));
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(best.y) : "v"(t.y));
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(best.z) : "v"(t.z));
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(best.w) : "v"(t.w));
            }
        *(uint4*)(a.y + (((long long)img * a.Hp + P) * a.Wp + pc) * (a.Cout * 2) + c8 * 16) = best;
    }
}

// ------------------------------------------------------------------------------------------------ host side
namespace {

struct TileCfg {
    int bm, bn;
};
// tile ids are 1-based in the C-ABI
// ids 1-6: two operand stages; 7-12: deeper rings (3 for the 128-wide tiles, 4 for the small ones)
// ids 13-16: 8-wave workgroups (one per CU): 256x128 with 2 / 3 stages, 256x256, 128x256 with 3 stages
// id 20: 256x64 for the 64-channel layers (8 waves)
// ids 17-19: 16-wave workgroups (four waves per SIMD, 64x64 / 64x32 wave tiles): 256x256, 256x128 with 2 / 3 stages
// ids 21-32: the 3x3 patch kernel (bf16, stride 1, pad 1, Cin % 64 == 0), BM x BN / waves / weight stages:
//   21: 256x64/8/4   22: 256x128/16/3   23: 256x256/16/2   24: 256x64/8/2   25: 256x128/16/4   26: 256x128/16/2
//   27: 256x64/8/all 9 taps resident   28: 512x64/16/resident   29: 128x64/4/2   30: 128x128/4/2   31: 256x64/4/2   32: 256x128/8/2
//   (auto: 24, 30, 23 by Cout; the others are the tuning record: deeper rings and the barrier-free resident variants are slower)
constexpr TileCfg kTiles[] = {
    {128, 128}, {128, 64}, {64, 64}, {64, 128}, {32, 64}, {32, 32},                                  // 1-6
    {128, 128}, {64, 128}, {64, 64}, {32, 64}, {32, 32}, {128, 64},                                  // 7-12
    {256, 128}, {256, 128}, {256, 256}, {128, 256}, {256, 256}, {256, 128}, {256, 128}, {256, 64},   // 13-20
    {256, 64}, {256, 128}, {256, 256}, {256, 64}, {256, 128}, {256, 128},                            // 21-26: conv3x3_patch_kernel
    {256, 64}, {512, 64}, {128, 64}, {128, 128}, {256, 64}, {256, 128},                              // 27-32: conv3x3_patch_kernel
    {256, 64}, {256, 64},                                                                            // 33 / 34: stem patch kernel, persistent form
    {32, 32}, {32, 64}, {32, 32}, {32, 32},                                                          // 35-38: generic kernel, K-split groups (4, 4, 8, 4 with 3 stages)
    {64, 64}, {64, 64}, {64, 64}, {64, 128}};                                                        // 39-42: K-split groups on the 64-row tiles (2, 4, 2 with 3 stages, 2): whole-video TCN layers
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <typename T, int BM, int BN, int WM_, int WN_, int STAGES, bool OUT_F32, int KS = 1>
int launch_tile(const ConvK& k, bool fast, hipStream_t s) {
    const int m_tiles = cdiv(k.M, BM);
    ConvK kk = k;
    constexpr bool CAN_STATS = KS == 1 && WM_ * WN_ * BN * 16 <= (BM + BN) * 128;   // the staged epilogue's LDS holds the waves' partial rows
    if (k.stat_sums && (!CAN_STATS || (k.Cout % (OUT_F32 ? 4 : 8)) != 0)) return MT4_EUNSUPPORTED;
    kk.n_tiles = cdiv(k.Cout, BN);
    kk.total_tiles = m_tiles * kk.n_tiles;
    kk.nt_epi = 1;
    {   // outputs that fit the 256 MB Infinity Cache stay cacheable for the next layer (+0.8 % over always-nt; MT4_NT_MIN_MB overrides)
        const long long min_mb = MT4_NT_MIN_MB;
        if ((long long)k.M * k.Cout * 2 < min_mb * 1000000LL) kk.nt_epi = 0;
    }
    // LDS: two operand stages, or ONE when the whole K fits a single step (then only the epilogue staging may need
    // more than a stage): the short-K layers are memory-bound and want as many workgroups per CU as possible
    constexpr int stage = (BM + BN) * 128;
    constexpr int os = OUT_F32 ? 4 : 2;
    constexpr int rowb = BN * 4 + 16;
    constexpr int passes = (BM * rowb > 4 * stage) ? 4 : (BM * rowb > 2 * stage) ? 2 : 1;
    constexpr int epi = BM / passes * rowb;
    constexpr int threads = WM_ * WN_ * 64;
    (void)os;
    if constexpr (KS > 1) {   // K-split groups (LDS-DMA path only): every group has its own ring of stages
        if (!fast) return MT4_EUNSUPPORTED;
        constexpr int lds_ks = KS * STAGES * stage > epi ? KS * STAGES * stage : epi;
        auto fn = igemm_conv_kernel<T, BM, BN, WM_, WN_, STAGES, true, OUT_F32, KS>;
        if (lds_ks > 65536) {
            MT4_RAISE_LDS(fn);
        }
        hipLaunchKernelGGL(fn, dim3(kk.total_tiles), dim3(threads * KS), lds_ks, s, kk);
        return mt4_check_launch();
    } else {
    const int lds = k.nsteps > 1 ? (fast ? STAGES : 2) * stage : (stage > epi ? stage : epi);
    const int grid = kk.total_tiles;  // one tile per workgroup (see PERSIST in the kernel)
    if constexpr (CAN_STATS) {
        if (k.stat_sums) {
            if (fast) {
                auto fn = igemm_conv_kernel<T, BM, BN, WM_, WN_, STAGES, true, OUT_F32, 1, false, true>;
                if (lds > 65536) {
                    MT4_RAISE_LDS(fn);
                }
                hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, kk);
            } else {
                auto fn = igemm_conv_kernel<T, BM, BN, WM_, WN_, 2, false, OUT_F32, 1, false, true>;
                if (lds > 65536) {
                    MT4_RAISE_LDS(fn);
                }
                hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, kk);
            }
            return mt4_check_launch();
        }
    }
    if (fast) {
        auto fn = igemm_conv_kernel<T, BM, BN, WM_, WN_, STAGES, true, OUT_F32>;
        if (lds > 65536) {
            MT4_RAISE_LDS(fn);
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, kk);
    } else {
        auto fn = igemm_conv_kernel<T, BM, BN, WM_, WN_, 2, false, OUT_F32>;
        if (lds > 65536) {
            MT4_RAISE_LDS(fn);
        }
        hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, kk);
    }
    return mt4_check_launch();
    }
}

// tile 17 (bf16 256 x 256, 16 waves) with the second K source (ConvK::x2)
int launch_dual(const ConvK& k, hipStream_t s) {
    constexpr int BM = 256, BN = 256, stage = (BM + BN) * 128;
    ConvK kk = k;
    kk.n_tiles = cdiv(k.Cout, BN);
    kk.total_tiles = cdiv(k.M, BM) * kk.n_tiles;
    kk.nt_epi = 1;
    if ((long long)k.M * k.Cout * 2 < (long long)MT4_NT_MIN_MB * 1000000LL) kk.nt_epi = 0;
    auto fn = igemm_conv_kernel<u16, BM, BN, 4, 4, 2, true, false, 1, true>;
    MT4_RAISE_LDS(fn);
    hipLaunchKernelGGL(fn, dim3(kk.total_tiles), dim3(1024), 2 * stage, s, kk);
    return mt4_check_launch();
}

template <typename T, bool OUT_F32>
int launch_dtype(const ConvK& k, int tile, bool fast, hipStream_t s) {
    switch (tile) {
        case 1: return launch_tile<T, 128, 128, 2, 2, 2, OUT_F32>(k, fast, s);
        case 2: return launch_tile<T, 128, 64, 2, 2, 2, OUT_F32>(k, fast, s);
        case 3: return launch_tile<T, 64, 64, 2, 2, 2, OUT_F32>(k, fast, s);
        case 4: return launch_tile<T, 64, 128, 2, 2, 2, OUT_F32>(k, fast, s);
        case 5: return launch_tile<T, 32, 64, 1, 4, 2, OUT_F32>(k, fast, s);
        case 6: return launch_tile<T, 32, 32, 2, 2, 2, OUT_F32>(k, fast, s);
        case 7: return launch_tile<T, 128, 128, 2, 2, 3, OUT_F32>(k, fast, s);
        case 8: return launch_tile<T, 64, 128, 2, 2, 3, OUT_F32>(k, fast, s);
        case 9: return launch_tile<T, 64, 64, 2, 2, 4, OUT_F32>(k, fast, s);
        case 10: return launch_tile<T, 32, 64, 1, 4, 4, OUT_F32>(k, fast, s);
        case 11: return launch_tile<T, 32, 32, 2, 2, 4, OUT_F32>(k, fast, s);
        case 12: return launch_tile<T, 128, 64, 2, 2, 3, OUT_F32>(k, fast, s);
        case 13: return launch_tile<T, 256, 128, 4, 2, 2, OUT_F32>(k, fast, s);
        case 14: return launch_tile<T, 256, 128, 4, 2, 3, OUT_F32>(k, fast, s);
        case 15: return launch_tile<T, 256, 256, 2, 4, 2, OUT_F32>(k, fast, s);
        case 16: return launch_tile<T, 128, 256, 2, 4, 3, OUT_F32>(k, fast, s);
        case 17: return launch_tile<T, 256, 256, 4, 4, 2, OUT_F32>(k, fast, s);
        case 18: return launch_tile<T, 256, 128, 4, 4, 2, OUT_F32>(k, fast, s);
        case 19: return launch_tile<T, 256, 128, 4, 4, 3, OUT_F32>(k, fast, s);
        case 20: return launch_tile<T, 256, 64, 4, 2, 2, OUT_F32>(k, fast, s);
        case 35: return launch_tile<T, 32, 32, 2, 2, 2, OUT_F32, 4>(k, fast, s);   // K-split x4 (16 waves)
        case 36: return launch_tile<T, 32, 64, 1, 4, 2, OUT_F32, 4>(k, fast, s);
        case 37: return launch_tile<T, 32, 32, 1, 2, 2, OUT_F32, 8>(k, fast, s);   // 8 groups of 2 waves
        case 38: return launch_tile<T, 32, 32, 2, 2, 3, OUT_F32, 4>(k, fast, s);   // 4 groups, 3 stages each
        case 39: return launch_tile<T, 64, 64, 2, 2, 2, OUT_F32, 2>(k, fast, s);   // 2 groups of 4 waves
        case 40: return launch_tile<T, 64, 64, 2, 2, 2, OUT_F32, 4>(k, fast, s);   // 4 groups of 4 waves
        case 41: return launch_tile<T, 64, 64, 2, 2, 3, OUT_F32, 2>(k, fast, s);   // 2 groups, 3 stages each
        case 42: return launch_tile<T, 64, 128, 2, 2, 2, OUT_F32, 2>(k, fast, s);  // 2 groups of 4 waves
    }
    return MT4_EINVAL;
}


// geometry the patch kernel covers (everything else runs the generic kernel)
bool patch3x3_ok(const mt4_conv_desc* d, const ConvK& k, bool fast) {
    return fast && d->dtype == MT4_BF16 && d->out_dtype == MT4_BF16 && d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 &&
           d->dil_h == 1 && d->dil_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->Ho == d->H && d->Wo == d->W && k.pix_bytes == d->Cin * 2 &&
           !d->out_row_map && d->relu <= 1 && (d->Cout % 8) == 0 && k.nsteps == 9 * k.SPT && (!d->residual || (k.res_ld * 2) % 16 == 0) &&
           k.x_total_bytes + (long long)(2 * d->W + 1024) * k.pix_bytes < 0x7fffffffLL;
}

template <int BM, int BN, int WM_, int WN_, int WS, bool EXPAND = false>
int launch_patch3x3(const ConvK& k, hipStream_t s) {
    ConvK kk = k;
    kk.n_tiles = cdiv(k.Cout, BN);
    kk.total_tiles = cdiv(k.M, BM) * kk.n_tiles;
    kk.nt_epi = 1;
    {
        const long long min_mb = MT4_NT_MIN_MB;
        if ((long long)k.M * (EXPAND ? k.f_cout : k.Cout) * 2 < min_mb * 1000000LL) kk.nt_epi = 0;
    }
    constexpr int threads = WM_ * WN_ * 64;
    constexpr int rpp = threads / 8;
    const int pra = (BM + 2 * k.W + 2 + 2 + 7) / 8 * 8;      // + 2: the rows of zeros out-of-image taps read
    const int npatch = k.SPT > 1 ? 2 : 1;
    constexpr int epi = BM / (BN >= 256 ? 4 : 2) * (BN * 4 + 16);
    int lds = npatch * pra * 128 + WS * BN * 128;
    if (lds < epi) lds = epi;
    if (EXPAND && lds < 2 * BM * 128 + epi) lds = 2 * BM * 128 + epi;   // the bf16 tile (two 64-channel planes) + the staging of a pass
    if (WS == 9 && k.SPT != 1) return MT4_EUNSUPPORTED;
    if (lds > 160 * 1024 || (k.SPT > 1 && cdiv(pra, rpp) > 11 - WS)) return MT4_EUNSUPPORTED;   // (next-slice patch pieces ride along with taps 0..)
    if (k.stat_sums && (EXPAND || (k.Cout & 7))) return MT4_EUNSUPPORTED;
    if constexpr (!EXPAND) {
        if (k.stat_sums) {       // the train-mode launch: its own instantiation (float64 channel sums in the epilogue)
            auto fs = conv3x3_patch_kernel<BM, BN, WM_, WN_, WS, false, true>;
            if (lds > 65536) {
                MT4_RAISE_LDS(fs);
            }
            hipLaunchKernelGGL(fs, dim3(kk.total_tiles), dim3(threads), lds, s, kk, pra, npatch);
            return mt4_check_launch();
        }
    }
    auto fn = conv3x3_patch_kernel<BM, BN, WM_, WN_, WS, EXPAND, false>;
    if (lds > 65536) {
        MT4_RAISE_LDS(fn);
    }
    hipLaunchKernelGGL(fn, dim3(kk.total_tiles), dim3(threads), lds, s, kk, pra, npatch);
    return mt4_check_launch();
}

int launch_patch_tile(const ConvK& k, int tile, hipStream_t s) {
    switch (tile) {
        case 23: return launch_patch3x3<256, 256, 4, 4, 2>(k, s);
        case 24: return launch_patch3x3<256, 64, 4, 2, 2>(k, s);
        case 26: return launch_patch3x3<256, 128, 4, 4, 2>(k, s);
        case 30: return launch_patch3x3<128, 128, 2, 2, 2>(k, s);
        case 32: return launch_patch3x3<256, 128, 4, 2, 2>(k, s);
    }
    return MT4_EUNSUPPORTED;   // ids 21, 22, 25, 27, 28, 29, 31: retired variants of the tuning record (deeper weight rings, weights-resident forms:
                               // every one measured slower, profiles/r01_tile_tuning_patch3x3.txt)
}

// the space-to-depth stem: KH x 1 kernel over runs of 4 pixels x 16 channels, stride 1, valid, Cout <= 64
bool stem_patch_ok(const mt4_conv_desc* d, const ConvK& k, bool fast) {
    return fast && d->dtype == MT4_BF16 && d->out_dtype == MT4_BF16 && d->KW == 1 && d->KH <= 8 && d->Cin == 64 && k.pix_bytes == 32 &&
           d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pad_h == 0 && d->pad_w == 0 &&
           d->W == d->Wo + 3 && d->H == d->Ho + d->KH - 1 && !d->residual && !d->out_row_map && d->relu <= 1 && d->Cout <= 64 &&
           (d->Cout % 8) == 0 && k.HoWo >= 256 && k.nsteps == d->KH && k.x_total_bytes < 0x70000000LL;
}

int launch_stem_patch(const ConvK& k, hipStream_t s) {
    constexpr int BM = 256;
    ConvK kk = k;
    kk.n_tiles = 1;
    kk.total_tiles = cdiv(k.M, BM);
    kk.nt_epi = 1;
    {
        const long long min_mb = MT4_NT_MIN_MB;
        if ((long long)k.M * k.Cout * 2 < min_mb * 1000000LL) kk.nt_epi = 0;
    }
    // frame pixels a tile of 256 consecutive output pixels of ONE image can span: its own run, 3 extra pixels per output row it crosses,
    // the KH-1 rows and 3 pixels of the kernel footprint
    kk.n_tiles = cdiv(k.HoWo, BM);                 // tiles per image
    kk.total_tiles = k.B * kk.n_tiles;
    const int rows_crossed = cdiv(BM, k.Wo) + 1;
    const int span = BM + 3 * rows_crossed + (k.KH - 1) * k.W + 4;
    const int pra = (span + 31) / 32 * 32;
    int lds = pra * 32 + k.KH * 64 * 128;
    constexpr int epi = BM / 2 * (64 * 4 + 16);
    if (lds < epi) lds = epi;
    if (lds > 160 * 1024) return MT4_EUNSUPPORTED;
    auto fn = stem_patch_kernel<BM, 4, 2>;
    if (lds > 65536) {
        MT4_RAISE_LDS(fn);
    }
    hipLaunchKernelGGL(fn, dim3(kk.total_tiles), dim3(512), lds, s, kk, pra);
    return mt4_check_launch();
}

int auto_tile(int M, int N, int nsteps, int es) {
    // Measured on MI355X over the ResNet-50 layer set (tools/tune_conv.py, profiles/r01_tile_tuning.txt):
    // long-K layers want the 128x128 tile (most MFMA per LDS byte); short-K (memory-bound) layers want the
    // smaller 64x128 / 128x64 footprints (more workgroups per CU -> more loads and stores in flight).
    auto tiles = [&](int t) { return (long long)cdiv(M, kTiles[t - 1].bm) * cdiv(N, kTiles[t - 1].bn); };
    const long long fill = 256;  // one workgroup per CU
    // 8-wave 256-row tiles (bf16, profiles/r01_tile_tuning_8wave.txt): one workgroup per CU with the same 2 waves per SIMD, but
    // 0.5-0.75x the operand bytes per FLOP through L2 -> LDS-DMA, whose issue cost is what the K-loop waits on
    if (es == 2 && N >= 256) {   // (fp32 launches are bound by the fp32 MFMA rate: the same tiles change nothing there, same-box A/B)
        if (nsteps == 1 && tiles(13) >= fill) return 13;
        if (nsteps >= 2 && tiles(15) >= 190) return 17;
    }
    if (es == 2 && N > 64 && N <= 128 && nsteps < 4 && tiles(20) >= 8 * fill) {   // short K: 256x64 (Swin stage-1 proj)
        return 20;
    }
    if (es == 2 && N > 64 && N <= 128 && nsteps >= 4 && tiles(19) >= 8 * fill) {   // 16-wave 256x128, 3 stages; many rounds: small tail
        return 19;
    }
    // single K-step: smallest footprint, most workgroups per CU
    if (nsteps == 1 && tiles(3) >= 4 * fill) return 3;
    if (N > 64) {
        if (nsteps >= 4 && tiles(1) >= fill) return 1;
        if (tiles(4) >= fill) return 4;
        if (tiles(1) >= fill) return 1;
    } else if (N > 32) {
        // 64-channel layers at many rounds: 256 pixels x 64 channels, 8 waves (+2.8 % frames/s, same-box A/B)
        if (es == 2 && tiles(20) >= 8 * fill) return 20;
        if (tiles(2) >= fill) return 2;
        if (tiles(3) >= fill) return 3;
    }
    // small output, long K (the Q2L decoders' linear2: 768 ... 1920 rows x 1024 channels from K = 8192): the 64 x 64 tile with the deep ring as soon
    // as it gives 3/4 of a round -- a quarter of the operand bytes the 32 x 32 tile pulls through L2 (1280 rows: 99.7 -> 60.3 us, 1920: 149.8 -> 61.3)
    if (nsteps >= 32 && N > 32 && tiles(3) >= 192) return 9;
    // not enough work to fill the chip: take the tile with the most blocks
    int best = 6;
    long long best_tiles = -1;
    for (int t = 1; t <= 6; ++t) {
        if (kTiles[t - 1].bn > 64 && N <= 64) continue;
        if (kTiles[t - 1].bn > 32 && N <= 32) continue;
        if (tiles(t) > best_tiles) {
            best_tiles = tiles(t);
            best = t;
        }
    }
    // few workgroups, long K (a TCN layer over one short video): nothing else hides the per-step DMA latency, so take
    // the 4-stage ring of the same tile (slower than 2 stages whenever the chip is full: it halves workgroups per CU)
    if (nsteps >= 8) {
        if (best == 5) return 10;
        if (best == 6) return 11;
        if (best == 3) return 9;
    }
    return best;
}

}  // namespace

extern "C" int mt4_conv_tile_count(void) { return kNumTiles; }

extern "C" int64_t mt4_conv_packed_k(int32_t Cin, int32_t KH, int32_t KW, int32_t dtype) {
    const int es = dtype == MT4_BF16 ? 2 : 4;
    const int cpt = cdiv(Cin * es, 16);
    const int chunks = cdiv(KH * KW * cpt, 8) * 8;
    return (int64_t)chunks * (16 / es);
}

extern "C" int mt4_conv_nhwc(const mt4_conv_desc* d, void* stream) {
    mt4_clear_error();
    if (!d || !d->x || !d->w || !d->y) return MT4_EINVAL;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->KH <= 0 ||
        d->KW <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->dil_h <= 0 || d->dil_w <= 0)
        return MT4_EINVAL;
    if (d->dtype != MT4_F32 && d->dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (d->out_dtype != MT4_F32 && d->out_dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    if (d->dtype == MT4_F32 && d->out_dtype != MT4_F32) return MT4_EUNSUPPORTED;
    const int es = d->dtype == MT4_BF16 ? 2 : 4;
    if ((d->Cin * es) % 16 != 0) return MT4_EALIGN;
    if (((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->y | (uintptr_t)d->residual | (uintptr_t)d->bias) & 15) return MT4_EALIGN;
    // (Ho/Wo are taken as given: every read is bounds-checked, out-of-image taps contribute zeros -- the sub-pixel phases of
    //  a strided conv's data gradient ask for one output row/column more than the 'valid' formula yields)
    const long long M = (long long)d->B * d->Ho * d->Wo;
    if (M > 0x7fffffffLL || M * d->Cout * 4 > (1LL << 40)) return MT4_EUNSUPPORTED;

    ConvK k{};
    k.x = (const char*)d->x; k.w = (const char*)d->w; k.bias = d->bias; k.res = (const char*)d->residual; k.y = (char*)d->y;
    k.B = d->B; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.KH = d->KH; k.KW = d->KW; k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w;
    k.dh = d->dil_h; k.dw = d->dil_w; k.relu = d->relu;
    if (d->relu < 0 || d->relu > 3) return MT4_EINVAL;
    if (d->relu == 3 && !d->residual) return MT4_EINVAL;
    k.row_map = d->out_row_map; k.map_len = d->out_row_map_len;
    k.map_img = d->out_rows_per_image > 0 ? d->out_rows_per_image : d->out_row_map_len;
    k.y_ld = d->y_ld > 0 ? d->y_ld : d->Cout;
    k.res_ld = d->res_ld > 0 ? d->res_ld : d->Cout;
    if (k.y_ld < d->Cout || k.res_ld < d->Cout) return MT4_EINVAL;
    if (d->residual_float && !(d->dtype == MT4_BF16 && d->out_dtype == MT4_F32 && !d->fuse_w)) return MT4_EUNSUPPORTED;
    k.res_f32 = d->residual_float ? 1 : 0;
    {
        const int oes = d->out_dtype == MT4_BF16 ? 2 : 4;
        const int res_es = k.res_f32 ? 4 : es;
        if ((d->Cout * oes) % 16 == 0 && ((k.y_ld * oes) % 16 != 0 || (d->residual && (k.res_ld * res_es) % 16 != 0))) return MT4_EALIGN;
    }
    if (d->out_row_map && d->out_row_map_len <= 0) return MT4_EINVAL;
    k.M = (int)M; k.HoWo = d->Ho * d->Wo;
    if (d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0) {
        // pure GEMM: every pixel is its own "image" (skips the per-row index decode in the kernel)
        k.B = (int)M; k.H = k.W = k.Ho = k.Wo = 1; k.HoWo = 1;
    }
    k.CPT = (d->Cin * es) / 16;
    k.taps = d->KH * d->KW;
    k.nsteps = cdiv(k.taps * k.CPT, 8);
    k.w_row_bytes = k.nsteps * 128;
    // x_pixel_stride (elements, 0 = Cin): a K-row of Cin elements may span several pixels of a narrower image -- the ResNet stem on
    // the space-to-depth frame reads 4 pixels x 16 channels = one 128-byte run per kernel row (overlapping between neighbouring outputs)
    const int pix = (d->x_pixel_stride > 0 ? d->x_pixel_stride : d->Cin) * es;
    if (pix % 16 != 0 || pix > d->Cin * es) return MT4_EALIGN;
    if (pix != d->Cin * es && (d->KW != 1 || d->pad_w != 0 || d->pad_h != 0)) return MT4_EUNSUPPORTED;
    k.pix_bytes = pix;
    k.x_img_bytes = (long long)k.H * k.W * pix;
    const long long xb = (long long)d->B * d->H * d->W * pix, wb = (long long)d->Cout * k.w_row_bytes;
    // FAST = LDS-DMA staging: whole 128-byte K-steps per tap, 32-bit buffer offsets, one validity bit per kh / kw
    // (x: offsets are relative to the image of a tile's first pixel; a tile of <= 256 output pixels spans at most 256/HoWo + 2 images)
    const long long tile_span = ((long long)(256 / k.HoWo) + 2) * k.x_img_bytes +
                                ((long long)(d->KH - 1) * d->dil_h * d->W + (long long)(d->KW - 1) * d->dil_w) * pix + (long long)k.CPT * 16 + 4096;
    const bool fast = (k.CPT % 8) == 0 && tile_span + (((long long)d->pad_h * d->W + d->pad_w) * pix) < 0x70000000LL && wb < 0x7fffffffLL &&
                      d->KH <= 8 && d->KW <= 8;
    k.x_total_bytes = xb;
    k.x_bias = (unsigned)(((long long)d->pad_h * d->W + d->pad_w) * pix);
    k.x_bytes = (unsigned)(xb < 0x7fffffffLL ? xb : 0);
    k.w_bytes = (unsigned)(wb < 0x7fffffffLL ? wb : 0);
    k.SPT = fast ? k.CPT / 8 : 1;
    if (d->stat_sums) {
        // the statistics of a train-mode BatchNorm behind this convolution: plain launches of the generic and the 3x3 patch kernel only
        if ((uintptr_t)d->stat_sums & 7) return MT4_EALIGN;
        if (d->fuse_w || d->x2 || d->out_row_map || d->tile == -1 || d->tile == 33 || pix != d->Cin * es) return MT4_EUNSUPPORTED;
        k.stat_sums = d->stat_sums;
    }
    if (d->fuse_w && d->fuse_expand) {
        // the Bottleneck's conv3 + bn3 + add + ReLU behind its 3x3 conv, in the patch kernel: only where that kernel runs (many tiles: the
        // caller launches the two convs otherwise -- the results are bit-identical either way)
        if (!d->fuse_y || !d->fuse_bias || d->fuse_cout <= 0 || (d->fuse_cout % 128) != 0) return MT4_EINVAL;
        if (((uintptr_t)d->fuse_w | (uintptr_t)d->fuse_y | (uintptr_t)d->fuse_bias) & 15) return MT4_EALIGN;
        if (!(patch3x3_ok(d, k, fast) && d->Cin == 128 && d->Cout == 128 && d->tile == 0 && !d->x2 &&
              (long long)cdiv(k.M, 256) * cdiv(d->fuse_cout, 256) >= 256))
            return MT4_EUNSUPPORTED;
        k.f_w = (const char*)d->fuse_w; k.f_bias = d->fuse_bias; k.f_y = (char*)d->fuse_y; k.f_cout = d->fuse_cout; k.f_relu = d->fuse_relu ? 1 : 0;
        return d->W <= 31 ? launch_patch3x3<128, 128, 2, 2, 2, true>(k, (hipStream_t)stream) : launch_patch3x3<256, 128, 4, 2, 2, true>(k, (hipStream_t)stream);
    }
    if (d->fuse_w) return MT4_EUNSUPPORTED;   // (fuse_w exists with fuse_expand only; two dependent 1x1 convs in one launch: mt4_chain_gemm_bf16)
    if (d->x2) {
        // second K source: y = act([W | W2] . [x ; x2 gathered at stride x2_stride] + bias); w rows hold both K ranges back to back
        if (d->x2_H <= 0 || d->x2_W <= 0 || d->x2_C <= 0 || d->x2_stride <= 0) return MT4_EINVAL;
        if (!(fast && d->dtype == MT4_BF16 && d->out_dtype == MT4_BF16 && d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 &&
              d->pad_h == 0 && d->pad_w == 0 && !d->out_row_map && !d->residual && d->relu <= 1 && d->tile == 0 && pix == d->Cin * es &&
              (d->x2_C * 2) % 128 == 0 && (d->Cout % 8) == 0 && (k.y_ld * 2) % 16 == 0))
            return MT4_EUNSUPPORTED;
        if ((long long)(d->Ho - 1) * d->x2_stride >= d->x2_H || (long long)(d->Wo - 1) * d->x2_stride >= d->x2_W) return MT4_EINVAL;
        if ((uintptr_t)d->x2 & 15) return MT4_EALIGN;
        k.x2 = (const char*)d->x2;
        k.x2_pix_bytes = d->x2_C * 2;
        k.x2_img_bytes = (long long)d->x2_H * d->x2_W * k.x2_pix_bytes;
        k.x2_total_bytes = (long long)d->B * k.x2_img_bytes;
        k.x2_W = d->x2_W; k.x2_s = d->x2_stride; k.x2_HoWo = d->Ho * d->Wo; k.x2_Wo = d->Wo;
        if (((long long)(256 / k.x2_HoWo) + 2) * k.x2_img_bytes >= 0x70000000LL) return MT4_EUNSUPPORTED;
        k.nsteps1 = k.nsteps;
        k.nsteps += d->x2_C * 2 / 128;
        k.w_row_bytes = k.nsteps * 128;
        const long long wb2 = (long long)d->Cout * k.w_row_bytes;
        if (wb2 >= 0x7fffffffLL) return MT4_EUNSUPPORTED;
        k.w_bytes = (unsigned)wb2;
        return launch_dual(k, (hipStream_t)stream);
    }
    int tile = d->tile;
    if (tile < -1 || tile > kNumTiles) return MT4_EINVAL;
    const bool latency = tile == -1;   // automatic choice, K-split tiles allowed
    if (latency) tile = 0;
    hipStream_t s = (hipStream_t)stream;
    if (tile == 34) return MT4_EUNSUPPORTED;   // (retired id: the persistent form of the stem patch kernel, measured no faster in the bench)
    if (tile == 33 || (tile == 0 && stem_patch_ok(d, k, fast))) {   // the space-to-depth stem
        if (!stem_patch_ok(d, k, fast)) return MT4_EUNSUPPORTED;
        const int rc = launch_stem_patch(k, s);
        if (rc != MT4_EUNSUPPORTED || tile != 0) return rc;
    }
    if (tile >= 21 && tile <= 32) {   // explicit request for the 3x3 patch kernel
        if (!patch3x3_ok(d, k, fast)) return MT4_EUNSUPPORTED;
        return launch_patch_tile(k, tile, s);
    }
    if (tile == 0 && patch3x3_ok(d, k, fast) && (long long)cdiv(k.M, 256) * cdiv(k.Cout, 256) >= 256) {
        // 3x3 stride-1 layers at many rounds of the chip: the patch kernel (same-box sweep at 1336 frames,
        // profiles/r01_tile_tuning_patch3x3.txt: layer1 conv2 0.462 -> 0.375 ms (256x64), layer2 0.343 -> 0.300 (128x128, 4 waves with 64x64
        // wave tiles), layer3 0.242 -> 0.244 and layer4 0.232 -> 0.230 (256x256: even); ResNet-50 bench, alternating runs on one box:
        // 67.8 k frames/s generic, 69.0 k with the tiles below): 256x64 for Cout <= 64, 128x128 for Cout <= 128 (only while the 2W+2 halo stays
        // small -- W = 56 at 256x448 frames: tile 30 0.475 ms, tile 32 0.310, generic 0.325), 256x256 above
        const int pt = d->Cout <= 64 ? 24 : d->Cout <= 128 ? (d->W <= 31 ? 30 : 32) : 23;
        const int rc = launch_patch_tile(k, pt, s);
        if (rc != MT4_EUNSUPPORTED) return rc;   // (patch too large for LDS: generic tiles)
    }
    if (tile == 0) {
        tile = auto_tile(k.M, k.Cout, k.nsteps, d->dtype == MT4_F32 ? 4 : 2);
        // few tiles and a long K: the 4-stage ring of the small tiles (10 / 11), or -- when the caller asked for latency (tile -1) and the
        // geometry is on the LDS-DMA path -- eight K-split groups of two waves per workgroup (36 / 37)
        // (measured, 4-stage TCN, fp32, T = 256: 1.23 ms with the rings, 1.07 with four groups, 0.96 with eight; config 1 0.357 -> 0.277;
        //  with more than one workgroup per CU -- T = 2000 -- the 16-wave workgroups lose 27 %, and bf16 (half the K-steps; 0.640 -> 0.630 ms with eight groups: the 84 dependent launches are the floor) gains nothing:
        //  fp32 launches of at most 256 tiles only)
        if (latency && fast && d->dtype == MT4_F32 && (tile == 10 || tile == 11)) {
            const int kt = tile == 11 ? 37 : 36;
            if ((long long)cdiv(k.M, kTiles[kt - 1].bm) * cdiv(k.Cout, kTiles[kt - 1].bn) <= 256) tile = kt;
        }
        // a whole video (T ~ 2000 frames x 512 channels: `Temporal_tenco/run.py:369-379` runs batch 1 on full videos): 64 x 64 tiles are one
        // round of the 256 CUs and pull the fewest operand bytes per CU ((BM + BN) x K, minimal for square tiles); what they lack is waves to
        // hide the per-K-step DMA round trip -- K-split groups supply them.  Same-box sweep, 4-stage head, T = 2000, hipGraph replay
        // (tools/tcn_long_sweep.py): fp32 2.35 ms (32 x 32, 4-stage ring) -> 2.13 (two groups) -> 2.04 (four groups, tile 40);
        // bf16 1.00 -> 0.83 (tile 40) -> 0.74 (two groups with 3-stage rings, tile 41).  At T = 1000 (128 such tiles: half the chip) the
        // small tiles stay, bf16 on the 4-stage 32 x 64 ring (0.74 -> 0.69 ms)
        // several short videos per forward (the temporal head's throughput mode: 8192 rows x 512 channels at 32 videos): 128 x 128 tiles are a
        // single round of 256 four-wave workgroups; 64 x 128 gives two per CU -- 36.1 -> 30.4 us (dilated conv), 24.4 -> 21.0 (1 x 1) in bf16,
        // 135 -> 123 / 56.5 -> 51.8 in fp32 (profiles/r04_tcn_batched_tile_sweep.txt).  Latency callers only: the spatial paths keep their tuned table.
        if (latency && tile == 1) {
            const long long t128 = (long long)cdiv(k.M, 128) * cdiv(k.Cout, 128), t64 = (long long)cdiv(k.M, 64) * cdiv(k.Cout, 128);
            if (t128 < 512 && t64 >= 256) tile = 4;
        }
        if (latency && fast && k.nsteps >= 8 && (tile == 5 || tile == 6 || tile == 10 || tile == 11 || tile == 3 || tile == 9)) {
            const long long t64 = (long long)cdiv(k.M, 64) * cdiv(k.Cout, 64);
            if (t64 >= 192 && t64 <= 512) tile = d->dtype == MT4_F32 ? 40 : 41;
            else if (d->dtype == MT4_BF16 && tile == 11 && (long long)cdiv(k.M, 32) * cdiv(k.Cout, 64) >= 256) tile = 10;
        }
    }
    if (d->dtype == MT4_F32) return launch_dtype<float, true>(k, tile, fast, s);
    if (d->out_dtype == MT4_F32) return launch_dtype<u16, true>(k, tile, fast, s);
    return launch_dtype<u16, false>(k, tile, fast, s);
}

// ------------------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                                        int Cout, int Cin, int KH, int KW, int CPT_E, long long Kpad) {
    // one thread per packed element
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Kpad) return;
    const int n = (int)(idx / Kpad);
    const long long k = idx - (long long)n * Kpad;
    const int tap = (int)(k / CPT_E);
    const int c = (int)(k - (long long)tap * CPT_E);
    float v = 0.f;
    if (tap < KH * KW && c < Cin) {
        const int kh = tap / KW, kw = tap - kh * KW;
        v = w[(((long long)n * Cin + c) * KH + kh) * KW + kw];
        if (scale) v *= scale[n];
    }
    if constexpr (sizeof(T) == 2) out[idx] = f32_to_bf16(v);
    else out[idx] = v;
}

extern "C" int mt4_pack_conv_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t Cin,
                                    int32_t KH, int32_t KW, int32_t dtype, void* stream) {
    mt4_clear_error();
    if (!w_oihw || !w_packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const int es = dtype == MT4_BF16 ? 2 : 4;
    const int E = 16 / es;
    const int cpt = cdiv(Cin * es, 16);
    const long long Kpad = mt4_conv_packed_k(Cin, KH, KW, dtype);
    const long long total = (long long)Cout * Kpad;
    const int grid = (int)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MT4_BF16)
        hipLaunchKernelGGL(pack_conv_weight_kernel<u16>, dim3(grid), dim3(256), 0, s, w_oihw, scale, (u16*)w_packed, Cout, Cin,
                           KH, KW, cpt * E, Kpad);
    else
        hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w_oihw, scale, (float*)w_packed, Cout,
                           Cin, KH, KW, cpt * E, Kpad);
    return mt4_check_launch();
}

// relu(stem conv on the space-to-depth frame) -> 3x3 / 2 max-pool, pad 1, one launch (stem_pool_kernel).  x_s2d [B][Hs][Ws][16] bf16
// (mt4_preprocess_u8_s2d), w_packed [Cout][KH * 64] bf16 (the 4 x 1 kernel over runs of 4 pixels), y [B][Hp][Wp][Cout] bf16 with
// Ho = Hs - KH + 1, Wo = Ws - 3, Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1.  Conv rows in whole 16-pixel tiles, up to 112 pixels (224-pixel
// frames) as one tile per pair of pooled rows, up to 224 (448-pixel frames) as two column segments; wider: MT4_EUNSUPPORTED (the caller runs
// mt4_conv_nhwc + mt4_maxpool3x3s2_nhwc).
extern "C" int mt4_stem_maxpool_bf16(const void* x_s2d, const void* w_packed, const float* bias, void* y, int32_t B, int32_t Hs, int32_t Ws,
                                     int32_t Cout, int32_t KH, void* stream) {
    mt4_clear_error();
    if (!x_s2d || !w_packed || !y || B <= 0 || Hs <= 0 || Ws <= 3 || KH <= 0 || KH > Hs) return MT4_EINVAL;
    if (((uintptr_t)x_s2d | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)bias) & 15) return MT4_EALIGN;
    const int Ho = Hs - KH + 1, Wo = Ws - 3;
    if (KH > 8 || Cout != 64 || (Wo & 15) || Wo > 224) return MT4_EUNSUPPORTED;
    const long long xb = (long long)B * Hs * Ws * 32;
    if (xb >= 0x70000000LL) return MT4_EUNSUPPORTED;
    StemPoolK k{};
    k.x = (const char*)x_s2d; k.w = (const char*)w_packed; k.bias = bias; k.y = (char*)y;
    k.x_bytes = (unsigned)xb;
    k.w_row_bytes = KH * 128;
    k.w_bytes = (unsigned)(Cout * k.w_row_bytes);
    k.Cout = Cout; k.KH = KH; k.Hs = Hs; k.Ws = Ws; k.Ho = Ho; k.Wo = Wo;
    k.Hp = (Ho - 1) / 2 + 1; k.Wp = (Wo - 1) / 2 + 1;
    const bool wide = Wo > 112;       // two column segments of 128 conv columns (the second starts at 2 Q0 - 1), each for half the pooled columns
    k.segs = wide ? 2 : 1;
    k.cw = wide ? 128 : Wo;
    k.rs = wide ? 132 : Ws;           // 128 + 3 pixels of kernel footprint, padded
    k.wp_seg = wide ? cdiv(k.Wp, 2) : k.Wp;
    k.ctw = k.cw < 2 * k.wp_seg + 8 ? k.cw : 2 * k.wp_seg + 8;
    k.tiles_per_img = cdiv(k.Hp, 2) * k.segs;
    k.pra = ((4 + KH) * k.rs + 31) / 32 * 32;      // five conv rows: 4 + KH frame rows
    int lds = k.pra * 32 + KH * 64 * 128;
    if (lds < 5 * k.ctw * 128) lds = 5 * k.ctw * 128;
    if (lds > 80 * 1024) return MT4_EUNSUPPORTED;
    if ((long long)B * k.tiles_per_img > 0x7fffffffLL) return MT4_EUNSUPPORTED;
    if (wide) {
        auto fn = stem_pool_kernel<10, 2>;
        MT4_RAISE_LDS(fn);
        hipLaunchKernelGGL(fn, dim3((unsigned)(B * k.tiles_per_img)), dim3(512), lds, (hipStream_t)stream, k);
    } else {
        auto fn = stem_pool_kernel<9, 2>;      // (two passes also here: 1.17 -> 1.10 ms per 1336 frames of 224 x 224, same box)
        if (lds > 65536) MT4_RAISE_LDS(fn);
        hipLaunchKernelGGL(fn, dim3((unsigned)(B * k.tiles_per_img)), dim3(512), lds, (hipStream_t)stream, k);
    }
    return mt4_check_launch();
}

// stem: [64][3][7][7] -> taps (kh, kwp) with 8-element slots (kw%2)*4 + c, kw = 2*kwp + slot/4
template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ out,
                                        int Cout, long long Kpad) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Kpad) return;
    const int n = (int)(idx / Kpad);
    const int k = (int)(idx - (long long)n * Kpad);
    const int tap = k >> 3, slot = k & 7;
    const int kh = tap >> 2, kwp = tap & 3;
    const int kw = 2 * kwp + (slot >> 2), c = slot & 3;
    float v = 0.f;
    if (kh < 7 && kw < 7 && c < 3) {
        v = w[(((long long)n * 3 + c) * 7 + kh) * 7 + kw];
        if (scale) v *= scale[n];
    }
    if constexpr (sizeof(T) == 2) out[idx] = f32_to_bf16(v);
    else out[idx] = v;
}

extern "C" int mt4_pack_stem_weight(const float* w_oihw, const float* scale, void* w_packed, int32_t Cout, int32_t dtype,
                                    void* stream) {
    mt4_clear_error();
    if (!w_oihw || !w_packed || Cout <= 0) return MT4_EINVAL;
    if (dtype != MT4_F32 && dtype != MT4_BF16) return MT4_EUNSUPPORTED;
    const long long Kpad = mt4_conv_packed_k(8, 7, 4, dtype);
    const long long total = (long long)Cout * Kpad;
    const int grid = (int)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MT4_BF16)
        hipLaunchKernelGGL(pack_stem_weight_kernel<u16>, dim3(grid), dim3(256), 0, s, w_oihw, scale, (u16*)w_packed, Cout, Kpad);
    else
        hipLaunchKernelGGL(pack_stem_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w_oihw, scale, (float*)w_packed, Cout,
                           Kpad);
    return mt4_check_launch();
}
