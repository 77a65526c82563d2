// PNG decode on the device (the reference decodes every frame with PIL inside its DataLoader workers,
// `Spatial_cnn/dataloader.py:257-261`, three processes, `Spatial_cnn/test.py:240-241`).  Two launches per batch of frames:
//   mt4_png_inflate       zlib / DEFLATE (RFC 1950 / 1951) streams -> the filtered scanlines, ONE THREAD PER FRAME: a DEFLATE stream is
//                         a serial bit stream, the parallelism is across the frames of a video (hundreds per batch);
//   mt4_png_unfilter_rgb8 the five PNG scanline filters (PNG spec 9.2) undone for 8-bit RGB -> [B][H][W][3] uint8.
// The host side (pngdec.py) only walks the chunk list of each file (IHDR, IDAT concatenation) -- no inflate, no pixel work on the CPU.
// Decoder structure: canonical-Huffman decode by code length with per-length counts and a sorted symbol list (the classic table-free
// scheme: at most 15 steps per symbol, no per-block lookup table to build); the count / symbol arrays of a thread live in LDS, index-major
// ([entry][thread]) so that the lanes of a wave reading the same entry hit different banks.
#include "mt4_common.h"

namespace {

constexpr int PNG_THREADS = 64;
constexpr int MAXBITS = 15, MAXL = 288, MAXD = 32;

__device__ const unsigned short kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const unsigned short kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const unsigned char kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    unsigned long long buf;
    int cnt;
    int bad;      // ran past the end of the stream
};

__device__ __forceinline__ void refill(Bits& b) {
    if (b.cnt <= 32) {
        if (b.p + 4 <= b.end) {
            struct __attribute__((packed)) U4 { uint32_t v; };
            b.buf |= (unsigned long long)((const U4*)b.p)->v << b.cnt;
            b.p += 4;
            b.cnt += 32;
        } else {
            while (b.cnt <= 56 && b.p < b.end) { b.buf |= (unsigned long long)(*b.p++) << b.cnt; b.cnt += 8; }
        }
    }
}

__device__ __forceinline__ unsigned getbits(Bits& b, int n) {   // n <= 16
    if (b.cnt < n) { refill(b); if (b.cnt < n) { b.bad = 1; return 0; } }
    const unsigned v = (unsigned)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// tables of one thread, in LDS: entry-major
struct Huff {
    unsigned short* count;    // [MAXBITS + 1] entries, stride PNG_THREADS
    unsigned short* symbol;   // [n] entries, stride PNG_THREADS
};
#define HC(h, i) (h).count[(i) * PNG_THREADS]
#define HS(h, i) (h).symbol[(i) * PNG_THREADS]

// canonical code from the code lengths: count[len] = number of codes of that length, symbol[] = symbols ordered by (length, value).
// Returns < 0 for an over-subscribed set, > 0 for an incomplete one (allowed only for a single-code distance tree), 0 for a complete one.
__device__ int construct(Huff& h, const unsigned char* lens, int n) {
    for (int len = 0; len <= MAXBITS; ++len) HC(h, len) = 0;
    for (int s = 0; s < n; ++s) HC(h, lens[s]) = HC(h, lens[s]) + 1;
    if (HC(h, 0) == n) return 0;          // no codes: complete, but decoding any symbol fails
    int left = 1;
    for (int len = 1; len <= MAXBITS; ++len) {
        left <<= 1;
        left -= HC(h, len);
        if (left < 0) return left;
    }
    unsigned short offs[MAXBITS + 1];
    offs[1] = 0;
    for (int len = 1; len < MAXBITS; ++len) offs[len + 1] = offs[len] + HC(h, len);
    for (int s = 0; s < n; ++s)
        if (lens[s]) { HS(h, offs[lens[s]]) = (unsigned short)s; offs[lens[s]]++; }
    return left;
}

__device__ __forceinline__ int decode_sym(Bits& b, const Huff& h) {
    refill(b);
    int code = 0, first = 0, index = 0;
    unsigned long long buf = b.buf;
    const int avail = b.cnt < MAXBITS ? b.cnt : MAXBITS;
    for (int len = 1; len <= avail; ++len) {
        code |= (int)(buf & 1);
        buf >>= 1;
        const int count = HC(h, len);
        if (code - count < first) {
            b.buf = buf;
            b.cnt -= len;
            return HS(h, index + (code - first));
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    b.bad = 1;
    return -1;
}

// error codes written to status[]: 0 ok, 1 truncated input, 2 bad block type, 3 stored-length mismatch, 4 bad code lengths,
// 5 invalid symbol / distance, 6 output overflow, 7 output short of H * (1 + 3 W)
__global__ __launch_bounds__(PNG_THREADS) void png_inflate_kernel(const uint8_t* __restrict__ streams, const long long* __restrict__ offsets,
                                                                   const int* __restrict__ lengths, uint8_t* __restrict__ raw, int B,
                                                                   long long raw_stride, long long raw_len, int* __restrict__ status) {
    __shared__ unsigned short lcount[(MAXBITS + 1) * PNG_THREADS], lsym[MAXL * PNG_THREADS];
    __shared__ unsigned short dcount[(MAXBITS + 1) * PNG_THREADS], dsym[MAXD * PNG_THREADS];
    const int t = threadIdx.x;
    const int img = blockIdx.x * PNG_THREADS + t;
    if (img >= B) return;
    Huff lc{lcount + t, lsym + t}, dc{dcount + t, dsym + t};
    Bits b;
    b.p = streams + offsets[img];
    b.end = b.p + lengths[img];
    b.buf = 0; b.cnt = 0; b.bad = 0;
    uint8_t* const out = raw + (long long)img * raw_stride;
    long long pos = 0;
    int err = 0, last = 0;
    unsigned char lens[MAXL + MAXD];
    while (!last && !err) {
        last = (int)getbits(b, 1);
        const int type = (int)getbits(b, 2);
        if (b.bad) { err = 1; break; }
        if (type == 0) {                       // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const unsigned len = getbits(b, 16), nlen = getbits(b, 16);
            if (b.bad) { err = 1; break; }
            if ((len ^ 0xffffu) != nlen) { err = 3; break; }
            if (pos + len > raw_len) { err = 6; break; }
            for (unsigned i = 0; i < len; ++i) {
                const unsigned v = getbits(b, 8);
                out[pos++] = (uint8_t)v;
            }
            if (b.bad) { err = 1; break; }
            continue;
        }
        if (type == 3) { err = 2; break; }
        if (type == 1) {                       // fixed codes
            int s = 0;
            for (; s < 144; ++s) lens[s] = 8;
            for (; s < 256; ++s) lens[s] = 9;
            for (; s < 280; ++s) lens[s] = 7;
            for (; s < 288; ++s) lens[s] = 8;
            construct(lc, lens, 288);
            for (s = 0; s < 30; ++s) lens[s] = 5;
            construct(dc, lens, 30);
        } else {                               // dynamic codes
            const int nlen = (int)getbits(b, 5) + 257, ndist = (int)getbits(b, 5) + 1, ncode = (int)getbits(b, 4) + 4;
            if (b.bad) { err = 1; break; }
            if (nlen > 286 || ndist > 30) { err = 4; break; }
            for (int i = 0; i < 19; ++i) lens[i] = 0;
            for (int i = 0; i < ncode; ++i) lens[kClOrder[i]] = (unsigned char)getbits(b, 3);
            if (construct(lc, lens, 19) != 0) { err = 4; break; }
            int idx = 0;
            while (idx < nlen + ndist) {
                const int sym = decode_sym(b, lc);
                if (sym < 0) { err = 1; break; }
                if (sym < 16) { lens[idx++] = (unsigned char)sym; continue; }
                int rep, val = 0;
                if (sym == 16) {
                    if (idx == 0) { err = 4; break; }
                    val = lens[idx - 1];
                    rep = 3 + (int)getbits(b, 2);
                } else if (sym == 17) rep = 3 + (int)getbits(b, 3);
                else rep = 11 + (int)getbits(b, 7);
                if (idx + rep > nlen + ndist) { err = 4; break; }
                while (rep--) lens[idx++] = (unsigned char)val;
            }
            if (err) break;
            if (b.bad) { err = 1; break; }
            if (lens[256] == 0) { err = 4; break; }
            // (the code-length table is dead: build the literal / length table over it)
            unsigned char dl[MAXD];
            for (int i = 0; i < ndist; ++i) dl[i] = lens[nlen + i];
            int r = construct(lc, lens, nlen);
            if (r < 0 || (r > 0 && nlen - HC(lc, 0) != 1)) { err = 4; break; }
            r = construct(dc, dl, ndist);
            if (r < 0 || (r > 0 && ndist - HC(dc, 0) != 1)) { err = 4; break; }
        }
        // literal / length + distance codes of the block
        for (;;) {
            int sym = decode_sym(b, lc);
            if (sym < 0) { err = 1; break; }
            if (sym < 256) {
                if (pos >= raw_len) { err = 6; break; }
                out[pos++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = 5; break; }
            const int len = kLenBase[sym] + (int)getbits(b, kLenExtra[sym]);
            const int ds = decode_sym(b, dc);
            if (ds < 0) { err = 1; break; }
            if (ds >= 30) { err = 5; break; }
            const long long dist = kDistBase[ds] + (long long)getbits(b, kDistExtra[ds]);
            if (b.bad) { err = 1; break; }
            if (dist > pos) { err = 5; break; }
            if (pos + len > raw_len) { err = 6; break; }
            const uint8_t* src = out + pos - dist;
            for (int i = 0; i < len; ++i) out[pos + i] = src[i];      // (overlapping copies repeat the pattern, byte by byte as DEFLATE defines)
            pos += len;
        }
    }
    if (!err && pos != raw_len) err = 7;
    status[img] = err;
}

// PNG filters, 8-bit RGB (bytes per pixel = 3): raw row = filter type + 3 W bytes.  One thread per (frame, channel): the Sub / Average /
// Paeth predictors chain along the row per channel, every row needs the finished row above.
__global__ void png_unfilter_rgb8_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ out, int B, int H, int W, long long raw_stride,
                                         int* __restrict__ status) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 3) return;
    const int img = idx / 3, c = idx - img * 3;
    const uint8_t* r = raw + (long long)img * raw_stride;
    uint8_t* o = out + (long long)img * H * W * 3;
    const int rowb = 1 + 3 * W;
    int bad = 0;
    for (int y = 0; y < H; ++y) {
        const uint8_t* row = r + (long long)y * rowb;
        const int ft = row[0];
        uint8_t* cur = o + (long long)y * W * 3;
        const uint8_t* up = y ? cur - W * 3 : nullptr;
        int left = 0, upleft = 0;
        if (ft > 4) bad = 1;
        for (int x = 0; x < W; ++x) {
            const int v = row[1 + 3 * x + c];
            const int a = left, bb = up ? up[3 * x + c] : 0, cc = upleft;
            int pred;
            if (ft == 0) pred = 0;
            else if (ft == 1) pred = a;
            else if (ft == 2) pred = bb;
            else if (ft == 3) pred = (a + bb) >> 1;
            else {
                const int p = a + bb - cc;
                const int pa = abs(p - a), pb = abs(p - bb), pc = abs(p - cc);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : cc);
            }
            const int px = (v + pred) & 255;
            cur[3 * x + c] = (uint8_t)px;
            left = px;
            upleft = bb;
        }
    }
    if (bad && c == 0) status[img] = 8;     // unknown filter type
}

}  // namespace

// B DEFLATE streams (the zlib payload of each PNG's IDAT chunks, 2-byte zlib header already stripped): stream i = streams[offsets[i] ..
// offsets[i] + lengths[i]) -> raw + i * raw_stride, exactly raw_len = H * (1 + 3 W) bytes each.  status[i] = 0 or a decode error code
// (device array of B int32; the caller checks it after the stream has drained).  Enqueue only.
extern "C" int mt4_png_inflate(const uint8_t* streams, const int64_t* offsets, const int32_t* lengths, uint8_t* raw, int32_t B,
                               int64_t raw_stride, int64_t raw_len, int32_t* status, void* stream) {
    mt4_clear_error();
    if (!streams || !offsets || !lengths || !raw || !status || B <= 0 || raw_len <= 0 || raw_stride < raw_len) return MT4_EINVAL;
    hipLaunchKernelGGL(png_inflate_kernel, dim3((unsigned)((B + PNG_THREADS - 1) / PNG_THREADS)), dim3(PNG_THREADS), 0, (hipStream_t)stream, streams,
                       (const long long*)offsets, (const int*)lengths, raw, B, (long long)raw_stride, (long long)raw_len, status);
    return mt4_check_launch();
}

// the filtered scanlines of mt4_png_inflate -> uint8 frames [B][H][W][3] (PNG color type 2, bit depth 8, no interlace)
extern "C" int mt4_png_unfilter_rgb8(const uint8_t* raw, uint8_t* out, int32_t B, int32_t H, int32_t W, int64_t raw_stride, int32_t* status,
                                     void* stream) {
    mt4_clear_error();
    if (!raw || !out || !status || B <= 0 || H <= 0 || W <= 0 || raw_stride < (int64_t)H * (1 + 3 * W)) return MT4_EINVAL;
    const int n = B * 3;
    hipLaunchKernelGGL(png_unfilter_rgb8_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, raw, out, B, H, W,
                       (long long)raw_stride, status);
    return mt4_check_launch();
}
