// PNG decode on the device (the reference decodes every frame with PIL inside its DataLoader workers,
// `Spatial_cnn/dataloader.py:257-261`, three processes, `Spatial_cnn/test.py:240-241`).  Two launches per batch of frames:
//   mt4_png_inflate       zlib / DEFLATE (RFC 1950 / 1951) streams -> the filtered scanlines, ONE WAVE PER FRAME;
//   mt4_png_unfilter_rgb8 the five PNG scanline filters (PNG spec 9.2) undone for 8-bit RGB -> [B][H][W][3] uint8, one wave per frame.
// The host side (pngdec.py) only walks the chunk list of each file (IHDR, IDAT concatenation) -- no inflate, no pixel work on the CPU.
// A DEFLATE stream is a serial bit stream; what a wave can do in parallel is everything around the bit decode, and that is where a
// thread-per-frame decoder (the first version: 0.4 k frames/s at 480 x 854, every output byte a dependent global store / load) spends its time:
//   * the 64 lanes run the SAME decode (uniform control flow, no divergence): Huffman tables, the most recent 8 KB of the output (a ring) and the code-length
//     scratch are the wave's own LDS, read at one address by every lane (broadcast);
//   * symbols are decoded through first-level lookup tables (11 bits literal / length, 10 bits distance: one LDS read per symbol), filled by
//     all lanes in parallel for every dynamic block; longer codes fall back to the canonical decode by code length;
//   * a match is copied by the lanes in parallel inside the ring (from the frame's output in memory when it reaches further back); finished
//     4 KB pieces of the ring go to memory as 16-byte vectors;
//   * a run of literals is looked up by the lanes in parallel (lane l: the code that starts l bits ahead), see the literal run below.
// The unfilter kernel keeps the raw and the previous row in LDS: rows of type None / Up are done by all lanes, Sub / Average / Paeth (a
// recurrence along the row per channel) by three lanes -- from LDS, so the chain runs at register speed.
#include "mt4_common.h"

namespace {

constexpr int MAXBITS = 15, MAXL = 288, MAXD = 32;
// The wave keeps the last WIN bytes of its output in an LDS ring and writes them out in pieces of FLUSH bytes; DEFLATE's 32 KB history beyond
// the ring is the frame's own output in memory (a match that reaches further back than the ring reads it from there: 0.2 % of the symbols are
// matches at all, and a scanline of a 854-pixel frame is 2.5 KB).  A 32 KB ring allowed four frames per CU; this size allows ten, and a wave
// that waits ~100 cycles per literal hop needs neighbours on its SIMD.
constexpr int WIN = 8192, FLUSH = 4096, NEAR = WIN - 512;      // matches up to NEAR bytes back are copied inside the ring
constexpr int LBITS = 11, DBITS = 10;
constexpr int EMPTY = 256;       // lookup entry without a code

__device__ const unsigned short kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const unsigned short kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const unsigned char kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Bit input.  The wave holds 512 bytes of the stream in registers (lane i: the two dwords 2i, 2i + 1 of the current chunk, aligned to 4 bytes in
// memory); a refill takes the next dword with a lane read -- one memory round trip per 512 bytes instead of one per 4 (which was most of a
// symbol's time).  Everything else is identical in every lane.
struct Bits {
    const uint32_t* base;      // aligned dword that holds the first byte of the stream
    int nbits;                 // bits in the stream (< 2^31: host-checked)
    int mis;                   // bytes in front of the stream inside base[0]
    unsigned long long buf;
    int cnt;
    int dw;                    // index (from base) of the next dword to take
    int chunk0;                // dword index of the chunk in registers
    uint32_t w0, w1;           // this lane's two dwords of the chunk
    int bad;
};

__device__ __forceinline__ void load_chunk(Bits& b, int lane) {
    b.chunk0 = b.dw & ~127;
    // a chunk that STARTS behind the stream's last byte is never read: a stream with intact PNG chunks but a corrupt DEFLATE body (no CRC /
    // Adler check here) would otherwise run the literal loops up to raw_len x 15 bits past its end -- far beyond the 1 KB pad behind the last
    // stream of a batch.  The wave decodes zeros from there on and `bad` is reported at the next check (block header, match, flush point, end)
    if (b.chunk0 > ((b.nbits >> 3) + b.mis + 3) >> 2) {
        b.w0 = b.w1 = 0;
        b.bad = 1;
        return;
    }
    const uint2 t = *(const uint2*)(b.base + b.chunk0 + 2 * lane);      // (the host pads the blob by 1 KB: the last chunk of the last stream is in bounds)
    b.w0 = t.x; b.w1 = t.y;
}

__device__ __forceinline__ void refill(Bits& b, int lane) {
    if (b.cnt <= 32) {
        if (b.dw - b.chunk0 >= 128) load_chunk(b, lane);
        const int j = __builtin_amdgcn_readfirstlane(b.dw - b.chunk0);
        const uint32_t lo = __builtin_amdgcn_readlane(b.w0, j >> 1), hi = __builtin_amdgcn_readlane(b.w1, j >> 1);
        b.buf |= (unsigned long long)((j & 1) ? hi : lo) << b.cnt;
        b.cnt += 32;
        b.dw += 1;
    }
}

// bits of the stream consumed so far (derived, not maintained: the per-symbol path carries as little state as it can -- a wave issues one
// instruction every 4-5 cycles, the instruction count of the literal loop IS its time)
__device__ __forceinline__ int bits_used(const Bits& b) { return 32 * b.dw - b.cnt - 8 * b.mis; }

__device__ __forceinline__ unsigned getbits(Bits& b, int n, int lane) {   // n <= 16
    refill(b, lane);
    const unsigned v = (unsigned)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// the wave's LDS (one wave per workgroup)
__shared__ __attribute__((aligned(16))) uint8_t g_win[WIN + 64];      // (+ 64 spare bytes: where lanes 1..63 put their copy of a literal)
__shared__ unsigned short g_lcount[MAXBITS + 1], g_lsym[MAXL], g_ltab[1 << LBITS];
__shared__ unsigned short g_dcount[MAXBITS + 1], g_dsym[MAXD], g_dtab[1 << DBITS];
__shared__ unsigned char g_lens[384];
__shared__ unsigned short s_lbase[32], s_dbase[32];
__shared__ unsigned char s_lext[32], s_dext[32];

// canonical Huffman code of one alphabet, in LDS: count[len], symbols sorted by (length, value), first-level lookup table
struct Huff {
    unsigned short* count;    // [MAXBITS + 1]
    unsigned short* symbol;   // [n]
    unsigned short* tab;      // [1 << tbits]: (code length << 9) | symbol; EMPTY (length 0, bit 8 set like every non-literal) = longer than tbits
                              // (or unused): canonical decode
    int tbits;
};

__device__ __forceinline__ unsigned rev_bits(unsigned c, int len) { return __brev(c) >> (32 - len); }
// every lane reads the same LDS address in the decode path: telling the compiler so (readfirstlane) moves the whole bit-stream state machine --
// buffer, counters, branches -- to the scalar unit; as vector values every branch waited on a v_cmp and every 64-bit shift ran at quarter rate
#define UNI(x) __builtin_amdgcn_readfirstlane((int)(x))

// from the code lengths lens[0..n) (LDS): counts, sorted symbols, lookup table.  Called by the whole wave; returns (uniformly) < 0 for an
// over-subscribed set, > 0 for an incomplete one, 0 for a complete one.
// (not inlined: five call sites, and inlined its loops kept the kernel's scalar registers spilling into vector lanes on the per-symbol path)
__device__ __noinline__ int construct(int which, int lens_off, int n, int lane) {
    Huff h = which ? Huff{g_dcount, g_dsym, g_dtab, DBITS} : Huff{g_lcount, g_lsym, g_ltab, LBITS};
    const unsigned char* lens = g_lens + lens_off;
    __syncthreads();          // (single-wave block: orders the LDS traffic of the previous user of these arrays)
    if (lane <= MAXBITS) h.count[lane] = 0;
    for (int i = lane; i < (1 << h.tbits); i += 64) h.tab[i] = EMPTY;
    __syncthreads();
    if (lane == 0)
        for (int s = 0; s < n; ++s) h.count[lens[s]] = h.count[lens[s]] + 1;
    __syncthreads();
    if (UNI(h.count[0]) == n) return 0;
    int left = 1;
    unsigned short offs[MAXBITS + 2], first[MAXBITS + 1];
    offs[1] = 0;
    int code = 0;
    for (int len = 1; len <= MAXBITS; ++len) {
        const int c = UNI(h.count[len]);
        left = (left << 1) - c;
        if (left < 0) return left;
        first[len] = (unsigned short)code;          // canonical code of the first symbol of this length
        code = (code + c) << 1;
        offs[len + 1] = offs[len] + c;
    }
    if (lane == 0) {
        unsigned short o[MAXBITS + 1];
        for (int len = 1; len <= MAXBITS; ++len) o[len] = offs[len];
        for (int s = 0; s < n; ++s)
            if (lens[s]) { h.symbol[o[lens[s]]] = (unsigned short)s; o[lens[s]]++; }
    }
    __syncthreads();
    const int nsym = offs[MAXBITS + 1];
    for (int p = lane; p < nsym; p += 64) {      // lookup entries of sorted position p
        int len = 1;
        while (p >= offs[len + 1]) ++len;
        if (len <= h.tbits) {
            const unsigned c = first[len] + (p - offs[len]);
            const unsigned r = rev_bits(c, len);
            const unsigned short e = (unsigned short)((len << 9) | h.symbol[p]);
            for (unsigned k = r; k < (1u << h.tbits); k += (1u << len)) h.tab[k] = e;
        }
    }
    __syncthreads();
    return left;
}

__device__ __forceinline__ int decode_sym(Bits& b, const Huff& h, int lane) {
    refill(b, lane);
    const int e = UNI(h.tab[b.buf & ((1u << h.tbits) - 1)]);
    if (e >> 9) {
        const int elen = e >> 9;
        b.buf >>= elen;
        b.cnt -= elen;
        return e & 511;
    }
    int code = 0, first = 0, index = 0;      // a code longer than the table's index: canonical decode by length
    unsigned long long buf = b.buf;
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)(buf & 1);
        buf >>= 1;
        const int count = UNI(h.count[len]);
        if (code - count < first) {
            b.buf = buf;
            b.cnt -= len;
            return UNI(h.symbol[index + (code - first)]);
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
__global__ __launch_bounds__(64) void png_inflate_kernel(const uint8_t* __restrict__ streams, const long long* __restrict__ offsets,
                                                         const int* __restrict__ lengths, uint8_t* __restrict__ raw, int B, long long raw_stride,
                                                         long long raw_len64, int* __restrict__ status) {
    uint8_t* const win = g_win;
    unsigned char* const lens = g_lens;
    const int lane = threadIdx.x;
    const int img = blockIdx.x;
    if (lane < 29) { s_lbase[lane] = kLenBase[lane]; s_lext[lane] = kLenExtra[lane]; }      // (as __device__ tables every match paid four memory round trips)
    if (lane < 30) { s_dbase[lane] = kDistBase[lane]; s_dext[lane] = kDistExtra[lane]; }
    __syncthreads();
    const Huff lc{g_lcount, g_lsym, g_ltab, LBITS}, dc{g_dcount, g_dsym, g_dtab, DBITS};
    Bits b;
    {
        const uint8_t* p0 = streams + offsets[img];
        const int mis = (int)((uintptr_t)p0 & 3);
        b.base = (const uint32_t*)(p0 - mis);
        b.nbits = lengths[img] * 8;
        b.mis = mis; b.buf = 0; b.cnt = 0; b.bad = 0; b.dw = 0; b.chunk0 = 0;
        load_chunk(b, lane);
        refill(b, lane);
        b.buf >>= 8 * mis;           // bytes in front of the stream inside its first dword
        b.cnt -= 8 * mis;
    }
    uint8_t* const out = raw + (long long)img * raw_stride;      // (16-byte aligned: raw_stride % 16 == 0, host-checked)
    int pos = 0, flushed = 0;                 // (32-bit: raw_len < 2 GiB, host-checked; 64-bit compares have no scalar form)
    const int raw_len = (int)raw_len64;
    int err = 0, last = 0;
    auto flush_to = [&](int upto) {       // window bytes [flushed, upto) -> memory; whole 16-byte vectors while they last
        while (flushed + 16 * 64 <= upto) {
            *(uint4*)(out + flushed + lane * 16) = *(const uint4*)(win + ((flushed + lane * 16) & (WIN - 1)));
            flushed += 16 * 64;
        }
        for (int i = flushed + lane; i < upto; i += 64) out[i] = win[i & (WIN - 1)];
        flushed = upto;
    };
    while (!last && !err) {
        last = (int)getbits(b, 1, lane);
        const int type = (int)getbits(b, 2, lane);
        if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
        if (type == 0) {                       // stored
            const int drop = b.cnt & 7;          // to the next byte boundary of the stream
            b.buf >>= drop; b.cnt -= drop;
            const unsigned len = getbits(b, 16, lane), nlen = getbits(b, 16, lane);
            if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
            if ((len ^ 0xffffu) != nlen) { err = 3; break; }
            if (pos + (int)len > raw_len) { err = 6; break; }
            const int used = bits_used(b);       // (a whole number of bytes here)
            if (used + 8 * (int)len > b.nbits) { err = 1; break; }
            // the lanes copy the block from the stream, then the bit reader restarts behind it
            const uint8_t* sp = (const uint8_t*)b.base + b.mis + (used >> 3);
            unsigned done = 0;
            while (done < len) {
                const unsigned chunk = min(len - done, (unsigned)FLUSH);
                for (unsigned i = lane; i < chunk; i += 64) win[(pos + i) & (WIN - 1)] = sp[done + i];
                pos += chunk;
                done += chunk;
                __syncthreads();
                if (pos - flushed >= FLUSH) flush_to(flushed + (pos - flushed) / FLUSH * FLUSH);      // (whole pieces: `flushed` stays 16-byte aligned)
                __syncthreads();
            }
            {   // restart the bit reader at the byte behind the block
                const int byte = b.mis + (used >> 3) + (int)len;
                b.dw = byte >> 2;
                b.buf = 0; b.cnt = 0;
                load_chunk(b, lane);
                refill(b, lane);
                const int mis2 = byte & 3;
                b.buf >>= 8 * mis2;
                b.cnt -= 8 * mis2;
            }
            continue;
        }
        if (type == 3) { err = 2; break; }
        if (type == 1) {                       // fixed codes
            __syncthreads();
            for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            __syncthreads();
            construct(0, 0, 288, lane);
            __syncthreads();
            if (lane < 30) lens[lane] = 5;
            __syncthreads();
            construct(1, 0, 30, lane);
        } else {                               // dynamic codes
            const int nlen = (int)getbits(b, 5, lane) + 257, ndist = (int)getbits(b, 5, lane) + 1, ncode = (int)getbits(b, 4, lane) + 4;
            if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
            if (nlen > 286 || ndist > 30) { err = 4; break; }
            __syncthreads();
            if (lane < 19) lens[lane] = 0;
            __syncthreads();
            for (int i = 0; i < ncode; ++i) {
                const unsigned v = getbits(b, 3, lane);
                if (lane == 0) lens[kClOrder[i]] = (unsigned char)v;
            }
            __syncthreads();
            if (UNI(construct(0, 0, 19, lane)) != 0) { err = 4; break; }      // (a call's result counts as divergent unless told otherwise -- and a
                                                                                   //  divergent loop exit drags the whole bit-stream state into vector registers)
            // the code lengths of both alphabets, run-length coded: decoded by every lane, kept in LDS behind the 19 entries in use
            unsigned char* cl = lens + 24;
            int idx = 0, prev = 0;
            while (idx < nlen + ndist) {
                const int sym = decode_sym(b, lc, lane);
                if (sym < 0) { err = 1; break; }
                if (sym < 16) {
                    if (lane == 0) cl[idx] = (unsigned char)sym;
                    prev = sym;
                    ++idx;
                    continue;
                }
                int rep, val = 0;
                if (sym == 16) {
                    if (idx == 0) { err = 4; break; }
                    val = prev;
                    rep = 3 + (int)getbits(b, 2, lane);
                } else if (sym == 17) rep = 3 + (int)getbits(b, 3, lane);
                else rep = 11 + (int)getbits(b, 7, lane);
                if (idx + rep > nlen + ndist) { err = 4; break; }
                for (int i = lane; i < rep; i += 64) cl[idx + i] = (unsigned char)val;
                idx += rep;
                prev = val;
            }
            if (err) break;
            if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
            __syncthreads();
            if (UNI(cl[256]) == 0) { err = 4; break; }
            // (cl sits at lens + 24: move it down to lens[0..nlen) and the distance lengths behind it, 8-byte apart from the code-length use)
            unsigned char mine[6];
            const int total = nlen + ndist;
            for (int k = 0; k < 6; ++k) mine[k] = (lane + 64 * k) < total ? cl[lane + 64 * k] : 0;
            __syncthreads();
            for (int k = 0; k < 6; ++k) if ((lane + 64 * k) < total) lens[lane + 64 * k] = mine[k];
            __syncthreads();
            int r = UNI(construct(0, 0, nlen, lane));
            if (r < 0 || (r > 0 && nlen - UNI(lc.count[0]) != 1)) { err = 4; break; }
            r = UNI(construct(1, nlen, ndist, lane));
            if (r < 0 || (r > 0 && ndist - UNI(dc.count[0]) != 1)) { err = 4; break; }
        }
        // literal / length + distance codes of the block
        for (;;) {
            {   // run of literals; leaves the loop on anything else (a length code, a code longer than the table's index, the end of the block,
                // the flush point, the end of the output)
                const int limit = min(raw_len, flushed + FLUSH);
                // Lane l looks up the code that would start l bits further on, so ONE LDS round trip serves every literal that starts within
                // the buffered bits (a refill leaves 33..64 of them, a table hit is <= 11 long): the scalar unit then walks the lanes' entries
                // (readlane at the running bit offset).  Lanes 1..63 write their copy of a literal to a spare byte instead of sitting behind
                // an exec mask that would be set and restored per hop: 11 instructions per literal.  (One lookup per literal: 33.  Collecting
                // the literals by v_writelane -- 8 instructions -- measured the same: the hop is bound by its vector -> scalar -> vector
                // dependency, not by issue.)
                // The lanes past the buffered bits hold a sentinel (bit 8 set, no length, not EMPTY), so the walk has ONE exit test per literal.
                const int spare = WIN + lane;
                constexpr unsigned SENT = EMPTY + 1;
                bool other = false;
                while (pos + 56 <= limit) {              // (a walk consumes at most 64 - 11 + 1 one-bit codes)
                    refill(b, lane);
                    const unsigned short t = g_ltab[(unsigned)(b.buf >> lane) & ((1u << LBITS) - 1)];
                    const int ev = lane <= min(b.cnt - LBITS, 63 - LBITS) ? (int)t : (int)SENT;   // an entry is good while its LBITS index bits are
                                                                                                 // stream bits (and the walk ends on a lane <= 63)
                    unsigned off = 0;
                    unsigned e = (unsigned)__builtin_amdgcn_readlane(ev, 0);
                    while (!(e & 256)) {
                        win[lane == 0 ? (pos & (WIN - 1)) : spare] = (uint8_t)e;
                        ++pos;
                        off += e >> 9;
                        e = (unsigned)__builtin_amdgcn_readlane(ev, off);
                    }
                    b.buf >>= off;
                    b.cnt -= off;
                    if (e != SENT) { other = true; break; }
                }
                while (!other) {           // next to the flush point / the end of the output: one lookup per literal
                    refill(b, lane);
                    const int e = UNI(g_ltab[b.buf & ((1u << LBITS) - 1)]);
                    if ((e & 256) || pos >= limit) break;
                    if (lane == 0) win[pos & (WIN - 1)] = (uint8_t)e;
                    ++pos;
                    b.buf >>= (e >> 9);
                    b.cnt -= (e >> 9);
                }
                if (pos >= limit && pos < raw_len) {        // flush point
                    if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
                    __syncthreads();
                    flush_to(flushed + FLUSH);
                    __syncthreads();
                    continue;
                }
            }
            int sym = decode_sym(b, lc, lane);
            if (sym < 0) { err = 1; break; }
            if (sym < 256) {
                if (pos >= raw_len) { err = 6; break; }
                if (lane == 0) win[pos & (WIN - 1)] = (uint8_t)sym;
                ++pos;
            } else {
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) { err = 5; break; }
                const int len = UNI(s_lbase[sym]) + (int)getbits(b, UNI(s_lext[sym]), lane);
                const int ds = decode_sym(b, dc, lane);
                if (ds < 0) { err = 1; break; }
                if (ds >= 30) { err = 5; break; }
                const int dist = UNI(s_dbase[ds]) + (int)getbits(b, UNI(s_dext[ds]), lane);
                if (b.bad || bits_used(b) > b.nbits) { err = 1; break; }
                if (dist > pos) { err = 5; break; }
                if (pos + len > raw_len) { err = 6; break; }
                // the lanes copy the match inside the window; an overlapping match (dist < len) repeats its first dist bytes
                if (dist > NEAR) {
                    // behind the ring: those bytes left in a flush (pos - dist + len < flushed: FLUSH + 258 + len < NEAR); the fence completes
                    // this wave's stores and drops what its L1 holds of the lines
                    __threadfence();
                    for (int i = lane; i < len; i += 64) win[(pos + i) & (WIN - 1)] = out[pos - dist + i];
                } else if (dist >= len) {
                    for (int i = lane; i < len; i += 64) win[(pos + i) & (WIN - 1)] = win[(pos - dist + i) & (WIN - 1)];
                } else {
                    for (int i = lane; i < len; i += 64) win[(pos + i) & (WIN - 1)] = win[(pos - dist + (i % dist)) & (WIN - 1)];
                }
                pos += len;
            }
            if (pos - flushed >= FLUSH) {
                __syncthreads();
                flush_to(flushed + FLUSH);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (!err) flush_to(pos);
    if (!err && bits_used(b) > b.nbits) err = 1;
    if (!err && pos != raw_len) err = 7;
    if (lane == 0) status[img] = err;
}

// PNG filters, 8-bit RGB (bytes per pixel = 3): raw row = filter type + 3 W bytes.  One wave per frame; the raw row and the two decoded rows
// (previous, current) live in LDS.
constexpr int UNF_MAXROW = 3 * 4096;
__global__ __launch_bounds__(64) void png_unfilter_rgb8_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ out, int B, int H, int W,
                                                               long long raw_stride, int* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t rrow[UNF_MAXROW + 16], rows[2][UNF_MAXROW + 16];
    const int lane = threadIdx.x, img = blockIdx.x;
    const uint8_t* r = raw + (long long)img * raw_stride;
    uint8_t* o = out + (long long)img * H * W * 3;
    const int nb = 3 * W, rowb = 1 + nb;
    int bad = 0;
    for (int y = 0; y < H; ++y) {
        const uint8_t* src = r + (long long)y * rowb;
        uint8_t* cur = rows[y & 1];
        const uint8_t* up = rows[(y & 1) ^ 1];
        const int ft = src[0];
        for (int i = lane; i < nb; i += 64) rrow[i] = src[1 + i];
        __syncthreads();
        if (ft == 0) {
            for (int i = lane; i < nb; i += 64) cur[i] = rrow[i];
        } else if (ft == 2) {
            for (int i = lane; i < nb; i += 64) cur[i] = (uint8_t)(rrow[i] + (y ? up[i] : 0));
        } else if (ft <= 4) {
            if (lane < 3) {
                int left = 0, upleft = 0;
                for (int x = 0; x < W; ++x) {
                    const int i = 3 * x + lane;
                    const int v = rrow[i], a = left, bb = y ? up[i] : 0, cc = upleft;
                    int pred;
                    if (ft == 1) pred = a;
                    else if (ft == 3) pred = (a + bb) >> 1;
                    else {
                        const int p = a + bb - cc;
                        const int pa = abs(p - a), pb = abs(p - bb), pc = abs(p - cc);
                        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : cc);
                    }
                    const int px = (v + pred) & 255;
                    cur[i] = (uint8_t)px;
                    left = px;
                    upleft = bb;
                }
            }
        } else {
            bad = 1;
        }
        __syncthreads();
        uint8_t* dst = o + (long long)y * nb;
        for (int i = lane; i < nb; i += 64) dst[i] = cur[i];
    }
    if (bad && lane == 0) status[img] = 8;     // unknown filter type
}

}  // namespace

// B DEFLATE streams (the zlib payload of each PNG's IDAT chunks, 2-byte zlib header already stripped): stream i = streams[offsets[i] ..
// offsets[i] + lengths[i]) -> raw + i * raw_stride, exactly raw_len = H * (1 + 3 W) bytes each.  status[i] = 0 or a decode error code
// (device array of B int32; the caller checks it after the stream has drained).  Enqueue only.
extern "C" int mt4_png_inflate(const uint8_t* streams, const int64_t* offsets, const int32_t* lengths, uint8_t* raw, int32_t B,
                               int64_t raw_stride, int64_t raw_len, int32_t* status, void* stream) {
    mt4_clear_error();
    if (!streams || !offsets || !lengths || !raw || !status || B <= 0 || raw_len <= 0 || raw_stride < raw_len) return MT4_EINVAL;
    if (raw_len >= 0x7fffff00LL) return MT4_EUNSUPPORTED;
    if ((raw_stride & 15) || ((uintptr_t)raw & 15)) return MT4_EALIGN;
    hipLaunchKernelGGL(png_inflate_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, streams, (const long long*)offsets, (const int*)lengths,
                       raw, B, (long long)raw_stride, (long long)raw_len, status);
    return mt4_check_launch();
}

// byte spans src[src_off[i] .. + len[i]) -> dst[dst_off[i] ..): the IDAT payloads of whole PNG files (uploaded as they lie on disk) packed into
// the contiguous zlib streams mt4_png_inflate reads -- the host neither copies nor touches the compressed bytes.  One workgroup per span.
namespace {
__global__ __launch_bounds__(256) void copy_spans_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const long long* __restrict__ so,
                                                         const long long* __restrict__ dof, const int* __restrict__ len) {
    const uint8_t* s = src + so[blockIdx.x];
    uint8_t* d = dst + dof[blockIdx.x];
    const int l = len[blockIdx.x];
    for (int i = threadIdx.x; i < l; i += 256) d[i] = s[i];
}
}  // namespace

extern "C" int mt4_copy_spans_u8(const uint8_t* src, uint8_t* dst, const int64_t* src_off, const int64_t* dst_off, const int32_t* len, int32_t n,
                                 void* stream) {
    mt4_clear_error();
    if (!src || !dst || !src_off || !dst_off || !len || n <= 0) return MT4_EINVAL;
    hipLaunchKernelGGL(copy_spans_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, src, dst, (const long long*)src_off,
                       (const long long*)dst_off, len);
    return mt4_check_launch();
}

// the filtered scanlines of mt4_png_inflate -> uint8 frames [B][H][W][3] (PNG color type 2, bit depth 8, no interlace)
extern "C" int mt4_png_unfilter_rgb8(const uint8_t* raw, uint8_t* out, int32_t B, int32_t H, int32_t W, int64_t raw_stride, int32_t* status,
                                     void* stream) {
    mt4_clear_error();
    if (!raw || !out || !status || B <= 0 || H <= 0 || W <= 0 || raw_stride < (int64_t)H * (1 + 3 * W)) return MT4_EINVAL;
    if (3 * W > UNF_MAXROW) return MT4_EUNSUPPORTED;
    hipLaunchKernelGGL(png_unfilter_rgb8_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, raw, out, B, H, W, (long long)raw_stride, status);
    return mt4_check_launch();
}
