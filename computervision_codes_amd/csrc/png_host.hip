// Host side of the PNG input path (no device work): read a load's files into ONE staging buffer with native threads and walk their chunk lists.
//
// The reference decodes every frame with PIL inside three DataLoader worker PROCESSES (`Spatial_cnn/dataloader.py:257-261`,
// `Spatial_cnn/test.py:240-241`).  With inflate + unfiltering on the GPU (png_kernels.hip) what is left on the host is open / read / a walk over
// the chunk headers -- ~60 us of interpreter time per file when done from Python threads, all of it under the GIL: 2048 files cost 0.12-0.15 s
// however many threads ran, which capped the extraction loop at ~5 k frames/s (profiles/r02_e2e_decode.txt, BENCH_r03 e2e_script).  Here the
// same steps run on `threads` native threads; the ctypes call releases the GIL for its whole duration.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "mt4_common.h"

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

template <typename F>
void parallel_for(int n, int threads, F&& fn) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                const int i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= n) return;
                fn(i);
            }
        });
    for (auto& th : pool) th.join();
}

}  // namespace

// sizes[i] = size in bytes of paths[i] (-1: cannot stat)
extern "C" int mt4_png_stat_files(const char* const* paths, int32_t n, int64_t* sizes, int32_t threads) {
    if (!paths || !sizes || n < 0) return MT4_EINVAL;
    parallel_for(n, threads, [&](int i) {
        struct stat st;
        sizes[i] = (paths[i] && ::stat(paths[i], &st) == 0) ? (int64_t)st.st_size : -1;
    });
    return MT4_OK;
}

// File i is read to dst + file_off[i] (sizes[i] bytes, from mt4_png_stat_files) and its chunk list walked (PNG specification 5.3):
//   status[i]  0 ok | 1 cannot open / short read | 2 not a PNG / truncated chunk | 3 not 8-bit RGB non-interlaced (IHDR) | 4 no IHDR / IDAT |
//              5 more than max_spans IDAT chunks | 6 bad zlib header (RFC 1950: CM = 8, no preset dictionary)
//   width[i], height[i]; nspans[i]; span_off[i * max_spans + s] = byte offset of IDAT payload s INSIDE the file, span_len[...] its length
//   (empty IDAT chunks are skipped).  The zlib header's two bytes are still part of the first span: the caller skips them.
// Returns MT4_OK when the call itself ran (per-file results are in status), MT4_EINVAL on bad arguments.
extern "C" int mt4_png_read_files(const char* const* paths, const int64_t* sizes, const int64_t* file_off, int32_t n, uint8_t* dst, int32_t* width,
                                  int32_t* height, int64_t* span_off, int32_t* span_len, int32_t* nspans, int32_t max_spans, int32_t* status,
                                  int32_t threads) {
    if (!paths || !sizes || !file_off || !dst || !width || !height || !span_off || !span_len || !nspans || !status || n < 0 || max_spans <= 0)
        return MT4_EINVAL;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    parallel_for(n, threads, [&](int i) {
        width[i] = height[i] = nspans[i] = 0;
        status[i] = 1;
        const int64_t size = sizes[i];
        if (size < 0 || !paths[i]) return;
        uint8_t* buf = dst + file_off[i];
        const int fd = ::open(paths[i], O_RDONLY | O_CLOEXEC);
        if (fd < 0) return;
        int64_t got = 0;
        while (got < size) {
            const ssize_t k = ::read(fd, buf + got, (size_t)(size - got));
            if (k <= 0) break;
            got += k;
        }
        ::close(fd);
        if (got != size) return;
        status[i] = 2;
        if (size < 33 || std::memcmp(buf, sig, 8) != 0) return;
        int64_t pos = 8;
        int ns = 0;
        bool have_ihdr = false, overflow = false;
        while (pos + 8 <= size) {
            const int64_t len = be32(buf + pos);
            const uint8_t* typ = buf + pos + 4;
            if (pos + 12 + len > size) return;                       // truncated chunk
            if (std::memcmp(typ, "IHDR", 4) == 0) {
                if (len < 13) return;
                const uint8_t* b = buf + pos + 8;
                width[i] = (int32_t)be32(b);
                height[i] = (int32_t)be32(b + 4);
                if (b[8] != 8 || b[9] != 2 || b[10] != 0 || b[11] != 0 || b[12] != 0 || width[i] <= 0 || height[i] <= 0) { status[i] = 3; return; }
                have_ihdr = true;
            } else if (std::memcmp(typ, "IDAT", 4) == 0) {
                if (len > 0) {
                    if (ns < max_spans) {
                        span_off[(int64_t)i * max_spans + ns] = pos + 8;
                        span_len[(int64_t)i * max_spans + ns] = (int32_t)len;
                        ++ns;
                    } else {
                        overflow = true;
                    }
                }
            } else if (std::memcmp(typ, "IEND", 4) == 0) {
                break;
            }
            pos += 12 + len;
        }
        if (!have_ihdr || ns == 0) { status[i] = 4; return; }
        if (overflow) { status[i] = 5; return; }
        nspans[i] = ns;
        const uint8_t* z = buf + span_off[(int64_t)i * max_spans];
        // (a first IDAT shorter than 2 bytes would split the zlib header over chunks: legal, never written by an encoder; left to the host path)
        if (span_len[(int64_t)i * max_spans] < 2 || (z[0] & 0x0F) != 8 || (((unsigned)z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) { status[i] = 6; return; }
        status[i] = 0;
    });
    return MT4_OK;
}
