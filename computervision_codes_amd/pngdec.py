"""PNG decode on the GPU.  The reference decodes every frame with PIL inside its DataLoader workers
(`Spatial_cnn/dataloader.py:257-261`, three worker processes, `Spatial_cnn/test.py:240-241`); here the host only walks the chunk
list of each file -- signature, IHDR, the concatenated IDAT payload -- and the inflate + scanline unfiltering run on the device
(`mt4_png_inflate`: one thread per frame, `mt4_png_unfilter_rgb8`).  8-bit RGB, non-interlaced files (what CholecT45 ships);
anything else raises `UnsupportedPng` and the caller keeps the PIL path for that file."""
from __future__ import annotations

import struct
import threading
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._lib import check, lib

_SIG = b"\x89PNG\r\n\x1a\n"


class UnsupportedPng(ValueError):
    pass


class DecodeError(RuntimeError):
    """the device decoder reported a bad stream (status codes of mt4_png_inflate / mt4_png_unfilter_rgb8)"""


class MixedSizes(UnsupportedPng):
    """the files of one call do not share a frame size (the caller groups them)"""


def parse_png(data: bytes) -> Tuple[int, int, bytes]:
    """(width, height, DEFLATE stream) of an 8-bit RGB, non-interlaced PNG: chunk walk only (PNG spec 5.3), the zlib header
    (RFC 1950: CM = 8, no preset dictionary) stripped, the Adler-32 trailer left in place behind the last block"""
    if len(data) < 33 or data[:8] != _SIG:
        raise UnsupportedPng("not a PNG file")
    pos = 8
    width = height = None
    idat: List[bytes] = []
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n:
            raise UnsupportedPng("truncated chunk")
        if typ == b"IHDR":
            width, height, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", body)
            if depth != 8 or ctype != 2 or comp != 0 or filt != 0 or inter != 0:
                raise UnsupportedPng(f"bit depth {depth}, colour type {ctype}, interlace {inter}: only 8-bit RGB, non-interlaced")
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    if width is None or not idat:
        raise UnsupportedPng("no IHDR / IDAT")
    z = b"".join(idat)
    if len(z) < 6 or (z[0] & 0x0F) != 8 or ((z[0] << 8) | z[1]) % 31 != 0 or (z[1] & 0x20):
        raise UnsupportedPng("bad zlib header")
    return width, height, z[2:]


MAX_WIDTH = 4096                 # `mt4_png_unfilter_rgb8` keeps three rows of 3 W bytes in LDS (csrc/png_kernels.hip: UNF_MAXROW)
MAX_RAW_BYTES = (1 << 31) - 256  # raw scanline bytes of one frame the kernels index with 32 bits
MAX_STREAM_BYTES = 1 << 28       # compressed bytes of one frame: its bit count must stay below 2^31


def _idat_spans(data: bytes) -> Tuple[int, int, List[Tuple[int, int]]]:
    """(width, height, [(offset, length) of every IDAT payload]) -- `parse_png` without copying the payloads.  Everything the device decoder
    cannot take is an `UnsupportedPng` HERE, before any launch or allocation sized by the header: malformed chunk headers, sizes beyond the
    kernels' limits, streams too long for their 32-bit bit counter."""
    if len(data) < 33 or data[:8] != _SIG:
        raise UnsupportedPng("not a PNG file")
    pos = 8
    width = height = None
    spans: List[Tuple[int, int]] = []
    while pos + 8 <= len(data):
        n, typ = struct.unpack_from(">I4s", data, pos)
        if pos + 12 + n > len(data):
            raise UnsupportedPng("truncated chunk")
        if typ == b"IHDR":
            if n < 13:
                raise UnsupportedPng("IHDR shorter than 13 bytes")
            width, height, depth, ctype, comp, filt, inter = struct.unpack_from(">IIBBBBB", data, pos + 8)
            if depth != 8 or ctype != 2 or comp != 0 or filt != 0 or inter != 0:
                raise UnsupportedPng(f"bit depth {depth}, colour type {ctype}, interlace {inter}: only 8-bit RGB, non-interlaced")
            if not (0 < width <= MAX_WIDTH) or height <= 0 or height * (1 + 3 * width) > MAX_RAW_BYTES:
                raise UnsupportedPng(f"frame {width} x {height}: the device decoder takes widths up to {MAX_WIDTH} and < 2 GiB of scanlines")
        elif typ == b"IDAT":
            if n:
                spans.append((pos + 8, n))
        elif typ == b"IEND":
            break
        pos += 12 + n
    if width is None or not spans:
        raise UnsupportedPng("no IHDR / IDAT")
    if sum(l for _, l in spans) >= MAX_STREAM_BYTES:
        raise UnsupportedPng("zlib stream of 256 MB or more")
    return width, height, spans


_TLS = threading.local()


def _staging(nbytes: int) -> torch.Tensor:
    """pinned host buffer for the compressed bytes, kept between calls (page-locking 0.7 GB per call cost as much as the decode); one per
    calling thread -- the extraction driver gathers the next spans on helper threads while an earlier one is still uploading"""
    buf = getattr(_TLS, "pinned", None)
    if buf is None or buf.numel() < nbytes:
        buf = _TLS.pinned = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8).pin_memory()
    return buf[:nbytes]


def decode_batch(files: Sequence[bytes], device="cuda", timings: dict = None, workers: int = 8) -> torch.Tensor:
    """PNG files (bytes) of ONE frame size -> uint8 [N,H,W,3] on the device, equal to `np.asarray(PIL.Image.open(f).convert('RGB'))`.
    Raises `UnsupportedPng` before any launch if a file is not 8-bit RGB / non-interlaced or the sizes differ, `RuntimeError` if a
    stream is corrupt.  The IDAT payloads are gathered straight into one pinned buffer (one copy on the host, `workers` threads)."""
    if len(files) == 0:
        raise UnsupportedPng("empty batch")
    metas = [_idat_spans(f) for f in files]
    w, h = metas[0][0], metas[0][1]
    if any((m[0], m[1]) != (w, h) for m in metas):
        raise MixedSizes("frames of different sizes in one batch")
    n = len(metas)
    lengths = np.array([sum(l for _, l in m[2]) - 2 for m in metas], dtype=np.int32)      # without the 2-byte zlib header
    if int(lengths.min()) < 4:
        raise UnsupportedPng("empty zlib stream")
    offsets = np.zeros(n, dtype=np.int64)
    offsets[1:] = np.cumsum(lengths[:-1].astype(np.int64))
    total = int(offsets[-1] + lengths[-1])
    blob = _staging(total + 1024)                  # (+ 1 KB: the decoder prefetches the stream in 512-byte pieces)
    dst = blob.numpy()

    def gather(i):
        f, m = files[i], metas[i]
        o, skip = int(offsets[i]), 2
        mv = memoryview(f)
        first = mv[m[2][0][0]:m[2][0][0] + 2] if m[2][0][1] >= 2 else None
        if first is None or (first[0] & 0x0F) != 8 or ((first[0] << 8) | first[1]) % 31 != 0 or (first[1] & 0x20):
            raise UnsupportedPng("bad zlib header")
        for (so, sl) in m[2]:
            if skip:
                so, sl, skip = so + skip, sl - skip, 0
            dst[o:o + sl] = np.frombuffer(mv[so:so + sl], dtype=np.uint8)
            o += sl

    if workers > 1 and n >= 4 * workers:        # (the copies release the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(gather, range(n)))
    else:
        for i in range(n):
            gather(i)
    dev = torch.device(device)
    streams = blob.to(dev, non_blocking=True)
    offs = torch.from_numpy(offsets).to(dev)
    lens = torch.from_numpy(lengths).to(dev)
    return _decode_streams(streams, offs, lens, n, h, w, dev, timings)


def _decode_streams(streams, offs, lens, n, h, w, dev, timings=None) -> torch.Tensor:
    """inflate + unfilter of n zlib streams that lie in `streams` (device) at `offs` / `lens` (device arrays)"""
    raw_len = h * (1 + 3 * w)
    raw_stride = (raw_len + 15) // 16 * 16
    raw = torch.empty((n, raw_stride), dtype=torch.uint8, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    s = ops._stream()
    if timings is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
    check(lib.mt4_png_inflate(streams.data_ptr(), offs.data_ptr(), lens.data_ptr(), raw.data_ptr(), n, raw_stride, raw_len, status.data_ptr(), s),
          "mt4_png_inflate")
    if timings is not None:
        ev[1].record()
    check(lib.mt4_png_unfilter_rgb8(raw.data_ptr(), out.data_ptr(), n, h, w, raw_stride, status.data_ptr(), s), "mt4_png_unfilter_rgb8")
    if timings is not None:
        ev[2].record()
    st = status.cpu()
    if timings is not None:
        timings["inflate_ms"], timings["unfilter_ms"] = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    if int(st.abs().max()) != 0:
        bad = int(torch.nonzero(st)[0])
        raise DecodeError(f"PNG decode failed: frame {bad} of the batch, code {int(st[bad])} (include/mt4hip.h: mt4_png_inflate)")
    return out


_READ_STATUS = {1: "cannot open / short read", 2: "not a PNG file / truncated chunk", 3: "only 8-bit RGB, non-interlaced", 4: "no IHDR / IDAT",
                5: "too many IDAT chunks", 6: "bad zlib header"}


def read_files(paths: Sequence[str], workers: int = 8):
    """The host part of `decode_files`, on `workers` NATIVE threads outside the interpreter lock (`mt4_png_stat_files` / `mt4_png_read_files`,
    csrc/png_host.hip): the files are read into this thread's pinned staging buffer as they lie on disk and their chunk lists walked.
    -> (blob pinned uint8 [total], w, h, src, dst, ln, offsets, lengths): span i of the upload goes from blob[src[i] .. + ln[i]) to
    streams[dst[i] ..); zlib stream f = streams[offsets[f] .. + lengths[f]) (2-byte zlib header stripped).  Raises `UnsupportedPng` /
    `MixedSizes` before anything is launched."""
    import ctypes as C
    import os
    n = len(paths)
    if n == 0:
        raise UnsupportedPng("empty batch")
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    sizes = np.empty(n, dtype=np.int64)
    check(lib.mt4_png_stat_files(arr, n, sizes.ctypes.data, workers), "mt4_png_stat_files")
    if int(sizes.min()) < 0:
        raise FileNotFoundError(paths[int(np.argmin(sizes))])
    foff = np.zeros(n, dtype=np.int64)
    foff[1:] = np.cumsum((sizes[:-1] + 15) // 16 * 16)
    total = int(foff[-1] + sizes[-1])
    blob = _staging(total)
    max_spans = int(min(4096, int(sizes.max()) // 4096 + 8))          # (libpng's default IDAT size is 8 KB, Pillow's 64 KB)
    width, height, nspans, status = (np.empty(n, dtype=np.int32) for _ in range(4))
    span_off = np.empty((n, max_spans), dtype=np.int64)
    span_len = np.empty((n, max_spans), dtype=np.int32)
    check(lib.mt4_png_read_files(arr, sizes.ctypes.data, foff.ctypes.data, n, blob.data_ptr(), width.ctypes.data, height.ctypes.data,
                                 span_off.ctypes.data, span_len.ctypes.data, nspans.ctypes.data, max_spans, status.ctypes.data, workers),
          "mt4_png_read_files")
    if int(status.max()) != 0:
        bad = int(np.flatnonzero(status)[0])
        raise UnsupportedPng(f"{paths[bad]}: {_READ_STATUS.get(int(status[bad]), status[bad])}")
    w, h = int(width[0]), int(height[0])
    if bool((width != w).any()) or bool((height != h).any()):
        raise MixedSizes("frames of different sizes in one batch")
    if not (0 < w <= MAX_WIDTH) or h * (1 + 3 * w) > MAX_RAW_BYTES:
        raise UnsupportedPng(f"frame {w} x {h}: the device decoder takes widths up to {MAX_WIDTH} and < 2 GiB of scanlines")
    live = np.arange(max_spans)[None, :] < nspans[:, None]
    sl = np.where(live, span_len, 0).astype(np.int64)
    so = span_off + foff[:, None]
    so[:, 0] += 2                                                       # the zlib header (checked by the reader)
    sl[:, 0] -= 2
    lengths64 = sl.sum(axis=1)
    if int(lengths64.min()) < 4:
        raise UnsupportedPng("empty zlib stream")
    if int(lengths64.max()) >= MAX_STREAM_BYTES:
        raise UnsupportedPng("zlib stream of 256 MB or more")
    offsets = np.zeros(n, dtype=np.int64)
    offsets[1:] = np.cumsum(lengths64[:-1])
    do = offsets[:, None] + np.cumsum(sl, axis=1) - sl                  # destination of every span inside its stream
    keep = live & (sl > 0)
    return blob, w, h, so[keep], do[keep], sl[keep].astype(np.int32), offsets, lengths64.astype(np.int32)


def decode_files(paths: Sequence[str], device="cuda", timings: dict = None, workers: int = 8) -> torch.Tensor:
    """PNG files on disk (ONE frame size) -> uint8 [N,H,W,3] on the device, as `decode_batch`, without a host copy of the compressed bytes:
    the files are read straight into one pinned buffer (`read_files`: native threads), uploaded as they lie on disk, and their IDAT
    payloads are packed into contiguous zlib streams on the device (`mt4_copy_spans_u8`).  Everything is enqueued on the CURRENT stream; the
    call ends with one blocking read of the decoder's status words (so the staging buffer is free again when it returns)."""
    blob, w, h, src, dst, ln, offsets, lengths = read_files(paths, workers)
    n = len(paths)
    dev = torch.device(device)
    files_dev = blob.to(dev, non_blocking=True)
    meta = np.concatenate([src, dst, offsets]).astype(np.int64)        # one upload for the three int64 arrays, one for the two int32 ones
    meta_dev = torch.from_numpy(meta).to(dev, non_blocking=True)
    m32_dev = torch.from_numpy(np.concatenate([ln, lengths])).to(dev, non_blocking=True)
    ns = len(ln)
    streams = torch.empty(int(offsets[-1]) + int(lengths[-1]) + 1024, dtype=torch.uint8, device=dev)     # (+ 1 KB: the decoder prefetches 512-byte pieces)
    streams[-1024:].zero_()
    check(lib.mt4_copy_spans_u8(files_dev.data_ptr(), streams.data_ptr(), meta_dev[:ns].data_ptr(), meta_dev[ns:2 * ns].data_ptr(), m32_dev[:ns].data_ptr(), ns,
                                ops._stream()), "mt4_copy_spans_u8")
    return _decode_streams(streams, meta_dev[2 * ns:], m32_dev[ns:], n, h, w, dev, timings)
