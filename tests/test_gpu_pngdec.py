"""GPU PNG decode (`mt4_png_inflate` + `mt4_png_unfilter_rgb8`) against Pillow, byte for byte."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _png(arr, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr, "RGB").save(b, format="PNG", **kw)
    return b.getvalue()


def _frames(n, h, w, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":                      # incompressible: literals, stored blocks at level 0
        return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == "flat":                       # long matches, distance 1 runs
        return np.broadcast_to(rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8), (n, h, w, 3)).copy()
    y, x = np.mgrid[0:h, 0:w]                # smooth gradients + a little noise: what video frames compress like (all filter types)
    base = (np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(h + w - 2, 1)], -1)).astype(np.int32)
    return np.clip(base[None] + rng.integers(-6, 7, (n, h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind", ["photo", "noise", "flat"])
@pytest.mark.parametrize("h,w,level", [(64, 96, 6), (37, 53, 9), (480, 854, 6), (8, 8, 1), (33, 40, 0)])
def test_png_decode_matches_pillow(cuda, kind, h, w, level):
    from PIL import Image
    from computervision_codes_amd import pngdec
    n = 5 if h < 400 else 3
    frames = _frames(n, h, w, 7 + h + level, kind)
    files = [_png(frames[i], compress_level=level) for i in range(n)]
    got = pngdec.decode_batch(files, cuda).cpu().numpy()
    for i, f in enumerate(files):
        ref = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
        assert np.array_equal(ref, frames[i])
        assert np.array_equal(got[i], ref), (kind, h, w, level, i)


@pytest.mark.parametrize("w,period", [(1024, 3), (1024, 4), (1024, 10), (700, 15), (854, 2), (854, 3), (853, 3), (2600, 1)])
def test_png_decode_matches_beyond_the_lds_ring(cuda, w, period):
    """random rows repeating every `period` rows: the only matches are `period` scanlines back (6 ... 31 KB), i.e. behind the 8 KB ring the
    decoder keeps in LDS -- those are copied from the frame's own output in memory.  (854 / 853 x 3: 7689 / 7680 bytes back, either side of the limit for a copy inside the ring.)"""
    import zlib
    from PIL import Image
    from computervision_codes_amd import pngdec
    rng = np.random.default_rng(w + period)
    h = 6 * period + 5
    files, frames = [], []
    for k in range(3):
        rows = rng.integers(0, 256, (period, w, 3), dtype=np.uint8)
        fr = np.concatenate([rows] * (h // period + 1))[:h].copy()
        fr[-1, : w // 2] = rng.integers(0, 256, (w // 2, 3), dtype=np.uint8)      # a row that matches only in part
        f = _png(fr, compress_level=9)
        # the encoder did find the far matches: the stream is far smaller than the random rows repeated
        assert len(f) < 0.45 * fr.size if period * (3 * w + 1) < 32768 - 300 else True
        files.append(f); frames.append(fr)
    got = pngdec.decode_batch(files, cuda).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], frames[i]), (w, period, i)


def test_png_decode_optimized_and_many_frames(cuda):
    """`optimize=True` (maximum-effort deflate, dynamic blocks only) and a batch larger than one workgroup of decoder threads"""
    from computervision_codes_amd import pngdec
    frames = _frames(150, 24, 40, 3, "photo")
    files = [_png(frames[i], optimize=True) for i in range(150)]
    got = pngdec.decode_batch(files, cuda).cpu().numpy()
    assert np.array_equal(got, frames)


def test_png_decode_rejects_what_it_does_not_cover(cuda):
    from PIL import Image
    from computervision_codes_amd import pngdec
    b = io.BytesIO()
    Image.fromarray(np.zeros((8, 8), np.uint8), "L").save(b, format="PNG")
    with pytest.raises(pngdec.UnsupportedPng):
        pngdec.decode_batch([b.getvalue()], cuda)
    good = _png(_frames(1, 16, 16, 1, "photo")[0])
    with pytest.raises(RuntimeError):                       # a corrupted stream is reported, not decoded into garbage silently
        bad = bytearray(good)
        i = bad.index(b"IDAT") + 4 + 10
        bad[i] ^= 0xFF
        bad[i + 1] ^= 0x5A
        pngdec.decode_batch([bytes(bad), good], cuda)


def test_png_decode_files_from_disk_equals_decode_batch(cuda, tmp_path):
    """`decode_files`: the files read straight into pinned memory, uploaded whole, IDAT payloads packed on the device (`mt4_copy_spans_u8`) --
    same frames as `decode_batch` on the same bytes (multi-chunk IDAT: frames large enough for several 64 KB chunks); mixed sizes are refused"""
    from computervision_codes_amd import pngdec
    frames = _frames(19, 200, 320, 3, "photo")
    paths = []
    for i in range(19):
        f = _png(frames[i], compress_level=(1, 6, 9)[i % 3])
        p = tmp_path / f"{i:06d}.png"
        p.write_bytes(f)
        paths.append(str(p))
    assert len(pngdec._idat_spans(open(paths[0], "rb").read())[2]) > 1
    a = pngdec.decode_files(paths, cuda, workers=4)
    b = pngdec.decode_batch([open(p, "rb").read() for p in paths], cuda)
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), frames)
    odd = tmp_path / "odd.png"
    odd.write_bytes(_png(_frames(1, 64, 96, 5, "photo")[0]))
    with pytest.raises(pngdec.MixedSizes):
        pngdec.decode_files(paths[:3] + [str(odd)], cuda)


def test_corrupt_stream_last_in_the_batch_is_reported_without_reading_past_the_blob(cuda):
    """A frame whose PNG chunks are intact but whose DEFLATE body is damaged, placed LAST in the batch (behind it only the 1 KB pad): the
    decoder must report it (`DecodeError`, the caller's signal to hand the batch to Pillow) -- it stops taking input at the stream's end
    (`load_chunk`) instead of running its literal loops megabytes past the blob.  Several damage patterns, incl. a stream cut to a fraction of
    its length with the chunk header rewritten to match (the symbols keep coming from beyond the end)."""
    import struct
    import zlib
    from computervision_codes_amd import pngdec
    frames = _frames(3, 120, 160, 5, "photo")
    good = [_png(f) for f in frames]

    def rebuild(f, mutate):
        """the file with its (single) IDAT payload replaced by mutate(payload), length and CRC fields consistent"""
        i = f.index(b"IDAT") - 4
        n = struct.unpack(">I", f[i:i + 4])[0]
        payload = mutate(bytes(f[i + 8:i + 8 + n]))
        chunk = b"IDAT" + payload
        return f[:i] + struct.pack(">I", len(payload)) + chunk + struct.pack(">I", zlib.crc32(chunk)) + f[i + 12 + n:]

    def flip(p):
        q = bytearray(p)
        for k in range(40, len(q), 97):
            q[k] ^= 0xA5
        return bytes(q)
    for mutate in (flip, lambda p: p[:len(p) // 3], lambda p: p[:64]):
        bad = rebuild(good[2], mutate)
        assert pngdec._idat_spans(bad)[:2] == (160, 120)                 # the chunk walk finds nothing wrong
        with pytest.raises(pngdec.DecodeError):
            pngdec.decode_batch([good[0], good[1], bad], cuda)
    out = pngdec.decode_batch(good, cuda)                                # and the decoder is fine afterwards
    assert np.array_equal(out.cpu().numpy(), np.stack(frames))
