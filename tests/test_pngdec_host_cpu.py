"""host side of the PNG decoder (`pngdec._idat_spans` / `parse_png`): chunk walk and format checks, no GPU"""
import io
import zlib

import numpy as np
import pytest


def _load():
    # (pngdec imports the HIP library at module import: the parser itself is pure Python)
    from computervision_codes_amd import pngdec
    return pngdec


def _png(arr, mode="RGB", **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr, mode).save(b, format="PNG", **kw)
    return b.getvalue()


def test_chunk_walk_recovers_the_zlib_stream():
    pngdec = _load()
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    f = _png(arr, compress_level=6)
    w, h, z = pngdec.parse_png(f)
    assert (w, h) == (23, 19)
    raw = zlib.decompressobj(wbits=-15).decompress(z)            # the 2-byte header is stripped: a raw DEFLATE stream (+ Adler-32 behind it)
    assert len(raw) == h * (1 + 3 * w)
    w2, h2, spans = pngdec._idat_spans(f)
    assert (w2, h2) == (w, h) and b"".join(f[o:o + n] for o, n in spans)[2:] == z


def test_unsupported_files_are_refused():
    pngdec = _load()
    with pytest.raises(pngdec.UnsupportedPng):
        pngdec.parse_png(b"not a png at all, just bytes" * 4)
    with pytest.raises(pngdec.UnsupportedPng):
        pngdec.parse_png(_png(np.zeros((4, 4), np.uint8), "L"))
    with pytest.raises(pngdec.UnsupportedPng):
        pngdec.parse_png(_png(np.zeros((4, 4, 4), np.uint8), "RGBA"))
    good = _png(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(pngdec.UnsupportedPng):
        pngdec.parse_png(good[:40])
