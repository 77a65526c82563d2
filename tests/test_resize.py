"""PIL-exact uint8 resize (`transforms.Resize((256,448))` on the decoded PNG, `Spatial_cnn/dataloader.py:155-159`): the host
coefficient tables against Pillow itself on the CPU (a numpy emulation of the kernel's integer arithmetic), and the HIP pass on the
GPU, both byte-identical to `PIL.Image.resize(size, BILINEAR)`."""
import numpy as np
import pytest
import torch
from PIL import Image

from computervision_codes_amd.ops import pil_resize_tables

CASES = [(480, 854, 256, 448), (97, 131, 256, 448), (300, 300, 224, 224), (64, 96, 64, 200), (500, 333, 100, 77), (256, 448, 384, 384)]


def _pil(img, oh, ow):
    return np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))


def _emulate(img, oh, ow):
    x = img
    for axis, (n_in, n_out) in enumerate(((img.shape[1], ow), (img.shape[0], oh))):
        if n_in == n_out:
            continue
        bd, kk = pil_resize_tables(n_in, n_out)
        src = x.astype(np.int64)
        shape = (x.shape[0], n_out, x.shape[2]) if axis == 0 else (n_out, x.shape[1], x.shape[2])
        out = np.zeros(shape, np.int64)
        for o in range(n_out):
            lo, n = bd[o]
            if axis == 0:
                out[:, o] = (1 << 21) + (src[:, lo:lo + n] * kk[o, :n, None].astype(np.int64)[None]).sum(1)
            else:
                out[o] = (1 << 21) + (src[lo:lo + n] * kk[o, :n, None, None].astype(np.int64)).sum(0)
        x = np.clip(out >> 22, 0, 255).astype(np.uint8)
    return x


@pytest.mark.parametrize("h,w,oh,ow", CASES)
def test_tables_reproduce_pillow_bytes(h, w, oh, ow):
    img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(_emulate(img, oh, ow), _pil(img, oh, ow))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,oh,ow", CASES)
def test_hip_resize_is_byte_identical_to_pillow(cuda, h, w, oh, ow):
    from computervision_codes_amd import ops
    rng = np.random.default_rng(h + w)
    imgs = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    imgs[1] = np.clip(np.add.outer(np.arange(h), np.arange(w))[..., None] % 256 + np.arange(3), 0, 255)   # smooth ramps
    imgs[2, ::2] = 255                                                                                          # extremes: clip8 on both ends
    imgs[2, 1::2] = 0
    got = ops.resize_bilinear_u8(torch.from_numpy(imgs).to(cuda), oh, ow).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], _pil(imgs[i], oh, ow)), i


@pytest.mark.gpu
def test_device_frame_loader_equals_host_loader(cuda, tmp_path):
    """`cholect.load_frames_device` (PNG decode on the host, Resize on the GPU) returns the bytes of `load_frames_u8` (all Pillow),
    also for a video that mixes native sizes and one already at the target size"""
    import os
    from computervision_codes_amd import cholect
    d = tmp_path / "data" / "VID01"
    os.makedirs(d)
    rng = np.random.default_rng(4)
    sizes = [(120, 214), (120, 214), (64, 96), (200, 150), (120, 214)]
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"{i:06d}.png")
    ids = list(range(len(sizes)))
    ref = cholect.load_frames_u8(str(tmp_path), "VID01", ids, 64, 96)
    got = cholect.load_frames_device(str(tmp_path), "VID01", ids, 64, 96, device=cuda)
    assert got.dtype == torch.uint8 and tuple(got.shape) == ref.shape and np.array_equal(got.cpu().numpy(), ref)
    dev = cholect.load_frames_device(str(tmp_path), "VID01", ids, 64, 96, device=cuda, decode="device")      # inflate + unfilter on the GPU too
    assert dev.dtype == torch.uint8 and np.array_equal(dev.cpu().numpy(), ref)
    Image.fromarray(rng.integers(0, 256, (64, 96), dtype=np.uint8), "L").save(d / "000005.png")     # a grey-scale file: not covered by the device
    ids6 = ids + [5]                                                                                    # decoder -> that batch falls back to Pillow
    ref6 = cholect.load_frames_u8(str(tmp_path), "VID01", ids6, 64, 96)
    dev6 = cholect.load_frames_device(str(tmp_path), "VID01", ids6, 64, 96, device=cuda, decode="device")
    assert np.array_equal(dev6.cpu().numpy(), ref6)
    # a stream the device decoder reports as bad is handed to Pillow as well (here: the decoder is made to say so for good files)
    from computervision_codes_amd import pngdec

    def refuse(*a, **k):
        raise pngdec.DecodeError("PNG decode failed: frame 0 of the batch, code 5")
    keep = pngdec.decode_files, pngdec.decode_batch
    pngdec.decode_files = pngdec.decode_batch = refuse
    try:
        again = cholect.load_frames_device(str(tmp_path), "VID01", ids, 64, 96, device=cuda, decode="device")
    finally:
        pngdec.decode_files, pngdec.decode_batch = keep
    assert np.array_equal(again.cpu().numpy(), ref)
