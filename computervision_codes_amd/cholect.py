"""CholecT45/T50 on-disk layout (SURVEY 8(b) "Labels"): `<data_dir>/data/VIDnn/%06d.png`,
`<data_dir>/{triplet,instrument,verb,target}/VIDnn.txt` (CSV int: col 0 frame id, cols 1.. multi-hot).
Split tables as in `Spatial_cnn/dataloader.py:112-148`; the extraction variant lists ALL videos
(`dataloader_test.py:87-88`).  CPU side only (PNG decode + resize): the normalisation runs on the GPU."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

# official cross-validation folds / fixed splits (dataset metadata)
_SPLITS = {
    "cholect45-crossval": {1: [79, 2, 51, 6, 25, 14, 66, 23, 50], 2: [80, 32, 5, 15, 40, 47, 26, 48, 70],
                           3: [31, 57, 36, 18, 52, 68, 10, 8, 73], 4: [42, 29, 60, 27, 65, 75, 22, 49, 12],
                           5: [78, 43, 62, 35, 74, 1, 56, 4, 13]},
    "cholect50-crossval": {1: [79, 2, 51, 6, 25, 14, 66, 23, 50, 111], 2: [80, 32, 5, 15, 40, 47, 26, 48, 70, 96],
                           3: [31, 57, 36, 18, 52, 68, 10, 8, 73, 103], 4: [42, 29, 60, 27, 65, 75, 22, 49, 12, 110],
                           5: [78, 43, 62, 35, 74, 1, 56, 4, 13, 92]},
    "cholect50": {"train": [1, 15, 26, 40, 52, 65, 79, 2, 18, 27, 43, 56, 66, 92, 4, 22, 31, 47, 57, 68, 96, 5, 23, 35, 48, 60, 70,
                            103, 13, 25, 36, 49, 62, 75, 110], "val": [8, 12, 29, 50, 78], "test": [6, 51, 10, 73, 14, 74, 32, 80, 42, 111]},
    "cholect50-challenge": {"train": [1, 15, 26, 40, 52, 79, 2, 27, 43, 56, 66, 4, 22, 31, 47, 57, 68, 23, 35, 48, 60, 70, 13, 25, 49, 62,
                                      75, 8, 12, 29, 50, 78, 6, 51, 10, 73, 14, 32, 80, 42], "val": [5, 18, 36, 65, 74],
                            "test": [92, 96, 103, 110, 111]},
    "cholect45-challenge": {"train": [1, 15, 26, 40, 52, 79, 2, 27, 43, 56, 66, 4, 22, 31, 47, 57, 5, 23, 35, 48, 60, 18, 13, 25, 49, 62,
                                      65, 8, 12, 29, 50, 78, 6, 51, 10, 36, 14, 32, 80, 42], "val": [68, 70, 73, 74, 75],
                            "test": [92, 96, 103, 110, 111]},
}
_SPLITS["cholect45"] = _SPLITS["cholect45-crossval"]


def video_name(v: int) -> str:
    return "VID{}".format(str(v).zfill(2))


def split_videos(variant: str, test_fold: int) -> Tuple[List[str], List[str], List[str]]:
    """(train, val, test) video names; crossval: val = last 5 train videos (`dataloader.py:77-84`)."""
    if variant not in _SPLITS:
        raise ValueError(f"{variant} is not a valid dataset variant")
    sp = _SPLITS[variant]
    if "crossval" in variant or variant == "cholect45":
        train = [v for k in sp if k != test_fold for v in sp[k]]
        test = list(sp[test_fold])
        train, val = train[:-5], train[-5:]
    else:
        train, val, test = list(sp["train"]), list(sp["val"]), list(sp["test"])
    return [video_name(v) for v in train], [video_name(v) for v in val], [video_name(v) for v in test]


def extraction_videos(variant: str, test_fold: int) -> List[str]:
    """all videos in the order train + test + val (`dataloader_test.py:88`)"""
    tr, va, te = split_videos(variant, test_fold)
    return tr + te + va


def load_labels(data_dir: str, video: str) -> Dict[str, np.ndarray]:
    out = {}
    for key, sub in (("ivt", "triplet"), ("i", "instrument"), ("v", "verb"), ("t", "target")):
        a = np.loadtxt(os.path.join(data_dir, sub, f"{video}.txt"), dtype=np.int64, delimiter=",", ndmin=2)
        out[key] = a
    return out


_WARNED_PNG = False


def load_frames_u8(data_dir: str, video: str, frame_ids, height: int = 256, width: int = 448) -> np.ndarray:
    """decode + `Resize((height,width))` (bilinear, as torchvision does on PIL images; `dataloader.py:155-159`) -> uint8 [N,H,W,3]"""
    from PIL import Image
    out = np.empty((len(frame_ids), height, width, 3), np.uint8)
    for i, fid in enumerate(frame_ids):
        with Image.open(os.path.join(data_dir, "data", video, "{}.png".format(str(int(fid)).zfill(6)))) as im:
            im = im.convert("RGB")
            if im.size != (width, height):
                im = im.resize((width, height), Image.BILINEAR)
            out[i] = np.asarray(im)
    return out


def load_frames_device(data_dir: str, video: str, frame_ids, height: int = 256, width: int = 448, device="cuda", workers: int = 0,
                       decode: str = "host"):
    """Same bytes as `load_frames_u8`, as a uint8 [N,H,W,3] tensor on the GPU: the PNGs are decoded on the host at their native size
    and `Resize((height,width))` runs on the device (`ops.resize_bilinear_u8`, byte-identical to Pillow's bilinear resize).  Frames
    of one native size go through one launch pair; a video mixing sizes falls back to one group per size.  workers > 1: the PNGs
    are decoded by that many threads (Pillow releases the GIL while it inflates; the reference's loader has 3 worker processes,
    `Spatial_cnn/test.py:240-241`)."""
    import struct

    import torch
    from PIL import Image

    from . import _lib, ops

    if len(frame_ids) == 0:
        return torch.empty((0, height, width, 3), dtype=torch.uint8, device=device)
    if decode == "device":
        # decode = "device": the files are only read; inflate + unfiltering run on the GPU (`pngdec.decode_batch`, 8-bit RGB non-interlaced PNGs --
        # what the dataset ships; anything else raises `pngdec.UnsupportedPng`).  Same bytes as the Pillow path.
        from . import pngdec
        paths = [os.path.join(data_dir, "data", video, "{}.png".format(str(int(fid)).zfill(6))) for fid in frame_ids]
        try:
            try:      # one frame size (a video): the files go to the device as they lie on disk, no host copy of the compressed bytes
                x = pngdec.decode_files(paths, device, workers=max(8, workers))
                return x if tuple(x.shape[1:3]) == (height, width) else ops.resize_bilinear_u8(x, height, width)
            except pngdec.MixedSizes:
                pass
            files = []
            for p in paths:
                with open(p, "rb") as fh:
                    files.append(fh.read())
            sizes: Dict[tuple, List[int]] = {}
            for i, f in enumerate(files):
                w0, h0, _ = pngdec._idat_spans(f)
                sizes.setdefault((h0, w0), []).append(i)
            out = torch.empty((len(files), height, width, 3), dtype=torch.uint8, device=device)
            for (h0, w0), idx in sizes.items():
                x = pngdec.decode_batch([files[i] for i in idx], device, workers=max(8, workers))
                y = x if (h0, w0) == (height, width) else ops.resize_bilinear_u8(x, height, width)
                out[torch.tensor(idx, device=device)] = y
            return out
        except (pngdec.UnsupportedPng, pngdec.DecodeError, _lib.Mt4Error, struct.error) as e:
            # a PNG flavour the device decoder does not cover (palette, grey, alpha, 16-bit, interlaced, wider than 4096), a malformed chunk
            # list, a stream it reports as bad, or a launch the library refuses (Pillow then has the last word on the file): this batch goes
            # through Pillow; said once
            global _WARNED_PNG
            if not _WARNED_PNG:
                print(f"[cholect] --png_decode device: {e}; such files are decoded by Pillow on the host", flush=True)
                _WARNED_PNG = True
            decode = "host"
    assert decode == "host"

    def decode_one(fid):
        with Image.open(os.path.join(data_dir, "data", video, "{}.png".format(str(int(fid)).zfill(6)))) as im:
            return np.asarray(im.convert("RGB"))
    if workers > 1 and len(frame_ids) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            raw = list(ex.map(decode_one, frame_ids))
    else:
        raw = [decode_one(fid) for fid in frame_ids]
    out = torch.empty((len(raw), height, width, 3), dtype=torch.uint8, device=device)
    groups: Dict[tuple, List[int]] = {}
    for i, a in enumerate(raw):
        groups.setdefault(a.shape[:2], []).append(i)
    for (h0, w0), idx in groups.items():
        x = torch.from_numpy(np.stack([raw[i] for i in idx])).to(device)
        y = x if (h0, w0) == (height, width) else ops.resize_bilinear_u8(x, height, width)
        out[torch.tensor(idx, device=device)] = y
    return out
