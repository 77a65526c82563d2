#!/usr/bin/env python3
"""Extraction from PNG FILES (decode on the device) as the driver runs it -- `extract.extract_videos_device` over several videos, one loader pipeline
across them -- swept over (frames per load, loads in flight), beside the host reader alone (`pngdec.read_files`: native threads) and the device
decoder alone.  480 x 854 synthetic frames (16 distinct contents, every file its own inode), ResNet-50 bf16, Resize to 256 x 448.
  python tools/png_pipeline_sweep.py > profiles/r04_png_pipeline_sweep.txt"""
import io
import os
import shutil
import sys
import tempfile
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computervision_codes_amd import cholect, extract, pngdec, shapes, synth  # noqa: E402
from computervision_codes_amd.spatial_cnn import VideoNas  # noqa: E402


def main():
    from PIL import Image
    dev = torch.device("cuda:0")
    nvid, n = 3, 2048
    root = tempfile.mkdtemp(prefix="mt4_png_")
    try:
        h, w = 480, 854
        rng = np.random.default_rng(0)
        y, x = np.mgrid[0:h, 0:w]
        blobs = []
        for i in range(16):
            base = (np.stack([(x + 7 * i) % 256, (y * 2 + 3 * i) % 256, ((x + y) // 2) % 256], -1)).astype(np.int32)
            wave = 40 * np.sin(x[..., None] / (17.0 + i) + np.arange(3)) * np.cos(y[..., None] / (23.0 + i))
            fr = np.clip(base * 0.5 + 60 + wave + rng.normal(0, 3.0, (h, w, 3)), 0, 255).astype(np.uint8)
            b = io.BytesIO()
            Image.fromarray(fr, "RGB").save(b, format="PNG")
            blobs.append(b.getvalue())
        vids = [f"VID{v + 1:02d}" for v in range(nvid)]
        for v in vids:
            os.makedirs(os.path.join(root, "data", v))
            for i in range(n):
                with open(os.path.join(root, "data", v, f"{i:06d}.png"), "wb") as f:
                    f.write(blobs[(i + 5 * vids.index(v)) % 16])
        print(f"# {nvid} videos x {n} frames of {w} x {h}, {len(blobs[0]) // 1000} KB per PNG; host cores {len(os.sched_getaffinity(0))}")
        paths = [os.path.join(root, "data", vids[0], f"{i:06d}.png") for i in range(n)]
        for workers in (4, 8, 16, 32):
            pngdec.read_files(paths[:256], workers)
            t0 = time.perf_counter()
            for _ in range(3):
                pngdec.read_files(paths, workers)
            dt = (time.perf_counter() - t0) / 3
            print(f"host reader alone, {workers:2d} threads: {n / dt:9.0f} files/s  ({n * len(blobs[0]) / dt / 1e9:.2f} GB/s)")
        for cnt in (512, 1024, 2048):
            pngdec.decode_files(paths[:cnt], dev, workers=16)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tm = {}
            pngdec.decode_files(paths[:cnt], dev, timings=tm, workers=16)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"decode_files alone, {cnt:4d} frames per call: {cnt / dt:8.0f} frames/s  (inflate {tm['inflate_ms']:.1f} ms, unfilter {tm['unfilter_ms']:.1f} ms)")
        args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
        m = VideoNas(args=args, dtype=torch.bfloat16, device=str(dev)).eval().load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=3))
        ids = np.arange(n)
        ref = None
        print("load_batch  in_flight  workers   frames/s (3 videos, one pipeline)   frames/s (ONE video)   same features")
        for lb, depth, workers in ((1024, 2, 16), (1024, 3, 16), (512, 3, 16), (512, 4, 16), (256, 6, 16), (512, 4, 32)):
            mk = lambda v: (lambda s, e: cholect.load_frames_device(root, v, ids[s:e], 256, 448, device=dev, workers=workers, decode="device"))
            plan = [(v, n, mk(v)) for v in vids]
            run = lambda pl: list(extract.extract_videos_device(m, pl, 512, prefetch=depth, load_batch=lb))
            run(plan[:2])                                   # (warm: helper threads, their pinned staging, the side streams' allocator pools)
            torch.cuda.synchronize()
            dt = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                out = run(plan)
                dt = min(dt, time.perf_counter() - t0)
            t0 = time.perf_counter()
            one = run(plan[:1])
            dt1 = time.perf_counter() - t0
            ref = out[0][1].copy() if ref is None else ref
            same = bool(np.array_equal(ref, out[0][1]) and np.array_equal(ref, one[0][1]))
            print(f"{lb:10d}  {depth:9d}  {workers:7d}   {nvid * n / dt:12.0f}                       {n / dt1:12.0f}           {same}", flush=True)
        # the same loop from frames already decoded (what the trunk alone allows) and with Pillow on 16 threads
        frames = cholect.load_frames_device(root, vids[0], ids, 256, 448, device=dev, workers=16, decode="device")
        t0 = time.perf_counter()
        extract.extract_video_device(m, n, lambda s, e: frames[s:e], 512)
        print(f"from decoded frames on the device: {n / (time.perf_counter() - t0):.0f} frames/s")
        t0 = time.perf_counter()
        extract.extract_video_device(m, n, lambda s, e: cholect.load_frames_device(root, vids[0], ids[s:e], 256, 448, device=dev, workers=16, decode="host"), 512, prefetch=1)
        print(f"Pillow on 16 host threads: {n / (time.perf_counter() - t0):.0f} frames/s")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
