#!/usr/bin/env python3
"""Two extraction streams on DISJOINT halves of the CUs (hipExtStreamCreateWithCUMask), half a pass apart, against one stream on the whole
chip: does the HBM-bound head of one part overlap the MFMA-bound tail of the other when they cannot fight for the same CUs?  (GPU box)
python tools/cu_mask_probe.py [--frames 1336] [--passes 12]"""
import argparse, ctypes, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from computervision_codes_amd.spatial_cnn import VideoNas
from computervision_codes_amd import shapes, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1336)
ap.add_argument("--passes", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * 8)(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=None, teacher_dim=1536, train=False)
model = VideoNas(args=args, dtype=torch.bfloat16).eval()
model.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=47))
fa = bench.device_frames(a.frames, 224, 224, 1, dev, nbase=64)
fb = bench.device_frames(a.frames, 224, 224, 2, dev, nbase=64)
fhalf = fb[: a.frames // 2].contiguous()
fboth = torch.cat([fa, fb], 0).contiguous()


def run(streams, frames, passes, skew=None):
    """enqueue `passes` passes per stream, alternating streams from this thread; returns frames/s"""
    for st, f in zip(streams, frames):           # warm-up
        with torch.cuda.stream(st):
            model.extract_u8(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if skew is not None:
        with torch.cuda.stream(streams[1]):
            model.extract_u8(skew)
    for _ in range(passes):
        for st, f in zip(streams, frames):
            with torch.cuda.stream(st):
                model.extract_u8(f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = passes * sum(f.shape[0] for f in frames) + (skew.shape[0] if skew is not None else 0)
    return n / dt


full = torch.cuda.Stream()
print(f"one stream, whole chip, {2 * a.frames} frames per pass : {run([full], [fboth], a.passes):9.0f} frames/s", flush=True)
print(f"one stream, whole chip, {a.frames} frames per pass : {run([full], [fa], 2 * a.passes):9.0f} frames/s", flush=True)
two = [torch.cuda.Stream(), torch.cuda.Stream()]
print(f"two plain streams, skewed half a pass         : {run(two, [fa, fb], a.passes, skew=fhalf):9.0f} frames/s", flush=True)
lo = masked_stream(range(0, 128))      # bit i -> XCD i % 8, CU i / 8 of it: the lower 16 CUs of every XCD
hi = masked_stream(range(128, 256))
print(f"one stream on half the CUs                    : {run([lo], [fa], a.passes):9.0f} frames/s", flush=True)
print(f"two streams on disjoint CU halves, in phase   : {run([lo, hi], [fa, fb], a.passes):9.0f} frames/s", flush=True)
print(f"two streams on disjoint CU halves, skewed     : {run([lo, hi], [fa, fb], a.passes, skew=fhalf):9.0f} frames/s", flush=True)
ev = masked_stream([b for b in range(256) if (b // 8) % 2 == 0])    # every other CU of each XCD
od = masked_stream([b for b in range(256) if (b // 8) % 2 == 1])
print(f"two streams on interleaved CU halves, skewed  : {run([ev, od], [fa, fb], a.passes, skew=fhalf):9.0f} frames/s", flush=True)
