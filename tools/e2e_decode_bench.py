"""The extraction driver's per-video loop on PNG files on disk, decode included: Pillow on host threads vs the device decoder, frames of the
dataset's native 480 x 854 resized to 256 x 448, ResNet-50 student in bf16.  Writes a synthetic video of N frames (16 distinct frames) under
/tmp first.   python tools/e2e_decode_bench.py [N]"""
import io, os, shutil, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from computervision_codes_amd import cholect, extract, shapes, synth
from computervision_codes_amd.spatial_cnn import VideoNas

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
root = "/tmp/e2e_png"
shutil.rmtree(root, ignore_errors=True)
os.makedirs(os.path.join(root, "data", "VID01"))
h, w = 480, 854
rng = np.random.default_rng(0)
y, x = np.mgrid[0:h, 0:w]
blobs16 = []
for i in range(16):
    base = (np.stack([(x + 7 * i) % 256, (y * 2 + 3 * i) % 256, ((x + y) // 2) % 256], -1)).astype(np.int32)
    blobs = 40 * np.sin(x[..., None] / (17.0 + i) + np.arange(3)) * np.cos(y[..., None] / (23.0 + i))
    fr = np.clip(base * 0.5 + 60 + blobs + rng.normal(0, 3.0, (h, w, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(fr, "RGB").save(b, format="PNG"); blobs16.append(b.getvalue())
for i in range(n):
    with open(os.path.join(root, "data", "VID01", f"{i:06d}.png"), "wb") as f:
        f.write(blobs16[i % 16])
print(f"{n} PNG files of {len(blobs16[0]) // 1000} KB written", flush=True)
args = types.SimpleNamespace(network="resnet50", loss_type="all", student_dim=2048, teacher_dim=1536, train=False)
m = VideoNas(args=args, dtype=torch.bfloat16).eval()
m.load_state_dict(synth.fill_from_shapes(shapes.spatial_cnn_shapes("resnet50"), seed=1))
ids = np.arange(n)
ref = None
for name, decode, load_batch, depth in (("host, 16 threads, loads of 512, one ahead", "host", None, 1),
                                        ("device, loads of 2560, one ahead", "device", 2560, 1),
                                        ("device, loads of 1024, one ahead", "device", 1024, 1),
                                        ("device, loads of 1024, two ahead", "device", 1024, 2),
                                        ("device, loads of 1024, three ahead", "device", 1024, 3),
                                        ("device, loads of 512, two ahead", "device", 512, 2),
                                        ("device, loads of 512, three ahead", "device", 512, 3)):
    load = lambda s, e: cholect.load_frames_device(root, "VID01", ids[s:e], 256, 448, workers=16, decode=decode)
    run = lambda: extract.extract_video_device(m, n, load, 512, prefetch=depth, load_batch=load_batch)
    run()                                                   # (page cache, pinned buffers, first launches)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    feat, _ = run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if ref is None: ref = feat.copy()
    print(f"{name:45s} {n / dt:8.0f} frames/s ({dt * 1e3:.0f} ms per {n}-frame video)   same features: {np.array_equal(ref, feat)}", flush=True)
shutil.rmtree(root, ignore_errors=True)
