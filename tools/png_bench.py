"""GPU PNG decode rate against Pillow on the host (threads), frames of the CholecT45 native size (480 x 854).  python tools/png_bench.py [n]"""
import io, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from computervision_codes_amd import pngdec

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h, w = 480, 854
rng = np.random.default_rng(0)
y, x = np.mgrid[0:h, 0:w]
files = []
for i in range(16):      # 16 distinct synthetic "video frames": smooth structure + sensor-like noise (real frames: 400-700 KB per PNG)
    base = (np.stack([(x + 7 * i) % 256, (y * 2 + 3 * i) % 256, ((x + y) // 2) % 256], -1)).astype(np.int32)
    blobs = 40 * np.sin(x[..., None] / (17.0 + i) + np.arange(3)) * np.cos(y[..., None] / (23.0 + i))
    fr = np.clip(base * 0.5 + 60 + blobs + rng.normal(0, 3.0, (h, w, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(fr, "RGB").save(b, format="PNG"); files.append(b.getvalue())
files = [files[i % 16] for i in range(n)]
print(f"{n} frames of {h}x{w}, {sum(map(len, files)) / n / 1e3:.0f} KB per PNG", flush=True)
dev = torch.device("cuda:0")
out = pngdec.decode_batch(files, dev); torch.cuda.synchronize()      # (first call: page-locks the staging buffer)
tm = {}
t0 = time.perf_counter(); out = pngdec.decode_batch(files, dev, timings=tm); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"device decode (parse + H2D of the compressed bytes + inflate + unfilter): {n / (t1 - t0):.0f} frames/s ({(t1 - t0) * 1e3:.1f} ms; "
      f"inflate kernel {tm['inflate_ms']:.1f} ms, unfilter kernel {tm['unfilter_ms']:.1f} ms = {n / (tm['inflate_ms'] + tm['unfilter_ms']) * 1e3:.0f} frames/s of GPU time)", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dec = lambda f: np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
for workers in (1, 16):
    t0 = time.perf_counter()
    if workers == 1: ref = [dec(f) for f in files[:64]]; m = 64
    else:
        with ThreadPoolExecutor(workers) as ex: ref = list(ex.map(dec, files)); m = n
    t1 = time.perf_counter()
    print(f"Pillow, {workers} thread(s): {m / (t1 - t0):.0f} frames/s", flush=True)
assert np.array_equal(out[:16].cpu().numpy(), np.stack(ref[:16]))
