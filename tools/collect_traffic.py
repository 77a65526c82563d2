#!/usr/bin/env python3
"""HBM traffic of the conv launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: python tools/collect_traffic.py <fetch_dir> <write_dir> <launches_per_step> <key> [out=profiles/traffic.json]
Per MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of a wide
(16 B/lane) coalesced stream -- doubled here; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections, csv, glob, json, os, sys

import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("srcdigest", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision_codes_amd", "srcdigest.py"))
_sd = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_sd)
KSHA = _sd.kernels_digest(_sd.CONV_SOURCES)          # the sources the profiled library was built from (bench.py nulls the figure when they change)

CONV_KERNELS = ("igemm_conv_kernel", "conv3x3_patch_kernel", "stem_patch_kernel", "stem_pool_kernel", "bottleneck64_fused_kernel", "chain_gemm_kernel")   # everything mt4_conv_nhwc launches

def per_dispatch(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # (the newest pass)
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in CONV_KERNELS):
            out[int(r["Dispatch_Id"])] = out.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return list(out.values())

fetch_dir, write_dir, per_step, key = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
out_path = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
fe, wr = per_dispatch(fetch_dir, "FETCH_SIZE"), per_dispatch(write_dir, "WRITE_SIZE")
n = min(len(fe), len(wr)) // per_step * per_step
steps = n // per_step
# last full step (steady state)
fe_s, wr_s = fe[n - per_step:n], wr[n - per_step:n]
rd = 2.0 * sum(fe_s) * 1024
wb = sum(wr_s) * 1024
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = {"hbm_bytes_per_step": rd + wb, "read_bytes": rd, "write_bytes": wb, "fetch_size_kib_raw": sum(fe_s),
             "write_size_kib_raw": sum(wr_s), "launches": per_step, "steps_profiled": steps, "kernels_sha": KSHA,
             "note": "conv launches of one step; read = 2 x FETCH_SIZE (gfx950 half-count), units KiB"}
json.dump(data, open(out_path, "w"), indent=1)
print(key, "read GB", rd / 1e9, "write GB", wb / 1e9)
