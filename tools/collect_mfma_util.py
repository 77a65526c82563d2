#!/usr/bin/env python3
"""MFMA utilisation of the conv launches from one rocprofv3 --pmc pass of bench.py
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE in one pass: all three fit the SQ/GRBM counter blocks).
usage: python tools/collect_mfma_util.py <pmc_dir> <launches_per_step> <key> [out=profiles/mfma_util.json]
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD (= 16 x the number of 16x16x32 bf16 MFMAs,
MI355X_MICROARCH.md counter-unit table; checked: layer1 conv2 = 18.9 M MFMAs -> 301.7 M); GRBM_GUI_ACTIVE comes back as ONE row
per dispatch summed over the 8 XCDs (value / launch duration = 18.3-19.4 cycles per ns = 8 x 2.29-2.43 GHz), so the launch's
duration in shader-clock cycles is value / 8.  utilisation = busy / (1024 SIMDs x cycles)."""
import collections, csv, glob, json, os, sys

import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("srcdigest", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision_codes_amd", "srcdigest.py"))
_sd = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_sd)
KSHA = _sd.kernels_digest(_sd.CONV_SOURCES)          # the sources the profiled library was built from (bench.py nulls the figure when they change)

CONV_KERNELS = ("igemm_conv_kernel", "conv3x3_patch_kernel", "stem_patch_kernel", "stem_pool_kernel", "bottleneck64_fused_kernel", "chain_gemm_kernel")   # every conv launch of a step

N_SIMD = 256 * 4
N_XCD = 8


def per_dispatch(d):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # (the newest pass)
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in CONV_KERNELS):
            continue
        rec = out.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(list))
        rec[r["Counter_Name"]].append(float(r["Counter_Value"]))
        rec["_name"] = r["Kernel_Name"]
    return list(out.values())


pmc_dir, per_step, key = sys.argv[1], int(sys.argv[2]), sys.argv[3]
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "mfma_util.json")
d = per_dispatch(pmc_dir)
n = len(d) // per_step * per_step
last = d[n - per_step:n]
rows, busy_t, cyc_t = [], 0.0, 0.0
for i, r in enumerate(last):
    busy = sum(r["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = r["GRBM_GUI_ACTIVE"]
    cyc = sum(gui) / N_XCD
    busy_t += busy
    cyc_t += cyc
    rows.append(dict(launch=i, kernel=r["_name"].split("(")[0][-60:], mfma_busy_cycles=busy, gpu_cycles=cyc, grbm_rows=len(gui),
                     sq_busy_cycles=sum(r.get("SQ_BUSY_CYCLES", [0.0])), mfma_util=round(busy / (N_SIMD * cyc), 4)))
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = dict(mfma_util=round(busy_t / (N_SIMD * cyc_t), 4), mfma_busy_cycles=busy_t, gpu_cycles=cyc_t, launches=per_step, kernels_sha=KSHA,
                 steps_profiled=n // per_step, per_launch=rows,
                 note="conv launches of one step; utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)")
json.dump(data, open(out_path, "w"), indent=1)
print(key, "MFMA utilisation", data[key]["mfma_util"], "cycles/step", cyc_t)
