#!/usr/bin/env python3
"""HBM traffic of one Temporal_tenco forward from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/tenco_infer_prof.py.
usage: python tools/collect_tcn_traffic.py <fetch_dir> <write_dir> <key> [out=profiles/traffic.json]
Per MI355X_MICROARCH.md (HBM): KiB units; FETCH_SIZE counts half the bytes of a wide coalesced stream on gfx950 (doubled here)."""
import csv, glob, json, os, sys

import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("srcdigest", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision_codes_amd", "srcdigest.py"))
_sd = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_sd)
KSHA = _sd.kernels_digest(_sd.TCN_SOURCES)          # the sources the profiled library was built from (bench.py nulls the figure when they change)

KERNELS = ("tcn_conv_kernel", "fpn_topdown_kernel", "igemm_conv_kernel", "tcn_layer_fused_kernel")


def last_forward(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS)]
    per = {}
    for did, name, v in rows:
        per[did] = (name, per.get(did, (name, 0.0))[1] + v)
    ids = sorted(per)
    # a forward ends with the heads launch (the only 1-tap launch with fp32 output after the FPN kernel): cut at the fpn_topdown dispatches
    cuts = [i for i, did in enumerate(ids) if "fpn_topdown" in per[did][0]]
    if len(cuts) >= 2:
        per_fwd = cuts[-1] - cuts[-2]
        sel = ids[cuts[-1] + 2 - per_fwd:cuts[-1] + 2]          # fpn kernel + the heads launch close a forward
    else:
        sel = ids
    return sum(per[did][1] for did in sel), len(sel)


fetch_dir, write_dir, key = sys.argv[1], sys.argv[2], sys.argv[3]
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
fe, n1 = last_forward(fetch_dir, "FETCH_SIZE")
wr, n2 = last_forward(write_dir, "WRITE_SIZE")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = {"hbm_bytes_per_video": 2.0 * fe * 1024 + wr * 1024, "read_bytes": 2.0 * fe * 1024, "write_bytes": wr * 1024, "launches": n1, "kernels_sha": KSHA,
             "note": "all launches of one forward (graph replay); read = 2 x FETCH_SIZE (gfx950 half-count), KiB units"}
json.dump(data, open(out_path, "w"), indent=1)
print(key, data[key])
