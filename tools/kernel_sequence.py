#!/usr/bin/env python3
"""The launches of the LAST forward in a `rocprofv3 --kernel-trace --output-format csv` run, in order, with durations: which launch of a model costs what.
usage: python tools/kernel_sequence.py <dir with *_kernel_trace.csv> <launches per forward> [first=0]"""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2])
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "pack_conv_weight" not in r["Kernel_Name"] and "bneck_pack" not in r["Kernel_Name"]]
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0.0
for i, r in enumerate(last):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.index("(")] if "(" in name else name
    print(f"{i:4d} {(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {dur:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8s} x{r.get('Grid_Size_Y', '?'):>6s} x{r.get('Grid_Size_Z', '?'):>4s}  wg {r.get('Workgroup_Size_X', '?'):>4s}  {name[:90]}")
print(f"sum of durations {tot / 1e3:.3f} ms, span {(int(last[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
