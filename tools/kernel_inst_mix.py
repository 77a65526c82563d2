#!/usr/bin/env python3
"""Instruction mix per kernel from one rocprofv3 --pmc pass (SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_INSTS_MFMA): instructions per wave by class, and wave cycles per wave.
usage: python tools/kernel_inst_mix.py <pmc_dir> <substr> [<substr> ...]"""
import collections, csv, glob, os, sys
d, subs = sys.argv[1], sys.argv[2:]
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    for s in subs:
        if s in name:
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
print(f"# {f}")
print("kernel                                                         waves   per wave: cycles    VALU    SALU     LDS  VMEM_RD  VMEM_WR    MFMA")
for k, c in acc.items():
    w = max(c["SQ_WAVES"], 1.0)
    print(f"{k:60s} {int(w):9d}  {c['SQ_WAVE_CYCLES'] / w:15.0f} {c['SQ_INSTS_VALU'] / w:7.0f} {c['SQ_INSTS_SALU'] / w:7.0f} {c['SQ_INSTS_LDS'] / w:7.0f} "
          f"{c['SQ_INSTS_VMEM_RD'] / w:8.0f} {c['SQ_INSTS_VMEM_WR'] / w:8.0f} {c['SQ_INSTS_MFMA'] / w:7.0f}")
