#!/usr/bin/env python3
"""Per-kernel SQ counters from one rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE): for every kernel whose name contains one of the given substrings, summed over its
dispatches -- the share of wave time parked (s_waitcnt / barrier), stalled at issue, issuing; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles) and the bank-conflict share of the LDS cycles
(MI355X_MICROARCH.md: rocprofv3 PMC slots, LDS).
usage: python tools/kernel_counters.py <pmc_dir> <substr> [<substr> ...]"""
import collections
import csv
import glob
import os
import sys

d, subs = sys.argv[1], sys.argv[2:]
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    for s in subs:
        if s in name:
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-70:]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[short].add(r["Dispatch_Id"])
print(f"# {f}")
print("kernel                                                                  launches   parked  issue-stall  issuing   MFMA-busy   LDS-busy  LDS-conflict-share  us/launch(at 2.1 GHz)")
for k, c in acc.items():
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    n = len(seen[k])
    print(f"{k:70s} {n:9d}  {c['SQ_WAIT_ANY'] / wc:7.2f}  {c['SQ_WAIT_INST_ANY'] / wc:11.2f}  {c['SQ_ACTIVE_INST_ANY'] / wc:7.2f}  "
          f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * max(cyc, 1)):9.3f}  {c['SQ_LDS_IDX_ACTIVE'] / (256 * max(cyc, 1)):9.3f}  "
          f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):18.3f}  {cyc / n / 2100.0:10.1f}")
