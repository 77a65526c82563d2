#!/usr/bin/env python3
"""kernel totals of the last `window_ms` of a rocprofv3 --kernel-trace database (one training step): python tools/step_breakdown.py results.db 20.3 [pattern]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6
pat = sys.argv[3] if len(sys.argv) > 3 else ""
end = db.execute("select max(end) from kernels").fetchone()[0]
tot = defaultdict(lambda: [0, 0.0])
for n, s, e in db.execute("select name, start, end from kernels where start > ? order by start", (end - win,)):
    if pat in n:
        tot[n[:100]][0] += 1
        tot[n[:100]][1] += (e - s) / 1e6
print(f"busy {sum(v[1] for v in tot.values()):.3f} ms of {win / 1e6:.2f}")
for n, v in sorted(tot.items(), key=lambda x: -x[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print(f"{n:100s} {v[0]:4d} {v[1]:7.3f}")
