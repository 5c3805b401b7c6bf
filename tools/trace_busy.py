#!/usr/bin/env python3
"""GPU occupancy of a pipelined bench run from a rocprofv3 kernel trace: union of the kernel intervals vs wall time,
time with 1 / 2 / 3+ kernels resident, and per-kernel mean durations.  Usage: trace_busy.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steady state: drop the first and last 15 % of the kernels
n = len(rows)
rows = rows[int(0.15 * n):int(0.85 * n)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last, depth = t, depth + d
wall = t1 - t0
print(f"kernels {len(rows)}  wall {wall / 1e3:.1f} us")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ' '} kernels resident: {100 * hist[k] / wall:5.1f} %")
dur = defaultdict(list)
for s, e, name in rows:
    dur[name.split("(")[0][-60:]].append(e - s)
tot = sum(sum(v) for v in dur.values())
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v) / tot * 100:5.1f} %  mean {sum(v) / len(v) / 1e3:7.1f} us  x{len(v):5d}  {name}")
print(f"sum of kernel durations / wall = {tot / wall:.2f}")
