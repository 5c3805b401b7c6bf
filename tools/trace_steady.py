#!/usr/bin/env python3
"""Steady-state stretch of a pipelined bench run in a rocprofv3 kernel trace: between the N1-th and N2-th launch of the kernel
whose name contains KEY (one per step), the time per step, the share of the time with 0 / 1 / 2 / 3+ launches resident, the idle
gaps, and every kernel's mean residence (dispatch -> end) against its count.
usage: trace_steady.py <kernel_trace.csv> KEY [N1 N2]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2]
ends = [e for s, e, k in rows if key in k]
n1, n2 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (len(ends) // 5, len(ends) // 5 + 60)
t0, t1 = ends[n1], ends[n2]
sel = [(max(s, t0), min(e, t1), k) for s, e, k in rows if e > t0 and s < t1]
ev = []
for s, e, _ in sel:
    ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, hist, gaps = 0, t0, defaultdict(int), []
for t, d in ev:
    hist[min(depth, 3)] += t - last
    if depth == 0 and t - last > 1000:
        gaps.append((t - last) / 1e3)
    last, depth = t, depth + d
wall = t1 - t0
print(f"steps {n2 - n1}  wall {wall / 1e3:.1f} us  = {wall / 1e3 / (n2 - n1):.2f} us per step")
print("launches resident: " + "  ".join(f"{k}{'+' if k == 3 else ''}: {100 * hist[k] / wall:.1f} %" for k in sorted(hist)))
print(f"idle gaps > 1 us: {len(gaps)}, total {sum(gaps):.1f} us = {100 * sum(gaps) * 1e3 / wall:.1f} % of the wall; longest {max(gaps, default=0):.1f} us")
dur = defaultdict(list)
for s, e, k in rows:
    if e > t0 and s < t1:
        dur[k.replace("vp::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]].append(e - s)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:44s} x{len(v):4d}  mean residence {sum(v) / len(v) / 1e3:8.1f} us")
