#!/usr/bin/env python3
"""One steady training step as a timeline, from a rocprofv3 kernel trace of tools/bench_train.py: per queue the time inside
kernels and the gaps between one kernel's end and the next one's start; the union over queues; the largest gaps.
usage: train_timeline.py <kernel_trace.csv> [launches per step]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 136
short = lambda k: k.split("(")[0].split("::")[-1][:40]
# a step starts at the first-layer conv of the forward pass: find the kernel name that occurs exactly once per step and first
marks = [i for i, r in enumerate(rows) if "load_rows" in r[2] or "ConvCfg<3, 0, 8" in r[2]]
starts = [m for j, m in enumerate(marks) if j == 0 or m - marks[j - 1] > per // 2]
assert len(starts) >= 4, "no step marks found"
# the shortest of the steady steps: under rocprofv3 the host may fall behind the device for a step (gaps that are the tool's)
cands = [(rows[starts[i + 1]][0] - rows[starts[i]][0], starts[i], starts[i + 1]) for i in range(1, len(starts) - 1)]
_, a, b = min(cands)
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
print(f"step: {len(step)} launches, {(t1 - t0) / 1e3:.1f} us from its first kernel's start to the next step's")
queues = sorted({r[3] for r in step})
for q in queues:
    ks = [r for r in step if r[3] == q]
    busy = sum(e - s for s, e, *_ in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], short(ks[i][2]), short(ks[i + 1][2])) for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g[0] < 20000]
    print(f"queue {q}: {len(ks)} launches, {busy / 1e3:.1f} us inside kernels, span {(ks[-1][1] - ks[0][0]) / 1e3:.1f} us; "
          f"{len(small)} back-to-back gaps, sum {sum(g[0] for g in small) / 1e3:.1f} us, median {sorted(g[0] for g in small)[len(small) // 2] / 1e3:.2f} us")
    for g in sorted(gaps, reverse=True)[:6]:
        print(f"      gap {g[0] / 1e3:8.2f} us  after {g[1]}  before {g[2]}")
ev = sorted([(s, 1) for s, e, *_ in step] + [(e, -1) for s, e, *_ in step])
depth, last, union, two = 0, t0, 0, 0
for t, d in ev:
    if depth > 0:
        union += t - last
    if depth > 1:
        two += t - last
    depth += d
    last = t
print(f"union of kernels {union / 1e3:.1f} us ({two / 1e3:.1f} us with two or more running), idle {(t1 - t0 - union) / 1e3:.1f} us")
if len(sys.argv) > 3:
    for s, e, k, q in step:
        print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f} us  ({(e - s) / 1e3:7.2f})  queue {q:>3} {short(k)}")
