#!/usr/bin/env python3
"""How a timed region of K steps starts and ends: from a rocprofv3 kernel trace of a bench run (tools/pipe_trace.sh with
STEPS / REPEATS), the LAST region (launches between two gaps of > 80 us with no kernel resident) as CU-weighted occupancy in bins
of 50 us -- the CUs a launch can hold = min(256, its workgroups) (every EQTransformer forward workgroup holds a CU's LDS).
usage: trace_region_edges.py <kernel_trace.csv> [bin_us] [K]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // max(1, int(r["Workgroup_Size_X"]))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g))
rows.sort()
BIN = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 50e3
# regions: split where no kernel is resident for > 80 us
regions, cur, end = [], [], None
for s, e, k, g in rows:
    if end is not None and s - end > 80e3:
        regions.append(cur)
        cur = []
    cur.append((s, e, k, g))
    end = e if end is None else max(end, e)
regions.append(cur)
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
steps_of = lambda reg: sum(1 for r in reg if "eqt_tail" in r[2] or "pn_window" in r[2])
big = [r for r in regions if steps_of(r) == K]
reg = big[-1]
t0, t1 = reg[0][0], max(r[1] for r in reg)
n_tail = sum(1 for r in reg if "eqt_tail" in r[2] or "pn_window" in r[2])
print(f"regions of {K} steps: {len(big)}; the last one: {len(reg)} launches, {n_tail} steps, {(t1 - t0) / 1e3:.1f} us = {(t1 - t0) / 1e3 / max(1, n_tail):.1f} us per step")
nb = int((t1 - t0) / BIN) + 1
occ = [0.0] * nb
for s, e, k, g in reg:
    cus = min(256, g)
    b0, b1 = int((s - t0) / BIN), int((e - t0) / BIN)
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * BIN), min(e, t0 + (b + 1) * BIN)
        if hi > lo:
            occ[b] += cus * (hi - lo) / BIN
mid = sorted(occ[nb // 4: 3 * nb // 4])
print(f"CU-weighted occupancy per {BIN / 1e3:.0f}-us bin (CUs held, may exceed 256 where launches queue behind each other); median of the middle half: {mid[len(mid) // 2]:.0f}")
print("first bins:", " ".join(f"{o:.0f}" for o in occ[:14]))
print("last bins: ", " ".join(f"{o:.0f}" for o in occ[-14:]))
short = lambda k: k.replace("void ", "").replace("vp::(anonymous namespace)::", "").replace("vp::", "").split("(")[0].split("<")[0][:20]
print("last launches:")
for s, e, k, g in sorted(reg, key=lambda r: r[1])[-14:]:
    print(f"  {(s - t1) / 1e3:9.1f} -> {(e - t1) / 1e3:8.1f} us  {short(k):20s} {min(256, g):4d} CUs")
print("first launches:")
for s, e, k, g in reg[:10]:
    print(f"  {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:8.1f} us  {short(k):20s} {min(256, g):4d} CUs")
