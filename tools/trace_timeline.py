#!/usr/bin/env python3
"""A stretch of the pipelined bench run as a timeline: start / end of every launch relative to the first, from a rocprofv3
kernel trace (tools/pipe_trace.sh).  usage: trace_timeline.py <kernel_trace.csv> [first launch index] [count]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = rows[i0][0]
short = lambda k: ("pn_window" if "pn_window" in k else k.split("(")[0].split("::")[-1])[:22]
prev_pn_end = None
for s, e, k, q, st in rows[i0:i0 + n]:
    note = ""
    if "pn_window" in k:
        if prev_pn_end is not None:
            note = f"   starts {(s - prev_pn_end) / 1e3:+.2f} us after the previous pn_window's end"
        prev_pn_end = e
    print(f"{(s - t0) / 1e3:9.2f} -> {(e - t0) / 1e3:9.2f} us  ({(e - s) / 1e3:7.2f})  queue {q:>3} {short(k):22s}{note}")
