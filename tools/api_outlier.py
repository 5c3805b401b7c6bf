#!/usr/bin/env python3
"""What stalls one classify() call in five?  (BENCH_r03: 65 ms among 24-27 ms calls of the EQTransformer station-day.)
40 calls with the garbage collector on, its collections timed through gc.callbacks, then 40 with it frozen + off.
usage: api_outlier.py [phasenet|eqtransformer]"""
import gc
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eqtransformer"
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
n = 8_640_000
data = synthetic_stream_array(n, seed=1004, n_events=600)[0]
t0 = va.UTCDateTime("2021-01-01T00:00:00")
st = va.Stream([va.Trace(data[i], dict(network="XX", station="DAY", location="", channel=f"HH{c}", starttime=t0, sampling_rate=100.0))
                for i, c in enumerate("ZNE")])
kw = dict(overlap=1500, blinding=(0, 0)) if name == "phasenet" else dict(overlap=5500, blinding=(500, 500))
m = cls.from_pretrained("volpick").cuda()
res = m.classify(st, batch_size=256, stacking="avg", **kw)

events, t_gc = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(cb)
for label in ("gc on", "gc frozen + off"):
    if label != "gc on":
        gc.collect()
        gc.freeze()
        gc.disable()
    walls, gcs = [], []
    for _ in range(40):
        del events[:]
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = m.classify(st, batch_size=256, stacking="avg", **kw)
        walls.append((time.perf_counter() - t) * 1e3)
        gcs.append([(g, round(ms, 1)) for g, ms in events if ms > 0.5])
    s = sorted(walls)
    print(f"{name} {label}: median {s[len(s) // 2]:.2f} ms, max {s[-1]:.2f}; calls > 1.3 x median: "
          f"{[(i, round(w, 1), gcs[i]) for i, w in enumerate(walls) if w > 1.3 * s[len(s) // 2]]}", flush=True)
print("gc counts", gc.get_count(), "objects", len(gc.get_objects()))
