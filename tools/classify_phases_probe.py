#!/usr/bin/env python3
"""Where classify() of a DEVICE-resident station-day spends its time beyond the forward passes: cProfile of the host side of
one call (the GPU work is queued asynchronously; the host blocks in the collects)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from bench import station_day_mseed  # noqa: E402

buf = station_day_mseed(24)[0]
m = va.PhaseNet.from_pretrained("volpick").cuda()
st = va.read(buf, device_resident=True)
kw = dict(batch_size=256, overlap=1500, blinding=(0, 0), stacking="avg")
for _ in range(3):
    m.classify(st, **kw)
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter()
    m.classify(st, **kw)
    ts.append((time.perf_counter() - t0) * 1e3)
print("classify(device-resident station-day) ms:", [round(t, 3) for t in sorted(ts)])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.classify(st, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
