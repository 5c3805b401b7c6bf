#!/usr/bin/env python3
"""classify() of a HOST 24 h three-component stream against the number of windows per forward launch (the user's batch_size
is a batching granularity: results do not depend on it).  usage: api_probe.py [phasenet|eqtransformer] [launch sizes ...]"""
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eqtransformer"
sizes = [int(v) for v in sys.argv[2:]] or [256, 1024, 4352]
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
n = 8_640_000
data = synthetic_stream_array(n, seed=1004, n_events=600)[0]
t0 = va.UTCDateTime("2021-01-01T00:00:00")
st = va.Stream([va.Trace(data[i], dict(network="XX", station="DAY", location="", channel=f"HH{c}", starttime=t0, sampling_rate=100.0))
                for i, c in enumerate("ZNE")])
kw = dict(overlap=1500, blinding=(0, 0)) if name == "phasenet" else dict(overlap=5500, blinding=(500, 500))
ref = None
for B in sizes:
    m = cls.from_pretrained("volpick")
    m._max_batch = B
    m.cuda()
    res = m.classify(st, batch_size=B, stacking="avg", **kw)
    walls = []
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = m.classify(st, batch_size=B, stacking="avg", **kw)
        walls.append(time.perf_counter() - t)
    m._timing = {}
    m.classify(st, batch_size=B, stacking="avg", **kw)
    ph = {k: round(v, 2) for k, v in m._timing.items() if k.endswith("_ms")}
    key = [(p.phase, p.peak_time.timestamp, p.peak_value) for p in res.picks]
    same = "" if ref is None else ("  picks identical to the first size" if key == ref else "  PICKS DIFFER")
    ref = ref or key
    print(f"{name} windows per launch {B:5d}: wall {statistics.median(walls) * 1e3:6.2f} ms  phases {ph}  picks {len(res.picks)}{same}", flush=True)
    m._release()
