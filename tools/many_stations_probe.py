#!/usr/bin/env python3
"""classify() on a host Stream of N station-days, pageable / pinned rows.  usage: many_stations_probe.py [phasenet|eqtransformer] [stations]
(Round 6 also measured a helper thread uploading station k + 1 beside station k's compute: 2.91 -> 2.77 ms per PhaseNet station-day
at 16 stations, SLOWER at 3 stations (2.85 -> 3.48) and for EQTransformer (21.6 -> 22.6): the call is bound by the host-to-device
rate either way (36-38 GB/s on these boxes); removed again, LOG.md.)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "phasenet"
n_st = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 8_640_000
data = synthetic_stream_array(n, seed=1004, n_events=600)[0]
t0 = va.UTCDateTime("2021-01-01T00:00:00")
kw = dict(overlap=1500, blinding=(0, 0)) if name == "phasenet" else dict(overlap=5500, blinding=(500, 500))
pinned = [va.pinned_array(n, np.float32) for _ in range(3)]
for r, s in zip(pinned, data):
    r[:] = s
for rows_name, rows in (("pageable", list(data)), ("pinned", pinned)):
    for _ in (0,):
        m = (va.PhaseNet if name == "phasenet" else va.EQTransformer).from_pretrained("volpick").cuda()
        if len(sys.argv) > 3:
            m._seg_per_context = int(sys.argv[3])  # segments of a long block per device context (default 2)
        st = va.Stream([va.Trace(rows[i], dict(network="XX", station=f"S{k:03d}", location="", channel=f"HH{c}", starttime=t0,
                                               sampling_rate=100.0)) for k in range(n_st) for i, c in enumerate("ZNE")])
        m.classify(st, batch_size=256, **kw)
        walls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = m.classify(st, batch_size=256, **kw)
            walls.append(time.perf_counter() - t)
        w = sorted(walls)[1]
        print(f"{name} {n_st} stations, {rows_name} rows: {w * 1e3 / n_st:.3f} ms per station-day, "
              f"{n_st * data.nbytes / w / 1e9:.1f} GB/s of host rows, {len(r.picks)} picks", flush=True)
        m._release()
