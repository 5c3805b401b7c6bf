#!/usr/bin/env python3
"""BASELINE.json configs[3]-style run on ONE GPU: classify a 24 h, 100 Hz, 3-component synthetic stream
(8.64 M samples -> 17,269 EQT windows at overlap 5500) through the public API and report wall time."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
name = sys.argv[2] if len(sys.argv) > 2 else "eqtransformer"
n = int(hours * 3600 * 100)
data, p_on, s_on = synthetic_stream_array(n, seed=1004, n_events=max(1, n // 10_000))
t0 = va.UTCDateTime("2024-01-01T00:00:00")
st = va.Stream([va.Trace(data[i], dict(network="XX", station="DAY", channel=f"HH{c}", starttime=t0, sampling_rate=100.0))
                for i, c in enumerate("ZNE")])
cls = va.EQTransformer if name == "eqtransformer" else va.PhaseNet
model = cls.from_pretrained("volpick").cuda()
kw = dict(batch_size=256, overlap=5500, blinding=(500, 500)) if name == "eqtransformer" else dict(batch_size=256)
model.classify(va.Stream([tr.copy() for tr in st]).__class__([va.Trace(tr.data[:60000], dict(tr.stats)) for tr in st]), **kw)  # warm-up
torch.cuda.synchronize()
t = time.perf_counter()
out = model.classify(st, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
T = model.in_samples
step = T - kw.get("overlap", 1500)
nwin = (n - T) // step + 1 + (1 if ((n - T) // step) * step + T < n else 0)
print(f"{name}: {hours:g} h stream, {n} samples, {nwin} windows: classify() {dt * 1e3:.1f} ms wall "
      f"({nwin / dt:,.0f} windows/s incl. host->device copy), {len(out.picks)} picks, {len(out.detections)} detections")
pk = {ph: np.array(sorted((p.peak_time - t0) * 100 for p in out.picks if p.phase == ph)) for ph in "PS"}
for ph, truth in (("P", p_on), ("S", s_on)):
    truth = truth[(truth > 1000) & (truth < n - 3000)]
    d = np.array([np.min(np.abs(pk[ph] - t)) if len(pk[ph]) else 1e9 for t in truth])
    print(f"  {ph}: {np.mean(d <= 15) * 100:.1f} % of {len(truth)} synthetic arrivals picked within 0.15 s")
