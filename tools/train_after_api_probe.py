#!/usr/bin/env python3
"""Does the training step slow down behind classify() calls of the kind bench.py's `api.many_stations` leg makes?
usage: train_after_api_probe.py [none|pageable|pinned|pinned_only]   (what runs before the trainer is timed)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from tools.bench_train import make_batch  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402
from volpick_amd.train import PhaseNetTrainer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    n = 8_640_000
    data = synthetic_stream_array(n, seed=1004, n_events=600)[0]
    t0 = va.UTCDateTime("2021-01-01T00:00:00")
    rows = list(data)
    if mode.startswith("pinned"):
        rows = [va.pinned_array(n, np.float32) for _ in range(3)]
        for r, s in zip(rows, data):
            r[:] = s
    if mode != "pinned_only":
        m = va.PhaseNet.from_pretrained("volpick").cuda()
        if len(sys.argv) > 2 and sys.argv[2] == "no_ahead":
            m._upload_ahead = lambda groups, args: groups
        st = va.Stream([va.Trace(rows[i], dict(network="XX", station=f"S{k}", location="", channel=f"HH{c}", starttime=t0,
                                               sampling_rate=100.0)) for k in range(1 if (len(sys.argv) > 2 and sys.argv[2] == "one_station") else 6) for i, c in enumerate("ZNE")])
        for _ in range(2):
            t = time.perf_counter()
            r = m.classify(st, overlap=1500, batch_size=256)
            print(f"classify of 6 station-days: {(time.perf_counter() - t) * 1e3:.1f} ms, {len(r.picks)} picks", flush=True)
        if not (len(sys.argv) > 2 and sys.argv[2] == "keep_alive"):
            m._release()
            del m
        del st
        if len(sys.argv) > 2 and sys.argv[2] == "empty_cache":
            import gc

            gc.collect()
            torch.cuda.empty_cache()
if mode == "none" and len(sys.argv) > 2 and sys.argv[2] == "churn":
    # device memory allocated, touched and returned to the driver before the trainer allocates its own
    bufs = [torch.full((256 << 20,), float(i), dtype=torch.float32, device="cuda") for i in range(4)]  # 4 x 1 GiB
    torch.cuda.synchronize()
    del bufs
    torch.cuda.empty_cache()
B = 512
x, y = make_batch(B)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
tr = PhaseNetTrainer(va.PhaseNet.from_pretrained("volpick"), max_batch=B, dtype="bf16")
for rep in range(3):
    for _ in range(10):
        tr.step(xd, yd, 1e-4, want_loss=False)
    tr.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        tr.step(xd, yd, 1e-4, want_loss=False)
    t_enq = time.perf_counter() - t
    tr.synchronize()
    print(f"mode {mode}: training step {(time.perf_counter() - t) / 30 * 1e3:.3f} ms (host enqueue {t_enq / 30 * 1e3:.3f} ms)", flush=True)
