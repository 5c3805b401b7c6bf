#!/usr/bin/env python3
"""Host-to-device rates of a station-day (3 rows x 34.5 MB): pageable rows one after the other / from three threads / pinned rows
asynchronously; whole rows and six segments per row (what classify() uploads)."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

n = 8_640_000
rows = [np.random.default_rng(i).standard_normal(n).astype(np.float32) for i in range(3)]
pinned = [torch.from_numpy(r).pin_memory() for r in rows]
dev = torch.device("cuda", 0)
x = torch.empty((3, n), dtype=torch.float32, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(3)]


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return dt * 1e3, 3 * n * 4 / dt / 1e9


def seq_pageable(nseg=1):
    L = n // nseg
    for s in range(nseg):
        for c in range(3):
            x[c, s * L:(s + 1) * L].copy_(torch.from_numpy(rows[c][s * L:(s + 1) * L]))


def thr_pageable(pool, nseg=1):
    L = n // nseg

    def one(c):
        with torch.cuda.stream(streams[c]):
            for s in range(nseg):
                x[c, s * L:(s + 1) * L].copy_(torch.from_numpy(rows[c][s * L:(s + 1) * L]))
    list(pool.map(one, range(3)))


def pinned_async(nseg=1, sync_each=False):
    L = n // nseg
    for s in range(nseg):
        for c in range(3):
            x[c, s * L:(s + 1) * L].copy_(pinned[c][s * L:(s + 1) * L], non_blocking=True)
        if sync_each:
            torch.cuda.current_stream().synchronize()


with ThreadPoolExecutor(3) as pool:
    for nseg in (1, 6):
        print(f"segments per row {nseg}:")
        print("  pageable, one thread          %.2f ms  %.1f GB/s" % timed(lambda: seq_pageable(nseg)))
        print("  pageable, three threads       %.2f ms  %.1f GB/s" % timed(lambda: thr_pageable(pool, nseg)))
        print("  pinned, async                 %.2f ms  %.1f GB/s" % timed(lambda: pinned_async(nseg)))
        print("  pinned, sync per segment      %.2f ms  %.1f GB/s" % timed(lambda: pinned_async(nseg, True)))
