#!/usr/bin/env python3
"""Host time of vp_classify_submit / vp_classify_collect at the start of a region of K steps (the bench's timed loop): how fast the
host fills the device contexts after a synchronisation.  usage: submit_time_probe.py [model] [K]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eqtransformer"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 256
lib = _lib.load()
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
model = cls.from_pretrained("volpick")
model._max_batch = B
model.cuda()
T = model.in_samples
overlap, blinding = (1500, (0, 0)) if name == "phasenet" else (5500, (500, 500))
n_samples = T + (T - overlap) * (B - 1)
data, _, _ = synthetic_stream_array(n_samples, seed=1002)
x = torch.from_numpy(data).cuda()
out = torch.empty((3, n_samples), dtype=torch.float32, device="cuda")
specs = model._trigger_specs({})
c_specs = (_lib.VpTriggerSpec * len(specs))(*[_lib.VpTriggerSpec(r, t_on, t_off) for r, _, t_on, t_off in specs])
cap = 8192
on, off, peak = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
val, spec_of = (C.c_float * cap)(), (C.c_int32 * cap)()
found = C.c_int()
fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()
NCTX, DEPTH = model.n_contexts, 2
ctxs = [model._context(k) for k in range(NCTX)]
outs = [out] + [torch.empty_like(out) for _ in range(NCTX - 1)]
log = []


def submit(i):
    k, slot = i % NCTX, (i // NCTX) % DEPTH
    t = time.perf_counter()
    _lib.check(lib.vp_classify_submit(ctxs[k], slot, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE, n_samples, overlap, blinding[0],
                                      blinding[1], _lib.VP_STACK_AVG, B, c_specs, len(specs), C.c_void_p(outs[k].data_ptr()),
                                      _lib.VP_MEM_DEVICE, cap), "submit")
    log.append(("submit", i, t, time.perf_counter()))


def collect(i):
    k, slot = i % NCTX, (i // NCTX) % DEPTH
    t = time.perf_counter()
    _lib.check(lib.vp_classify_collect(ctxs[k], slot, C.byref(fv), C.byref(lv), C.byref(nw), on, off, peak, val, spec_of, cap,
                                       C.byref(found)), "collect")
    log.append(("collect", i, t, time.perf_counter()))


def run_steps(k):
    inflight = []
    for i in range(k):
        if len(inflight) == NCTX * DEPTH:
            collect(inflight.pop(0))
        submit(i)
        inflight.append(i)
    while inflight:
        collect(inflight.pop(0))


for _ in range(30):
    run_steps(K)
torch.cuda.synchronize()
best = None
for _ in range(7):
    log.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(K)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if best is None or t1 - t0 < best[0]:
        best = (t1 - t0, t0, list(log))
dt, t0, lg = best
print(f"{name}: best region of {K} steps: {dt * 1e6:.0f} us = {dt / K * 1e6:.1f} us per step ({B * K / dt:.0f} windows/s)")
for kind, i, a, b in lg[:10] + lg[-10:]:
    print(f"  {kind:8s} step {i:3d}: called at {(a - t0) * 1e6:8.1f} us, took {(b - a) * 1e6:7.1f} us")
