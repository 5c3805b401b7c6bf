#!/usr/bin/env python3
"""What ONE forward launch over K batches buys (vp_classify_multi: K stream blocks of 256 windows each share one forward
launch of K x 256 workgroups, one stacking launch, one trigger-scan launch): us per 256-window block against K.
usage: coalesce_probe.py [phasenet|eqtransformer]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "phasenet"
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
T = cls.in_samples
overlap, blinding = (1500, (0, 0)) if name == "phasenet" else (5500, (500, 500))
n = T + (T - overlap) * 255
for K in (1, 2, 4, 8):
    m = cls.from_pretrained("volpick")
    m._max_batch = 256 * K
    m.cuda()
    args = m._argdict(dict(overlap=overlap, blinding=blinding, stacking="avg", batch_size=256 * K))
    specs = m._trigger_specs(args)
    groups = [{"data": torch.from_numpy(synthetic_stream_array(n, seed=1002 + k)[0]).cuda()} for k in range(K)]
    for _ in range(5):
        m._classify_blocks(groups, args, specs)
    torch.cuda.synchronize()
    reps = 60
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m._classify_blocks(groups, args, specs)
    dt = (time.perf_counter() - t0) / reps
    print(f"{name} K = {K}: {dt * 1e6:8.1f} us per call = {dt * 1e6 / K:7.1f} us per 256-window block "
          f"({256 * K / dt / 1e6:.3f} M windows/s, synchronous calls on one context), picks {sum(len(o) for o in out)}")
    m._release()
