#!/usr/bin/env python3
"""Interleaved A/B timing of plan variants in ONE process (HIP events per launch).
usage: ab_steps.py phasenet "0,0,0" "0,0,1" "1,0,0" [--rounds 5]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

model_name = sys.argv[1]
variants = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:] if not a.startswith("--")]
rounds = 5
B = 256
cls = va.PhaseNet if model_name == "phasenet" else va.EQTransformer
lib = _lib.load()
models = []
x = torch.from_numpy(synthetic_windows(B, cls.in_samples, seed=1)).cuda()
for flags in variants:
    m = cls.from_pretrained("volpick")
    m._plan_flags = flags
    m.cuda()
    m._forward_raw(x, preprocess=True)
    models.append(m)
res = {v: [] for v in variants}
names = {}
for r in range(rounds):
    for v, m in zip(variants, models):
        n = lib.vp_step_count(m._handle)
        ms = (C.c_float * n)()
        _lib.check(lib.vp_profile_steps(m._handle, B, 20, ms, n))
        res[v].append(list(ms))
        nm = []
        for i in range(n):
            s = C.c_char_p()
            lib.vp_step_info(m._handle, i, C.byref(s), None)
            nm.append(s.value.decode())
        names[v] = nm
for v in variants:
    a = np.array(res[v]) * 1e3
    med = np.median(a, axis=0)
    print(f"variant {v}: total median {med.sum():8.1f} us   min-of-rounds {a.sum(1).min():8.1f} us")
    if len(med) <= 40:
        for n_, t in zip(names[v], med):
            print(f"     {n_:50s} {t:8.1f} us")
