#!/usr/bin/env python3
"""Every step of a model timed IN the pipeline (vp_profile_step_in_pipeline) next to its back-to-back time.
usage: pipeline_steps.py phasenet|eqtransformer"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

B = 256
cls = va.PhaseNet if sys.argv[1] == "phasenet" else va.EQTransformer
m = cls.from_pretrained("volpick").cuda()
x = torch.from_numpy(synthetic_windows(B, cls.in_samples, seed=1)).cuda()
m._forward_raw(x, preprocess=True)
lib = _lib.load()
n = lib.vp_step_count(m._handle)
b2b = (C.c_float * n)()
_lib.check(lib.vp_profile_steps(m._handle, B, 20, b2b, n))
tot_p = tot_b = 0.0
for i in range(n):
    name, ms = C.c_char_p(), C.c_float()
    lib.vp_step_info(m._handle, i, C.byref(name), None)
    _lib.check(lib.vp_profile_step_in_pipeline(m._handle, B, 30, i, C.byref(ms)))
    tot_p += ms.value
    tot_b += b2b[i]
    print(f"{name.value.decode():48s} in pipeline {ms.value * 1e3:7.1f} us   back to back {b2b[i] * 1e3:7.1f} us")
print(f"{'sum':48s} in pipeline {tot_p * 1e3:7.1f} us   back to back {tot_b * 1e3:7.1f} us")
