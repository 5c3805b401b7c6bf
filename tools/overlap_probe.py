#!/usr/bin/env python3
"""Do two EQTransformer launches share the chip when both are ready?  Two handles (two HIP streams), two host
threads, each looping ONE launch of the plan 200 times (vp_profile_one_step); the per-launch time of each loop alone and
with the other loop running beside it.  Co-resident kernels keep their solo times, kernels that exclude each other
(LDS / VGPR budget of a CU) take about the sum.
usage: overlap_probe.py"""
import ctypes as C
import sys
import threading
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

B, ITERS = 256, 200
lib = _lib.load()
x = torch.from_numpy(synthetic_windows(B, 6000, seed=1)).cuda()
ms_ = []
for _ in range(2):
    m = va.EQTransformer.from_pretrained("volpick")
    m.cuda()
    m._forward_raw(x, preprocess=True)
    ms_.append(m)
n = lib.vp_step_count(ms_[0]._handle)
names = []
for i in range(n):
    s = C.c_char_p()
    lib.vp_step_info(ms_[0]._handle, i, C.byref(s), None)
    names.append(s.value.decode().split(" ")[0])


def loop(m, idx, out, k):
    t = C.c_float()
    _lib.check(lib.vp_profile_one_step(m._handle, B, ITERS, idx, C.byref(t)))
    out[k] = t.value * 1e3


solo = {}
for i in range(n):
    o = [0.0]
    loop(ms_[0], i, o, 0)
    solo[i] = o[0]
    print(f"solo {names[i]:14s} {o[0]:7.1f} us")
print("pairs: time per launch of each loop while both run (sum of the solo times in brackets)")
for i in range(n):
    for j in range(i, n):
        o = [0.0, 0.0]
        ts = [threading.Thread(target=loop, args=(ms_[0], i, o, 0)), threading.Thread(target=loop, args=(ms_[1], j, o, 1))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print(f"{names[i]:14s} || {names[j]:14s}: {o[0]:7.1f} / {o[1]:7.1f} us   (solo {solo[i]:6.1f} + {solo[j]:6.1f} = {solo[i] + solo[j]:6.1f})")
