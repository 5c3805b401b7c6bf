#!/usr/bin/env python3
"""Cycles per phase of the fused ResCNN kernel (eqt_res3t_kernel; eqt_res3_kernel with VOLPICK_PLAN_FLAGS=0,0,0,0,0,0,0,8192):
debug plan flag bit 1 = shader-clock stamps of workgroup 0 (a -DR3_CLOCK=1 build of the library, VOLPICK_HIP_LIB).
`res3_clock.py waves` reads a -DR3_CLOCK=2 build: wave 0 of EACH half (team) of workgroup 0, with stamps inside the phases (MFMAs
issued / last epilogue done / barrier passed).  Builds: make BUILD=build_x TARGET=../../exp/lib_x.so EXTRA=-DR3_CLOCK=2"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

B = 256
m = va.EQTransformer.from_pretrained("volpick")
m._plan_flags = (0, 2)
m.cuda()
x = torch.from_numpy(synthetic_windows(B, 6000, seed=1)).cuda()
for _ in range(5):
    m._forward_raw(x, preprocess=True)
clk = np.zeros(64 * 8, np.uint64)
lib = _lib.load()
lib.vp_debug_conv_clock(m._handle, clk.ctypes.data_as(C.c_void_p), 64)
if len(sys.argv) > 1 and sys.argv[1] == "waves":
    c = clk.astype(np.int64)
    t0 = c[0]
    print("per conv and team: cycles to [MFMAs issued, epilogue done, barrier passed], measured from the barrier before the conv")
    for team in (0, 1):
        w = c[team * 128: team * 128 + 4 + 42 + 1]
        print(f"team {team}: prologue stamps at", list(w[:4] - t0))
        for i in range(7):
            for j in (0, 1):
                k = 4 + 6 * i + 3 * j
                base = w[k - 1]
                print(f"  block {i} conv{j + 1}: start {base - t0:7d}   mac {w[k] - base:6d}  epilogue {w[k + 1] - w[k]:6d}  barrier {w[k + 2] - w[k + 1]:6d}   = {w[k + 2] - base:6d}")
        print(f"  store {w[46] - w[45]:6d}   total {w[46] - w[0]}")
    sys.exit(0)
c = clk.astype(np.int64)[:20]
names = ["requests issued (+ L2 warm-up)", "zero fill", "load x, act (split)"] + [f"block {i} conv{j}" for i in range(7) for j in (1, 2)] + ["store"]
for n, v in zip(names, np.diff(c)):
    print(f"{n:24s} {v:8d}")
print("total", c[len(names)] - c[0])
