#!/usr/bin/env python3
"""Cycles per stage of eqt_dec03_kernel<true>: waves 0 and 4 of workgroup 0 (the two waves of a SIMD), the first rows of the
workgroup.  Needs a -DD3_CLOCK=1 build (make BUILD=build_x TARGET=../../exp/lib_d3c.so EXTRA=-DD3_CLOCK=1; VOLPICK_HIP_LIB)."""
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
c = clk.astype(np.int64)
names = ["park + barrier", "stage 0 (fp32 MFMA)", "barrier", "stage 1", "request a2 + barrier", "stage 2", "edge samples", "request a3, next row + barrier", "stage 3"]
for w in (0, 1):
    s = c[w * 128: w * 128 + 30]
    print(f"wave {4 * w}:")
    for r in range(3):
        k = s[10 * r: 10 * r + 10]
        print(f"  row {r}: " + "  ".join(f"{n} {v}" for n, v in zip(names, np.diff(k))) + f"   = {k[9] - k[0]}")
