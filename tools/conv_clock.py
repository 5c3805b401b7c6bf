#!/usr/bin/env python3
"""Phases of one workgroup (tile 1, window 7) of every EQTransformer conv launch, in shader-clock cycles:
load (global -> LDS), MFMA loop, epilogue (debug plan flag bit 1; stamps of conv_mfma_kernel)."""
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
for _ in range(3):
    m._forward_raw(x, preprocess=True)
lib = _lib.load()
clk = np.zeros((B + 16, 32), np.uint64)  # [max_batch][32] window slots, then 8 stamps per conv layer
_lib.check(lib.vp_debug_core_clock(m._handle, B + 16, clk.ctypes.data_as(C.c_void_p)))
conv = clk.reshape(-1)[B * 32:].reshape(-1, 8).astype(np.int64)
n_steps = lib.vp_step_count(m._handle)
names = []
for i in range(n_steps):
    name, fl = C.c_char_p(), C.c_double()
    lib.vp_step_info(m._handle, i, C.byref(name), C.byref(fl))
    names.append(name.value.decode())
print("conv layer (plan order)      load     mfma   stage/store phases ...   total")
for li, row in enumerate(conv):
    if row[0] == 0:
        continue
    d = np.diff(row[:6])
    d = d[(row[1:6] != 0)]
    print(f"conv #{li:2d}  " + " ".join(f"{v:8d}" for v in d) + f"   {row[:6][row[:6] != 0].max() - row[0]:8d}")
print("steps:", names)
