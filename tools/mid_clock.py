#!/usr/bin/env python3
"""Cycles per stage of the fused EQT middle kernel (debug plan flag bit 1: shader-clock stamps per window)."""
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
clk = np.zeros((B, 32), np.uint64)
_lib.check(_lib.load().vp_debug_core_clock(m._handle, B, clk.ctypes.data_as(C.c_void_p)))
c = clk.reshape(-1)[: B * 8].reshape(B, 8).astype(np.int64)
d = np.diff(c[:, :7], axis=1)
names = ["bilstm.0", "bilstm.1", "bilstm.2", "transformer_d0", "transformer_d", "pick branches"]
for n, v in zip(names, np.median(d, axis=0)):
    print(f"{n:16s} {v:9.0f} cycles")
print("total", np.median(c[:, 6] - c[:, 0]))
