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
c = clk.astype(np.int64)
d = np.diff(c[:, :7], axis=1)
names = ["bilstm.0", "bilstm.1", "bilstm.2", "transformer_d0", "transformer_d", "pick branches"]
for n, v in zip(names, np.median(d, axis=0)):
    print(f"{n:16s} {v:9.0f} cycles")
print("total", np.median(c[:, 6] - c[:, 0]))


def seg(title, slots, labels):
    print(title)
    for (a, b), n in zip(zip(slots[:-1], slots[1:]), labels):
        print(f"  {n:34s} {np.median(c[:, b] - c[:, a]):9.0f}")


seg("bilstm.0 (stage start = slot 0)", [0, 8, 9, 10, 1],
    ["load x", "input projection", "47 recurrent steps", "1x1 conv + BN + store"])
seg("transformer_d (stage start = slot 4)", [4, 11, 21, 20, 22, 23, 24, 13, 14, 15, 16, 5],
    ["-", "q/k projection + exp (MFMA)", "requests for the next stage issued", "e = Wa . tanh", "softmax rows", "a . x (MFMA) + LN1", "-",
     "FF1 16->128 (MFMA)", "FF2 128->16 (MFMA)", "LN2 + store (3 waves)", "closing barrier"])
seg("pick branches (stage start = slot 5)", [5, 17, 18, 19],
    ["-", "weights + input projection (P and S, MFMA)", "47 recurrent steps (P | S)"])
seg("  S branch attention (band 3)", [27, 28, 29, 30], ["e", "softmax", "a . x (MFMA)"])
print(f"  per branch (q/k + attention + store) ~ {np.median(c[:, 6] - c[:, 19]) / 2:9.0f}")
