#!/usr/bin/env python3
"""Phase cycles (load / MFMA / stage / store) of one probe workgroup of every conv_mfma_kernel launch."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from tests.emulator import flat_weights, plan_conv  # noqa: E402
from volpick_amd import _lib  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "eqtransformer"
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
m = cls.from_pretrained("volpick")
m._plan_flags = (1 if name == "phasenet" else 0, 2)
m.cuda()
B = 256
x = torch.from_numpy(synthetic_windows(B, cls.in_samples, seed=1)).cuda()
for _ in range(5):
    m._forward_raw(x, preprocess=True)
lib = _lib.load()
clk = np.zeros((64, 8), np.uint64)
n = lib.vp_debug_conv_clock(m._handle, clk.ctypes.data_as(C.c_void_p), 64)
names = []
w = m._weights
i = 0
while True:
    L = plan_conv(m._kind, w, i)
    if L is None:
        break
    names.append(L["name"])
    i += 1
print(f"{'layer':16s} {'load':>8s} {'mfma':>8s} {'stage':>8s} {'store':>8s} {'total':>8s}  cycles (probe workgroup)")
for i in range(min(n, len(names))):
    d = np.diff(clk[i, :5].astype(np.int64))
    if clk[i, 0] == 0:
        continue
    print(f"{names[i]:16s} {d[0]:8d} {d[1]:8d} {d[2]:8d} {d[3]:8d} {d.sum():8d}")
