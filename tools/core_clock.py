#!/usr/bin/env python3
"""Per-layer cycle shares of the fused PhaseNet core kernel (debug plan, shader-clock stamps)."""
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
m = va.PhaseNet.from_pretrained("volpick")
# optional argument: plan flags "a,b,c,..." (the clock-stamp bit of [1] is added), e.g. "0,0,0,0,0,8" = level-0 down path on the VALU
_f = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 0]
_f += [0] * (2 - len(_f))
_f[1] |= 2
m._plan_flags = tuple(_f)
print("plan flags", m._plan_flags)
m.cuda()
x = torch.from_numpy(synthetic_windows(B, 3001, seed=1)).cuda()
for _ in range(3):
    m._forward_raw(x, preprocess=True)
# steady state: 200 back-to-back launches of every step, stamps of the last launch survive
lib = _lib.load()
n_steps = n = lib.vp_step_count(m._handle)
ms = (C.c_float * n)()
_lib.check(lib.vp_profile_steps(m._handle, B, 200, ms, n))
print("step ms:", [round(v * 1e3, 1) for v in ms])
clk = np.zeros((B, 32), np.uint64)
_lib.check(_lib.load().vp_debug_core_clock(m._handle, B, clk.ctypes.data_as(C.c_void_p)))
d = np.diff(clk[:, :15].astype(np.int64), axis=1)
names = ["load d0 / level-0 down", "d1same", "d1down", "d2same", "d2down", "d3same", "d3down", "d4same", "u0T", "u0same", "u1T",
         "u1same", "u2T", "u2same"]
med = np.median(d, axis=0)
tot = med.sum()
for n, c in zip(names, med):
    print(f"{n:10s} {c:10.0f} cycles  {100 * c / tot:5.1f} %")
wall = (clk[:, 17] - clk[:, 16]).astype(np.float64) / 100e6
cyc = (clk[:, 14] - clk[:, 0]).astype(np.float64)
print("in-kernel wall us (median)", np.median(wall) * 1e6, " shader clock GHz (median)", np.median(cyc / wall) / 1e9)
t_s = (clk[:, 16] - clk[:, 16].min()).astype(np.float64) / 100.0  # us after the first workgroup's start
t_e = (clk[:, 17] - clk[:, 16].min()).astype(np.float64) / 100.0
print("workgroup starts after the first (us): median %.2f  p90 %.2f  max %.2f;  ends: min %.2f  median %.2f  max %.2f;  wall min / max %.2f / %.2f"
      % (np.median(t_s), np.percentile(t_s, 90), t_s.max(), t_e.min(), np.median(t_e), t_e.max(), wall.min() * 1e6, wall.max() * 1e6))
if n_steps == 1:  # whole-network kernel: slot 1 - slot 0 is the level-0 down phase; 18..22 and 23..28 detail the two level-0 phases
    dphase = np.diff(clk[:, 18:23].astype(np.int64), axis=1)
    print("level-0 down phase median cycles: load x %d  inc (tiled form: its first phase) %d  down0.same (tiled form: the other phases of inc + down0.same, skip stores) %d  down0.down (MFMA) %d" % tuple(np.median(dphase, axis=0)))
    sub = clk[:, [18, 29, 30, 31, 19]].astype(np.int64)
    if (sub[:, 1:4] > 0).all():
        print("load x in detail (wave 0's stamps): window read + partial reductions %d  barrier + 3-lane finish + barrier %d  "
              "normalise + store x image %d  closing barrier %d" % tuple(np.median(np.diff(sub, axis=1), axis=0)))
    if (clk[:, 25] == 0).all():  # U3T form: stamps 23 (start), 24 (operands there, ring zeroed), 28 (last tile stored)
        print("level-0 up phase (tiled, bf16 matrix cores) median cycles: prepare %d  thirteen phases of 256 samples %d"
              % (np.median(clk[:, 24].astype(np.int64) - clk[:, 23].astype(np.int64)), np.median(clk[:, 28].astype(np.int64) - clk[:, 24].astype(np.int64))))
    else:
        uphase = np.diff(clk[:, 23:29].astype(np.int64), axis=1)
        print("level-0 up phase median cycles: load skip rows %d  up3.same(skip) %d  up3.convT (MFMA) %d  up3.same(convT) %d  1x1+softmax+store %d"
              % tuple(np.median(uphase, axis=0)))
    whole = (clk[:, 28] - clk[:, 0]).astype(np.float64)
    print("whole window median cycles %d (down %d, core %d, up %d)" % (np.median(whole), np.median(clk[:, 1] - clk[:, 0]),
          np.median(clk[:, 14] - clk[:, 1]), np.median(clk[:, 28] - clk[:, 14])))
else:
    u = np.diff(clk[:, 18:26].astype(np.int64), axis=1)
    print("up3 VALU form (tile 1) median cycles: load skip rows %d  same(skip) %d  up2.same regs->LDS+barrier %d  convT (MFMA) %d  "
          "barrier %d  same(convT) %d  1x1+softmax+store %d" % tuple(np.median(u, axis=0)))
    w0, w1 = clk[:, 26].astype(np.float64) / 100e6, clk[:, 27].astype(np.float64) / 100e6
    print("up3 VALU form (tile 1 of every window): workgroup wall us median %.2f; first start -> last end %.2f us; start spread %.2f us"
          % (np.median(w1 - w0) * 1e6, (w1.max() - w0.min()) * 1e6, (w0.max() - w0.min()) * 1e6))
print("total", tot, "cycles; window span min/med/max", (clk[:, 14] - clk[:, 0]).min(), np.median(clk[:, 14] - clk[:, 0]),
      (clk[:, 14] - clk[:, 0]).max())
