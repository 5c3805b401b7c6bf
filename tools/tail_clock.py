#!/usr/bin/env python3
"""Cycles per phase of the fused decoder tail (eqt_tail3_kernel; eqt_tail_kernel with `fp32` as the second argument):
debug plan flag bit 1 = shader-clock stamps per workgroup.  usage: tail_clock.py [sustain_rounds] [fp32]"""
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
FP32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
TILES = 9 if FP32 else 15  # tiles per workgroup at 256 windows (2000- / 1200-sample tiles)
m._plan_flags = (0, 2, 0, 0, 0, 0, 0, 64 if FP32 else 0)
m.cuda()
x = torch.from_numpy(synthetic_windows(B, 6000, seed=1)).cuda()
for _ in range(5):
    m._forward_raw(x, preprocess=True)
if len(sys.argv) > 1 and int(sys.argv[1]) > 0:  # sustained: the stamps of the LAST of many back-to-back launches of every step (DVFS settles)
    lib = _lib.load()
    n = lib.vp_step_count(m._handle)
    ms = (C.c_float * n)()
    for _ in range(int(sys.argv[1])):
        _lib.check(lib.vp_profile_steps(m._handle, B, 20, ms, n))
    print("sustained: last step (fused tail) %.1f us per launch by HIP events" % (ms[n - 1] * 1e3))
clk = np.zeros((B, 32), np.uint64)
_lib.check(_lib.load().vp_debug_tail_clock(m._handle, B, clk.ctypes.data_as(C.c_void_p)))
c = clk.astype(np.int64)
names = ["park + barrier", "stage 4", "stage 5", "stage 6", "heads"]
for t in range(4):
    s = c[:, 6 * t:6 * t + 6]
    d = np.diff(s, axis=1)
    print(f"tile {t}: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(names, np.median(d, axis=0))) +
          f"   total {np.median(s[:, 5] - s[:, 0]):7.0f}")
    if t < 3:
        print(f"        gap to the next tile {np.median(c[:, 6 * t + 6] - s[:, 5]):7.0f}")
wall = (c[:, 31] - c[:, 30]) * 10e-9  # 100 MHz ticks
cyc4 = c[:, 23] - c[:, 0]
print(f"kernel wall time per workgroup: median {np.median(wall) * 1e6:.1f} us ({TILES} tiles); first four tiles "
      f"{np.median(cyc4):.0f} cycles")
t_start = (c[:, 30] - c[:, 30].min()) * 10e-9 * 1e6
t_end = (c[:, 31] - c[:, 30].min()) * 10e-9 * 1e6
print(f"workgroup start after the first one (us): median {np.median(t_start):.1f}  p90 {np.percentile(t_start, 90):.1f}  max {t_start.max():.1f}")
print(f"workgroup end   after the first start (us): min {t_end.min():.1f}  median {np.median(t_end):.1f}  max {t_end.max():.1f}")
print(f"workgroup wall (us): min {wall.min() * 1e6:.1f}  p10 {np.percentile(wall, 10) * 1e6:.1f}  p90 {np.percentile(wall, 90) * 1e6:.1f}  max {wall.max() * 1e6:.1f}")
print(f"  => shader clock ~ {np.median(cyc4 / ((wall * 4 / TILES))) / 1e9:.2f} GHz if the tiles take equal time")
