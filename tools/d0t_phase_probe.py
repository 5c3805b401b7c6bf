#!/usr/bin/env python3
"""Per-phase cycles of the tiled level-0 down path of pn_window_kernel.  Needs a library built with -DD0T_PROBE
(cd volpick_amd/csrc && touch phasenet_fused.hip && make HIPCC="/opt/rocm/bin/hipcc -DD0T_PROBE"): slots 2 .. 14 then hold the phase ends."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import volpick_amd as va
from volpick_amd import _lib
from volpick_amd.synthetic import synthetic_windows
B=256
m = va.PhaseNet.from_pretrained("volpick"); m._plan_flags=(0,2); m.cuda()
x = torch.from_numpy(synthetic_windows(B, 3001, seed=1)).cuda()
for _ in range(3): m._forward_raw(x, preprocess=True)
lib=_lib.load(); n=lib.vp_step_count(m._handle); ms=(C.c_float*n)()
_lib.check(lib.vp_profile_steps(m._handle, B, 200, ms, n))
clk=np.zeros((B,32),np.uint64)
_lib.check(lib.vp_debug_core_clock(m._handle, B, clk.ctypes.data_as(C.c_void_p)))
c=clk.astype(np.int64)
if len(sys.argv) > 1 and sys.argv[1] == "up":  # library built with -DU3T_PROBE: slots 2 .. 14 = the thirteen phases of the up path
    print("up phase: start(23) -> operands loaded, ring zeroed (24):", np.median(c[:,24]-c[:,23]))
    prev=c[:,24]
    for j in range(13):
        print("phase",j, np.median(c[:,2+j]-prev)); prev=c[:,2+j]
    print("whole up phase:", np.median(c[:,28]-c[:,23]))
else:
    print("start->x loaded(19):", np.median(c[:,19]-c[:,0]))
    prev=c[:,19]
    for j in range(7):
        print("phase",j, np.median(c[:,2+j]-prev)); prev=c[:,2+j]
    print("after loop -> stamp21:", np.median(c[:,21]-prev), " down0.down:", np.median(c[:,22]-c[:,21]))
