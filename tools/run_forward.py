#!/usr/bin/env python3
"""Run N forward passes of one model (for rocprofv3 counter collection). usage: run_forward.py phasenet|eqtransformer [n] [plan flags, e.g. 0,0,0,0,0,3]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cls = va.PhaseNet if name == "phasenet" else va.EQTransformer
m = cls.from_pretrained("volpick")
if len(sys.argv) > 3:
    m._plan_flags = tuple(int(v) for v in sys.argv[3].split(","))
m.cuda()
x = torch.from_numpy(synthetic_windows(256, cls.in_samples, seed=1)).cuda()
for _ in range(n):
    y = m._forward_raw(x, preprocess=True)
torch.cuda.synchronize()
print("done", float(y.sum()))
