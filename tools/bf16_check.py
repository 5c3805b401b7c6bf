"""bf16-storage training step against torch autograd on the CPU oracle module: loss, every z / gz tensor, every
parameter gradient -- relative errors in units of the tensor's max.  Columns: the fp32 mode against fp32 autograd; the
bf16 mode against fp32 autograd (what the storage format itself costs); the bf16 mode against autograd with the same
rounding points (oracle/bf16_emulation.py: what is left is summation order).

    python tools/bf16_check.py [B]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from test_gpu_train import make_batch, rel, torch_step  # noqa: E402

from oracle.bf16_emulation import bf16_storage, round_bf16  # noqa: E402
from oracle.models import load_pretrained  # noqa: E402
from volpick_amd import PhaseNet  # noqa: E402
from volpick_amd.train import PhaseNetTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x, y = make_batch(B, 7)
net = load_pretrained("phasenet")
ref32 = torch_step(net, x, y)
with bf16_storage(load_pretrained("phasenet")) as net16:
    ref16 = torch_step(net16, x, y)
rows = {}
for col, dt, (want_loss, grads, z, gz, pred) in (("fp32", "fp32", ref32), ("bf16", "bf16", ref32), ("bf16/emu", "bf16", ref16)):
    tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=max(B, 8), dtype=dt)
    loss = tr.step(x, y, lr=0.0, update=False)
    t = tr.tensors(B)
    g = tr.gradients()
    rows[col] = {"loss": abs(loss - want_loss) / abs(want_loss), "pred": float(np.abs(tr.predictions(B) - pred).max())}
    for name in z:
        rows[col][name + ".z"] = rel(t[name + ".z"], z[name])
        rows[col][name + ".gz"] = rel(t[name + ".gz"], gz[name])
    for k, w in grads.items():
        rows[col]["grad " + k] = rel(g[k], w)
    tr.close()
print(f"{'':40s} {'fp32':>10s} {'bf16':>10s} {'bf16/emu':>10s}")
for k in rows["fp32"]:
    print(f"{k:40s} {rows['fp32'][k]:10.2e} {rows['bf16'][k]:10.2e} {rows['bf16/emu'][k]:10.2e}")
