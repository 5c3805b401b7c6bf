#!/usr/bin/env python3
"""Where does the wall time of a training step go?  tr.step() (torch event + stream wait per step) against the bare
vp_train_step call on the same device pointers, and the host time of the call alone (enqueue only)."""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tools.bench_train import make_batch  # noqa: E402
from volpick_amd import PhaseNet, _lib  # noqa: E402
from volpick_amd.train import PhaseNetTrainer  # noqa: E402

B = 512
x, y = make_batch(B)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=B, dtype="bf16")
lib = _lib.load()


def bare():
    _lib.check(lib.vp_train_step(tr._h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), _lib.VP_MEM_DEVICE, B, 1e-4, 1, None))


xd2, yd2 = torch.roll(xd, 1, 0).contiguous(), torch.roll(yd, 1, 0).contiguous()
torch.cuda.synchronize()
flip = [0]


def alternating():  # a fresh tensor pair every step: what a loader hands over (bench.py's train leg)
    flip[0] ^= 1
    tr.step(*((xd, yd), (xd2, yd2))[flip[0]], 1e-4, want_loss=False)


for name, fn in (("tr.step", lambda: tr.step(xd, yd, 1e-4, want_loss=False)), ("bare vp_train_step", bare),
                 ("tr.step promised", lambda: tr.step(xd, yd, 1e-4, want_loss=False, inputs_unchanged=True)),
                 ("tr.step alternating", alternating), ("tr.step", lambda: tr.step(xd, yd, 1e-4, want_loss=False))):
    for _ in range(5):
        fn()
    tr.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(30):
        t1 = time.perf_counter()
        fn()
        host += time.perf_counter() - t1
    t_enq = time.perf_counter() - t0
    tr.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:20s}: wall {dt / 30 * 1e3:.3f} ms/step, host enqueue {host / 30 * 1e3:.3f} ms/step (loop {t_enq / 30 * 1e3:.3f})", flush=True)
