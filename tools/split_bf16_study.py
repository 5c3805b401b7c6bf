#!/usr/bin/env python3
"""Parity study for a SPLIT-operand bf16 matrix path ("known gaps"): what happens to the probabilities if
every convolution sees its input activations with 16 significant bits (x = hi + lo, two bfloat16 pieces; the dropped
remainder is 2^-17 relative) while the weights stay exact (w = hi + mid + lo, three pieces) and products accumulate in
fp32?  CPU only: the oracle modules with a rounding hook in front of every Conv1d / ConvTranspose1d, on the bench's
synthetic windows.  Prints max / 99.9th percentile |dp| against the unmodified fp32 forward pass, also for 8 + 8 + 0
(activations as ONE bfloat16, i.e. plain bf16 inputs) as a yardstick.

    python tools/split_bf16_study.py [n_windows]
"""
import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import pipeline as OP  # noqa: E402
from oracle.models import load_pretrained  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402


def pieces(x, n):
    out, r = torch.zeros_like(x), x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out, r = out + p, r - p
    return out


def run(name, T, n_win, n_pieces):
    net = load_pretrained(name).eval()
    x = torch.from_numpy(synthetic_windows(n_win, T, seed=1002))
    xn = OP.batch_pre(net, x)
    with torch.no_grad():
        ref = net(xn)
    hooks = [m.register_forward_pre_hook(lambda m, a: (pieces(a[0], n_pieces),) + tuple(a[1:]))
             for m in net.modules() if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d))]
    with torch.no_grad():
        got = net(xn)
    for h in hooks:
        h.remove()
    ref = torch.stack(ref, 1) if isinstance(ref, tuple) else ref
    got = torch.stack(got, 1) if isinstance(got, tuple) else got
    d = (got - ref).abs().numpy().ravel()
    return float(d.max()), float(np.percentile(d, 99.9)), float(d.mean())


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.set_num_threads(8)
    for name, T in (("phasenet", 3001), ("eqtransformer", 6000)):
        for k in (1, 2, 3):
            mx, p999, mean = run(name, T, n, k)
            print(f"{name:14s} conv inputs as {k} bfloat16 piece(s) ({8 * k} significant bits): max |dp| {mx:.2e}  p99.9 {p999:.2e}  mean {mean:.2e}")
