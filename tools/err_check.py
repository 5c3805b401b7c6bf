"""Max / 99.99th percentile / mean absolute difference between each PhaseNet / EQTransformer plan and the torch-CPU oracle on
512 / 256 synthetic windows (tolerance of the path: 1e-4 on probabilities).  Run on the GPU box: python tools/err_check.py"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import pipeline as OP
from oracle.models import load_pretrained
from volpick_amd import PhaseNet
from volpick_amd.synthetic import synthetic_windows
orc = load_pretrained("phasenet")
x = synthetic_windows(512, 3001, seed=4242)
xn = OP.batch_pre(orc, torch.from_numpy(x))
with torch.no_grad():
    want = orc(xn).double().numpy()
for name, flags in [("one launch, deep layers on bf16 pieces (default)", (0,)), ("one launch, all core layers fp32 MFMA", (0, 0, 0, 0, 0, 3)), ("three launches, VALU level 0", (0, 0, 0, 0, 0, 2)), ("three launches, all MFMA", (0, 0, 0, 0, 0, 1)), ("layer plan", (1, 0))]:
    m = PhaseNet.from_pretrained("volpick"); m._plan_flags = flags; m.cuda()
    got = m(xn).double().numpy()
    e = np.abs(got - want)
    print(f"{name:50s} max|err| {e.max():.2e}  99.99th pct {np.quantile(e, 0.9999):.2e}  mean {e.mean():.2e}")

from volpick_amd import EQTransformer
orc = load_pretrained("eqtransformer")
x = synthetic_windows(256, 6000, seed=4243)
xn = OP.batch_pre(orc, torch.from_numpy(x))
with torch.no_grad():
    want = [w.double().numpy() for w in orc(xn)]
for name, flags in [("EQT default (encoder 1-6, ResCNN, decoder 1-3, tail on bf16 pieces)", (0,)), ("EQT every fused kernel on the fp32 MFMA", (0, 0, 0, 0, 0, 0, 0, 496)),
                    ("EQT only the tail on bf16 pieces", (0, 0, 0, 0, 0, 0, 0, 432)), ("EQT only decoder 1-3 on bf16 pieces", (0, 0, 0, 0, 0, 0, 0, 464)),
                    ("EQT only encoder 3-6 on bf16 pieces", (0, 0, 0, 0, 0, 0, 0, 368)), ("EQT only encoder 1-2 on bf16 pieces", (0, 0, 0, 0, 0, 0, 0, 240)),
                    ("EQT six middle launches", (0, 0, 1)), ("EQT layer plan", (1, 0, 0, 0, 0, 0, 0, 15))]:
    m = EQTransformer.from_pretrained("volpick"); m._plan_flags = flags; m.cuda()
    got = [g.double().cpu().numpy() for g in m(xn.cuda())]
    e = np.concatenate([np.abs(g - w).ravel() for g, w in zip(got, want)])
    print(f"{name:34s} max|err| {e.max():.2e}  99.99th pct {np.quantile(e, 0.9999):.2e}  mean {e.mean():.2e}")
