"""EQTransformer plan variants against the torch-CPU oracle run in FLOAT64 (and in fp32): is a bf16-piece kernel less accurate
than its fp32-MFMA form, or only differently rounded?  64 synthetic windows.  Run on the GPU box: python tools/err_check_f64.py"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import pipeline as OP
from oracle.models import load_pretrained
from volpick_amd import EQTransformer
from volpick_amd.synthetic import synthetic_windows
orc = load_pretrained("eqtransformer")
x = synthetic_windows(64, 6000, seed=4243)
xn = OP.batch_pre(orc, torch.from_numpy(x))
with torch.no_grad():
    w32 = torch.stack(orc(xn), 1).double().numpy()
    orc64 = load_pretrained("eqtransformer").double()
    w64 = torch.stack(orc64(xn.double()), 1).numpy()
print("oracle fp32 vs fp64: max %.2e mean %.2e" % (np.abs(w32 - w64).max(), np.abs(w32 - w64).mean()))
for name, flags in [("default (bf16 pieces)", (0,)), ("all fp32 MFMA", (0, 0, 0, 0, 0, 0, 0, 496)), ("only encoder 3-6 bf16", (0, 0, 0, 0, 0, 0, 0, 368)), ("only encoder 1-2 bf16", (0, 0, 0, 0, 0, 0, 0, 240)), ("layer plan", (1, 0, 0, 0, 0, 0, 0, 15))]:
    m = EQTransformer.from_pretrained("volpick"); m._plan_flags = flags; m.cuda()
    got = torch.stack(list(m(xn.cuda())), 1).double().cpu().numpy()
    e64 = np.abs(got - w64); e32 = np.abs(got - w32)
    print("%-24s vs fp64 oracle: max %.2e mean %.2e   vs fp32 oracle: max %.2e mean %.2e" % (name, e64.max(), e64.mean(), e32.max(), e32.mean()))
