"""Loss curves of the HIP training step and of torch (same module on the GPU through PyTorch-ROCm, fp32, Adam)
from the same random initialisation over the same sequence of synthetic batches.

    python tools/train_curve.py [--steps 300] [--batch 64] [--lr 1e-3]

Evidence that the step is the same optimisation, not just the same first gradient: the two curves track each
other (they cannot coincide digit for digit: Adam amplifies fp32 rounding differences)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from oracle.models import PhaseNet as TorchPhaseNet  # noqa: E402  (torch restatement = the comparison baseline)
from volpick_amd import PhaseNet  # noqa: E402
from volpick_amd.synthetic import synthetic_windows  # noqa: E402
from volpick_amd.train import PhaseNetLit, gaussian_labels  # noqa: E402


def batch(B, seed):
    rng = np.random.default_rng(seed)
    x = synthetic_windows(B, 3001, seed=seed)
    x = x - x.mean(-1, keepdims=True)
    x = x / (np.abs(x).max(-1, keepdims=True) + 1e-10)
    p = rng.integers(300, 1500, B).astype(float)
    s = p + rng.integers(200, 1200, B)
    return x.astype(np.float32), gaussian_labels(p, s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--n-batches", type=int, default=16)
    a = ap.parse_args()
    torch.manual_seed(0)
    net = TorchPhaseNet(phases="PSN", norm="peak").cuda().train()
    model = PhaseNet(phases="PSN", norm="peak")
    model.load_state_dict({k: v.detach().cpu().numpy() for k, v in net.state_dict().items()})
    lit = PhaseNetLit(lr=a.lr, max_batch=a.batch, model=model)
    opt = torch.optim.Adam(net.parameters(), lr=a.lr)
    data = [batch(a.batch, 5000 + i) for i in range(a.n_batches)]
    dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in data]
    ours, theirs = [], []
    for k in range(a.steps):
        xd, yd = dev[k % a.n_batches]
        ours.append(lit.training_step({"X": xd, "y": yd}, k))
        for pg in opt.param_groups:
            pg["lr"] = lit.learning_rate(k)
        opt.zero_grad(set_to_none=True)
        pred = net(xd)
        loss = -(yd * torch.log(pred + 1e-5)).mean(-1).sum(-1).mean()
        loss.backward()
        opt.step()
        theirs.append(float(loss.detach()))
    ours, theirs = np.array(ours), np.array(theirs)
    w = max(1, a.steps // 10)
    rel = np.abs(ours - theirs) / theirs
    print(json.dumps({
        "steps": a.steps, "batch": a.batch, "lr": a.lr,
        "loss_first": [float(ours[0]), float(theirs[0])],
        "loss_mean_last_tenth": [float(ours[-w:].mean()), float(theirs[-w:].mean())],
        "max_rel_diff_first_20_steps": float(rel[:20].max()),
        "median_rel_diff": float(np.median(rel)),
        "curve_every_10th": [[float(o), float(t)] for o, t in zip(ours[::10], theirs[::10])],
    }))


if __name__ == "__main__":
    main()
