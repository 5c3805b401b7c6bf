#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded synthetic inputs and the CPU oracle's outputs.

These are SELF-golden vectors (oracle/__init__.py: parity unpinned) — the reference repo
holds no fixtures for this path and SeisBench/ObsPy cannot be imported here, so they pin
the oracle against drift and give the GPU tests a fixed target; they do not prove parity
with SeisBench.  Re-run on a SeisBench-equipped machine with `--seisbench` to upgrade them
to reference-golden (the script then calls sbm.<Model>.from_pretrained('volpick') instead).

    python tools/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import pipeline as OP  # noqa: E402
from oracle.models import load_pretrained  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array, synthetic_windows  # noqa: E402

OUT = ROOT / "tests" / "golden"


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(1)
    for model, T, seed in [("phasenet", 3001, 4101), ("eqtransformer", 6000, 4102)]:
        net = load_pretrained(model)
        x = synthetic_windows(2, T, seed=seed)
        xn = OP.batch_pre(net, torch.from_numpy(x))
        with torch.no_grad():
            y = net(xn)
        y = torch.stack(y, 1).numpy() if isinstance(y, tuple) else y.numpy()
        # stream-level: annotate + picks on a short stream
        n = 3 * T + 700
        data, p_on, s_on = synthetic_stream_array(n, seed=seed + 10, n_events=3)
        kw = dict(overlap=T // 2, blinding=(250, 250), stacking="avg")
        res = OP.classify_array(net, data, **kw)
        ann = {f"ann_{lab}": tr for lab, off, tr in res["annotations"]}
        offs = np.array([off for _, off, _ in res["annotations"]])
        picks = np.array([(("PS".index(ph)), on, off, pk, v) for ph, on, off, pk, v in res["picks"]], dtype=np.float64)
        np.savez_compressed(
            OUT / f"{model}_volpick.npz",
            windows=x, windows_pre=xn.numpy(), forward=y.astype(np.float32),
            stream=data, overlap=kw["overlap"], blinding=np.array(kw["blinding"]), ann_offsets=offs,
            picks=picks.reshape(-1, 5), true_p=p_on, true_s=s_on, **ann)
        print(model, "windows", x.shape, "forward", y.shape, "picks", len(res["picks"]))
    # trigger_onset known answers (hand-checked against the ObsPy rule)
    x = np.array([0, .1, .5, .6, .2, .05, 0, .7, .8, .1, np.nan, .9, .4, .4, .9], dtype=np.float32)
    np.savez(OUT / "trigger_cases.npz", x=x,
             t_03_03=OP.trigger_onset(x, .3, .3), t_03_015=OP.trigger_onset(x, .3, .15),
             t_05_01=OP.trigger_onset(x, .5, .1))


if __name__ == "__main__":
    main()
