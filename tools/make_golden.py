#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded synthetic inputs and the CPU oracle's outputs.

These are SELF-golden vectors (oracle/__init__.py: parity unpinned) -- the reference repo holds no fixtures for this
path and SeisBench/ObsPy cannot be imported in the build image, so they pin the oracle against drift and give the GPU
tests a fixed target; they do not prove parity with SeisBench.

    python tools/make_golden.py                 (here: oracle -> tests/golden/{phasenet,eqtransformer}_volpick.npz)
    python tools/make_golden.py --seisbench     (on a machine WITH seisbench + obspy installed)

`--seisbench` is the upgrade path to reference-golden vectors: the same seeded inputs go through
`seisbench.models.<Model>` objects that carry the released weights (read from volpick_amd/weights/*.npz, which are
bit-identical to Final_models/**/*.pt.v1) -- `annotate_batch_pre`, the forward pass and `classify` on an ObsPy stream --
and the arrays are written under the same keys, so every test that reads the fixtures then checks the HIP path and the
oracle against SeisBench itself.  That branch has never run here (no seisbench in this image, no network); it is kept
short and literal so that the first person with such a machine can read what it does before trusting it.
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


def seisbench_outputs(model, x, data, kw):
    """The same quantities as the oracle branch of main(), computed by SeisBench (see the module docstring)."""
    import json

    import obspy
    import seisbench.models as sbm

    meta = json.loads((ROOT / "volpick_amd" / "weights" / model / "volpick.json").read_text())
    cls = sbm.PhaseNet if model == "phasenet" else sbm.EQTransformer
    net = cls(**meta["model_args"])
    with np.load(ROOT / "volpick_amd" / "weights" / model / "volpick.npz") as z:
        net.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    net.eval()
    xn = net.annotate_batch_pre(torch.from_numpy(x.copy()), {})
    with torch.no_grad():
        y = net(xn)
    y = torch.stack(y, 1).numpy() if isinstance(y, tuple) else y.numpy()
    st = obspy.Stream([obspy.Trace(data[i].copy(), dict(network="XX", station="GOLD", channel=f"HH{c}", sampling_rate=100.0))
                       for i, c in enumerate("ZNE")])
    thr = dict(P_threshold=meta["default_args"]["P_threshold"], S_threshold=meta["default_args"]["S_threshold"])
    ann = net.annotate(st, overlap=kw["overlap"], blinding=list(kw["blinding"]), stacking=kw["stacking"])
    out = net.classify(st, overlap=kw["overlap"], blinding=list(kw["blinding"]), stacking=kw["stacking"], **thr)
    picks = getattr(out, "picks", out)
    t0 = st[0].stats.starttime
    labels = [tr.stats.channel.split("_")[-1] for tr in ann]
    annd = {f"ann_{lab}": np.asarray(tr.data, np.float32) for lab, tr in zip(labels, ann)}
    offs = np.array([int(round((tr.stats.starttime - t0) * 100)) for tr in ann])
    rows = [("PS".index(p.phase), round((p.start_time - t0) * 100), round((p.end_time - t0) * 100),
             round((p.peak_time - t0) * 100), p.peak_value) for p in picks if p.phase in "PS"]
    rows.sort(key=lambda r: (r[1], r[0]))
    return xn.numpy(), y.astype(np.float32), annd, offs, np.array(rows, dtype=np.float64).reshape(-1, 5)


def main():
    use_seisbench = "--seisbench" in sys.argv[1:]
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(1)
    for model, T, seed in [("phasenet", 3001, 4101), ("eqtransformer", 6000, 4102)]:
        x = synthetic_windows(2, T, seed=seed)
        n = 3 * T + 700
        data, p_on, s_on = synthetic_stream_array(n, seed=seed + 10, n_events=3)
        kw = dict(overlap=T // 2, blinding=(250, 250), stacking="avg")
        if use_seisbench:
            xn, y, ann, offs, picks = seisbench_outputs(model, x, data, kw)
        else:
            net = load_pretrained(model)
            xn = OP.batch_pre(net, torch.from_numpy(x))
            with torch.no_grad():
                y = net(xn)
            y = torch.stack(y, 1).numpy() if isinstance(y, tuple) else y.numpy()
            xn = xn.numpy()
            res = OP.classify_array(net, data, **kw)  # stream-level: annotate + picks on a short stream
            ann = {f"ann_{lab}": tr for lab, off, tr in res["annotations"]}
            offs = np.array([off for _, off, _ in res["annotations"]])
            picks = np.array([(("PS".index(ph)), on, off, pk, v) for ph, on, off, pk, v in res["picks"]],
                             dtype=np.float64).reshape(-1, 5)
        np.savez_compressed(
            OUT / f"{model}_volpick.npz",
            windows=x, windows_pre=xn, forward=y.astype(np.float32),
            stream=data, overlap=kw["overlap"], blinding=np.array(kw["blinding"]), ann_offsets=offs,
            picks=picks, true_p=p_on, true_s=s_on, source=np.array("seisbench" if use_seisbench else "oracle"), **ann)
        print(model, "windows", x.shape, "forward", y.shape, "picks", len(picks), "source:",
              "seisbench" if use_seisbench else "oracle")
    # trigger_onset known answers (hand-checked against the ObsPy rule)
    x = np.array([0, .1, .5, .6, .2, .05, 0, .7, .8, .1, np.nan, .9, .4, .4, .9], dtype=np.float32)
    np.savez(OUT / "trigger_cases.npz", x=x,
             t_03_03=OP.trigger_onset(x, .3, .3), t_03_015=OP.trigger_onset(x, .3, .15),
             t_05_01=OP.trigger_onset(x, .5, .1))


if __name__ == "__main__":
    main()
