#!/usr/bin/env python3
"""Pick parity over many streams: the HIP path (picks through vp_classify) against the CPU oracle on the same
samples, for both models and a list of seeds.  Per model: picks compared, streams whose pick lists differ in length or
phase order, max |dt| (samples) of the peaks, max |dvalue|, and max |dp| of the stacked probability rows.

    python tools/parity_sweep.py [n_seeds] [windows_per_stream]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
from oracle import pipeline as OP  # noqa: E402
from oracle.models import load_pretrained  # noqa: E402
from volpick_amd.synthetic import synthetic_stream_array  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 96
torch.set_num_threads(16)
for name, cls, overlap, blinding in (("phasenet", va.PhaseNet, 1500, (0, 0)), ("eqtransformer", va.EQTransformer, 5500, (500, 500))):
    model = cls.from_pretrained("volpick").cuda()
    net = load_pretrained(name)
    T = net.in_samples
    n = T + (T - overlap) * (n_win - 1)
    tot = bad = 0
    max_dt = 0
    max_dv = max_dp = 0.0
    for seed in range(n_seeds):
        data, _, _ = synthetic_stream_array(n, seed=7000 + seed, n_events=max(3, n // 6000))
        ref = OP.classify_array(net, data, overlap=overlap, blinding=blinding, batch_size=256)
        want = sorted(ref["picks"])
        args = model._argdict(dict(overlap=overlap, blinding=blinding, stacking="avg", batch_size=256))
        specs = [s for s in model._trigger_specs(args) if s[1] != "Detection"]
        got, _ = model._classify_block(data, args, specs)
        got = sorted((specs[si][1], on, off, pk, v) for si, on, off, pk, v in got)
        rows = model._annotate_block(data, args)[0].cpu().numpy()
        order = list(model.labels) if name == "phasenet" else ["Detection", "P", "S"]
        for label, off, tr in ref["annotations"]:
            tr = np.asarray(tr)
            mine = rows[order.index(label), off:off + len(tr)]
            m = np.isfinite(tr) & np.isfinite(mine)
            assert m.sum() > 0.99 * len(tr), (label, m.sum(), len(tr))
            max_dp = max(max_dp, float(np.abs(tr[m] - mine[m]).max()))
        if len(got) != len(want) or any(g[0] != w[0] for g, w in zip(got, want)):
            bad += 1
            continue
        tot += len(got)
        max_dt = max([max_dt] + [abs(g[3] - w[3]) for g, w in zip(got, want)])
        max_dv = max([max_dv] + [abs(g[4] - w[4]) for g, w in zip(got, want)])
    print(f"{name:14s} {n_seeds} streams x {n_win} windows: {tot} picks compared, {bad} streams with differing pick lists, "
          f"max |dt| {max_dt} samples, max |dvalue| {max_dv:.2e}, max |dp| of the stacked rows {max_dp:.2e}")
