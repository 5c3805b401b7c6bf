#!/usr/bin/env python3
"""Why does bench.py's `train` leg run slower than tools/bench_train.py?  The same timed loop (bench.bench_train) in a fresh
process, after the forward models of the bench have been built and used (their device contexts hold HIP streams), and after
those models have been released again.  usage: train_after_forward_probe.py [fresh|after|after_del]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
if mode != "fresh":
    import torch

    import volpick_amd as va
    from volpick_amd.synthetic import synthetic_stream_array

    data = synthetic_stream_array(2_000_000, seed=1004, n_events=100)[0]
    t0 = va.UTCDateTime("2021-01-01T00:00:00")
    st = va.Stream([va.Trace(data[i], dict(network="XX", station="DAY", location="", channel=f"HH{c}", starttime=t0,
                                           sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    models = []
    for cls, kw in ((va.PhaseNet, dict(overlap=1500, blinding=(0, 0))), (va.EQTransformer, dict(overlap=5500, blinding=(500, 500)))):
        m = cls.from_pretrained("volpick").cuda()
        n = len(m.classify(st, batch_size=256, stacking="avg", **kw).picks)
        models.append(m)
    torch.cuda.synchronize()
    if mode == "after_del":
        for m in models:
            m._release()
        del models
        gc.collect()
r = bench.bench_train(torch_baseline=False)
print(mode, "ms_per_step", round(r["ms_per_step"], 4), [round(t, 3) for t in r["ms_per_step_all"]], "launches", r["launches_per_step"])
