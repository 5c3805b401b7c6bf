#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 counter passes -> profiles/r03_traffic.json (or the name given as the second argument).

    tools/pmc_traffic.sh            (on the GPU box: four --pmc runs of tools/run_forward.py, counters only)
    python tools/pmc_traffic.py gpurun_out/pmc

bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB, and FETCH_SIZE under-reports by 2x on
gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).  The first launch of every kernel (cold caches, plan
warm-up) is dropped; kernels are mapped to plan step names by a substring of their name.  "_step_total" is the sum
over every kernel of one forward pass (256 windows: annotate_batch_pre + the model)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / "profiles" / (sys.argv[2] if len(sys.argv) > 2 else "r03_traffic.json")


def per_kernel(path, counter):
    vals = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)][1:] for k, vs in vals.items()}  # drop the first launch


NAMES = {
    "phasenet": [("pn_window_kernel", "fused.window (whole PhaseNet, one workgroup per window)")],
    "eqtransformer": [("eqt_front_kernel", "fused.front (encoder.0-2, time-tiled)"),
                      ("eqt_enc36_b3_kernel", "fused.enc36 (encoder.3-6, one window per workgroup)"), ("eqt_enc36_kernel", "fused.enc36 (encoder.3-6, one window per workgroup)"),
                      ("eqt_res3t_kernel", "fused.rescnn (7 residual blocks)"), ("eqt_res3k_kernel", "fused.rescnn (7 residual blocks)"), ("eqt_res3_kernel", "fused.rescnn (7 residual blocks)"), ("eqt_res_kernel", "fused.rescnn (7 residual blocks)"),
                      ("eqt_mid4_kernel", "fused.mid (3 BiLSTM + 2 transformer blocks + pick branches)"),
                      ("eqt_mid_kernel", "fused.mid (3 BiLSTM + 2 transformer blocks + pick branches)"),
                      ("eqt_dec03_kernel", "fused.dec03 (decoder.0-3, one row per workgroup)"),
                      ("eqt_tail3_kernel", "fused.tail (decoder.4-6 + heads, time-tiled)"),
                      ("eqt_tail_kernel", "fused.tail (decoder.4-6 + heads, time-tiled)")],
}


def main():
    d = Path(sys.argv[1])
    out = {"_note": "HBM bytes per launch at 256 windows = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, rocprofv3 --pmc FETCH_SIZE / "
                    "--pmc WRITE_SIZE in separate passes of tools/run_forward.py (tools/pmc_traffic.sh, tools/pmc_traffic.py), "
                    "gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md; _step_total = all kernels of one forward pass"}
    for model, tag in (("phasenet", "pn"), ("eqtransformer", "eqt")):
        fetch = per_kernel(next(d.glob(f"{tag}_fetch/**/*counter_collection.csv")), "FETCH_SIZE")
        write = per_kernel(next(d.glob(f"{tag}_write/**/*counter_collection.csv")), "WRITE_SIZE")
        out[model] = {}
        total = 0.0
        for k in fetch:
            if not fetch[k] or k not in write or not write[k]:
                continue
            total += (2 * sum(fetch[k]) / len(fetch[k]) + sum(write[k]) / len(write[k])) * 1024
        for needle, step in NAMES[model]:
            ks = [k for k in fetch if needle in k]
            if not ks:
                continue
            k = ks[0]
            f = sum(fetch[k]) / len(fetch[k])
            w = sum(write[k]) / len(write[k])
            out[model][step] = int(round((2 * f + w) * 1024, -5))
        out[model]["_step_total"] = int(round(total, -5))
    OUT.write_text(json.dumps(out, indent=2) + "\n")
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
