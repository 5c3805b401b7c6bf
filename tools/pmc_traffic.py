#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 counter passes -> profiles/r01_traffic.json.

    tools/pmc_traffic.sh            (on the GPU box: four --pmc runs of tools/run_forward.py, counters only)
    python tools/pmc_traffic.py gpurun_out/pmc

bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB, and FETCH_SIZE under-reports by 2x on
gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).  The first launch of every kernel (cold caches, plan
warm-up) is dropped; kernels are mapped to plan step names by their order in the forward pass."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def per_kernel(path, counter):
    vals = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)][1:] for k, vs in vals.items()}  # drop the first launch


def main():
    d = Path(sys.argv[1])
    out = {"_note": "HBM bytes per launch at 256 windows = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, rocprofv3 --pmc FETCH_SIZE / "
                    "--pmc WRITE_SIZE in separate passes of tools/run_forward.py (tools/pmc_traffic.sh, tools/pmc_traffic.py), "
                    "gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md"}
    names = {
        "phasenet": [("pn_window_kernel", "fused.window (whole PhaseNet, one workgroup per window)")],
        "eqtransformer": [("ConvCfg<16, 0, 8, 2, 7, 1, -3, 0, 1, 4, 6, 1, 6, 1>", "decoder.6+heads"),
                          ("eqt_res_kernel", "fused.rescnn (7 residual blocks)"),
                          ("eqt_mid_kernel", "fused.mid (3 BiLSTM + 2 transformer blocks + pick branches)")],
    }
    for model, tag in (("phasenet", "pn"), ("eqtransformer", "eqt")):
        fetch = per_kernel(next(d.glob(f"{tag}_fetch/**/*counter_collection.csv")), "FETCH_SIZE")
        write = per_kernel(next(d.glob(f"{tag}_write/**/*counter_collection.csv")), "WRITE_SIZE")
        out[model] = {}
        for needle, step in names[model]:
            k = next(k for k in fetch if needle in k)
            f = sum(fetch[k]) / len(fetch[k])
            w = sum(write[k]) / len(write[k])
            out[model][step] = int(round((2 * f + w) * 1024, -5))
    (ROOT / "profiles" / "r01_traffic.json").write_text(json.dumps(out, indent=2) + "\n")
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
