#!/usr/bin/env python3
"""Writes tests/golden/bench_steim2_6min.mseed: six minutes of one three-component station (100 Hz, Steim-2, 4096-byte
records, big-endian), made with the oracle's encoder (oracle/mseed.py) from seeded random-walk counts
(tests/mseed_util.three_component, seed 8).  bench.py's "mseed" object tiles it to a station-day by patching the records'
start times, so that the bench itself manufactures nothing through the oracle.

    python tools/make_mseed_fixture.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import mseed as OM  # noqa: E402
from tests.mseed_util import three_component  # noqa: E402

N = 36_000
traces = three_component(N, np.random.default_rng(8), spikes=False)
blob = b"".join(OM.write_mseed([t], reclen=4096, encoding=11) for t in traces)
out = ROOT / "tests" / "golden" / "bench_steim2_6min.mseed"
out.write_bytes(blob)
np.savez_compressed(ROOT / "tests" / "golden" / "bench_steim2_6min_samples.npz", **{t["channel"]: t["data"] for t in traces})
print(f"{out}: {len(blob)} bytes, {len(OM.scan_records(blob))} records, {3 * N} samples")
