"""Decode throughput of vp_mseed_decode on one station-day of Steim-2 records, inputs resident in
HBM (vp_mseed_decode_bench: HIP events around `iters` launches), next to the oracle's CPU decoder
on a bounded sample and to the end-to-end volpick_amd.read() wall time (PCIe + host included).

    python tools/bench_mseed.py [--hours 24] [--reclen 4096] [--encoding 11]
"""
import argparse
import ctypes as C
import json
import struct
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import volpick_amd as va  # noqa: E402
import volpick_amd.io as vio  # noqa: E402
from oracle import mseed as OM  # noqa: E402  (input generator + CPU baseline only)
from tests.mseed_util import T0, three_component  # noqa: E402
from volpick_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=int, default=24)
    ap.add_argument("--reclen", type=int, default=4096)
    ap.add_argument("--encoding", type=int, default=11)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    rng = np.random.default_rng(8)
    hour = 360_000
    base = three_component(hour, rng, spikes=False)
    parts = []
    for t in base:
        blob0 = OM.write_mseed([t], reclen=a.reclen, encoding=a.encoding)
        recs = OM.scan_records(blob0)
        for h in range(a.hours):
            blob = bytearray(blob0)
            for rec in recs:
                y, doy, hh, mm, ss, fr, _ = OM.us_to_btime(rec["start_us"] + h * 3_600_000_000)
                struct.pack_into(">HHBBBBH", blob, rec["offset"] + 20, y, doy, hh, mm, ss, 0, fr)
            parts.append(bytes(blob))
    buf = b"".join(parts)
    lib = _lib.load()
    t0 = time.perf_counter()
    recs = vio.scan_mseed(buf)
    t_scan = time.perf_counter() - t0
    ns = recs["nsamples"].astype(np.int64)
    index = (np.cumsum(ns) - ns).astype(np.int64)
    total = int(ns.sum())
    dbuf = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    dout = torch.empty(total, dtype=torch.int32, device="cuda")
    recs_c = (_lib.VpMseedRecord * len(recs)).from_buffer_copy(np.ascontiguousarray(recs).tobytes())
    ms = C.c_float(0)
    _lib.check(lib.vp_mseed_decode_bench(0, dbuf.data_ptr(), len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)),
                                         len(recs), _lib.VP_SAMPLES_INT32, dout.data_ptr(), total, a.iters, C.byref(ms)))
    payload = int((recs["reclen"] - recs["data_offset"]).sum())
    algo_bytes = payload + 4 * total
    # end-to-end read(): host scan + H2D + decode + D2H + Stream
    va.read(buf[: len(buf) // 24])
    t0 = time.perf_counter()
    st = va.read(buf)
    t_read = time.perf_counter() - t0
    assert sum(tr.stats.npts for tr in st) == total
    va.read(buf[: len(buf) // 24], device_resident=True)
    t0 = time.perf_counter()
    std = va.read(buf, device_resident=True)
    torch.cuda.synchronize()
    t_read_dev = time.perf_counter() - t0
    assert sum(tr.stats.npts for tr in std) == total
    # CPU baseline: the oracle's decoder on a bounded sample of records
    k = min(len(recs), 200)
    orecs = OM.scan_records(buf[: int(recs["offset"][k - 1] + recs["reclen"][k - 1])])
    t0 = time.perf_counter()
    n_cpu = sum(len(OM.decode_record(buf, r)) for r in orecs)
    t_cpu = time.perf_counter() - t0
    print(json.dumps({
        "metric": "miniSEED samples decoded per second (Steim-2, one station-day, inputs in HBM)",
        "value": total / (ms.value * 1e-3), "unit": "samples/s", "kernel_ms": ms.value, "records": int(len(recs)),
        "samples": total, "file_bytes": len(buf), "bytes_per_sample": len(buf) / total,
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (ms.value * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": algo_bytes / (ms.value * 1e-3) / 8e12, "algorithmic_bytes": algo_bytes},
        "read_wall_ms": t_read * 1e3, "read_device_resident_wall_ms": t_read_dev * 1e3, "scan_ms": t_scan * 1e3,
        "cpu_baseline": {"value": n_cpu / t_cpu, "unit": "samples/s", "cores": 1, "kind": "port",
                         "sample": f"{k} records through oracle/mseed.py (numpy/Python)"},
    }))


if __name__ == "__main__":
    main()
