#!/usr/bin/env python3
"""Where does `va.read()` of a station-day spend its time?  (VERDICT r4 missing-3: the decode kernel takes 0.05 ms, read()
29-43 ms.)  Phases of the host path and of the device-resident path, each timed alone with synchronisations around it, then
the whole file -> picks call.  Run on a GPU box:  python tools/mseed_read_probe.py [hours]"""
import ctypes as C
import importlib.util
import statistics
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def med(fn, n=7):
    import torch

    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return statistics.median(ts), min(ts), r


def main():
    import torch

    import volpick_amd as va
    import volpick_amd.io as vio
    from volpick_amd import _lib

    hours = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    buf, blob0, recs0, want = bench.station_day_mseed(hours)
    lib = _lib.load()
    print(f"file {len(buf) / 1e6:.1f} MB, {hours} h")
    va.read(blob0)
    rows = []
    m, lo, recs = med(lambda: vio.scan_mseed(buf))
    rows.append(("scan_mseed (host)", m, lo))
    m, lo, (r, seg) = med(lambda: vio._segments(recs))
    rows.append(("_segments (lexsort, chain)", m, lo))
    m, lo, _ = med(lambda: (_lib.VpMseedRecord * len(r)).from_buffer_copy(np.ascontiguousarray(r).tobytes()))
    rows.append(("record table -> ctypes", m, lo))
    recs_c = (_lib.VpMseedRecord * len(r)).from_buffer_copy(np.ascontiguousarray(r).tobytes())
    ns = r["nsamples"].astype(np.int64)
    index = (np.cumsum(ns) - ns).astype(np.int64)
    total = int(ns.sum())
    status = np.zeros(len(r), np.int32)

    def dec_host():
        out = np.empty(total, np.int32)
        _lib.check(lib.vp_mseed_decode(0, buf, _lib.VP_MEM_HOST, len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                       len(r), _lib.VP_SAMPLES_INT32, out.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, total, 0,
                                       status.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    m, lo, _ = med(dec_host)
    rows.append(("vp_mseed_decode host file -> fresh host array", m, lo))
    keep = np.empty(total, np.int32)

    def dec_host_reused():
        _lib.check(lib.vp_mseed_decode(0, buf, _lib.VP_MEM_HOST, len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                       len(r), _lib.VP_SAMPLES_INT32, keep.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, total, 0,
                                       status.ctypes.data_as(C.POINTER(C.c_int32))))

    m, lo, _ = med(dec_host_reused)
    rows.append(("  same, into a touched host array", m, lo))

    def dec_dev():
        out = torch.empty(total, dtype=torch.int32, device="cuda")
        _lib.check(lib.vp_mseed_decode(0, buf, _lib.VP_MEM_HOST, len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                       len(r), _lib.VP_SAMPLES_INT32, C.c_void_p(out.data_ptr()), _lib.VP_MEM_DEVICE, total, 0,
                                       status.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    m, lo, _ = med(dec_dev)
    rows.append(("vp_mseed_decode host file -> device array", m, lo))
    dbuf = torch.empty(len(buf), dtype=torch.uint8, device="cuda")
    hsrc = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
    m, lo, _ = med(lambda: dbuf.copy_(hsrc))
    rows.append(("torch H2D of the file (pageable)", m, lo))
    pin = hsrc.pin_memory()
    m, lo, _ = med(lambda: dbuf.copy_(pin, non_blocking=True))
    rows.append(("torch H2D of the file (pinned)", m, lo))

    def dec_devfile():
        out = torch.empty(total, dtype=torch.int32, device="cuda")
        _lib.check(lib.vp_mseed_decode(0, C.c_void_p(dbuf.data_ptr()), _lib.VP_MEM_DEVICE, len(buf), recs_c,
                                       index.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                       len(r), _lib.VP_SAMPLES_INT32, C.c_void_p(out.data_ptr()), _lib.VP_MEM_DEVICE, total, 0,
                                       status.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    m, lo, _ = med(dec_devfile)
    rows.append(("vp_mseed_decode device file -> device array", m, lo))
    m, lo, st = med(lambda: va.read(buf))
    rows.append(("va.read(buf) -> host Stream", m, lo))
    m, lo, std = med(lambda: va.read(buf, device_resident=True))
    rows.append(("va.read(buf, device_resident=True)", m, lo))
    model = va.PhaseNet.from_pretrained("volpick").cuda()
    kw = dict(batch_size=256, overlap=1500, blinding=(0, 0), stacking="avg")
    model.classify(std, **kw)
    m, lo, res = med(lambda: len(model.classify(std, **kw).picks))
    rows.append(("PhaseNet.classify(device-resident stream)", m, lo))
    m, lo, res = med(lambda: len(model.classify(st, **kw).picks))
    rows.append(("PhaseNet.classify(host stream, int32 counts)", m, lo))
    m, lo, res = med(lambda: len(model.classify(va.read(buf), **kw).picks))
    rows.append(("file -> picks through a HOST stream (read + classify)", m, lo))
    m, lo, res = med(lambda: len(model.classify(va.read(buf, device_resident=True), **kw).picks))
    rows.append(("file -> picks (read device-resident + classify)", m, lo))
    if hasattr(va, "read_and_classify"):
        pass
    print(f"{'phase':58s} {'median ms':>10s} {'min ms':>8s}")
    for name, m, lo in rows:
        print(f"{name:58s} {m:10.3f} {lo:8.3f}")
    print("picks:", res)


if __name__ == "__main__":
    main()
