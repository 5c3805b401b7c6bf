"""GPU parity tests of miniSEED ingestion: vp_mseed_decode (through the C ABI and
through volpick_amd.read) against the oracle's decoder on the same bytes — bit-exact."""
import ctypes as C
import struct

import numpy as np
import pytest

import volpick_amd as va
import volpick_amd.io as vio
from oracle import mseed as OM
from tests.mseed_util import T0, file_bytes, seismogram, three_component
from volpick_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("encoding", [1, 3, 4, 5, 10, 11])
@pytest.mark.parametrize("byteorder", ["<", ">"])
@pytest.mark.parametrize("reclen", [256, 512, 4096])
def test_read_matches_oracle(encoding, byteorder, reclen):
    rng = np.random.default_rng(encoding * 1000 + reclen)
    traces = three_component(7001, rng, spikes=encoding != 1)
    if encoding == 1:
        for t in traces:
            t["data"] = (t["data"] % 30000 - 15000).astype(np.int32)
    buf = file_bytes(traces, reclen=reclen, encoding=encoding, byteorder=byteorder, with_b1001=True)
    st = va.read(buf)
    want = OM.read_mseed(buf)
    assert len(st) == len(want) == 3
    for tr, w in zip(st, want):
        assert tr.id == f"{w['network']}.{w['station']}.{w['location']}.{w['channel']}"
        assert tr.stats.starttime._us == w["start_us"] and tr.stats.sampling_rate == w["rate"]
        assert tr.stats.npts == len(w["data"]) == 7001
        if encoding == 5:
            assert tr.data.dtype == np.float32 and np.array_equal(tr.data, w["data"].astype(np.float32))
        else:
            assert tr.data.dtype == w["data"].dtype and np.array_equal(tr.data, w["data"])
        assert tr.stats.mseed["steim_integrity_errors"] == 0
        assert tr.stats.mseed["record_length"] == reclen and tr.stats.mseed["byteorder"] == byteorder


def test_integers_decode_straight_to_float32():
    rng = np.random.default_rng(3)
    traces = three_component(5000, rng)
    buf = file_bytes(traces, encoding=11)
    st = va.read(buf, dtype=np.float32)
    for tr, t in zip(st.select(channel="*Z"), traces[:1]):
        assert tr.data.dtype == np.float32 and np.array_equal(tr.data, t["data"].astype(np.float32))


def test_gaps_out_of_order_records_and_mixed_files():
    rng = np.random.default_rng(4)
    x = seismogram(9000, rng)
    mk = lambda s, d, ch="HHZ": dict(network="XX", station="GAP", location="", channel=ch, start_us=s, rate=100.0, data=d)
    a = file_bytes([mk(T0, x[:3000])], encoding=11)
    b = file_bytes([mk(T0 + 30_000_000, x[3000:5000])], encoding=10, reclen=4096, byteorder="<")  # contiguous
    c = file_bytes([mk(T0 + 70_000_000, x[5000:])], encoding=3)                                   # after a 20 s gap
    d = file_bytes([mk(T0, x[:100].astype(np.float32) * 0.5, "HHN")], encoding=4)
    buf = c + a + d + b
    st = va.read(buf)
    want = OM.read_mseed(buf)
    assert [tr.id for tr in st] == ["XX.GAP..HHN", "XX.GAP..HHZ", "XX.GAP..HHZ"]
    assert [tr.stats.npts for tr in st] == [100, 5000, 4000] == [len(w["data"]) for w in want]
    for tr, w in zip(st, want):
        assert tr.stats.starttime._us == w["start_us"] and np.array_equal(tr.data, w["data"])
    assert np.array_equal(st[1].data, x[:5000]) and np.array_equal(st[2].data, x[5000:])
    assert st[1].stats.mseed["number_of_records"] == len(a) // 512 + len(b) // 4096


def _decode_raw(buf, recs, index, count, kind, out, zero_fill, want_status=True):
    lib = _lib.load()
    recs_c = (_lib.VpMseedRecord * len(recs)).from_buffer_copy(np.ascontiguousarray(recs).tobytes())
    status = np.full(len(recs), -7, np.int32)
    rc = lib.vp_mseed_decode(0, buf, _lib.VP_MEM_HOST, len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)),
                             None if count is None else count.ctypes.data_as(C.POINTER(C.c_int64)), len(recs), kind,
                             out.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, out.size, zero_fill,
                             status.ctypes.data_as(C.POINTER(C.c_int32)) if want_status else None)
    return rc, status


def test_c_abi_placement_clipping_skip_and_zero_fill():
    """out_index / out_count place records anywhere (the (3, N) station array of stream_to_array);
    samples outside [0, out_len) are dropped; index -1 skips; zero_fill clears the gaps."""
    rng = np.random.default_rng(6)
    x = seismogram(1200, rng)
    buf = file_bytes([dict(network="XX", station="P", location="", channel="HHZ", start_us=T0, rate=100.0, data=x)])
    recs = vio.scan_mseed(buf)
    ns = recs["nsamples"].astype(np.int64)
    starts = np.cumsum(ns) - ns
    out = np.full(2000, 12345, np.int32)
    index = (starts + 500).astype(np.int64)
    index[1] = -1                                     # skipped record
    count = ns.copy()
    count[2] = 5                                      # only five samples of record 2
    rc, status = _decode_raw(buf, recs, index, count, _lib.VP_SAMPLES_INT32, out, 1)
    assert rc == 0 and (status == 0).all()
    want = np.zeros(2000, np.int32)
    for r in range(len(recs)):
        if r == 1:
            continue
        n = 5 if r == 2 else int(ns[r])
        want[500 + starts[r]:500 + starts[r] + n] = x[starts[r]:starts[r] + n]
    assert np.array_equal(out, want)
    # clipping at the end of `out`, no zero fill: untouched cells keep their old content
    out = np.full(300, 777, np.int32)
    index = (starts + 100).astype(np.int64)
    rc, _ = _decode_raw(buf, recs, index, None, _lib.VP_SAMPLES_INT32, out, 0, want_status=False)
    assert rc == 0
    want = np.full(300, 777, np.int32)
    want[100:] = x[:200]
    assert np.array_equal(out, want)


def test_status_reports_integrity_and_short_payload_and_argument_errors():
    rng = np.random.default_rng(7)
    x = seismogram(2000, rng)
    tr = dict(network="XX", station="BAD", location="", channel="HHZ", start_us=T0, rate=100.0, data=x)
    buf = bytearray(file_bytes([tr], encoding=11))
    buf[512 + 64 + 8: 512 + 64 + 12] = b"\x7f\x00\x00\x01"   # record 1: wrong reverse integration constant
    recs = vio.scan_mseed(bytes(buf))
    ns = recs["nsamples"].astype(np.int64)
    recs["nsamples"][2] += 900                                # record 2 claims more samples than its frames hold
    index = (np.cumsum(ns) - ns).astype(np.int64)
    out = np.zeros(int(ns.sum()) + 900, np.int32)
    rc, status = _decode_raw(bytes(buf), recs, index, ns, _lib.VP_SAMPLES_INT32, out, 0)
    assert rc == 0
    assert status[1] == 1 and status[2] == 2 and (np.delete(status, [1, 2]) == 0).all()
    assert np.array_equal(out[: int(ns.sum())], x)            # the samples themselves are unaffected
    with pytest.raises(ValueError, match="fewer samples"):
        b2 = bytearray(buf)
        b2[2 * 512 + 30: 2 * 512 + 32] = int(recs["nsamples"][2]).to_bytes(2, "big")
        va.read(bytes(b2))
    fl = file_bytes([dict(tr, data=x.astype(np.float32))], encoding=4)
    r4 = vio.scan_mseed(fl)
    rc, _ = _decode_raw(fl, r4, np.zeros(len(r4), np.int64), None, _lib.VP_SAMPLES_INT32, out, 0)
    assert rc == -1 and b"floating-point" in _lib.load().vp_last_error()
    r4["offset"][0] = len(fl)
    rc, _ = _decode_raw(fl, r4, np.zeros(len(r4), np.int64), None, _lib.VP_SAMPLES_FLOAT32, out.view(np.float32), 0)
    assert rc == -1 and b"outside the buffer" in _lib.load().vp_last_error()


def test_day_long_round_trip_property():
    """Full size: one station-day (3 x 8.64 M samples).  decode(encode(x)) == x exactly.  The
    oracle's encoder is a Python loop, so the day is 24 copies of one encoded hour with the record
    start times moved (every Steim record carries its own integration constant, so the copies
    stay valid)."""
    rng = np.random.default_rng(8)
    hour = 360_000
    base = three_component(hour, rng)
    tile = {t["channel"]: file_bytes([t], reclen=4096, encoding=11) for t in base}
    parts, want = [], {}
    for t in base:
        recs_per_tile = len(tile[t["channel"]]) // 4096
        for h in range(24):
            # same payload, start time moved by h hours: patch the BTIME of every record of the tile
            blob = bytearray(tile[t["channel"]])
            r = OM.scan_records(bytes(blob))
            for rec in r:
                y, doy, hh, mm, ss, fr, _ = OM.us_to_btime(rec["start_us"] + h * 3_600_000_000)
                struct.pack_into(">HHBBBBH", blob, rec["offset"] + 20, y, doy, hh, mm, ss, 0, fr)
            parts.append(bytes(blob))
        want[t["channel"]] = np.tile(t["data"], 24)
        assert recs_per_tile > 0
    buf = b"".join(parts)
    st = va.read(buf)
    assert len(st) == 3 and all(tr.stats.npts == 24 * hour for tr in st)
    for tr in st:
        assert np.array_equal(tr.data, want[tr.stats.channel])
        assert tr.stats.mseed["steim_integrity_errors"] == 0


def test_read_then_classify_equals_classify_of_the_arrays():
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.synthetic import synthetic_stream_array

    data, _, _ = synthetic_stream_array(60_000, seed=1001, n_events=6)
    counts = np.round(data * 2000).astype(np.int32)
    traces = [dict(network="XX", station="VOLC", location="", channel="HH" + c, start_us=T0, rate=100.0, data=counts[i])
              for i, c in enumerate("ZNE")]
    st = va.read(file_bytes(traces, encoding=11, reclen=4096))
    ref = Stream([Trace(counts[i].astype(np.float32), dict(network="XX", station="VOLC", location="", channel="HH" + c,
                                                           starttime=UTCDateTime._from_us(T0), sampling_rate=100.0))
                  for i, c in enumerate("ZNE")])
    m = va.PhaseNet.from_pretrained("volpick").cuda()
    a, b = m.classify(st).picks, m.classify(ref).picks
    assert len(a) == len(b) > 0
    for p, q in zip(a, b):
        assert p.phase == q.phase and p.peak_time == q.peak_time and p.peak_value == q.peak_value
    # device-resident traces: decoded, assembled and classified without visiting the host
    buf = file_bytes(traces, encoding=11, reclen=4096)
    dst = va.read(buf, device_resident=True)
    assert all(tr._data is None and tr._dev.is_cuda for tr in dst) and [len(tr) for tr in dst] == [60_000] * 3
    c = m.classify(dst).picks
    assert all(tr._data is None for tr in dst)  # classify did not materialise host copies
    assert len(c) == len(b)
    for p, q in zip(c, b):
        assert p.phase == q.phase and p.peak_time == q.peak_time and p.peak_value == q.peak_value
    ann_d, ann_h = m.annotate(dst), m.annotate(ref)
    for x, y in zip(ann_d, ann_h):
        assert x.stats.channel == y.stats.channel and np.array_equal(x.data, y.data, equal_nan=True)
    for tr, w in zip(dst, st):  # .data materialises the same samples on demand
        assert np.array_equal(tr.data, w.data) and tr.stats.mseed == w.stats.mseed
    # gaps between device-resident segments are zero-filled / split exactly like host traces
    mk = lambda s, d, ch: dict(network="XX", station="GAP", location="", channel=ch, start_us=s, rate=100.0, data=d)
    segs = []
    for i, ch in enumerate(("HHZ", "HHN", "HHE")):
        segs += [mk(T0, counts[i, :20_000], ch), mk(T0 + 250_000_000, counts[i, 25_000:], ch)]  # 50 s gap
    gbuf = file_bytes(segs, encoding=11)
    pg_d, pg_h = m.classify(va.read(gbuf, device_resident=True)).picks, m.classify(va.read(gbuf)).picks
    assert len(pg_d) == len(pg_h) > 0
    for p, q in zip(pg_d, pg_h):
        assert p.phase == q.phase and p.peak_time == q.peak_time and p.peak_value == q.peak_value


def test_decode_scratch_can_be_released_and_is_rebuilt():
    """vp_mseed_decode keeps its device scratch per device between calls; vp_mseed_release_scratch frees it and the
    next read allocates again with the same result."""
    rng = np.random.default_rng(77)
    buf = file_bytes(three_component(9001, rng), reclen=512, encoding=11, byteorder=">")
    first = va.read(buf)
    assert vio.release_decode_scratch(0) > 0
    assert vio.release_decode_scratch(0) == 0  # nothing left to free
    again = va.read(buf)
    for a, b in zip(first, again):
        assert np.array_equal(a.data, b.data)
