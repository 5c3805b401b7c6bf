"""miniSEED 3 (FDSN 2020) and the rarer miniSEED encodings (0 text, 2 int24): the oracle's restatement, the library's host
scanner against it, and a record assembled BY HAND in this file from the published layout -- independent of the oracle's
encoder.  The CRC-32C routine is pinned by the standard check value.  The GPU decode of the same bytes: the tests marked gpu."""
import struct

import numpy as np
import pytest

import volpick_amd.io as vio
from oracle import mseed as OM
from tests.mseed_util import T0, seismogram, three_component


def _crc32c_bits(data):
    """CRC-32C written a third time, MSB-first over the bit-reversed message (polynomial 0x1EDC6F41), so that neither the
    oracle's nor the library's reflected table form is what checks the hand-built record."""
    rev8 = lambda b: int(f"{b:08b}"[::-1], 2)
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= rev8(byte) << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1EDC6F41) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    crc ^= 0xFFFFFFFF
    return int(f"{crc:032b}"[::-1], 2)


def hand_built_record(samples, encoding, sid=b"FDSN:XX_TEST__L_H_Z", extra=b'{"FDSN":{"Time":{"Quality":100}}}'):
    """2012-01-01T00:00:00.123456789Z, 1 Hz, publication version 4: byte for byte from the specification's table."""
    s = np.asarray(samples)
    payload = s.astype({1: "<i2", 3: "<i4", 4: "<f4", 5: "<f8"}[encoding]).tobytes()
    rec = bytearray()
    rec += b"MS"                              # 0   record header indicator
    rec += bytes([3])                         # 2   format version
    rec += bytes([0b100])                     # 3   flags: clock locked
    rec += struct.pack("<I", 123456789)       # 4   nanosecond
    rec += struct.pack("<H", 2012)            # 8   year
    rec += struct.pack("<H", 1)               # 10  day of year
    rec += bytes([0, 0, 0])                   # 12  hour, minute, second
    rec += bytes([encoding])                  # 15  data payload encoding
    rec += struct.pack("<d", 1.0)             # 16  sample rate (Hz)
    rec += struct.pack("<I", len(s))          # 24  number of samples
    rec += struct.pack("<I", 0)               # 28  CRC (zero while it is computed)
    rec += bytes([4])                         # 32  data publication version
    rec += bytes([len(sid)])                  # 33  length of identifier
    rec += struct.pack("<H", len(extra))      # 34  length of extra headers
    rec += struct.pack("<I", len(payload))    # 36  length of data payload
    rec += sid + extra + payload
    struct.pack_into("<I", rec, 28, _crc32c_bits(bytes(rec)))
    return bytes(rec)


def test_crc32c_check_value():
    # the check value every CRC catalogue lists for CRC-32C (iSCSI): the message b"123456789"
    assert OM.crc32c(b"123456789") == OM.crc32c_fast(b"123456789") == _crc32c_bits(b"123456789") == 0xE3069283
    assert OM.crc32c(b"") == 0
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, 1000).astype(np.uint8).tobytes()
    assert OM.crc32c(blob) == OM.crc32c_fast(blob) == _crc32c_bits(blob)


def _scan_both(buf):
    got = vio.scan_mseed(buf)
    want = OM.scan_records(buf)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert int(g["offset"]) == w["offset"] and int(g["start_us"]) == w["start_us"]
        assert float(g["sample_rate"]) == w["rate"]
        assert (int(g["reclen"]), int(g["data_offset"]), int(g["nsamples"]), int(g["encoding"])) == (
            w["reclen"], w["data_offset"], w["nsamples"], w["encoding"])
        assert bool(g["big_endian"]) == w["big_endian"]
        assert (g["network"].decode(), g["station"].decode(), g["location"].decode(), g["channel"].decode()) == (
            w["network"], w["station"], w["location"], w["channel"])
        if w.get("format_version") == 3:
            assert int(g["quality"]) == 0x300 | w["pubversion"]
    return got


def test_hand_built_record_is_read_by_oracle_and_host_scanner():
    x = np.array([1, -2, 300, -40000, 5, 2_000_000_000, -2_000_000_000], np.int64)
    buf = hand_built_record(x, 3)
    (w,) = OM.scan_records(buf)
    assert (w["network"], w["station"], w["location"], w["channel"]) == ("XX", "TEST", "", "LHZ")
    assert w["start_us"] == 1_325_376_000_123_456 and w["rate"] == 1.0 and w["nsamples"] == 7 and not w["big_endian"]
    assert w["data_offset"] == 40 + 19 + 33 and w["reclen"] == len(buf) and w["pubversion"] == 4
    assert np.array_equal(OM.decode_record(buf, w), x.astype(np.int32))
    (g,) = _scan_both(buf)
    assert int(g["start_us"]) == 1_325_376_000_123_456 and g["channel"] == b"LHZ" and int(g["quality"]) == 0x304
    y = np.array([0.5, -1.25, 3e10], np.float64)
    for enc, dt in ((4, np.float32), (5, np.float64)):
        b = hand_built_record(y, enc)
        (w,) = OM.scan_records(b)
        assert np.array_equal(OM.decode_record(b, w), y.astype(dt))
        _scan_both(b)


def test_a_flipped_bit_anywhere_fails_the_crc():
    buf = bytearray(hand_built_record(np.arange(20), 1))
    for pos in (3, 9, 16, 24, 33, 45, len(buf) - 1):
        bad = bytearray(buf)
        bad[pos] ^= 0x10
        with pytest.raises(Exception, match="CRC-32C mismatch|past the end|implausible"):
            vio.scan_mseed(bytes(bad))
        with pytest.raises(ValueError):
            OM.scan_records(bytes(bad))
    bad = bytearray(buf)
    bad[28] ^= 1  # the CRC field itself
    with pytest.raises(Exception, match="CRC-32C mismatch"):
        vio.scan_mseed(bytes(bad))


@pytest.mark.parametrize("encoding", [0, 1, 2, 3, 4, 5, 10, 11])
def test_oracle_round_trip_and_scanner(encoding):
    rng = np.random.default_rng(20 + encoding)
    traces = three_component(3001, rng, spikes=encoding in (3, 10, 11), sta="V3", loc="00")
    for t in traces:
        if encoding == 0:
            t["data"] = (t["data"] % 96 + 32).astype(np.int32)
        elif encoding == 1:
            t["data"] = (t["data"] % 60000 - 30000).astype(np.int32)
        elif encoding == 2:
            t["data"] = (t["data"] % (1 << 24) - (1 << 23)).astype(np.int32)
        elif encoding in (4, 5):
            t["data"] = (t["data"] * 0.37).astype(np.float32 if encoding == 4 else np.float64)
    buf = OM.write_mseed3(traces, encoding=encoding, max_payload=640, extra_headers=b'{"a":1}', pubversion=2)
    got = OM.read_mseed(buf)
    assert len(got) == 3
    for g, t in zip(sorted(got, key=lambda s: s["channel"]), sorted(traces, key=lambda s: s["channel"])):
        assert g["start_us"] == t["start_us"] and g["rate"] == t["rate"] and g["location"] == "00"
        assert np.array_equal(g["data"], t["data"])
    recs = _scan_both(buf)
    assert len(recs) > 3


def test_mixed_v2_v3_period_rate_and_nanoseconds():
    rng = np.random.default_rng(31)
    x = seismogram(2000, rng)
    mk = lambda s, d, rate=100.0: dict(network="XX", station="MIX", location="", channel="HHZ", start_us=s, rate=rate, data=d)
    a = OM.write_mseed([mk(T0, x[:1000])], encoding=11)
    b = OM.write_mseed3([mk(T0 + 10_000_000, x[1000:])], encoding=10, nanosecond=999)  # truncated to the microsecond
    got = _scan_both(a + b + OM.write_mseed3([mk(T0, x[:50], rate=0.1)], encoding=3, period_rate=True, sid="FDSN:YY_SLOW_10_L_H_Z"))
    assert float(got["sample_rate"][-1]) == pytest.approx(0.1) and got["location"][-1] == b"10"
    segs = OM.read_mseed(a + b)
    assert len(segs) == 1 and np.array_equal(segs[0]["data"], x)  # the v3 records continue the v2 ones
    r, seg = vio._segments(vio.scan_mseed(a + b))
    assert len(set(seg)) == 1 and int(r["nsamples"].sum()) == 2000


def test_identifiers_that_do_not_fit_are_refused_loudly():
    rng = np.random.default_rng(32)
    tr = [dict(network="XX", station="S", location="", channel="HHZ", start_us=T0, rate=100.0, data=seismogram(100, rng))]
    for sid in ("FDSN:XX_S__HH_H_Z", "FDSN:LONGNET_S__H_H_Z", "XFDSN:XX_S__H_H_Z", "FDSN:XX_S_H_H_Z", "FDSN:XX_S__H_H_Z_Q"):
        with pytest.raises(Exception, match="source identifier"):
            vio.scan_mseed(OM.write_mseed3(tr, sid=sid))
        if "LONGNET" not in sid:  # the field widths of vp_mseed_record are the library's limit, not the format's
            with pytest.raises(ValueError):
                OM.scan_records(OM.write_mseed3(tr, sid=sid))
    ok = vio.scan_mseed(OM.write_mseed3(tr, sid="FDSN:XX_STATION_00_B_H_Z"))
    assert ok["station"][0] == b"STATION" and ok["channel"][0] == b"BHZ"


def test_truncated_v3_record_is_an_error_not_a_silent_stop():
    rng = np.random.default_rng(33)
    tr = [dict(network="XX", station="S", location="", channel="HHZ", start_us=T0, rate=100.0, data=seismogram(500, rng))]
    buf = OM.write_mseed3(tr, encoding=3, max_payload=4096)
    with pytest.raises(Exception, match="past the end"):
        vio.scan_mseed(buf[:-7])


@pytest.mark.parametrize("byteorder", ["<", ">"])
@pytest.mark.parametrize("encoding", [0, 2])
def test_v2_text_and_int24_round_trip_in_the_oracle(encoding, byteorder):
    rng = np.random.default_rng(40 + encoding)
    traces = three_component(2000, rng, spikes=False)
    for t in traces:
        t["data"] = (t["data"] % 96 + 32 if encoding == 0 else t["data"] * 5 % (1 << 24) - (1 << 23)).astype(np.int32)
    buf = OM.write_mseed(traces, encoding=encoding, byteorder=byteorder)
    got = OM.read_mseed(buf)
    for g, t in zip(sorted(got, key=lambda s: s["channel"]), sorted(traces, key=lambda s: s["channel"])):
        assert np.array_equal(g["data"], t["data"])
    _scan_both(buf)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("encoding", [0, 1, 2, 3, 4, 5, 10, 11])
def test_gpu_read_of_v3_matches_oracle(encoding):
    import volpick_amd as va

    rng = np.random.default_rng(50 + encoding)
    traces = three_component(7001, rng, spikes=encoding in (3, 10, 11))
    for t in traces:
        if encoding == 0:
            t["data"] = (t["data"] % 96 + 32).astype(np.int32)
        elif encoding == 1:
            t["data"] = (t["data"] % 60000 - 30000).astype(np.int32)
        elif encoding == 2:
            t["data"] = (t["data"] % (1 << 24) - (1 << 23)).astype(np.int32)
        elif encoding in (4, 5):
            t["data"] = (t["data"] * 0.37).astype(np.float32 if encoding == 4 else np.float64)
    for extra in (b"", b"{}", b'{"a":1}', b'{"ab":12}'):  # every payload alignment 0 .. 3
        buf = OM.write_mseed3(traces, encoding=encoding, max_payload=1000, extra_headers=extra)
        st = va.read(buf)
        want = OM.read_mseed(buf)
        assert len(st) == len(want) == 3
        for tr, w in zip(st, want):
            assert tr.id == f"{w['network']}.{w['station']}.{w['location']}.{w['channel']}"
            assert tr.stats.starttime._us == w["start_us"] and tr.stats.sampling_rate == w["rate"] and tr.stats.npts == 7001
            assert tr.stats.mseed["format_version"] == 3 and tr.stats.mseed["publication_version"] == 1
            if encoding == 0:
                assert tr.data.dtype == np.dtype("S1") and np.array_equal(tr.data.view(np.uint8), w["data"].astype(np.uint8))
            elif encoding == 5:
                assert tr.data.dtype == np.float32 and np.array_equal(tr.data, w["data"].astype(np.float32))
            else:
                assert tr.data.dtype == w["data"].dtype and np.array_equal(tr.data, w["data"])
            assert tr.stats.mseed["steim_integrity_errors"] == 0


@pytest.mark.gpu
def test_gpu_decodes_the_hand_built_records():
    import volpick_amd as va

    x = np.array([1, -2, 300, -40000, 5, 2_000_000_000, -2_000_000_000], np.int64)
    (tr,) = va.read(hand_built_record(x, 3))
    assert tr.id == "XX.TEST..LHZ" and np.array_equal(tr.data, x.astype(np.int32)) and tr.stats.sampling_rate == 1.0
    assert tr.stats.starttime._us == 1_325_376_000_123_456
    (tr,) = va.read(hand_built_record(np.array([7, -8, 32767, -32768]), 1, extra=b"{}"))  # payload at an odd byte
    assert np.array_equal(tr.data, np.array([7, -8, 32767, -32768], np.int32))
    y = np.array([0.5, -1.25, 3e10], np.float64)
    (tr,) = va.read(hand_built_record(y, 5, extra=b"{ }"))
    assert np.array_equal(tr.data, y.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("byteorder", ["<", ">"])
@pytest.mark.parametrize("encoding", [0, 2])
def test_gpu_read_of_v2_text_and_int24(encoding, byteorder):
    import volpick_amd as va

    rng = np.random.default_rng(60 + encoding)
    traces = three_component(5000, rng, spikes=False)
    for t in traces:
        t["data"] = (t["data"] % 96 + 32 if encoding == 0 else t["data"] * 5 % (1 << 24) - (1 << 23)).astype(np.int32)
    buf = OM.write_mseed(traces, encoding=encoding, byteorder=byteorder, reclen=512)
    st = va.read(buf)
    want = OM.read_mseed(buf)
    assert len(st) == 3
    for tr, w in zip(st, want):
        got = tr.data.view(np.uint8).astype(np.int32) if encoding == 0 else tr.data
        assert np.array_equal(got, w["data"]) and tr.stats.mseed["format_version"] == 2


@pytest.mark.gpu
def test_gpu_v3_file_goes_the_whole_way_to_picks():
    """read -> classify of a miniSEED 3 file equals classify of the arrays it encodes, host and device-resident."""
    import volpick_amd as va
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.synthetic import synthetic_stream_array

    data, _, _ = synthetic_stream_array(60_000, seed=1003, n_events=6)
    counts = np.round(data * 2000).astype(np.int32)
    traces = [dict(network="XX", station="V3", location="", channel="HH" + c, start_us=T0, rate=100.0, data=counts[i])
              for i, c in enumerate("ZNE")]
    buf = OM.write_mseed3(traces, encoding=11, max_payload=4032, extra_headers=b"{}")
    ref = Stream([Trace(counts[i].astype(np.float32), dict(network="XX", station="V3", location="", channel="HH" + c,
                                                           starttime=UTCDateTime._from_us(T0), sampling_rate=100.0))
                  for i, c in enumerate("ZNE")])
    m = va.PhaseNet.from_pretrained("volpick").cuda()
    want = m.classify(ref).picks
    assert len(want) > 0
    for st in (va.read(buf), va.read(buf, device_resident=True)):
        got = m.classify(st).picks
        assert len(got) == len(want)
        for p, q in zip(got, want):
            assert p.phase == q.phase and p.peak_time == q.peak_time and p.peak_value == q.peak_value
