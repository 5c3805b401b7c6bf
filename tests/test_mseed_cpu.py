"""CPU tests of waveform-file ingestion: the oracle round trip, the library's host
record scanner against the oracle's, SAC parsing and the stream_to_array rule.  No GPU calls."""
import numpy as np
import pytest

import volpick_amd.io as vio
from oracle import mseed as OM
from tests.mseed_util import T0, file_bytes, seismogram, three_component
from volpick_amd import Stream, Trace, UTCDateTime


@pytest.mark.parametrize("encoding", [1, 3, 4, 5, 10, 11])
@pytest.mark.parametrize("byteorder", ["<", ">"])
@pytest.mark.parametrize("reclen", [512, 4096])
def test_oracle_round_trip(encoding, byteorder, reclen):
    rng = np.random.default_rng(encoding * 100 + reclen)
    x = seismogram(3000, rng, spikes=encoding != 1)
    if encoding == 1:
        x = (x % 30000).astype(np.int32)
    tr = dict(network="XX", station="ABC", location="00", channel="HHZ", start_us=T0, rate=100.0, data=x)
    buf = file_bytes([tr], reclen=reclen, encoding=encoding, byteorder=byteorder, with_b1001=True)
    assert len(buf) % reclen == 0
    out = OM.read_mseed(buf)
    assert len(out) == 1 and out[0]["start_us"] == T0 and out[0]["rate"] == 100.0
    assert (out[0]["network"], out[0]["station"], out[0]["location"], out[0]["channel"]) == ("XX", "ABC", "00", "HHZ")
    want = x.astype(np.float32) if encoding == 4 else x  # float32 records hold rounded counts
    assert np.array_equal(out[0]["data"].astype(np.float64), want.astype(np.float64))


def test_steim_word_kinds_all_occur():
    """The synthetic counts exercise every Steim-2 packing (1x30 ... 7x4 bits) and Steim-1's three."""
    rng = np.random.default_rng(5)
    x = seismogram(20000, rng)
    for version, enc in ((1, 10), (2, 11)):
        payload, n = OM.steim_encode(x, version, 63)
        w = np.frombuffer(payload, dtype=">u4").reshape(-1, 16)
        kinds = set()
        for f in range(w.shape[0]):
            for k in range(1, 16):
                nib = (int(w[f, 0]) >> (30 - 2 * k)) & 3
                kinds.add((nib, int(w[f, k]) >> 30 if (version == 2 and nib >= 2) else -1))
        want = {(0, -1), (1, -1), (2, -1), (3, -1)} if version == 1 else {(0, -1), (1, -1), (2, 1), (2, 2), (2, 3),
                                                                         (3, 0), (3, 1), (3, 2)}
        assert want <= kinds, (version, kinds)


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
    return got


@pytest.mark.parametrize("byteorder", ["<", ">"])
def test_host_scanner_matches_oracle(byteorder):
    rng = np.random.default_rng(11)
    traces = three_component(5000, rng) + three_component(2500, rng, start_us=T0 + 90_000_000, sta="AB", loc="01",
                                                          band="EH", rate=50.0)
    traces.append(dict(network="YY", station="SLOW", location="", channel="LHZ", start_us=T0, rate=0.1,
                       data=seismogram(300, rng)))
    buf = file_bytes(traces, reclen=512, encoding=11, byteorder=byteorder, with_b1001=True)
    buf += file_bytes(traces[:1], reclen=4096, encoding=10, byteorder=byteorder)
    got = _scan_both(buf)
    assert set(got["reclen"]) == {512, 4096}


def test_scanner_time_correction_microseconds_and_noise_blocks():
    rng = np.random.default_rng(12)
    tr = three_component(800, rng, start_us=T0 + 37)[:1]  # 37 us go to blockette 1001
    a = file_bytes(tr, with_b1001=True, time_correction=25)            # +2.5 ms, not yet applied
    b = file_bytes(tr, with_b1001=True, time_correction=25, activity_flags=0x02)  # already applied
    junk = b"\0" * 128
    got = _scan_both(a + junk + b)
    na = len(a) // 512
    assert int(got["start_us"][0]) == T0 + 37 + 2500 and int(got["start_us"][na]) == T0 + 37


def test_scanner_rejects_truncated_or_headerless_records():
    lib_buf = file_bytes(three_component(500, np.random.default_rng(1))[:1])
    broken = bytearray(lib_buf)
    broken[46:48] = b"\0\0"  # no blockette chain
    with pytest.raises(Exception, match="blockette 1000"):
        vio.scan_mseed(bytes(broken))
    assert len(vio.scan_mseed(b"")) == 0 and len(vio.scan_mseed(b"\0" * 4096)) == 0


def test_segments_follow_gaps_overlaps_and_record_order():
    rng = np.random.default_rng(13)
    x = seismogram(4000, rng)
    mk = lambda s, data: dict(network="XX", station="GAP", location="", channel="HHZ", start_us=s, rate=100.0, data=data)
    first = file_bytes([mk(T0, x[:1500])])
    second = file_bytes([mk(T0 + 15_000_000, x[1500:2600])])      # contiguous with `first`
    third = file_bytes([mk(T0 + 40_000_000, x[2600:])])           # 14 s gap
    buf = third + first + second                                  # records out of time order in the file
    r, seg = vio._segments(vio.scan_mseed(buf))
    bounds = np.flatnonzero(np.diff(seg)) + 1
    assert len(bounds) == 1 and int(r["start_us"][bounds[0]]) == T0 + 40_000_000
    assert np.all(np.diff(r["start_us"]) > 0)
    want = OM.read_mseed(buf)
    assert [len(w["data"]) for w in want] == [2600, 1400]


def test_sac_both_byte_orders_and_autodetect():
    x = np.sin(np.arange(1234) * 0.01).astype(np.float32)
    for bo in "<>":
        buf = OM.write_sac(x, 100.0, T0 + 37, "NC", "MMT", "", "EHZ", byteorder=bo)
        st = vio.read(buf)
        assert len(st) == 1
        tr = st[0]
        assert tr.id == "NC.MMT..EHZ" and tr.stats.npts == 1234 and tr.stats.sampling_rate == 100.0
        assert tr.stats.starttime._us == T0 + 37
        assert tr.data.dtype == np.float32 and np.array_equal(tr.data, x)
        w = OM.read_sac(buf)
        assert w["start_us"] == T0 + 37 and np.array_equal(w["data"], x)
    with pytest.raises(ValueError):
        vio.read(b"\1" * 700)


def test_stream_to_array_rule():
    """volpick/data/convert.py:26-70 — common span, zero fill, shortest trace first, demean, completeness."""
    rng = np.random.default_rng(14)
    mk = lambda ch, s, n: dict(network="XX", station="S", location="", channel=ch, start_us=T0 + s, rate=100.0,
                               data=rng.normal(size=n) + 3.0)
    tds = [mk("HHZ", 0, 1000), mk("HHN", 2_000_000, 700), mk("HHN", 9_500_000, 40), mk("HHE", 500_000, 1000)]
    st = Stream([Trace(t["data"], dict(network="XX", station="S", location="", channel=t["channel"],
                                       starttime=UTCDateTime._from_us(t["start_us"]), sampling_rate=100.0)) for t in tds])
    t0, data, comp = vio.stream_to_array(st, "ZNE")
    w0, wdata, wcomp = OM.stream_to_array(tds, "ZNE")
    assert t0._us == w0 and data.shape == wdata.shape == (3, 1050)
    assert np.allclose(data, wdata, atol=1e-12) and abs(comp - wcomp) < 1e-12
    assert np.abs(data.mean(axis=1)).max() < 1e-12
