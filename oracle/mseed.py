"""TEST INFRASTRUCTURE ONLY -- CPU restatement of waveform-file ingestion.

PARITY UNPINNED.  The reference reads waveform files with ``obspy.read`` (libmseed)
(volpick/data/convert.py:7,  ``read(...)`` call sites in ``convert_mseed_to_seisbench``) and
turns the resulting Stream into a (3, N) array with ``stream_to_array``
(volpick/data/convert.py:26-70).  ObsPy/libmseed are absent from the build image and the
reference repository holds no waveform file, so this module restates the published formats
from the SEED 2.4 manual (fixed section of data header, blockettes 1000/1001, data
encodings 1/3/4/5/10/11 with the Steim-1/Steim-2 frame layout) and the SAC binary header
layout; nothing here has been checked against a libmseed-written file.  miniSEED 3 follows the
FDSN specification of 2020 ("miniSEED 3", fixed 40-byte little-endian header, source identifier
``FDSN:NET_STA_LOC_B_S_SS``, CRC-32C over the record with its CRC field zeroed); the CRC routine is
pinned by the standard check value (CRC-32C of b"123456789" = 0xE3069283, tests/test_mseed3.py).

Besides the decoder it holds a small *encoder* (Steim-1/2 packing and record writing) used
only to manufacture test inputs, and ``stream_to_array`` -- the in-repo array assembly rule.

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import struct
from datetime import datetime, timedelta, timezone

import numpy as np

ENC_INT16, ENC_INT32, ENC_FLOAT32, ENC_FLOAT64, ENC_STEIM1, ENC_STEIM2 = 1, 3, 4, 5, 10, 11
ENC_TEXT, ENC_INT24 = 0, 2
_EPOCH = datetime(1970, 1, 1, tzinfo=timezone.utc)


# ----------------------------------------------------------------------------- time helpers
def btime_to_us(year, doy, hour, minute, sec, fract) -> int:
    """SEED BTIME (fract in 0.0001 s) -> microseconds since 1970-01-01."""
    d = datetime(year, 1, 1, tzinfo=timezone.utc) + timedelta(days=doy - 1, hours=hour, minutes=minute, seconds=sec)
    delta = d - _EPOCH
    return (delta.days * 86400 + delta.seconds) * 1_000_000 + fract * 100


def us_to_btime(us: int):
    d = _EPOCH + timedelta(microseconds=int(us))
    return d.year, d.timetuple().tm_yday, d.hour, d.minute, d.second, d.microsecond // 100, d.microsecond % 100


def sample_rate(factor: int, mult: int) -> float:
    """SEED sample-rate factor/multiplier pair -> Hz."""
    if factor == 0:
        return 0.0
    r = float(factor) if factor > 0 else -1.0 / factor
    return r * mult if mult > 0 else (r / -mult if mult < 0 else r)


# ----------------------------------------------------------------------------- Steim encoder (test inputs)
def _fits(v, bits):
    return -(1 << (bits - 1)) <= v < (1 << (bits - 1))


def steim_encode(samples, version: int, nframes: int, prev: int = 0, byteorder: str = ">"):
    """Pack as many samples as fit into ``nframes`` 64-byte frames.

    Returns (payload bytes, number of samples packed).  d[0] = samples[0] - prev.
    """
    x = [int(v) for v in samples]
    diffs = [x[0] - prev] + [x[i] - x[i - 1] for i in range(1, len(x))]
    # wrap differences to int32 (the decoder integrates modulo 2^32)
    diffs = [((d + (1 << 31)) & 0xFFFFFFFF) - (1 << 31) for d in diffs]
    frames = np.zeros((nframes, 16), dtype=np.uint32)
    pos = 0
    for f in range(nframes):
        nibbles = 0
        for w in range(16):
            nib = 0
            if w == 0 or (f == 0 and w in (1, 2)):
                pass  # control word / X0 / Xn
            elif pos < len(diffs):
                rem = diffs[pos:pos + 7]
                word = None
                if version == 2:
                    options = [(7, 4, 3, 2), (6, 5, 3, 1), (5, 6, 3, 0), (4, 8, 1, None), (3, 10, 2, 3), (2, 15, 2, 2),
                               (1, 30, 2, 1)]
                else:
                    options = [(4, 8, 1, None), (2, 16, 2, None), (1, 32, 3, None)]
                for cnt, bits, nb, dnib in options:
                    if len(rem) >= cnt and all(_fits(v, bits) for v in rem[:cnt]):
                        val = 0
                        for v in rem[:cnt]:
                            val = (val << bits) | (v & ((1 << bits) - 1))
                        if dnib is not None:
                            val |= dnib << 30
                        word, nib = val, nb
                        pos += cnt
                        break
                if word is None:
                    raise ValueError(f"difference {rem[0]} does not fit Steim-{version}")
                frames[f, w] = word & 0xFFFFFFFF
            nibbles |= nib << (30 - 2 * w)
        frames[f, 0] = nibbles
    frames[0, 1] = x[0] & 0xFFFFFFFF
    frames[0, 2] = x[pos - 1] & 0xFFFFFFFF
    return frames.astype(byteorder + "u4").tobytes(), pos


# ----------------------------------------------------------------------------- record writer (test inputs)
def pack_plain(values, encoding, byteorder):
    """Payload bytes of the uncompressed encodings (0 text: one byte per sample; 2: 24-bit two's complement)."""
    v = np.asarray(values)
    if encoding == ENC_TEXT:
        return v.astype(np.uint8).tobytes()
    if encoding == ENC_INT24:
        b = (v.astype(np.int64) & 0xFFFFFF).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3]
        return (b[:, ::-1] if byteorder == ">" else b).tobytes()
    dt = {ENC_INT16: "i2", ENC_INT32: "i4", ENC_FLOAT32: "f4", ENC_FLOAT64: "f8"}[encoding]
    return v.astype(byteorder + dt).tobytes()


# ----------------------------------------------------------------------------- miniSEED 3
def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli polynomial 0x1EDC6F41, reflected), bit by bit: the slow textbook form."""
    crc ^= 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


_CRC_TABLE = None


def crc32c_fast(data: bytes) -> int:
    """Table form of ``crc32c`` (the encoder below writes day-long files)."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t.append(c)
        _CRC_TABLE = t
    crc = 0xFFFFFFFF
    t = _CRC_TABLE
    for byte in data:
        crc = (crc >> 8) ^ t[(crc ^ byte) & 0xFF]
    return crc ^ 0xFFFFFFFF


def write_mseed3(traces, encoding=ENC_STEIM2, max_payload=448, extra_headers=b"", pubversion=1, flags=0, sid=None,
                 period_rate=False, nanosecond=0):
    """miniSEED 3 records of ``traces`` (same dicts as ``write_mseed``): header fields little-endian, Steim frames
    big-endian, every other payload little-endian, variable record length (no padding but whole Steim frames)."""
    out = bytearray()
    for tr in traces:
        data = np.asarray(tr["data"])
        rate = float(tr["rate"])
        ident = sid if sid is not None else "FDSN:{}_{}_{}_{}".format(
            tr["network"], tr["station"], tr["location"], "_".join(tr["channel"].ljust(3)[:3]).replace(" ", ""))
        ident = ident.encode()
        pos, prev, n = 0, 0, len(data)
        while pos < n:
            if encoding in (ENC_STEIM1, ENC_STEIM2):
                payload, cnt = steim_encode(data[pos:], 1 if encoding == ENC_STEIM1 else 2, max_payload // 64, prev, ">")
                used = 0  # drop the unused frames at the end of the last record
                w = np.frombuffer(payload, ">u4").reshape(-1, 16)
                for f in range(w.shape[0]):
                    if w[f, 0] != 0 or f == 0:
                        used = f + 1
                payload = payload[: 64 * used]
            else:
                width = {ENC_TEXT: 1, ENC_INT16: 2, ENC_INT24: 3, ENC_INT32: 4, ENC_FLOAT32: 4, ENC_FLOAT64: 8}[encoding]
                cnt = min(n - pos, max_payload // width)
                payload = pack_plain(data[pos:pos + cnt], encoding, "<")
            us = tr["start_us"] + int(round(pos * 1e6 / rate))
            d = _EPOCH + timedelta(microseconds=us)
            hdr = bytearray(40)
            hdr[0:3] = b"MS\x03"
            struct.pack_into("<BIHHBBBBdIIBBHI", hdr, 3, flags, d.microsecond * 1000 + nanosecond, d.year,
                             d.timetuple().tm_yday, d.hour, d.minute, d.second, encoding,
                             -1.0 / rate if period_rate else rate, cnt, 0, pubversion, len(ident), len(extra_headers),
                             len(payload))
            rec = bytearray(hdr + ident + extra_headers + payload)
            struct.pack_into("<I", rec, 28, crc32c_fast(bytes(rec)))
            out += rec
            if encoding in (ENC_STEIM1, ENC_STEIM2):
                prev = int(data[pos + cnt - 1])
            pos += cnt
    return bytes(out)


def scan_record3(buf: bytes, off: int):
    """One miniSEED 3 record at ``off`` -> (record dict, record length); the CRC is verified."""
    (flags, nsec, year, doy, hh, mm, ss, enc, rate, ns, crc, pubversion, sid_len, extra_len,
     payload_len) = struct.unpack_from("<BIHHBBBBdIIBBHI", buf, off + 3)
    reclen = 40 + sid_len + extra_len + payload_len
    if off + reclen > len(buf):
        raise ValueError(f"miniSEED 3 record at byte {off} runs past the end of the buffer")
    rec = bytearray(buf[off:off + reclen])
    rec[28:32] = b"\0\0\0\0"
    if crc32c_fast(bytes(rec)) != crc:
        raise ValueError(f"miniSEED 3 record at byte {off}: CRC-32C mismatch")
    sid = buf[off + 40: off + 40 + sid_len].decode()
    if not sid.startswith("FDSN:") or sid.count("_") != 5:
        raise ValueError(f"source identifier {sid!r} is not FDSN:NET_STA_LOC_B_S_SS")
    net, sta, loc, band, source, sub = sid[5:].split("_")
    if max(len(band), len(source), len(sub)) > 1:
        raise ValueError(f"source identifier {sid!r}: channel codes longer than one character")
    start = btime_to_us(year, doy, hh, mm, ss, 0) + nsec // 1000
    return dict(offset=off, reclen=reclen, nsamples=ns, encoding=enc, big_endian=enc in (ENC_STEIM1, ENC_STEIM2),
                data_offset=40 + sid_len + extra_len, start_us=start,
                rate=rate if rate > 0 else (-1.0 / rate if rate < 0 else 0.0), network=net, station=sta, location=loc,
                channel=band + source + sub, format_version=3, pubversion=pubversion), reclen


def write_mseed(traces, reclen=512, encoding=ENC_STEIM2, byteorder=">", with_b1001=False, seq0=1,
                time_correction=0, activity_flags=0):
    """``traces``: list of dicts {network, station, location, channel, start_us, rate, data}.

    Returns the concatenated records of all traces (trace after trace).
    """
    out = bytearray()
    seq = seq0
    exp = int(np.log2(reclen))
    assert 1 << exp == reclen
    data_off = 64
    for tr in traces:
        data = np.asarray(tr["data"])
        rate = float(tr["rate"])
        if rate >= 1 and rate == int(rate):
            fac, mul = int(rate), 1
        else:
            fac, mul = -int(round(1.0 / rate)), 1
        pos, prev = 0, 0
        n = len(data)
        while pos < n:
            payload_bytes = reclen - data_off
            if encoding in (ENC_STEIM1, ENC_STEIM2):
                payload, cnt = steim_encode(data[pos:], 1 if encoding == ENC_STEIM1 else 2, payload_bytes // 64, prev,
                                            byteorder)
            elif encoding in (ENC_TEXT, ENC_INT24):
                cnt = min(n - pos, payload_bytes // (1 if encoding == ENC_TEXT else 3))
                payload = pack_plain(data[pos:pos + cnt], encoding, byteorder)
                payload += b"\0" * (payload_bytes - len(payload))
            else:
                dt = {ENC_INT16: "i2", ENC_INT32: "i4", ENC_FLOAT32: "f4", ENC_FLOAT64: "f8"}[encoding]
                cnt = min(n - pos, payload_bytes // np.dtype(dt).itemsize)
                payload = np.asarray(data[pos:pos + cnt]).astype(byteorder + dt).tobytes()
                payload += b"\0" * (payload_bytes - len(payload))
            us = tr["start_us"] + int(round(pos * 1e6 / rate))
            y, doy, hh, mm, ss, fract, usec = us_to_btime(us)
            hdr = bytearray(data_off)
            hdr[0:6] = f"{seq % 1000000:06d}".encode()
            hdr[6:8] = b"D "
            hdr[8:13] = tr["station"].ljust(5)[:5].encode()
            hdr[13:15] = tr["location"].ljust(2)[:2].encode()
            hdr[15:18] = tr["channel"].ljust(3)[:3].encode()
            hdr[18:20] = tr["network"].ljust(2)[:2].encode()
            nblk = 2 if with_b1001 else 1
            struct.pack_into(byteorder + "HHBBBBH", hdr, 20, y, doy, hh, mm, ss, 0, fract)
            struct.pack_into(byteorder + "HhhBBBBiHH", hdr, 30, cnt, fac, mul, activity_flags, 0, 0, nblk,
                             time_correction, data_off, 48)
            struct.pack_into(byteorder + "HHBBBB", hdr, 48, 1000, 56 if with_b1001 else 0, encoding,
                             1 if byteorder == ">" else 0, exp, 0)
            if with_b1001:
                struct.pack_into(byteorder + "HHBbBB", hdr, 56, 1001, 0, 100, usec, 0, payload_bytes // 64)
            out += hdr + payload
            if encoding in (ENC_STEIM1, ENC_STEIM2):
                prev = int(data[pos + cnt - 1])
            pos += cnt
            seq += 1
    return bytes(out)


# ----------------------------------------------------------------------------- decoder (the restatement)
def scan_records(buf: bytes):
    """Walk a miniSEED 2 / 3 byte string; returns one dict per data record."""
    recs = []
    off = 0
    n = len(buf)
    while off + 40 <= n:
        if buf[off:off + 3] == b"MS\x03":
            rec, reclen = scan_record3(buf, off)
            recs.append(rec)
            off += reclen
            continue
        if off + 48 > n:
            break
        h = buf[off:off + 48]
        if not (h[0:6].replace(b" ", b"0").isdigit() and h[6:7] in b"DRQM"):
            off += 64  # not a data record header: resynchronise on the smallest record unit
            continue
        year_be = struct.unpack_from(">H", h, 20)[0]
        bo = ">" if 1900 <= year_be <= 2100 else "<"
        y, doy, hh, mm, ss, _, fract = struct.unpack_from(bo + "HHBBBBH", h, 20)
        ns, fac, mul, act, _io, _dq, nblk, tcorr, data_off, blk_off = struct.unpack_from(bo + "HhhBBBBiHH", h, 30)
        enc, word_be, reclen, usec = None, None, None, 0
        b = blk_off
        guard = 0
        while b and off + b + 4 <= n and guard < 16:
            btype, bnext = struct.unpack_from(bo + "HH", buf, off + b)
            if btype == 1000:
                enc, wo, exp = struct.unpack_from("BBB", buf, off + b + 4)
                word_be, reclen = (wo == 1), 1 << exp
            elif btype == 1001:
                usec = struct.unpack_from("b", buf, off + b + 5)[0]
            b = bnext
            guard += 1
        if reclen is None:
            raise ValueError(f"record at byte {off} has no blockette 1000")
        start = btime_to_us(y, doy, hh, mm, ss, fract) + usec
        if tcorr and not (act & 0x02):
            start += tcorr * 100
        recs.append(dict(offset=off, reclen=reclen, nsamples=ns, encoding=enc, big_endian=word_be, data_offset=data_off,
                         start_us=start, rate=sample_rate(fac, mul), network=h[18:20].decode().strip(),
                         station=h[8:13].decode().strip(), location=h[13:15].decode().strip(),
                         channel=h[15:18].decode().strip()))
        off += reclen
    return recs


def _sext(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def decode_steim(payload: bytes, nsamples: int, version: int, big_endian: bool):
    w = np.frombuffer(payload[: len(payload) // 64 * 64], dtype=(">" if big_endian else "<") + "u4").astype(np.uint64)
    frames = w.reshape(-1, 16)
    diffs = []
    for f in range(frames.shape[0]):
        ctrl = int(frames[f, 0])
        for k in range(1, 16):
            nib = (ctrl >> (30 - 2 * k)) & 3
            word = int(frames[f, k])
            if nib == 0:
                continue
            if nib == 1:
                diffs += [_sext(word >> s, 8) for s in (24, 16, 8, 0)]
            elif version == 1:
                diffs += [_sext(word >> 16, 16), _sext(word, 16)] if nib == 2 else [_sext(word, 32)]
            else:
                dnib = word >> 30
                if nib == 2:
                    if dnib == 1:
                        diffs += [_sext(word, 30)]
                    elif dnib == 2:
                        diffs += [_sext(word >> 15, 15), _sext(word, 15)]
                    elif dnib == 3:
                        diffs += [_sext(word >> s, 10) for s in (20, 10, 0)]
                else:
                    if dnib == 0:
                        diffs += [_sext(word >> s, 6) for s in (24, 18, 12, 6, 0)]
                    elif dnib == 1:
                        diffs += [_sext(word >> s, 5) for s in (25, 20, 15, 10, 5, 0)]
                    elif dnib == 2:
                        diffs += [_sext(word >> s, 4) for s in (24, 20, 16, 12, 8, 4, 0)]
        if len(diffs) >= nsamples:
            break
    x0 = _sext(int(frames[0, 1]), 32)
    d = np.array(diffs[:nsamples], dtype=np.int64)
    if len(d) < nsamples:
        raise ValueError("Steim payload holds fewer samples than the header says")
    out = np.empty(nsamples, dtype=np.int64)
    if nsamples:
        d[0] = x0
        out = np.cumsum(d)
    return ((out + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)


def decode_record(buf: bytes, rec: dict):
    p = buf[rec["offset"] + rec["data_offset"]: rec["offset"] + rec["reclen"]]
    enc, n, bo = rec["encoding"], rec["nsamples"], ">" if rec["big_endian"] else "<"
    if enc in (ENC_STEIM1, ENC_STEIM2):
        return decode_steim(p, n, 1 if enc == ENC_STEIM1 else 2, rec["big_endian"])
    if enc == ENC_TEXT:
        return np.frombuffer(p, dtype=np.uint8, count=n).astype(np.int32)
    if enc == ENC_INT24:
        b = np.frombuffer(p, dtype=np.uint8, count=3 * n).reshape(-1, 3).astype(np.int32)
        if rec["big_endian"]:
            b = b[:, ::-1]
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        return np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.int32)
    dt = {ENC_INT16: "i2", ENC_INT32: "i4", ENC_FLOAT32: "f4", ENC_FLOAT64: "f8"}.get(enc)
    if dt is None:
        raise ValueError(f"unsupported encoding {enc}")
    a = np.frombuffer(p, dtype=bo + dt, count=n)
    return a.astype({"i2": np.int32, "i4": np.int32, "f4": np.float32, "f8": np.float64}[dt])


def read_mseed(buf: bytes):
    """Records -> continuous segments (libmseed trace-list rule: a record extends a segment of the
    same source id when it starts one sample period after the segment's last sample, within half
    a period).  Returns a list of dicts sorted by (id, start)."""
    recs = scan_records(buf)
    recs.sort(key=lambda r: (r["network"], r["station"], r["location"], r["channel"], r["start_us"], r["offset"]))
    segs = []
    for r in recs:
        if r["nsamples"] == 0:
            continue
        d = decode_record(buf, r)
        key = (r["network"], r["station"], r["location"], r["channel"])
        s = segs[-1] if segs else None
        period = 1e6 / r["rate"] if r["rate"] else 0.0
        if s is not None and s["key"] == key and s["rate"] == r["rate"] and s["dtype"] == d.dtype and abs(
                r["start_us"] - (s["start_us"] + round(s["n"] * period))) <= 0.5 * period:
            s["parts"].append(d)
            s["n"] += len(d)
        else:
            segs.append(dict(key=key, start_us=r["start_us"], rate=r["rate"], n=len(d), parts=[d], dtype=d.dtype))
    out = []
    for s in segs:
        net, sta, loc, cha = s["key"]
        out.append(dict(network=net, station=sta, location=loc, channel=cha, start_us=s["start_us"], rate=s["rate"],
                        data=np.concatenate(s["parts"])))
    return out


# ----------------------------------------------------------------------------- SAC
def write_sac(data, rate, start_us, network, station, location, channel, byteorder="<"):
    hf = np.full(70, -12345.0, dtype=np.float32)
    hi = np.full(40, -12345, dtype=np.int32)
    hs = [b"-12345  "] * 24
    d = _EPOCH + timedelta(microseconds=int(start_us))
    hf[0] = 1.0 / rate
    hf[5] = (d.microsecond % 1000) * 1e-6  # b: sub-millisecond remainder of the start time
    hf[6] = hf[5] + (len(data) - 1) / rate
    hi[0:6] = [d.year, d.timetuple().tm_yday, d.hour, d.minute, d.second, d.microsecond // 1000]
    hi[6] = 6      # nvhdr
    hi[9] = len(data)
    hi[15] = 1     # iftype = ITIME
    hi[35] = 1     # leven
    hs[0] = station.ljust(8)[:8].encode()
    hs[1], hs[2] = b"-12345  ", b"        "
    hs[3] = (location if location else "-12345").ljust(8)[:8].encode()
    hs[20] = channel.ljust(8)[:8].encode()
    hs[21] = network.ljust(8)[:8].encode()
    return (hf.astype(byteorder + "f4").tobytes() + hi.astype(byteorder + "i4").tobytes() + b"".join(hs)
            + np.asarray(data, dtype=byteorder + "f4").tobytes())


def read_sac(buf: bytes):
    nv_le = struct.unpack_from("<i", buf, 76 * 4)[0]
    bo = "<" if 1 <= nv_le <= 7 else ">"
    hf = np.frombuffer(buf, dtype=bo + "f4", count=70)
    hi = np.frombuffer(buf, dtype=bo + "i4", count=40, offset=280)
    ks = [buf[440 + 8 * i: 448 + 8 * i].decode("ascii", "replace").strip("\0 ") for i in range(24)]
    ks = ["" if k == "-12345" else k for k in ks]
    npts = int(hi[9])
    b = float(hf[5]) if hf[5] != -12345.0 else 0.0
    d = datetime(int(hi[0]), 1, 1, tzinfo=timezone.utc) + timedelta(days=int(hi[1]) - 1, hours=int(hi[2]),
                                                                    minutes=int(hi[3]), seconds=int(hi[4]))
    delta = d - _EPOCH
    start_us = (delta.days * 86400 + delta.seconds) * 1_000_000 + int(hi[5]) * 1000 + int(round(b * 1e6))
    data = np.frombuffer(buf, dtype=bo + "f4", count=npts, offset=632).astype(np.float32)
    return dict(network=ks[21], station=ks[0], location=ks[3], channel=ks[20], start_us=start_us,
                rate=float(np.float32(1.0) / hf[0]), data=data)


# ----------------------------------------------------------------------------- array assembly
def stream_to_array(traces, component_order="ZNE", demean=True):
    """volpick/data/convert.py:26-70 on trace dicts: common start/end over all traces, zero fill,
    ``*{c}`` channel match, multiple traces written shortest first, row-wise demean (:67),
    completeness as defined there.  Returns (start_us, data float64 (C, samples), completeness)."""
    rate = traces[0]["rate"]
    start = min(t["start_us"] for t in traces)
    end = max(t["start_us"] + (len(t["data"]) - 1) * 1e6 / rate for t in traces)
    samples = int((end - start) / 1e6 * rate) + 1
    data = np.zeros((len(component_order), samples), dtype=np.float64)
    completeness = 0.0
    for ci, c in enumerate(component_order):
        sel = [t for t in traces if t["channel"].endswith(c)]
        if len(sel) > 1:
            sel = sorted(sel, key=lambda t: len(t["data"]))
        cc = 0.0
        for t in sel:
            s0 = int((t["start_us"] - start) / 1e6 * rate)
            l = min(len(t["data"]), samples - s0)
            data[ci, s0:s0 + l] = t["data"][:l]
            cc += l
        completeness += min(1.0, cc / samples)
    if demean:
        data -= data.mean(axis=1, keepdims=True)
    return start, data, completeness / len(component_order)
