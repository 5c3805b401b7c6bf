"""Known-answer vectors for the miniSEED codec, assembled BY HAND from the bit layout in the SEED 2.4 manual
(chapter 8 "Fixed Section of Data Header", blockette 1000, appendix B "Steim compression") -- not produced by
oracle/mseed.py's encoder, so they pin the decoder (oracle, host scanner and GPU kernel alike) to the published
format rather than to the encoder's reading of it ("encoder <-> decoder symmetry by one author").

Every data word below is written out as a literal; the comments give the field arithmetic.  One 512-byte record
per case: 48-byte fixed header, blockette 1000 at byte 48, data at byte 64, one 64-byte Steim frame in use.

  frame = 16 words; W0 = sixteen 2-bit nibbles c0 .. c15, c_k describing word k (c0 = 00); in the first frame
  W1 = X0 (forward integration constant = first sample), W2 = Xn (reverse integration constant = last sample).
  Steim-1  c = 01: four 8-bit differences | 10: two 16-bit | 11: one 32-bit
  Steim-2  c = 01: four 8-bit | c = 10, dnib (top two bits of the word) 01: one 30-bit, 10: two 15-bit, 11: three 10-bit
           | c = 11, dnib 00: five 6-bit, 01: six 5-bit, 10: seven 4-bit
  The first difference of a record is x0 - x(-1) and is not used: x0 comes from X0.
"""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import mseed as OM
from volpick_amd import _lib

# ---- Steim-2: every word kind once -------------------------------------------------------------------------------
S2_DIFFS = [5, -3, 127, -128,              # W3  c=01  four 8-bit:   05 FD 7F 80
            511, -512, 7,                  # W4  c=10 dnib=11 three 10-bit: 1FF 200 007
            1, -1, 7, -8, 0, 2, -2,        # W5  c=11 dnib=10 seven 4-bit:  1 F 7 8 0 2 E
            -100000,                       # W6  c=10 dnib=01 one 30-bit:   2^30 - 100000 = 0x3FFE7960
            16383, -16384,                 # W7  c=10 dnib=10 two 15-bit:   3FFF 4000
            31, -32, 1, -1, 0,             # W8  c=11 dnib=00 five 6-bit:   1F 20 01 3F 00
            15, -16, 1, -1, 2, -2]         # W9  c=11 dnib=01 six 5-bit:    0F 10 01 1F 02 1E
S2_WORDS = [
    0x01BAF000,  # W0: nibbles 00 00 00 01 | 10 11 10 10 | 11 11 00 00 | 00 00 00 00
    None, None,  # W1 = X0, W2 = Xn (filled from the sample list below)
    0x05FD7F80,
    0xDFF80007,  # 11 | 0111111111 | 1000000000 | 0000000111
    0x81F7802E,  # 10 | 00 | 0001 1111 0111 1000 0000 0010 1110
    0x7FFE7960,  # 01 | 0x3FFE7960
    0x9FFFC000,  # 10 | 011111111111111 | 100000000000000
    0x1F801FC0,  # 00 | 011111 100000 000001 111111 000000
    0x5F00FC5E,  # 01 | 01111 10000 00001 11111 00010 11110
]
S2_X0 = 1000

# ---- Steim-1: every word kind once -------------------------------------------------------------------------------
S1_DIFFS = [5, -3, 127, -128,  # W3  c=01
            32767, -32768,     # W4  c=10: 7FFF 8000
            1234567]           # W5  c=11: 0x0012D687
S1_WORDS = [0x01B00000,        # nibbles 00 00 00 01 | 10 11 00 00 | ...
            None, None, 0x05FD7F80, 0x7FFF8000, 0x0012D687]
S1_X0 = -77


def _samples(x0, diffs):
    x = [x0]
    for d in diffs[1:]:  # the first difference is not used
        x.append(x[-1] + d)
    return np.array(x, dtype=np.int32)


def _record(words, x0, diffs, encoding, bo):
    """One 512-byte data record around a single Steim frame; every header field by hand (SEED 2.4 chapter 8)."""
    x = _samples(x0, diffs)
    w = list(words) + [0] * (16 - len(words))
    w[1], w[2] = x0 & 0xFFFFFFFF, int(x[-1]) & 0xFFFFFFFF
    hdr = b"000001" + b"D" + b" " + b"KNOWN" + b"00" + b"BHZ" + b"XX"          # seq, quality, reserved, sta, loc, cha, net
    hdr += struct.pack(bo + "HHBBBBH", 2005, 151, 21, 4, 52, 0, 1100)            # BTIME 2005-151 21:04:52.1100
    hdr += struct.pack(bo + "HhhBBBBiHH", len(x), 100, 1, 0, 0, 0, 1, 0, 64, 48)  # n, rate 100 x 1, flags, 1 blockette, data @64, b1000 @48
    assert len(hdr) == 48
    b1000 = struct.pack(bo + "HHBBBB", 1000, 0, encoding, 1 if bo == ">" else 0, 9, 0)  # 2^9 = 512-byte record
    frame = struct.pack(bo + "16I", *w)
    rec = hdr + b1000 + b"\0" * 8 + frame
    return rec + b"\0" * (512 - len(rec)), x


CASES = [("steim2", S2_WORDS, S2_X0, S2_DIFFS, 11), ("steim1", S1_WORDS, S1_X0, S1_DIFFS, 10)]


def test_hand_computed_words():
    """The literals above are what the layout says (field arithmetic spelled out once more, independently)."""
    def pack(fields, bits, dnib=None):
        v, sh = 0, (30 if dnib is not None else 32)
        for f in fields:
            sh -= bits
            v |= (f & ((1 << bits) - 1)) << sh
        return v | ((dnib << 30) if dnib is not None else 0)

    assert pack(S2_DIFFS[0:4], 8) == S2_WORDS[3] and pack(S2_DIFFS[4:7], 10, 3) == S2_WORDS[4]
    # seven 4-bit differences occupy the low 28 bits (two spare bits under the dnib)
    w5 = (2 << 30) | sum((d & 0xF) << (24 - 4 * i) for i, d in enumerate(S2_DIFFS[7:14]))
    assert w5 == S2_WORDS[5]
    assert (1 << 30) | (S2_DIFFS[14] & 0x3FFFFFFF) == S2_WORDS[6]
    assert pack(S2_DIFFS[15:17], 15, 2) == S2_WORDS[7]
    assert sum((d & 0x3F) << (24 - 6 * i) for i, d in enumerate(S2_DIFFS[17:22])) == S2_WORDS[8]
    assert (1 << 30) | sum((d & 0x1F) << (25 - 5 * i) for i, d in enumerate(S2_DIFFS[22:28])) == S2_WORDS[9]
    nib = [0, 0, 0, 1, 2, 3, 2, 2, 3, 3] + [0] * 6
    assert sum(c << (30 - 2 * k) for k, c in enumerate(nib)) == S2_WORDS[0]
    assert pack(S1_DIFFS[4:6], 16) == S1_WORDS[4] and S1_DIFFS[6] == S1_WORDS[5]
    assert sum(c << (30 - 2 * k) for k, c in enumerate([0, 0, 0, 1, 2, 3] + [0] * 10)) == S1_WORDS[0]
    assert len(_samples(S2_X0, S2_DIFFS)) == 28 and _samples(S2_X0, S2_DIFFS)[-1] == 1000 - 3 + 127 - 128 + 6 - 1 - 100000 - 1 - 1 - 1


@pytest.mark.parametrize("name,words,x0,diffs,enc", CASES)
@pytest.mark.parametrize("bo", [">", "<"])
def test_oracle_and_host_scanner_on_hand_built_records(lib, name, words, x0, diffs, enc, bo):
    rec, want = _record(words, x0, diffs, enc, bo)
    recs = OM.scan_records(rec)
    assert len(recs) == 1
    r = recs[0]
    assert (r["nsamples"], r["encoding"], r["reclen"], r["data_offset"], r["big_endian"]) == (len(want), enc, 512, 64, bo == ">")
    assert (r["network"], r["station"], r["location"], r["channel"], r["rate"]) == ("XX", "KNOWN", "00", "BHZ", 100.0)
    assert r["start_us"] == OM.btime_to_us(2005, 151, 21, 4, 52, 1100)
    assert np.array_equal(OM.decode_record(rec, r), want)
    c = (_lib.VpMseedRecord * 2)()
    n = C.c_int64()
    assert lib.vp_mseed_scan(rec, len(rec), c, 2, C.byref(n)) == 0 and n.value == 1
    assert (c[0].nsamples, c[0].encoding, c[0].reclen, c[0].data_offset, c[0].big_endian) == (len(want), enc, 512, 64, int(bo == ">"))
    assert (c[0].network, c[0].station, c[0].location, c[0].channel) == (b"XX", b"KNOWN", b"00", b"BHZ")
    assert c[0].start_us == r["start_us"] and c[0].sample_rate == 100.0
    # 2005-151 is May 31: the date of the reference's demo pick (Final_models/demo.ipynb:242)
    assert r["start_us"] == 1117573492_110000


@pytest.mark.gpu
@pytest.mark.parametrize("name,words,x0,diffs,enc", CASES)
@pytest.mark.parametrize("bo", [">", "<"])
def test_gpu_decoder_on_hand_built_records(name, words, x0, diffs, enc, bo):
    """vp_mseed_decode (HIP: one wavefront per record, DPP prefix scans over the frame's words) on the same bytes:
    the hand-computed samples, integrity constant accepted; and a flipped Xn is reported."""
    import volpick_amd as va

    lib = _lib.load()
    rec, want = _record(words, x0, diffs, enc, bo)
    st = va.read(rec * 3)  # three identical records: same channel, overlapping times -> decoded independently
    assert sum(tr.stats.npts for tr in st) == 3 * len(want)
    c = (_lib.VpMseedRecord * 1)()
    n = C.c_int64()
    assert lib.vp_mseed_scan(rec, len(rec), c, 1, C.byref(n)) == 0
    out = np.full(len(want) + 4, -999, np.int32)
    idx = np.array([2], np.int64)
    status = np.full(1, -7, np.int32)
    rc = lib.vp_mseed_decode(0, rec, _lib.VP_MEM_HOST, len(rec), c, idx.ctypes.data_as(C.POINTER(C.c_int64)), None, 1,
                             _lib.VP_SAMPLES_INT32, out.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, out.size, 0,
                             status.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0 and status[0] == 0
    assert np.array_equal(out[2:2 + len(want)], want) and (out[:2] == -999).all() and (out[2 + len(want):] == -999).all()
    bad = bytearray(rec)
    off = 64 + 8 + (3 if bo == ">" else 0)
    bad[off] ^= 1  # least significant bit of Xn
    rc = lib.vp_mseed_decode(0, bytes(bad), _lib.VP_MEM_HOST, len(bad), c, idx.ctypes.data_as(C.POINTER(C.c_int64)), None, 1,
                             _lib.VP_SAMPLES_INT32, out.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, out.size, 0,
                             status.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0 and status[0] == 1 and np.array_equal(out[2:2 + len(want)], want)
