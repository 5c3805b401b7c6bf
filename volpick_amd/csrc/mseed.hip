// miniSEED 2 / miniSEED 3 ingestion: host record scanner and a one-wavefront-per-record
// decoder.  Replaces obspy.read()'s libmseed unpacking ahead of stream_to_array
// (/root/reference volpick/data/convert.py:7,26-70).
//
// The decode is byte/integer work bound by HBM: ~1.3-2 compressed bytes in and 4 bytes out per
// sample.  Records are independent (every Steim record carries its own forward integration
// constant), so the grid is one wavefront per record; a wavefront takes 64 words (four frames)
// at a time, and the two serial steps -- where each word's differences start (1..7 per word)
// and the running sum -- are two DPP prefix scans over the lanes; the <= 7 differences of a
// word are integrated in registers and stored straight to their place in the output.
#include <mutex>

#include "vp_common.h"

namespace vp {
namespace {

struct DevRec {
  long long payload;    // byte offset of the data section in the buffer
  long long out_index;  // destination of the record's first sample
  int nbytes;           // payload bytes
  int nsamples;         // header sample count
  int nwrite;           // samples to store (<= nsamples)
  int enc;              // encoding | big_endian << 8
};

// 32-bit word of a payload that may start at any byte (miniSEED 3): two aligned loads and v_alignbyte_b32.  The second
// load is skipped when none of its bytes lies inside the buffer; when some do, the aligned word holding them cannot cross a
// page, so reading its last bytes beyond `nbytes` touches no other allocation's page.
__device__ __forceinline__ unsigned load_word(const uint8_t* __restrict__ buf, const long long byte, const long long nbytes) {
  const int sh = static_cast<int>(byte & 3);
  const unsigned* a = reinterpret_cast<const unsigned*>(buf + (byte - sh));
  const unsigned lo = a[0];
  if (sh == 0) return lo;
  const unsigned hi = byte - sh + 4 < nbytes ? a[1] : 0u;
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// Inclusive prefix sum over the 64 lanes in seven DPP adds (no LDS traffic): three row_shr steps
// inside each group of four lanes, two bank-masked row_shr steps inside each row of 16, then
// row_bcast:15 / row_bcast:31 carry the row totals across rows.  Masked-off lanes add `old` = 0.
__device__ __forceinline__ int wave_incl_scan(const int v) {
  int s = v;
  s += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  s += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  s += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, false);  // row_shr:3
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, false);  // row_shr:4, banks 1-3
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, false);  // row_shr:8, banks 2-3
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return s;
}

constexpr int REC_PER_WG = 4;

template <typename OutT>
__global__ __launch_bounds__(64 * REC_PER_WG) void mseed_decode_kernel(const uint8_t* __restrict__ buf,
                                                                       const DevRec* __restrict__ recs,
                                                                       const long long n_recs, OutT* __restrict__ out,
                                                                       const long long out_len,
                                                                       int* __restrict__ status, const long long nbytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * REC_PER_WG + wave;
  if (r >= n_recs) return;  // no workgroup barrier below: waves are independent
  const DevRec rec = recs[r];
  const int enc = rec.enc & 0xff;
  const bool be = (rec.enc >> 8) != 0;
  const int ns = rec.nsamples;
  auto store = [&](const int i, const auto v) {
    if (i < rec.nwrite) {
      const long long o = rec.out_index + i;
      if (o >= 0 && o < out_len) out[o] = static_cast<OutT>(v);
    }
  };
  if (enc == 10 || enc == 11) {
    const int nwords = (rec.nbytes >> 6) << 4;  // whole 64-byte frames
    int produced = 0, carry = 0;
    unsigned x0 = 0, xn = 0;
    const bool whole = rec.nwrite >= ns && rec.out_index >= 0 && rec.out_index + ns <= out_len;
    OutT* dst = out + rec.out_index;
    for (int base = 0; base < nwords && produced < ns; base += 64) {
      const int g = base + lane;
      unsigned word = 0;
      if (g < nwords) {
        word = load_word(buf, rec.payload + 4ll * g, nbytes);
        if (be) word = __builtin_bswap32(word);
      }
      const int idx = lane & 15;
      const unsigned ctrl = __shfl(word, lane & ~15, 64);  // word 0 of this lane's frame
      const int nib = idx == 0 ? 0 : (ctrl >> (30 - 2 * idx)) & 3;
      if (base == 0) {
        x0 = __shfl(word, 1, 64);  // forward integration constant: the record's first sample
        xn = __shfl(word, 2, 64);  // reverse integration constant: its last sample
      }
      int cnt = 0, bits = 0;
      const unsigned dnib = word >> 30;
      if (nib == 1) {
        cnt = 4, bits = 8;
      } else if (enc == 10) {
        if (nib == 2) cnt = 2, bits = 16;
        if (nib == 3) cnt = 1, bits = 32;
      } else if (nib == 2) {
        if (dnib == 1) cnt = 1, bits = 30;
        if (dnib == 2) cnt = 2, bits = 15;
        if (dnib == 3) cnt = 3, bits = 10;
      } else if (nib == 3) {
        if (dnib == 0) cnt = 5, bits = 6;
        if (dnib == 1) cnt = 6, bits = 5;
        if (dnib == 2) cnt = 7, bits = 4;
      }
      // where this word's differences start in the record
      const int cincl = wave_incl_scan(cnt);
      const int total = __shfl(cincl, 63, 64);
      const int pos = produced + cincl - cnt;
      // running sums of the word's own differences (most significant field first), in registers
      int p[7];
      int run = 0;
      int sh = bits * (cnt - 1);
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        int d = bits == 32 ? static_cast<int>(word) : __builtin_amdgcn_sbfe(static_cast<int>(word), sh, bits);  // v_bfe_i32
        if (j >= cnt) d = 0;
        if (j == 0 && pos == 0 && cnt > 0) d = static_cast<int>(x0);  // x[0] = X0; the first difference is not used
        run += d;
        p[j] = run;
        sh -= bits;
      }
      const int sincl = wave_incl_scan(run);
      const int before = carry + sincl - run;  // value of the sample preceding this word's first one
      if (whole && produced + total <= ns) {  // every sample of the chunk lands inside `out`: no per-sample checks
#pragma unroll
        for (int j = 0; j < 7; ++j)
          if (j < cnt) dst[pos + j] = static_cast<OutT>(before + p[j]);
        if (produced + total == ns && status && pos + cnt == ns && cnt > 0 && before + run != static_cast<int>(xn))
          status[r] = 1;
      } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int k = pos + j;
          if (j < cnt && k < ns) {
            const int v = before + p[j];
            store(k, v);
            if (k == ns - 1 && status && v != static_cast<int>(xn)) status[r] = 1;
          }
        }
      }
      carry += __shfl(sincl, 63, 64);
      produced += total;
    }
    if (produced < ns && lane == 0 && status) status[r] = 2;
    return;
  }
  const int width = enc == 0 ? 1 : enc == 1 ? 2 : enc == 2 ? 3 : enc == 5 ? 8 : 4;
  int n = rec.nbytes / width;
  if (n < ns) {
    if (lane == 0 && status) status[r] = 2;
  } else {
    n = ns;
  }
  const uint8_t* p = buf + rec.payload;
  for (int i = lane; i < n; i += 64) {
    if (enc == 0) {  // text: the byte's value
      store(i, static_cast<int>(p[i]));
    } else if (enc == 1) {
      unsigned h;
      if ((rec.payload & 1) == 0) {
        h = reinterpret_cast<const unsigned short*>(p)[i];
        if (be) h = __builtin_bswap16(static_cast<unsigned short>(h));
      } else {
        h = be ? (p[2 * i] << 8) | p[2 * i + 1] : (p[2 * i + 1] << 8) | p[2 * i];
      }
      store(i, static_cast<int>(static_cast<short>(h)));
    } else if (enc == 2) {  // 24-bit two's complement
      const unsigned b0 = p[3 * i], b1 = p[3 * i + 1], b2 = p[3 * i + 2];
      const unsigned u = be ? (b0 << 16) | (b1 << 8) | b2 : (b2 << 16) | (b1 << 8) | b0;
      store(i, __builtin_amdgcn_sbfe(static_cast<int>(u), 0, 24));
    } else if (enc == 3) {
      unsigned u = load_word(buf, rec.payload + 4ll * i, nbytes);
      if (be) u = __builtin_bswap32(u);
      store(i, static_cast<int>(u));
    } else if (enc == 4) {
      unsigned u = load_word(buf, rec.payload + 4ll * i, nbytes);
      if (be) u = __builtin_bswap32(u);
      store(i, __uint_as_float(u));
    } else {  // float64, as two words
      unsigned a = load_word(buf, rec.payload + 8ll * i, nbytes), b = load_word(buf, rec.payload + 8ll * i + 4, nbytes);
      unsigned long long u = be ? ((unsigned long long)__builtin_bswap32(a) << 32) | __builtin_bswap32(b)
                                : ((unsigned long long)b << 32) | a;
      store(i, __longlong_as_double(static_cast<long long>(u)));
    }
  }
}

// ------------------------------------------------------------------------------------------ host
inline unsigned rd16(const uint8_t* p, bool be) { return be ? (p[0] << 8) | p[1] : (p[1] << 8) | p[0]; }
inline unsigned rd32(const uint8_t* p, bool be) {
  return be ? ((unsigned)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]
            : ((unsigned)p[3] << 24) | (p[2] << 16) | (p[1] << 8) | p[0];
}

// days since 1970-01-01 of January 1st of `y` (proleptic Gregorian)
long long days_to_year(int y) {
  const long long yy = y - 1;
  return yy * 365 + yy / 4 - yy / 100 + yy / 400 - 719162;
}

void copy_code(char* dst, int cap, const uint8_t* src, int n) {
  int b = 0, e = n;
  while (b < e && (src[b] == ' ' || src[b] == 0)) ++b;
  while (e > b && (src[e - 1] == ' ' || src[e - 1] == 0)) --e;
  int k = 0;
  for (int i = b; i < e && k < cap - 1; ++i) dst[k++] = (char)src[i];
  while (k < cap) dst[k++] = 0;
}

double seed_rate(int factor, int mult) {
  if (factor == 0) return 0.0;
  double r = factor > 0 ? (double)factor : -1.0 / factor;
  if (mult > 0) r *= mult;
  if (mult < 0) r /= -mult;
  return r;
}

// CRC-32C (Castagnoli, reflected 0x82F63B78) as miniSEED 3 defines it over the record with its CRC field zeroed:
// slicing-by-8 tables built once.
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
uint32_t crc32c_update(uint32_t crc, const uint8_t* p, size_t n) {
  static const Crc32cTables T;
  while (n >= 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^ T.t[3][hi & 0xff] ^
          T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xff];
  return crc;
}
uint32_t mseed3_crc(const uint8_t* rec, size_t reclen) {  // bytes 28..31 count as zero
  static const uint8_t zero[4] = {0, 0, 0, 0};
  uint32_t c = 0xffffffffu;
  c = crc32c_update(c, rec, 28);
  c = crc32c_update(c, zero, 4);
  c = crc32c_update(c, rec + 32, reclen - 32);
  return ~c;
}

// "FDSN:NET_STA_LOC_BAND_SOURCE_SUBSOURCE" -> the four codes of a record; false when the identifier has another shape or a
// code does not fit its field
bool split_sid(const uint8_t* sid, int n, vp_mseed_record* m) {
  if (n < 5 || std::memcmp(sid, "FDSN:", 5) != 0) return false;
  const uint8_t* f[6];
  int len[6], k = 0;
  f[0] = sid + 5;
  for (int i = 5; i < n; ++i) {
    if (sid[i] == '_') {
      if (k == 5) return false;
      len[k] = (int)(sid + i - f[k]);
      f[++k] = sid + i + 1;
    }
  }
  if (k != 5) return false;
  len[5] = (int)(sid + n - f[5]);
  if (len[0] > 3 || len[1] > 7 || len[2] > 3) return false;
  copy_code(m->network, 4, f[0], len[0]);
  copy_code(m->station, 8, f[1], len[1]);
  copy_code(m->location, 4, f[2], len[2]);
  if (len[3] > 1 || len[4] > 1 || len[5] > 1) return false;  // the joined form would be B_S_SS: longer than the field
  uint8_t cha[3];
  int c = 0;
  for (int j = 3; j < 6; ++j)
    if (len[j]) cha[c++] = f[j][0];
  copy_code(m->channel, 4, cha, c);
  return true;
}

// One miniSEED 3 record at `off`.  Returns VP_OK and the record length, or an error with its message set.
int scan_mseed3(const uint8_t* buf, size_t nbytes, size_t off, vp_mseed_record* m, size_t* reclen) {
  const uint8_t* h = buf + off;
  const unsigned nsec = rd32(h + 4, false);
  const int year = (int)rd16(h + 8, false), doy = (int)rd16(h + 10, false);
  const int hh = h[12], mm = h[13], ss = h[14], enc = h[15];
  uint64_t rbits = 0;
  for (int i = 7; i >= 0; --i) rbits = (rbits << 8) | h[16 + i];
  double rate;
  std::memcpy(&rate, &rbits, 8);
  const unsigned ns = rd32(h + 24, false), crc = rd32(h + 28, false);
  const int sid_len = h[33], extra_len = (int)rd16(h + 34, false);
  const unsigned payload = rd32(h + 36, false);
  const size_t len = 40 + (size_t)sid_len + extra_len + payload;
  if (payload > (1u << 30) || off + len > nbytes) {
    set_error("miniSEED 3 record at byte %zu runs past the end of the buffer", off);
    return VP_ERR_INVALID;
  }
  if (year < 1900 || year > 2100 || doy < 1 || doy > 366 || nsec > 999999999u || ns > 0x7fffffffu) {
    set_error("miniSEED 3 record at byte %zu: implausible start time or sample count", off);
    return VP_ERR_INVALID;
  }
  if (mseed3_crc(h, len) != crc) {
    set_error("miniSEED 3 record at byte %zu: CRC-32C mismatch", off);
    return VP_ERR_INVALID;
  }
  *reclen = len;
  if (!m) return VP_OK;
  if (!split_sid(h + 40, sid_len, m)) {
    set_error("miniSEED 3 record at byte %zu: source identifier '%.*s' is not FDSN:NET_STA_LOC_B_S_SS with codes that fit "
              "vp_mseed_record", off, sid_len, (const char*)(h + 40));
    return VP_ERR_UNSUPPORTED;
  }
  m->offset = (int64_t)off;
  m->start_us = ((days_to_year(year) + doy - 1) * 86400LL + hh * 3600 + mm * 60 + ss) * 1000000LL + nsec / 1000;
  m->sample_rate = rate > 0 ? rate : rate < 0 ? -1.0 / rate : 0.0;
  m->reclen = (int32_t)len;
  m->data_offset = 40 + sid_len + extra_len;
  m->nsamples = (int32_t)ns;
  m->encoding = enc;
  m->big_endian = enc == 10 || enc == 11 || enc == 19;  // Steim frames are big-endian words; everything else little-endian
  m->quality = 0x300 | h[32];
  return VP_OK;
}

bool is_data_header(const uint8_t* h) {
  for (int i = 0; i < 6; ++i)
    if (!((h[i] >= '0' && h[i] <= '9') || h[i] == ' ')) return false;
  return h[6] == 'D' || h[6] == 'R' || h[6] == 'Q' || h[6] == 'M';
}

int build_dev_recs(const vp_mseed_record* recs, const int64_t* out_index, const int64_t* out_count, int64_t n_recs,
                   size_t nbytes, int out_kind, std::vector<DevRec>* dev, std::vector<int64_t>* origin) {
  for (int64_t r = 0; r < n_recs; ++r) {
    const vp_mseed_record& m = recs[r];
    if (out_index[r] < 0 || m.nsamples <= 0) continue;
    VP_REQUIRE(m.offset >= 0 && m.data_offset >= 40 && m.data_offset <= m.reclen &&
                   (size_t)(m.offset + m.reclen) <= nbytes,
               "mseed record %lld lies outside the buffer", (long long)r);
    const int e = m.encoding;
    if (!(e == 0 || e == 1 || e == 2 || e == 3 || e == 4 || e == 5 || e == 10 || e == 11)) {
      set_error("mseed record %lld: unsupported encoding %d", (long long)r, e);
      return VP_ERR_UNSUPPORTED;
    }
    VP_REQUIRE(out_kind == VP_SAMPLES_FLOAT32 || (e != 4 && e != 5),
               "mseed record %lld holds floating-point samples; decode with VP_SAMPLES_FLOAT32", (long long)r);
    DevRec d;
    d.payload = m.offset + m.data_offset;
    d.out_index = out_index[r];
    d.nbytes = m.reclen - m.data_offset;
    d.nsamples = m.nsamples;
    d.nwrite = m.nsamples;
    if (out_count && out_count[r] < d.nwrite) d.nwrite = (int)(out_count[r] < 0 ? 0 : out_count[r]);
    d.enc = e | (m.big_endian ? 256 : 0);
    dev->push_back(d);
    origin->push_back(r);
  }
  return VP_OK;
}

void launch_decode(const uint8_t* buf, long long nbytes, const DevRec* recs, long long n, int out_kind, void* out,
                   long long out_len, int* status, hipStream_t s) {
  const dim3 grid((unsigned)((n + REC_PER_WG - 1) / REC_PER_WG)), block(64 * REC_PER_WG);
  if (out_kind == VP_SAMPLES_INT32) {
    hipLaunchKernelGGL(mseed_decode_kernel<int>, grid, block, 0, s, buf, recs, n, (int*)out, out_len, status, nbytes);
  } else {
    hipLaunchKernelGGL(mseed_decode_kernel<float>, grid, block, 0, s, buf, recs, n, (float*)out, out_len, status, nbytes);
  }
}

// Device scratch of vp_mseed_decode, per device, grow-only, reused from call to call: the file image, the sample array (host
// destinations), the record table and the status words.  (hipMalloc + hipFree of 35 + 104 MB around every station-day cost
// more than the decode itself; a call holds the device's lock from its first use of the scratch to its last.)
struct MseedScratch {
  std::mutex mu;
  void* p[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  int grow(int slot, size_t bytes, void** out) {
    if (bytes > cap[slot]) {
      if (p[slot]) (void)hipFree(p[slot]);
      p[slot] = nullptr;
      cap[slot] = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      if (hipMalloc(&p[slot], want) != hipSuccess) {
        (void)hipGetLastError();  // the failure is reported here: do not leave HIP's sticky last error for an unrelated later check
        p[slot] = nullptr;
        set_error("vp_mseed_decode: cannot allocate %zu bytes of device scratch", want);
        return VP_ERR_HIP;
      }
      cap[slot] = want;
    }
    *out = p[slot];
    return VP_OK;
  }
  size_t release() {  // caller holds mu
    size_t freed = 0;
    for (int i = 0; i < 4; ++i) {
      if (p[i]) (void)hipFree(p[i]);
      freed += cap[i];
      p[i] = nullptr;
      cap[i] = 0;
    }
    return freed;
  }
};
MseedScratch& mseed_scratch(int device) {
  static MseedScratch pool[64];
  return pool[(unsigned)device % 64];
}

}  // namespace
}  // namespace vp

using namespace vp;

extern "C" int vp_mseed_release_scratch(int device_id, size_t* bytes_freed) {
  VP_REQUIRE(device_id >= 0, "vp_mseed_release_scratch: device index");
  MseedScratch& sc = mseed_scratch(device_id);
  std::lock_guard<std::mutex> lock(sc.mu);  // behind any decode call in flight on this device
  VP_HIP(hipSetDevice(device_id));
  const size_t freed = sc.release();
  if (bytes_freed) *bytes_freed = freed;
  return VP_OK;
}

extern "C" int vp_mseed_scan(const uint8_t* buf, size_t nbytes, vp_mseed_record* recs, int64_t cap,
                             int64_t* n_found) {
  VP_REQUIRE(buf && n_found && (recs || cap == 0), "vp_mseed_scan: null argument");
  int64_t n = 0;
  size_t off = 0;
  while (off + 40 <= nbytes) {
    const uint8_t* h = buf + off;
    if (h[0] == 'M' && h[1] == 'S' && h[2] == 3) {
      size_t len = 0;
      vp_mseed_record scratch;
      const int rc = scan_mseed3(buf, nbytes, off, n < cap ? &recs[n] : &scratch, &len);
      if (rc != VP_OK) return rc;
      ++n;
      off += len;
      continue;
    }
    if (off + 48 > nbytes) break;
    if (!is_data_header(h)) {
      off += 64;
      continue;
    }
    const unsigned year_be = rd16(h + 20, true);
    const bool be = year_be >= 1900 && year_be <= 2100;
    const int year = (int)rd16(h + 20, be), doy = (int)rd16(h + 22, be);
    const int hh = h[24], mm = h[25], ss = h[26], fract = (int)rd16(h + 28, be);
    const int ns = (int)rd16(h + 30, be);
    const int fac = (int16_t)rd16(h + 32, be), mul = (int16_t)rd16(h + 34, be);
    const int act = h[36];
    const int tcorr = (int32_t)rd32(h + 40, be);
    const int data_off = (int)rd16(h + 44, be);
    int blk = (int)rd16(h + 46, be);
    int enc = -1, wo = 1, reclen = 0, usec = 0;
    for (int guard = 0; blk && off + blk + 8 <= nbytes && guard < 16; ++guard) {
      const uint8_t* b = buf + off + blk;
      const int type = (int)rd16(b, be), next = (int)rd16(b + 2, be);
      if (type == 1000) {
        enc = b[4];
        wo = b[5];
        reclen = (b[6] >= 7 && b[6] <= 24) ? 1 << b[6] : 0;
      } else if (type == 1001) {
        usec = (int8_t)b[5];
      }
      if (next <= blk) break;
      blk = next;
    }
    if (reclen == 0) {
      set_error("mseed record at byte %zu has no (valid) blockette 1000", off);
      return VP_ERR_INVALID;
    }
    if (year < 1900 || year > 2100 || doy < 1 || doy > 366) {
      set_error("mseed record at byte %zu: implausible start time", off);
      return VP_ERR_INVALID;
    }
    if (n < cap) {
      vp_mseed_record& m = recs[n];
      m.offset = (int64_t)off;
      int64_t us = ((days_to_year(year) + doy - 1) * 86400LL + hh * 3600 + mm * 60 + ss) * 1000000LL + fract * 100LL + usec;
      if (tcorr != 0 && !(act & 0x02)) us += (int64_t)tcorr * 100;
      m.start_us = us;
      m.sample_rate = seed_rate(fac, mul);
      m.reclen = reclen;
      m.data_offset = data_off;
      m.nsamples = ns;
      m.encoding = enc;
      m.big_endian = wo == 1;
      m.quality = h[6];
      copy_code(m.network, 4, h + 18, 2);
      copy_code(m.station, 8, h + 8, 5);
      copy_code(m.location, 4, h + 13, 2);
      copy_code(m.channel, 4, h + 15, 3);
    }
    ++n;
    off += reclen;
  }
  *n_found = n;
  return VP_OK;
}

extern "C" int vp_mseed_decode(int device_id, const uint8_t* buf, int buf_mem, size_t nbytes,
                               const vp_mseed_record* recs, const int64_t* out_index, const int64_t* out_count,
                               int64_t n_recs, int out_kind, void* out, int out_mem, int64_t out_len, int zero_fill,
                               int32_t* status) {
  VP_REQUIRE(buf && out && out_len >= 0 && n_recs >= 0 && (n_recs == 0 || (recs && out_index)),
             "vp_mseed_decode: null argument");
  VP_REQUIRE(out_kind == VP_SAMPLES_INT32 || out_kind == VP_SAMPLES_FLOAT32, "vp_mseed_decode: bad out_kind");
  std::vector<DevRec> dev;
  std::vector<int64_t> origin;
  const int rc = build_dev_recs(recs, out_index, out_count, n_recs, nbytes, out_kind, &dev, &origin);
  if (rc != VP_OK) return rc;
  if (status)
    for (int64_t r = 0; r < n_recs; ++r) status[r] = 0;
  VP_HIP(hipSetDevice(device_id));
  hipStream_t s = nullptr;  // the null stream keeps this call self-contained (no handle, one sync at the end)
  MseedScratch& sc = mseed_scratch(device_id);
  std::lock_guard<std::mutex> lock(sc.mu);
  void *dbuf = nullptr, *dout = nullptr, *drec = nullptr, *dstat = nullptr;
  const uint8_t* bufp = buf;
  if (buf_mem == VP_MEM_HOST) {
    if (const int rc2 = sc.grow(0, nbytes ? nbytes : 4, &dbuf)) return rc2;
    VP_HIP(hipMemcpyAsync(dbuf, buf, nbytes, hipMemcpyHostToDevice, s));
    bufp = (const uint8_t*)dbuf;
  }
  VP_REQUIRE(((uintptr_t)bufp & 3) == 0, "vp_mseed_decode: buffer is not 4-byte aligned");
  void* outp = out;
  const size_t out_bytes = (size_t)out_len * 4;
  if (out_mem == VP_MEM_HOST) {
    if (const int rc2 = sc.grow(1, out_bytes ? out_bytes : 4, &dout)) return rc2;
    outp = dout;
    if (!zero_fill) VP_HIP(hipMemcpyAsync(outp, out, out_bytes, hipMemcpyHostToDevice, s));
  }
  if (zero_fill) VP_HIP(hipMemsetAsync(outp, 0, out_bytes, s));
  if (!dev.empty()) {
    if (const int rc2 = sc.grow(2, dev.size() * sizeof(DevRec), &drec)) return rc2;
    VP_HIP(hipMemcpyAsync(drec, dev.data(), dev.size() * sizeof(DevRec), hipMemcpyHostToDevice, s));
    if (status) {
      if (const int rc2 = sc.grow(3, dev.size() * sizeof(int), &dstat)) return rc2;
      VP_HIP(hipMemsetAsync(dstat, 0, dev.size() * sizeof(int), s));
    }
    launch_decode(bufp, (long long)nbytes, (const DevRec*)drec, (long long)dev.size(), out_kind, outp, out_len, (int*)dstat, s);
    VP_HIP(hipGetLastError());
  }
  if (out_mem == VP_MEM_HOST) VP_HIP(hipMemcpyAsync(out, outp, out_bytes, hipMemcpyDeviceToHost, s));
  std::vector<int> hstat(dev.size());
  if (status && !dev.empty())
    VP_HIP(hipMemcpyAsync(hstat.data(), dstat, dev.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  VP_HIP(hipStreamSynchronize(s));
  if (status)
    for (size_t i = 0; i < dev.size(); ++i) status[origin[i]] = hstat[i];
  return VP_OK;
}

extern "C" int vp_mseed_decode_bench(int device_id, const uint8_t* buf_dev, size_t nbytes, const vp_mseed_record* recs,
                                     const int64_t* out_index, int64_t n_recs, int out_kind, void* out_dev,
                                     int64_t out_len, int iters, float* ms) {
  VP_REQUIRE(buf_dev && out_dev && recs && out_index && ms && iters > 0 && n_recs > 0, "vp_mseed_decode_bench: bad argument");
  std::vector<DevRec> dev;
  std::vector<int64_t> origin;
  const int rc = build_dev_recs(recs, out_index, nullptr, n_recs, nbytes, out_kind, &dev, &origin);
  if (rc != VP_OK) return rc;
  VP_REQUIRE(!dev.empty(), "vp_mseed_decode_bench: nothing to decode");
  VP_HIP(hipSetDevice(device_id));
  struct Owned {  // frees on scope exit
    void* p = nullptr;
    ~Owned() {
      if (p) (void)hipFree(p);
    }
  } drec;
  VP_HIP(hipMalloc(&drec.p, dev.size() * sizeof(DevRec)));
  VP_HIP(hipMemcpy(drec.p, dev.data(), dev.size() * sizeof(DevRec), hipMemcpyHostToDevice));
  hipStream_t s;
  VP_HIP(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  VP_HIP(hipEventCreate(&e0));
  VP_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    launch_decode(buf_dev, (long long)nbytes, (const DevRec*)drec.p, (long long)dev.size(), out_kind, out_dev, out_len, nullptr, s);
  VP_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i)
    launch_decode(buf_dev, (long long)nbytes, (const DevRec*)drec.p, (long long)dev.size(), out_kind, out_dev, out_len, nullptr, s);
  VP_HIP(hipEventRecord(e1, s));
  VP_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  VP_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s);
  return VP_OK;
}
