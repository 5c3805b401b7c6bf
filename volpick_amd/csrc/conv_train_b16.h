// The training step's convs on the bf16 matrix cores.
//
// In the bf16 mode of the trainer the activations and gradients ARE bfloat16 (train_phasenet.hip), the weights fp32.  The
// fp32-MFMA kernel (conv_mfma.h) widens the rows and spends 32 matrix cycles per K = 4; its launches keep the fp32 matrix
// pipe 0.43-0.69 busy and are bound by it (tools/pmc_sq_train.sh).  Here the weights are cut into their three bf16 pieces
// (conv_b3.h: hi = rne(w), mid = rne(w - hi), lo = rne(w - hi - mid); every piece x activation product is exact in fp32)
// and one v_mfma_f32_16x16x32_bf16 per piece covers K = 32: 48 cycles where the fp32 form needs 256.
//
// K of one instruction = FOUR input channels x EIGHT consecutive taps (the fp32 kernel's channel block of four, all its taps;
// taps beyond the filter carry zero weights; TAPS = 11: two such groups).  A lane's eight K values are then eight
// CONSECUTIVE SAMPLES of one channel row -- the input tile stays row-major in LDS, as it lies in memory, no transposition:
// the B fragment is four 4-byte LDS reads.  Their start SN * column + SHIFT (+ 8 per tap group) must be even for that:
// unit-stride layers keep TWO images of the tile, the second shifted by one sample (lanes of odd start read it), the strided
// ones one image with the parity they need.  Row strides (in dwords) == 8 (two images) or 16 (one) mod 32 and the second
// image 16 banks off the first: the reads of a half-wave fall on disjoint banks for SN = 1 and 2.
// Same ConvCfg, tiles, epilogues and fp32 fragments as conv_mfma_kernel's BF16 / EPI_STORE path: the A operand is cut
// from those fragments on the device every step (conv_b16_pack_kernel).  Results agree with the fp32-MFMA form to fp32
// rounding (exact products, another summation order).
#pragma once
#include "conv_b3.h"
#include "conv_mfma.h"

namespace vp {

template <class C>
struct ConvB16 {
  // TPR = taps a lane takes from one channel row: 8 (filters of 3 .. 16 taps: K = four channels x eight taps) or 2 (the
  // two-tap phases of the stride-4 transposed layers: K = sixteen channels x two taps, a lane reads one dword from each of
  // four rows)
  static constexpr int TPR = C::TAPS <= 2 ? 2 : 8;
  static constexpr int TG = TPR == 8 ? (C::TAPS + 7) / 8 : 1;   // tap groups
  static constexpr int STEPS = TPR == 8 ? C::CB * TG : C::CB / 4;  // K-steps: (channel block, tap group) / four channel blocks
  static constexpr bool TWO = (C::SN % 2) == 1;   // both parities occur among the lanes' starts
  static constexpr int SIG = C::SHIFT & 1;        // single image: shifted by one sample or not
  static constexpr int WNEED = C::SN * (C::TN - 1) + TPR * TG + C::SHIFT;  // samples a row is read up to
  static constexpr int WQ = (WNEED + 1 + 3) / 4;                            // staged quads per row (W4 of them from memory, zeros behind)
  static constexpr int RS0 = 2 * WQ;                                        // dwords per row
  static constexpr int RSD = (RS0 + 31) / 32 * 32 + (TPR == 2 ? 2 : TWO ? 8 : 16);  // (TPR 2: lane groups sit four rows apart)
  static_assert(TPR == 8 || C::CB % 4 == 0, "two-tap layers: whole groups of sixteen channels");
  static constexpr int IMG = C::CINP * RSD + 16;  // second image: 16 banks off the first
  static constexpr int IN_DW = (TWO ? IMG : 0) + C::CINP * RSD + 4;
  static constexpr int OUT_DW = C::DIRECT ? 0 : C::LDS_OUT;
  static constexpr int LDS_DW = IN_DW > OUT_DW ? IN_DW : OUT_DW;
  static constexpr int STAT_DW = 2 * C::COUT * C::WAVES_N;  // BatchNorm sums of the tile: [COUT][WAVES_N][2], behind the images
  static constexpr size_t A_UINT4 = (size_t)C::MT * STEPS * 3 * 64;  // operand size [m-tile][step][piece][lane]
  static_assert(C::BF16 && C::EPI == EPI_STORE && !C::APRE && !C::AQ4, "the training step's plain bf16 layers");
  static_assert((LDS_DW + STAT_DW) * 4 <= 64 * 1024, "LDS budget");
};

// A operand of conv_b16_kernel from the fp32 fragments [m-tile][CB][TAPS][64] (fragment lane = channel-of-block * 16 + row):
// TPR 8: lane (row lane % 16, channel lane / 16 of block cb) takes taps 8 tg .. 8 tg + 7 of its row and channel; TPR 2: lane
// (row, group lane / 16) takes the four channels x two taps of channel block 4 step + group.  Cut into the three pieces.
struct ConvB16PackJob {
  const float* frag;
  uint4* out;
  int MT, CB, TAPS, TG, TPR, first_block;  // blocks of 256 threads = four (m-tile, K-step) pairs
};
constexpr int MAX_B16_JOBS = 40;
struct ConvB16PackJobs {
  ConvB16PackJob job[MAX_B16_JOBS];
  int count;
};
__global__ __launch_bounds__(256) void conv_b16_pack_kernel(const ConvB16PackJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;
  const ConvB16PackJob jb = jobs.job[j];
  const int trip = ((int)blockIdx.x - jb.first_block) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int steps = jb.TPR == 8 ? jb.CB * jb.TG : jb.CB / 4;
  if (trip >= jb.MT * steps) return;
  const int mt = trip / steps, st = trip - mt * steps;
  float w[8];
  if (jb.TPR == 8) {
    const int cb = st / jb.TG, tg = st - cb * jb.TG;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tap = 8 * tg + i;
      w[i] = tap < jb.TAPS ? jb.frag[(((long)mt * jb.CB + cb) * jb.TAPS + tap) * 64 + lane] : 0.f;
    }
  } else {  // lane (row lane % 16, group lane / 16): channel block 4 st + group, its four channels x two taps
    const int cb = 4 * st + (lane >> 4), m = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = i >> 1, tap = i & 1;
      w[i] = tap < jb.TAPS ? jb.frag[(((long)mt * jb.CB + cb) * jb.TAPS + tap) * 64 + kk * 16 + m] : 0.f;
    }
  }
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    h[p] = pack_bf16x2(w[2 * p], w[2 * p + 1]);
    const float ra = w[2 * p] - bf16_lo(h[p]), rb = w[2 * p + 1] - bf16_hi(h[p]);
    m[p] = pack_bf16x2(ra, rb);
    l[p] = pack_bf16x2(ra - bf16_lo(m[p]), rb - bf16_hi(m[p]));
  }
  uint4* o = jb.out + ((long)mt * steps + st) * 3 * 64 + lane;
  o[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o[64] = make_uint4(m[0], m[1], m[2], m[3]);
  o[128] = make_uint4(l[0], l[1], l[2], l[3]);
}

// `stat` (may be null): the layer's BatchNorm statistics leave with the tile -- per workgroup and output channel the sum and
// the sum of squares of the values AS STORED (rounded to bf16, inside [0, l_out)), [COUT][workgroups][2] floats, workgroup =
// blockIdx.y * gridDim.x + blockIdx.x; bnv_apply_kernel folds them in a fixed order (BnArgs::fpart).  Saves the statistics
// launch and its read of z.
template <class C>
__global__ __launch_bounds__(256) void conv_b16_kernel(const ConvArgs a, const uint4* __restrict__ a3, float* __restrict__ stat) {
  using W = ConvB16<C>;
  extern __shared__ float4 lds_raw[];
  unsigned* ldu = reinterpret_cast<unsigned*>(lds_raw);
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int win = blockIdx.y, col0 = (int)blockIdx.x * C::TN;
  const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
  const int g = lane >> 4, n = lane & 15;
  // ---- stage the tile: CINP rows, W4 quads from memory (as conv_mfma_kernel), zeros behind ------------------------------
  {
    const int a0 = HALO + C::SN * col0 + C::IN_OFF_F4;  // multiple of 4 by construction
    const bf16_t* s1 = reinterpret_cast<const bf16_t*>(a.src1) + (long)win * a.ws1 + a0;
    const bf16_t* s2 = (C::CIN2 > 0) ? reinterpret_cast<const bf16_t*>(a.src2) + (long)win * a.ws2 + a0 : nullptr;
    constexpr int TOT = C::CINP * W::WQ, N_IT = (TOT + 255) / 256;
    constexpr bool SHIFTED = W::TWO || W::SIG == 1, PLAIN = W::TWO || W::SIG == 0;
    uint2 v[N_IT];
    unsigned e[N_IT];  // the sample behind the quad (shifted image)
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int idx = tid + k * 256;
      v[k] = make_uint2(0u, 0u);
      e[k] = 0u;
      if (idx < TOT) {
        const int c = idx / W::WQ, q = idx - c * W::WQ;
        const bf16_t* row = (c < C::CIN1) ? s1 + (long)c * a.ls1 : s2 + (long)(c - C::CIN1) * a.ls2;
        if (c < C::CIN && q < C::W4) {
          v[k] = *reinterpret_cast<const uint2*>(row + 4 * q);
          if (SHIFTED && q + 1 < C::W4) e[k] = row[4 * q + 4];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int idx = tid + k * 256;
      if (idx < TOT) {
        const int c = idx / W::WQ, q = idx - c * W::WQ;
        if (PLAIN) *reinterpret_cast<uint2*>(ldu + c * W::RSD + 2 * q) = v[k];
        if (SHIFTED) {
          const uint2 s = make_uint2(__builtin_amdgcn_alignbit(v[k].y, v[k].x, 16), (v[k].y >> 16) | (e[k] << 16));
          *reinterpret_cast<uint2*>(ldu + (W::TWO ? W::IMG : 0) + c * W::RSD + 2 * q) = s;
        }
      }
    }
  }
  __syncthreads();
  // ---- main loop ---------------------------------------------------------------------------------------------------------
  f32x4 acc[C::MW][C::NW];
#pragma unroll
  for (int i = 0; i < C::MW; ++i)
#pragma unroll
    for (int j = 0; j < C::NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane's B start per n-tile j: sample SN * (local column) + SHIFT; image by its parity
  int boff[C::NW];
#pragma unroll
  for (int j = 0; j < C::NW; ++j) {
    const int pos = C::SN * ((wn * C::NW + j) * 16 + n) + C::SHIFT;
    const int par = W::TWO ? (pos & 1) : W::SIG;
    boff[j] = (W::TWO && par ? W::IMG : 0) + (W::TPR == 8 ? g : 4 * g) * W::RSD + ((pos - par) >> 1);
  }
  const uint4* ap = a3 + (long)(wm * C::MW) * W::STEPS * 3 * 64 + lane;
  // K-steps of A in flight ahead of the MFMAs.  The layers of >= 14 K-steps (64+ channels: levels 3-4) are bound by the
  // length of a workgroup's chain of fragment round trips: six in flight instead of two, 1.394 -> 1.381 ms per step same-box.
  constexpr int PF = (W::STEPS >= 14 && C::MW == 1) ? 6 : 2;
  uint4 q[PF + 1][C::MW][3];
  auto load_a = [&](const int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::MW; ++i)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) q[s % (PF + 1)][i][pc] = ap[(((long)i * W::STEPS + s) * 3 + pc) * 64];
  };
#pragma unroll
  for (int s = 0; s < PF && s < W::STEPS; ++s) load_a(s);
#pragma unroll
  for (int s = 0; s < W::STEPS; ++s) {
    if (s + PF < W::STEPS) load_a(s + PF);
    uint4 b[C::NW];
#pragma unroll
    for (int j = 0; j < C::NW; ++j) {
      if constexpr (W::TPR == 8) {
        const int cb = s / W::TG, tg = s - cb * W::TG;
        const unsigned* p = ldu + boff[j] + cb * 4 * W::RSD + 4 * tg;
        b[j] = make_uint4(p[0], p[1], p[2], p[3]);
      } else {
        const unsigned* p = ldu + boff[j] + s * 16 * W::RSD;
        b[j] = make_uint4(p[0], p[W::RSD], p[2 * W::RSD], p[3 * W::RSD]);
      }
    }
#pragma unroll
    for (int pc = 2; pc >= 0; --pc)  // smallest products first
#pragma unroll
      for (int i = 0; i < C::MW; ++i)
#pragma unroll
        for (int j = 0; j < C::NW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, q[s % (PF + 1)][i][pc]),
                                                            __builtin_bit_cast(bf16x8_b3, b[j]), acc[i][j], 0, 0, 0);
  }
  const float* bias = a.bias;
  if (stat) {  // (uniform)
    constexpr int RPC = C::P >= 4 ? 4 : C::P, NCH = 4 / RPC;  // rows of a lane's four that belong to one channel
    static_assert(C::P == 1 || C::P == 2 || C::P == 4, "rows (channel, phase)");
    float* lstat = lds + W::LDS_DW;
    float ss[C::MW][NCH], sq[C::MW][NCH];
#pragma unroll
    for (int i = 0; i < C::MW; ++i)
#pragma unroll
      for (int c = 0; c < NCH; ++c) ss[i][c] = sq[i][c] = 0.f;
#pragma unroll
    for (int i = 0; i < C::MW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wm * C::MW + i) * 16 + 4 * g + r;
        const int co = m / C::P, p = m - co * C::P;
        const float bs = bias[co];
#pragma unroll
        for (int j = 0; j < C::NW; ++j) {
          const int t = C::P * (col0 + (wn * C::NW + j) * 16 + n) + p + C::OUT_OFF;
          float x = acc[i][j][r] + bs;
          if (C::RELU) x = fmaxf(x, 0.f);
          x = bf16_lo(pack_bf16x2(x, 0.f));
          if (t >= 0 && t < a.l_out) {
            ss[i][r / RPC] += x;
            sq[i][r / RPC] = fmaf(x, x, sq[i][r / RPC]);
          }
        }
      }
#pragma unroll
    for (int i = 0; i < C::MW; ++i)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {  // the 16 columns of the lane row
          ss[i][c] += __shfl_xor(ss[i][c], o, 64);
          sq[i][c] += __shfl_xor(sq[i][c], o, 64);
        }
        if (n == 0) {
          const int co = ((wm * C::MW + i) * 16 + 4 * g + c * RPC) / C::P;
          lstat[(co * C::WAVES_N + wn) * 2] = ss[i][c];
          lstat[(co * C::WAVES_N + wn) * 2 + 1] = sq[i][c];
        }
      }
    __syncthreads();
    if (tid < C::COUT) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < C::WAVES_N; ++k) s0 += lstat[(tid * C::WAVES_N + k) * 2], s1 += lstat[(tid * C::WAVES_N + k) * 2 + 1];
      const long nwg = (long)gridDim.x * gridDim.y, wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
      *reinterpret_cast<float2*>(stat + ((long)tid * nwg + wg) * 2) = make_float2(s0, s1);
    }
  }
  // ---- epilogues: as conv_mfma_kernel's BF16 / EPI_STORE path -------------------------------------------------------------
  bf16_t* const dbase = reinterpret_cast<bf16_t*>(a.dst) + (long)win * a.wsd + a.dst_halo;
  if constexpr (C::DIRECT) {
#pragma unroll
    for (int i = 0; i < C::MW; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; rr += 2) {
        const int co = ((wm * C::MW + i) * 16 + 4 * g + rr) / 2;
        const float bs = bias[co];
        bf16_t* rowb = dbase + (long)co * a.lsd;
#pragma unroll
        for (int j = 0; j < C::NW; ++j) {
          const int t = 2 * (col0 + (wn * C::NW + j) * 16 + n) + C::OUT_OFF;
          float v0 = acc[i][j][rr] + bs, v1 = acc[i][j][rr + 1] + bs;
          if (C::RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
          if (t >= 0 && t + 1 < a.l_out) {
            *reinterpret_cast<unsigned*>(rowb + t) = pack_bf16x2(v0, v1);
          } else if (t >= 0 && t < a.l_out) {
            rowb[t] = to_bf16(v0);
          }
        }
      }
    }
    return;
  } else {
    __syncthreads();  // all B reads done; the LDS image is reused as the output staging tile
    if constexpr (C::P == 2) {
#pragma unroll
      for (int i = 0; i < C::MW; ++i) {
#pragma unroll
        for (int rr = 0; rr < 4; rr += 2) {
          const int co = ((wm * C::MW + i) * 16 + 4 * g + rr) / 2;
          const float bs = bias[co];
#pragma unroll
          for (int j = 0; j < C::NW; ++j) {
            const int col = (wn * C::NW + j) * 16 + n;
            float v0 = acc[i][j][rr] + bs, v1 = acc[i][j][rr + 1] + bs;
            if (C::RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
            *reinterpret_cast<float2*>(lds + co * C::OS + 2 * col) = make_float2(v0, v1);
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::MW; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (wm * C::MW + i) * 16 + 4 * g + r;
          const int co = m / C::P, p = m - co * C::P;
          const float bs = bias[co];
#pragma unroll
          for (int j = 0; j < C::NW; ++j) {
            float x = acc[i][j][r] + bs;
            if (C::RELU) x = fmaxf(x, 0.f);
            lds[co * C::OS + C::P * ((wn * C::NW + j) * 16 + n) + p] = x;
          }
        }
      }
    }
    __syncthreads();
    const int t0 = C::P * col0 + C::OUT_OFF;  // global output index of staged column 0
    if constexpr (C::OUT_OFF % 4 == 0) {
      for (int idx = tid; idx < C::COUT * (C::OW / 4); idx += 256) {
        const int co = idx / (C::OW / 4), qq = idx - co * (C::OW / 4);
        const int t = t0 + 4 * qq;
        if (t < a.l_out) {
          float4 x = *reinterpret_cast<const float4*>(lds + co * C::OS + 4 * qq);
          if (t + 1 >= a.l_out) x.y = 0.f;  // keep the right margin zero
          if (t + 2 >= a.l_out) x.z = 0.f;
          if (t + 3 >= a.l_out) x.w = 0.f;
          *reinterpret_cast<uint2*>(dbase + (long)co * a.lsd + t) = make_uint2(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w));
        }
      }
    } else {
      constexpr int PAR = ((C::OUT_OFF % 2) + 2) % 2;  // parity of t0 (P * col0 is even)
      static_assert(C::OUT_OFF % 4 == 0 || (C::P % 2 == 0 && C::OW % 2 == 0), "pair stores");
      constexpr int NP = C::OW / 2 + PAR;
      for (int idx = tid; idx < C::COUT * NP; idx += 256) {
        const int co = idx / NP, qq = 2 * (idx - co * NP) - PAR;
        const int t = t0 + qq;  // even
        const bool ok0 = qq >= 0 && t >= 0 && t < a.l_out, ok1 = qq + 1 < C::OW && t + 1 >= 0 && t + 1 < a.l_out;
        bf16_t* row = dbase + (long)co * a.lsd;
        if (ok0 && ok1) {
          *reinterpret_cast<unsigned*>(row + t) = pack_bf16x2(lds[co * C::OS + qq], lds[co * C::OS + qq + 1]);
        } else if (ok0) {
          row[t] = to_bf16(lds[co * C::OS + qq]);
        } else if (ok1) {
          row[t + 1] = to_bf16(lds[co * C::OS + qq + 1]);
        }
      }
    }
  }
}

template <class C>
int launch_conv_b16(const ConvArgs& a, const uint4* a3, float* stat, int cols, hipStream_t stream) {
  dim3 grid((cols + C::TN - 1) / C::TN, a.n_windows, 1);
  hipLaunchKernelGGL(conv_b16_kernel<C>, grid, dim3(256), (size_t)(ConvB16<C>::LDS_DW + ConvB16<C>::STAT_DW) * 4, stream, a, a3, stat);
  return 0;
}

}  // namespace vp
