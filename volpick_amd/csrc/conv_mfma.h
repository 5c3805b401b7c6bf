// 1-D convolution family as an implicit GEMM on the gfx950 fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, 64 FLOP/clk/SIMD = the fp32 peak).
//
// One kernel template covers every conv-shaped layer of PhaseNet and EQTransformer
//  through a "polyphase" formulation:
//
//   out[co][P*n + p + OUT_OFF] = act( bias[co] +
//        sum_{tap<TAPS} sum_{ci} A[(co,p)][(tap,ci)] * in[ci][SN*n + tap + IN_OFF] )
//
//   plain conv, stride s, left pad padL : P=1, SN=s, TAPS=K, IN_OFF=-padL
//   conv with few output channels       : P=2 output phases per GEMM row block so that
//                                         M = COUT*P fills the 16-row MFMA tile
//                                         (TAPS = K + s*(P-1), SN = s*P)
//   ConvTranspose1d(k=7, s=4) + crop    : P=4, TAPS=2, SN=1, IN_OFF=-1, OUT_OFF=-(crop)
//
// GEMM view per workgroup: M = COUT*P rows, N = TN output columns (time) of one window,
// K = TAPS * CIN.  A (the folded weights) is pre-packed on the host in MFMA lane order
// and streamed from L2; B is read straight out of an LDS image of the input tile — the
// im2col matrix is never materialised: for K-step (tap, 4 channels) lane l reads
// lds[(cb*4 + (l>>4))*S + (col(l&15))*SN + tap], i.e. a per-lane base + immediate offset.
// The input image is fetched with unguarded, 16-byte coalesced loads because every
// activation row carries zeroed halos (vp_common.h).
#pragma once
#include "bf16.h"
#include "vp_common.h"

namespace vp {

enum ConvEpi {
  EPI_STORE = 0,     // dst[co][t] = v
  EPI_POOL2 = 1,     // dst[co][j] = max(v[2j], v[2j+1])         (EQT encoder: ReLU + MaxPool1d(2))
  EPI_UP2 = 2,       // dst[co][2t] = dst[co][2t+1] = v[t]       (EQT decoder: producer-side Upsample(2))
  EPI_RES = 3,       // o = v + res[co][t]; dst = o; dst2 = relu(s2[co]*o + b2[co])   (EQT ResCNN)
  EPI_SOFTMAX3 = 4,  // y[c][t] = softmax_c( w[c][:] . v[:][t] + b[c] )             (PhaseNet out)
  EPI_POOL2_DUAL = 5,// POOL2, plus dst2 = relu(s2*pooled + b2)   (EQT encoder tail feeding ResCNN)
  EPI_HEAD = 6       // y[t] = sigmoid(b + sum_ci sum_k w[ci][k] v[ci][t+k-5])  (EQT decoder tail + Conv1d(8,1,11) head):
                     // tiles overlap by 8 columns so the 5-sample halo of the head is recomputed, not re-read
};

struct ConvArgs {
  const float* src1;  // [win][CIN1][ls1]
  const float* src2;  // [win][CIN2][ls2] (concat partner) or null
  int ls1, ls2;
  long ws1, ws2;      // window strides in floats
  float* dst;         // [win][COUT or 3][lsd]
  int lsd;
  long wsd;
  int dst_halo;       // HALO for internal tensors, 0 for the dense final output
  float* dst2;
  int lsd2;
  long wsd2;
  const float* afrag; // [set][MT][CB][TAPS][64]
  const float* bias;  // [set][COUT]
  long afrag_set_stride, bias_set_stride;
  int win_per_set;    // windows sharing one weight set (three EQT decoders run as one launch)
  int n_windows;
  int l_out;          // valid length of the (pre-pool / pre-upsample) conv output
  int l_dst;          // valid length of dst rows (after pool / upsample)
  const float* e0;    // EPI_RES: residual tensor (same geometry as dst); EPI_SOFTMAX3: w[3][8]
  const float* e1;    // EPI_RES/DUAL: s2[COUT];                          EPI_SOFTMAX3: b[3]
  const float* e2;    // EPI_RES/DUAL: b2[COUT]
  long e_set_stride;  // per-set stride of e1/e2
  int ls_res;         // EPI_RES: geometry of the residual tensor
  long ws_res;
  unsigned long long* clk;  // debug: 8 shader-clock stamps of workgroup (tile 1, window 7): start, loaded, mfma done, staged, stored
};

struct ConvGeom {  // runtime mirror of the template parameters (planning / packing / tests)
  int cin1, cin2, cout, P, taps, sn, in_off, out_off, waves_m, waves_n, nw, relu, epi;
  int cin() const { return cin1 + cin2; }
  int cinp() const { return round_up(cin1 + cin2, 4); }
  int M() const { return cout * P; }
  int tn() const { return waves_n * nw * 16; }
  int shift() const { return in_off - floor4(in_off); }
  int w_in() const { return sn * (tn() - 1) + taps + shift(); }
  int w4() const { return (w_in() + 3) / 4; }
  // max physical index (+1) of a source row read when `cols` output columns are produced
  int src_need(int cols) const {
    int tiles = (cols + tn() - 1) / tn();
    return HALO + sn * (tiles - 1) * tn() + floor4(in_off) + 4 * w4();
  }
  size_t afrag_floats() const { return (size_t)M() * cinp() * taps; }
};

template <int CIN1_, int CIN2_, int COUT_, int P_, int TAPS_, int SN_, int IN_OFF_, int OUT_OFF_, int WAVES_M_,
          int WAVES_N_, int NW_, int RELU_, int EPI_, int APRE_ = 0, int AQ4_ = 0, int BF16_ = 0>
struct ConvCfg {
  // BF16: source and destination rows rest in memory as bfloat16 (training step, bf16.h): the loader widens 8-byte
  // groups of four samples into the fp32 LDS image, the EPI_STORE epilogues round on the way out; strides (ls, ws) count
  // elements.  The MFMA loop, the fragments and the bias are fp32 either way.
  static constexpr bool BF16 = BF16_ != 0;
  static constexpr bool APRE = APRE_ != 0;  // A fragments of channel block 0 requested ahead of the input tile
  // AQ4: the A operand comes regrouped [m-tile][K-step / 4][lane][4] (regroup_afrag4) as one 16-byte load per four
  // K-steps -- the weight-heavy layers (64 -> 64 channels: 98 KB of fragments per workgroup) issue 4x fewer memory
  // instructions for it (tools/micro/micro_stream.hip: 19 vs 52-56 B/clk/CU out of L2).
  static constexpr bool AQ4 = AQ4_ != 0;
  static constexpr int CIN1 = CIN1_, CIN2 = CIN2_, CIN = CIN1_ + CIN2_, CINP = (CIN + 3) / 4 * 4, CB = CINP / 4;
  static constexpr int COUT = COUT_, P = P_, TAPS = TAPS_, SN = SN_, IN_OFF = IN_OFF_, OUT_OFF = OUT_OFF_;
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, NW = NW_, RELU = RELU_, EPI = EPI_;
  static constexpr int M = COUT * P, MT = M / 16, MW = MT / WAVES_M;
  static constexpr int NT = WAVES_N * NW, TN = NT * 16;
  static constexpr int IN_OFF_F4 = (IN_OFF >= 0) ? (IN_OFF / 4) * 4 : -(((-IN_OFF) + 3) / 4) * 4;
  static constexpr int SHIFT = IN_OFF - IN_OFF_F4;
  static constexpr int W_IN = SN * (TN - 1) + TAPS + SHIFT;
  static constexpr int W4 = (W_IN + 3) / 4;
  // LDS row stride: >= 4*W4, == 16 (mod 32) so the two 16-lane channel rows a ds_read_b32
  // half-wave touches fall on disjoint banks (unit-stride case).
  static constexpr int S = ((4 * W4 + 15) / 32) * 32 + 16;
  static constexpr int OW = P * TN;  // staged output columns per tile
  static constexpr int OS = OW + 4;
  static constexpr int LDS_IN = CINP * S, LDS_OUT = COUT * OS;
  // Plain two-phase layers store straight from the accumulators: registers (0,1) / (2,3) of a lane are two consecutive
  // samples of one channel, 16 lanes x 8 bytes = one full 128-byte line per channel row and store.  No staging tile
  // (it was the larger half of the LDS footprint of the EQT decoders: 50 -> 30 KB, 3 -> 5 workgroups per CU), no
  // staging / store phases and one barrier less per tile.
  static constexpr bool DIRECT = (EPI == EPI_STORE) && (P == 2) && (OUT_OFF % 2 == 0);
  static constexpr int LDS_FLOATS = DIRECT ? LDS_IN : (LDS_IN > LDS_OUT ? LDS_IN : LDS_OUT);
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroups");
  static_assert(M % (16 * WAVES_M) == 0, "M must tile into 16-row MFMA tiles per wave");
  static_assert(S >= 4 * W4 && S % 32 == 16, "LDS stride");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static_assert(!BF16 || EPI == EPI_STORE, "bf16 rows: plain stores only (training step)");
  static ConvGeom geom() {
    return ConvGeom{CIN1, CIN2, COUT, P, TAPS, SN, IN_OFF, OUT_OFF, WAVES_M, WAVES_N, NW, RELU, EPI};
  }
};

template <class C>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int win = blockIdx.y;
  // EPI_HEAD tiles advance by TN-8 columns and start 4 columns early (halo of the fused head conv)
  const int col0 = (C::EPI == EPI_HEAD) ? (int)blockIdx.x * (C::TN - 8) - 4 : (int)blockIdx.x * C::TN;
  const int set = win / a.win_per_set;
  int stamp_i = 0;
#define CONV_STAMP()                                                                                      \
  if (a.clk && tid == 0 && blockIdx.x == 1 && win == 7) a.clk[stamp_i] = __builtin_readcyclecounter(); \
  ++stamp_i;
  CONV_STAMP()

  // The A fragments of the first channel block are requested BEFORE the input tile (template flag of the layer): their
  // L2 round trip then runs under the tile's; otherwise it starts behind the barrier, fully exposed, and for the
  // layers with one or two channel blocks (encoder.0 / .1) it was most of the "MFMA" phase (tools/conv_clock.py).
  const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
  const int g = lane >> 4, n = lane & 15;
  const float* ap = a.afrag + (long)set * a.afrag_set_stride + (long)(wm * C::MW) * C::CB * C::TAPS * 64 + lane;
  float a0[C::TAPS][C::MW];
  if constexpr (C::APRE) {
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap)
#pragma unroll
      for (int i = 0; i < C::MW; ++i) a0[tap][i] = ap[((long)i * C::CB) * C::TAPS * 64 + tap * 64];
  }
  // ---- stage the input tile: CINP rows x 4*W4 floats, aligned 16-byte loads --------------
  {
    const int a0 = HALO + C::SN * col0 + C::IN_OFF_F4;  // multiple of 4 by construction
    // All of a thread's loads are issued before its first LDS write (chunks of 8 x 16 B in flight):
    // the rolled form serialised load -> wait -> ds_write and cost 5-17k cycles of pure latency per tile.
    // (a tile of 9 or 13 quads per thread -- decoder.6, decoder.2 -- takes them in ONE chunk: a second chunk of one or
    // five loads behind the first chunk's LDS writes was a second, fully exposed memory round trip: 13.7 k cycles of
    // load phase against 6 k for the layers with <= 8 quads, tools/conv_clock.py)
    constexpr int TOT = C::CINP * C::W4, N_IT = (TOT + 255) / 256, CH = (N_IT <= 13) ? N_IT : 8;
    if constexpr (C::BF16) {
      const bf16_t* s1 = reinterpret_cast<const bf16_t*>(a.src1) + (long)win * a.ws1 + a0;
      const bf16_t* s2 = (C::CIN2 > 0) ? reinterpret_cast<const bf16_t*>(a.src2) + (long)win * a.ws2 + a0 : nullptr;
#pragma unroll
      for (int it0 = 0; it0 < N_IT; it0 += CH) {
        uint2 v[CH];  // four samples each
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int idx = tid + (it0 + k) * 256;
          v[k] = make_uint2(0u, 0u);
          if (it0 + k < N_IT && idx < TOT) {
            const int c = idx / C::W4, q = idx - c * C::W4;
            if (c < C::CIN1) {
              v[k] = *reinterpret_cast<const uint2*>(s1 + (long)c * a.ls1 + 4 * q);
            } else if (c < C::CIN) {
              v[k] = *reinterpret_cast<const uint2*>(s2 + (long)(c - C::CIN1) * a.ls2 + 4 * q);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int idx = tid + (it0 + k) * 256;
          if (it0 + k < N_IT && idx < TOT) {
            const int c = idx / C::W4, q = idx - c * C::W4;
            *reinterpret_cast<float4*>(lds + c * C::S + 4 * q) =
                make_float4(bf16_lo(v[k].x), bf16_hi(v[k].x), bf16_lo(v[k].y), bf16_hi(v[k].y));
          }
        }
      }
    } else {
    const float* s1 = a.src1 + (long)win * a.ws1 + a0;
    const float* s2 = (C::CIN2 > 0) ? a.src2 + (long)win * a.ws2 + a0 : nullptr;
#pragma unroll
    for (int it0 = 0; it0 < N_IT; it0 += CH) {
      float4 v[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int idx = tid + (it0 + k) * 256;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it0 + k < N_IT && idx < TOT) {
          const int c = idx / C::W4, q = idx - c * C::W4;
          if (c < C::CIN1) {
            v[k] = *reinterpret_cast<const float4*>(s1 + (long)c * a.ls1 + 4 * q);
          } else if (c < C::CIN) {
            v[k] = *reinterpret_cast<const float4*>(s2 + (long)(c - C::CIN1) * a.ls2 + 4 * q);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int idx = tid + (it0 + k) * 256;
        if (it0 + k < N_IT && idx < TOT) {
          const int c = idx / C::W4, q = idx - c * C::W4;
          *reinterpret_cast<float4*>(lds + c * C::S + 4 * q) = v[k];
        }
      }
    }
    }
  }
  __syncthreads();
  CONV_STAMP()

  // ---- MFMA main loop ---------------------------------------------------------------
  f32x4 acc[C::MW][C::NW];
#pragma unroll
  for (int i = 0; i < C::MW; ++i)
#pragma unroll
    for (int j = 0; j < C::NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* bp = lds + g * C::S + (wn * C::NW * 16 + n) * C::SN + C::SHIFT;
  if constexpr (C::APRE) {  // channel block 0 out of the registers fetched above
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      float bv[C::NW];
#pragma unroll
      for (int j = 0; j < C::NW; ++j) bv[j] = bp[j * 16 * C::SN + tap];
#pragma unroll
      for (int i = 0; i < C::MW; ++i)
#pragma unroll
        for (int j = 0; j < C::NW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[tap][i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  if constexpr (C::AQ4) {
    static_assert(!C::AQ4 || (C::CB % 4 == 0 && !C::APRE), "AQ4 walks four channel blocks per trip");
    constexpr int SB = C::CB * C::TAPS / 4;  // 16-byte groups per m-tile
    const float4* ap4 = reinterpret_cast<const float4*>(a.afrag + (long)set * a.afrag_set_stride) +
                        (long)(wm * C::MW) * SB * 64 + lane;
    for (int cb0 = 0; cb0 < C::CB; cb0 += 4) {
      float4 a4[C::MW];
#pragma unroll
      for (int s = 0; s < 4 * C::TAPS; ++s) {  // the K-steps of these four channel blocks, same order as below
        const int cbl = s / C::TAPS, tap = s % C::TAPS;
        if ((s & 3) == 0) {
#pragma unroll
          for (int i = 0; i < C::MW; ++i) a4[i] = ap4[((long)i * SB + (cb0 * C::TAPS + s) / 4) * 64];
        }
        float bv[C::NW];
#pragma unroll
        for (int j = 0; j < C::NW; ++j) bv[j] = bp[(cb0 + cbl) * 4 * C::S + j * 16 * C::SN + tap];
#pragma unroll
        for (int i = 0; i < C::MW; ++i) {
          const float av = (s & 3) == 0 ? a4[i].x : (s & 3) == 1 ? a4[i].y : (s & 3) == 2 ? a4[i].z : a4[i].w;
#pragma unroll
          for (int j = 0; j < C::NW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  } else
  for (int cb = C::APRE ? 1 : 0; cb < C::CB; ++cb) {
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      float av[C::MW], bv[C::NW];
#pragma unroll
      for (int i = 0; i < C::MW; ++i) av[i] = ap[((long)i * C::CB + cb) * C::TAPS * 64 + tap * 64];
#pragma unroll
      for (int j = 0; j < C::NW; ++j) bv[j] = bp[cb * 4 * C::S + j * 16 * C::SN + tap];
#pragma unroll
      for (int i = 0; i < C::MW; ++i)
#pragma unroll
        for (int j = 0; j < C::NW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  if constexpr (C::DIRECT) {
    const float* bias = a.bias + (long)set * a.bias_set_stride;
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
#pragma unroll
    for (int i = 0; i < C::MW; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; rr += 2) {
        const int co = ((wm * C::MW + i) * 16 + 4 * g + rr) / 2;
        const float b = bias[co];
        float* row = d + (long)co * a.lsd;
#pragma unroll
        for (int j = 0; j < C::NW; ++j) {
          const int t = 2 * (col0 + (wn * C::NW + j) * 16 + n) + C::OUT_OFF;
          float v0 = acc[i][j][rr] + b, v1 = acc[i][j][rr + 1] + b;
          if (C::RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
          if constexpr (C::BF16) {
            bf16_t* rowb = reinterpret_cast<bf16_t*>(a.dst) + (long)win * a.wsd + a.dst_halo + (long)co * a.lsd;
            if (t >= 0 && t + 1 < a.l_out) {
              *reinterpret_cast<unsigned*>(rowb + t) = pack_bf16x2(v0, v1);
            } else if (t >= 0 && t < a.l_out) {
              rowb[t] = to_bf16(v0);
            }
          } else if (t >= 0 && t + 1 < a.l_out) {
            *reinterpret_cast<float2*>(row + t) = make_float2(v0, v1);
          } else if (t >= 0 && t < a.l_out) {
            row[t] = v0;  // odd length: the right margin stays zero
          }
        }
      }
    }
    CONV_STAMP()
    CONV_STAMP()
    CONV_STAMP()
    return;
  }
  __syncthreads();  // all B reads done; the LDS image is reused as the output staging tile
  CONV_STAMP()

  // ---- epilogue 1: bias (+ReLU), D fragments -> LDS [COUT][OS] ----------------------
  {
    const float* bias = a.bias + (long)set * a.bias_set_stride;
    if constexpr (C::P == 2) {
      // two-phase layers: registers (0,1) and (2,3) of a lane are the two phases of one channel, i.e. two CONSECUTIVE
      // staged samples — one 8-byte store each instead of four scalar stores two floats apart across the lanes
      // (2-way bank conflicts; the staging of decoder.6 took half as long as its MFMAs)
      static_assert(C::OS % 2 == 0, "8-byte aligned staging rows");
#pragma unroll
      for (int i = 0; i < C::MW; ++i) {
#pragma unroll
        for (int rr = 0; rr < 4; rr += 2) {
          const int co = ((wm * C::MW + i) * 16 + 4 * g + rr) / 2;
          const float b = bias[co];
#pragma unroll
          for (int j = 0; j < C::NW; ++j) {
            const int col = (wn * C::NW + j) * 16 + n;
            float v0 = acc[i][j][rr] + b, v1 = acc[i][j][rr + 1] + b;
            if (C::RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
            if constexpr (C::EPI == EPI_HEAD) {  // outside the signal the head must see zero padding
              const int tg = 2 * (col0 + col);
              if (tg < 0 || tg >= a.l_out) v0 = 0.f;
              if (tg + 1 < 0 || tg + 1 >= a.l_out) v1 = 0.f;
            }
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
          const float b = bias[co];
#pragma unroll
          for (int j = 0; j < C::NW; ++j) {
            float v = acc[i][j][r] + b;
            if (C::RELU) v = fmaxf(v, 0.f);
            lds[co * C::OS + C::P * ((wn * C::NW + j) * 16 + n) + p] = v;
          }
        }
      }
    }
  }
  __syncthreads();
  CONV_STAMP()

  // ---- epilogue 2: coalesced stores of the staged tile -----------------------------
  const int t0 = C::P * col0 + C::OUT_OFF;  // global output index of staged column 0
  if constexpr (C::EPI == EPI_STORE && C::BF16) {
    bf16_t* d = reinterpret_cast<bf16_t*>(a.dst) + (long)win * a.wsd + a.dst_halo;
    if constexpr (C::OUT_OFF % 4 == 0) {
      for (int idx = tid; idx < C::COUT * (C::OW / 4); idx += 256) {
        const int co = idx / (C::OW / 4), q = idx - co * (C::OW / 4);
        const int t = t0 + 4 * q;
        if (t < a.l_out) {
          float4 v = *reinterpret_cast<const float4*>(lds + co * C::OS + 4 * q);
          if (t + 1 >= a.l_out) v.y = 0.f;  // keep the right margin zero
          if (t + 2 >= a.l_out) v.z = 0.f;
          if (t + 3 >= a.l_out) v.w = 0.f;
          *reinterpret_cast<uint2*>(d + (long)co * a.lsd + t) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
      }
    } else {
      // the tile starts OUT_OFF samples off the four-sample grid: even-aligned pairs (4-byte stores); a pair that
      // straddles the tile's edge or the row's ends is written sample by sample (the other half belongs to the
      // neighbouring tile, or stays zero)
      constexpr int PAR = ((C::OUT_OFF % 2) + 2) % 2;  // parity of t0 (P * col0 is even)
      static_assert(C::P % 2 == 0 && C::OW % 2 == 0, "pair stores");
      constexpr int NP = C::OW / 2 + PAR;
      for (int idx = tid; idx < C::COUT * NP; idx += 256) {
        const int co = idx / NP, q = 2 * (idx - co * NP) - PAR;
        const int t = t0 + q;  // even
        const bool ok0 = q >= 0 && t >= 0 && t < a.l_out, ok1 = q + 1 < C::OW && t + 1 >= 0 && t + 1 < a.l_out;
        bf16_t* row = d + (long)co * a.lsd;
        if (ok0 && ok1) {
          *reinterpret_cast<unsigned*>(row + t) = pack_bf16x2(lds[co * C::OS + q], lds[co * C::OS + q + 1]);
        } else if (ok0) {
          row[t] = to_bf16(lds[co * C::OS + q]);
        } else if (ok1) {
          row[t + 1] = to_bf16(lds[co * C::OS + q + 1]);
        }
      }
    }
  } else if constexpr (C::EPI == EPI_STORE) {
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
    if constexpr (C::OUT_OFF % 4 == 0) {
      for (int idx = tid; idx < C::COUT * (C::OW / 4); idx += 256) {
        const int co = idx / (C::OW / 4), q = idx - co * (C::OW / 4);
        const int t = t0 + 4 * q;
        if (t < a.l_out) {
          float4 v = *reinterpret_cast<const float4*>(lds + co * C::OS + 4 * q);
          if (t + 1 >= a.l_out) v.y = 0.f;  // keep the right margin zero
          if (t + 2 >= a.l_out) v.z = 0.f;
          if (t + 3 >= a.l_out) v.w = 0.f;
          *reinterpret_cast<float4*>(d + (long)co * a.lsd + t) = v;
        }
      }
    } else {
      for (int idx = tid; idx < C::COUT * C::OW; idx += 256) {
        const int co = idx / C::OW, q = idx - co * C::OW;
        const int t = t0 + q;
        if (t >= 0 && t < a.l_out) d[(long)co * a.lsd + t] = lds[co * C::OS + q];
      }
    }
  } else if constexpr (C::EPI == EPI_POOL2 || C::EPI == EPI_POOL2_DUAL) {
    // MaxPool1d(2) over the ReLU output; an odd tail is pooled with the -1e10 pad, i.e. alone.
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
    float* d2 = (C::EPI == EPI_POOL2_DUAL) ? a.dst2 + (long)win * a.wsd2 + HALO : nullptr;
#pragma unroll 4
    for (int idx = tid; idx < C::COUT * (C::OW / 2); idx += 256) {  // four LDS reads in flight per trip
      const int co = idx / (C::OW / 2), q = idx - co * (C::OW / 2);
      const int t = t0 + 2 * q;
      if (t < a.l_out) {
        const float2 pr = *reinterpret_cast<const float2*>(lds + co * C::OS + 2 * q);  // one 8-byte read: scalar reads two
        float v = pr.x;                                                                 // floats apart collide 2-way
        if (t + 1 < a.l_out) v = fmaxf(v, pr.y);
        d[(long)co * a.lsd + (t >> 1)] = v;
        if constexpr (C::EPI == EPI_POOL2_DUAL) {
          const float s = a.e1[co], b = a.e2[co];
          d2[(long)co * a.lsd2 + (t >> 1)] = fmaxf(fmaf(s, v, b), 0.f);
        }
      }
    }
  } else if constexpr (C::EPI == EPI_UP2) {
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
    for (int idx = tid; idx < C::COUT * C::OW; idx += 256) {
      const int co = idx / C::OW, q = idx - co * C::OW;
      const int t = t0 + q;
      if (t < a.l_out) {
        const float v = lds[co * C::OS + q];
        float* row = d + (long)co * a.lsd;
        if (2 * t + 1 < a.l_dst) {
          *reinterpret_cast<float2*>(row + 2 * t) = make_float2(v, v);  // one coalesced 8-byte store per pair
        } else {
          row[2 * t] = v;
        }
      }
    }
  } else if constexpr (C::EPI == EPI_RES) {
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
    const float* res = a.e0 + (long)win * a.ws_res + HALO;
    float* d2 = a.dst2 ? a.dst2 + (long)win * a.wsd2 + HALO : nullptr;
    const float* s2 = a.e1 + (long)set * a.e_set_stride;
    const float* b2 = a.e2 + (long)set * a.e_set_stride;
    for (int idx = tid; idx < C::COUT * C::OW; idx += 256) {
      const int co = idx / C::OW, q = idx - co * C::OW;
      const int t = t0 + q;
      if (t < a.l_out) {
        const float o = lds[co * C::OS + q] + res[(long)co * a.ls_res + t];
        d[(long)co * a.lsd + t] = o;
        if (d2) d2[(long)co * a.lsd2 + t] = fmaxf(fmaf(s2[co], o, b2[co]), 0.f);
      }
    }
  } else if constexpr (C::EPI == EPI_HEAD) {
    static_assert(C::EPI != EPI_HEAD || (C::COUT == 8 && C::SN == 1 && C::P == 2), "EQT decoder tail");
    const int b = win - set * a.win_per_set;
    float* y = a.dst + ((long)b * 3 + set) * a.l_out;  // dense (B, 3, T): row = decoder index
    const float* w = a.e0 + set * 88;
    const float bias_h = a.e1[set];
    // Thread i owns the 4 staged columns 4i+5 .. 4i+8 and reads the 16 staged values 4i .. 4i+15 of each
    // channel with four aligned ds_read_b128 (8x fewer LDS instructions than one read per tap).  The tile
    // owns staged columns [8, 8 + 2*(TN-8)), i.e. outputs [2*tile*(TN-8), +2*(TN-8)).
    const int t_tile = C::P * col0;  // output index of staged column 0
    constexpr int N_THR = (C::P * (C::TN - 8) + 8 - 5 + 3) / 4;
    static_assert(N_THR <= 256 && 4 * (N_THR - 1) + 15 < C::OS, "head epilogue geometry");
    if (tid < N_THR) {
      float acc[4] = {bias_h, bias_h, bias_h, bias_h};
      // not unrolled over the channels: the fully unrolled form kept 8 x 16 staged values live and took the kernel
      // to 204 VGPRs (2 waves/SIMD); one channel at a time stays near the main loop's budget
#pragma unroll 1
      for (int ci = 0; ci < 8; ++ci) {
        float v[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 x4 = *reinterpret_cast<const float4*>(lds + ci * C::OS + 4 * tid + 4 * q4);
          v[4 * q4] = x4.x;
          v[4 * q4 + 1] = x4.y;
          v[4 * q4 + 2] = x4.z;
          v[4 * q4 + 3] = x4.w;
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const float wk = w[ci * 11 + k];
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk, v[o + k], acc[o]);  // staged col (4i+5+o) + k - 5
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int col = 4 * tid + 5 + o;
        const int t = t_tile + col;
        if (col >= 8 && col < 8 + C::P * (C::TN - 8) && t < a.l_out) y[t] = 1.f / (1.f + expf(-acc[o]));
      }
    }
  } else if constexpr (C::EPI == EPI_SOFTMAX3) {
    static_assert(C::EPI != EPI_SOFTMAX3 || C::COUT == 8, "PhaseNet head: 8 -> 3");
    float* d = a.dst + (long)win * a.wsd + a.dst_halo;
    float w[3][8], bb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bb[c] = a.e1[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) w[c][k] = a.e0[c * 8 + k];
    }
    for (int q = tid; q < C::OW; q += 256) {
      const int t = t0 + q;
      if (t < a.l_out) {
        float z[3] = {bb[0], bb[1], bb[2]};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = lds[k * C::OS + q];
#pragma unroll
          for (int c = 0; c < 3; ++c) z[c] = fmaf(w[c][k], v, z[c]);
        }
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
        const float inv = 1.f / (e0 + e1 + e2);
        d[t] = e0 * inv;
        d[(long)a.lsd + t] = e1 * inv;
        d[2l * a.lsd + t] = e2 * inv;
      }
    }
  }
  __syncthreads();
  CONV_STAMP()
#undef CONV_STAMP
}

template <class C>
int launch_conv(const ConvArgs& a, int cols, hipStream_t stream) {
  constexpr int STEP = (C::EPI == EPI_HEAD) ? C::TN - 8 : C::TN;
  dim3 grid((cols + STEP - 1) / STEP, a.n_windows, 1);
  hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(256), C::LDS_FLOATS * sizeof(float), stream, a);
  return 0;
}

// ---- host-side A-matrix builders + MFMA-order packing (conv_pack.cpp) ------------------
// Amat is row-major [M][TAPS*CINP], column index = tap*CINP + ci.
std::vector<float> amat_conv(const float* W, int cout, int cin, int K, int stride, int P, int cinp,
                             const float* row_scale);
std::vector<float> amat_convT_k7s4(const float* Wt, int cin, int cout, int cinp, const float* row_scale);
std::vector<float> amat_upconv(const float* W, int cout, int cin, int K, int cinp);
std::vector<float> pack_afrag(const std::vector<float>& amat, int M, int cinp, int taps);

}  // namespace vp
