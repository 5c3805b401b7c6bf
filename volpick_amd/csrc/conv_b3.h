// Convolutions on the bf16 matrix cores with EXACT operands (the PhaseNet core's five deepest layers; the EQTransformer
// ResCNN kernel in eqt_fused.hip is the same idea written out by hand).
//
// An fp32 number is exactly the sum of three bfloat16 pieces (hi = rne(x), mid = rne(x - hi), lo = x - hi - mid: 8 + 8 + 8
// significant bits), bf16 x bf16 products are exact in fp32, and v_mfma_f32_16x16x32_bf16 runs at 16x the rate of the
// fp32 MFMA.  Of the nine piece products of w x, the six with i + j <= 2 carry everything down to 2^-24 |w x| (what is
// dropped is of the order of the rounding of one fp32 product), so a conv computed this way agrees with the fp32-MFMA
// form to fp32 rounding at 6 / 16 of its matrix time.
//
// Layout.  K of an instruction = 32 input channels at ONE tap; a lane supplies 8 consecutive channels.  Conv inputs
// therefore rest in LDS as three bf16 images [piece][column][C + 8 channels]: a B fragment is one 16-byte read per piece
// (column stride (C + 8) * 2 bytes = 16 mod 32 bytes for C = 32, 64, 128: the 16 columns of a read fall on disjoint
// banks), and a lane's four accumulator rows -- four consecutive output channels of one column -- leave as one 8-byte
// store per piece.  Transposed (four-phase) layers order their GEMM rows (phase, channel) instead of (channel, phase) for
// that.  The A operand: [m-tile][tap * KS + channel step][piece][lane][8 bf16], streamed from L2 three K-steps ahead.
#pragma once
#include "bf16.h"
#include "conv_lds.h"

namespace vp {

typedef __bf16 bf16x8_b3 __attribute__((ext_vector_type(8)));

// a three-piece image: element (piece, column, channel) at img[piece * ps + column * CS + channel]; logical sample t
// sits in column t + c0
template <int C>
struct B3Image {
  static constexpr int CS = C + 8;
  bf16_t* img;
  int ps, c0;
};

__device__ __forceinline__ void b3_split(const float v, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = to_bf16(v);
  const float r1 = v - from_bf16(h);
  m = to_bf16(r1);
  l = to_bf16(r1 - from_bf16(m));
}
// four consecutive channels ch .. ch + 3 of column col: one 8-byte store per piece
__device__ __forceinline__ void b3_store4(bf16_t* img, const int ps, const int cs, const int col, const int ch, const float (&v)[4]) {
  // pairs at a time: one v_cvt_pk_bf16_f32 rounds two values and leaves them packed (the same pieces as b3_split's)
  const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
  const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
  const unsigned m0 = pack_bf16x2(r0, r1), m1 = pack_bf16x2(r2, r3);
  const unsigned l0 = pack_bf16x2(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), l1 = pack_bf16x2(r2 - bf16_lo(m1), r3 - bf16_hi(m1));
  bf16_t* p = img + col * cs + ch;
#ifdef B3_EXP
  if (B3_EXP & 4) {  // timing probe (see b3c_store4): the pieces are computed, nothing is stored
    asm volatile("" ::"v"(h0), "v"(h1), "v"(m0), "v"(m1), "v"(l0), "v"(l1), "v"(p));
    return;
  }
#endif
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + ps) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2 * ps) = make_uint2(l0, l1);
}

// Output side of a layer: a three-piece image.  Columns [col_lo, col_hi) are the ones the layer's tiles write (zeros
// where the sample lies outside [0, L)); zero_rest() clears every other column (the next layer's padding) and may run
// beside the layer: the two sets are disjoint.
template <int C>
struct B3Store {
  static constexpr int CS = C + 8;
  bf16_t* img;
  int ps, c0, L, ncols;
  __device__ __forceinline__ void quad(const int co, const int t, const float (&v)[4]) const {
    const int col = t + c0;
    if ((unsigned)col >= (unsigned)ncols) return;
    const bool in = (unsigned)t < (unsigned)L;
    const float z[4] = {in ? v[0] : 0.f, in ? v[1] : 0.f, in ? v[2] : 0.f, in ? v[3] : 0.f};
    b3_store4(img, ps, CS, col, co, z);
  }
  __device__ __forceinline__ void zero_rest(const int col_lo, const int col_hi, const int tid, const int nth) const {
    constexpr int Q = CS / 8;  // 16-byte chunks per column
    for (int i = tid; i < 3 * ncols * Q; i += nth) {
      const int pc = i / (ncols * Q), r = i - pc * (ncols * Q), col = r / Q, q = r - col * Q;
      if (col < col_lo || col >= col_hi) *reinterpret_cast<uint4*>(img + pc * ps + col * CS + 8 * q) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
};

// Output side: an fp32 image [channel][S], sample t at column B + t (the layers behind the bf16 chain)
template <int S, int B>
struct F32QuadStore {
  float* img;
  int L;
  __device__ __forceinline__ void quad(const int co, const int t, const float (&v)[4]) const {
    if ((unsigned)t < (unsigned)L) {
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(co + r) * S + B + t] = v[r];
    }
  }
};

// lds_epilogue hook (conv_lds.h) for an fp32-MFMA layer whose OUTPUT feeds the bf16 chain (P = 1)
template <int C>
struct B3BlockStore : B3Store<C> {
  static constexpr bool custom_block_epilogue = true;
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 1 && L::OUT_OFF == 0, "plain conv");
#pragma unroll
    for (int j = 0; j < L::NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];
        if (L::RELU) v[r] = fmaxf(v[r], 0.f);
      }
      this->quad(mt * 16 + 4 * g, colb + j * 16 + n, v);
    }
  }
};

// The conv.  L = LdsLayer (conv_lds.h) with CIN1, CIN2 multiples of 32; MPERM: GEMM rows ordered (phase, channel).
// Items = (m-tile, block of NB n-tiles); wave w takes items w, w + nwaves, ...
// PRE: the first PF K-steps of the wave's FIRST item are already on their way into q (conv_b3_request, called in front of the
// barrier that precedes this layer: a layer's first fragments otherwise make their trip to L2 with all sixteen waves waiting).
template <class L, bool MPERM, int C1, int C2, class Store, int PF_, bool PRE>
__device__ __forceinline__ void conv_b3_impl(const B3Image<C1> i1, const B3Image<C2> i2, const uint4* __restrict__ af3,
                                             const float* __restrict__ bias, const int cols, const Store store, const int wave,
                                             const int nwaves, const int lane, uint4 (&q)[PF_ + 1][3]) {
  static_assert(L::CIN1 % 32 == 0 && L::CIN2 % 32 == 0 && L::CIN1 == C1 && (L::CIN2 == 0 || L::CIN2 == C2), "32-channel K-steps");
  static_assert(MPERM || L::P == 1, "multi-phase layers order their rows (phase, channel)");
  constexpr int KS1 = L::CIN1 / 32, KS = (L::CIN1 + L::CIN2) / 32, STEPS = L::TAPS * KS, NB = L::NB;
  constexpr int PF = PF_;  // K-steps of A in flight ahead of the MFMAs (4 and 5 measured the same: the layers with one n-tile per
                         // item run at the rate the weights stream out of L2, 34-44 B/clk/CU)
  constexpr int MT = L::M / 16, MT_PER_PHASE = L::COUT / 16;
  const int NT = (cols + 15) >> 4, NBLK = (NT + NB - 1) / NB, items = MT * NBLK;
  const int g = lane >> 4, n = lane & 15;
  for (int item = wave; item < items; item += nwaves) {
    const int mt = item % MT, nblk = item / MT, colb = nblk * NB * 16;
    const int phase = MPERM ? mt / MT_PER_PHASE : 0, co0 = (MPERM ? (mt % MT_PER_PHASE) : mt) * 16 + 4 * g;
    f32x4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float biasv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = bias[co0 + r];
    const uint4* ap = af3 + (long)mt * (STEPS * 3 * 64) + lane;
    auto load_a = [&](const int s) {  // q: A pieces of K-steps s .. s + PF
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) q[s % (PF + 1)][pc] = ap[(s * 3 + pc) * 64];
    };
    if (!(PRE && item == wave)) {  // (uniform)
#pragma unroll
      for (int s = 0; s < PF && s < STEPS; ++s) load_a(s);
    }
    // B: the (K-step, n-tile) pairs as one sequence, the fragment of pair i + 1 read before the MFMAs of pair i
    uint4 b[2][3];
    auto load_b = [&](uint4 (&bv)[3], const int s, const int j) {
      const int tap = s / KS, ks = s - tap * KS;
      const int col = (colb + j * 16 + n) * L::SN + tap + L::IN_OFF;
      if (ks < KS1) {
        const bf16_t* p = i1.img + (col + i1.c0) * (C1 + 8) + ks * 32 + 8 * g;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bv[pc] = *reinterpret_cast<const uint4*>(p + pc * i1.ps);
      } else {
        const bf16_t* p = i2.img + (col + i2.c0) * (C2 + 8) + (ks - KS1) * 32 + 8 * g;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bv[pc] = *reinterpret_cast<const uint4*>(p + pc * i2.ps);
      }
    };
    load_b(b[0], 0, 0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + PF < STEPS) load_a(s + PF);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = s * NB + j;
        if (i + 1 < STEPS * NB) load_b(b[(i + 1) & 1], (i + 1) / NB, (i + 1) % NB);
        __builtin_amdgcn_sched_barrier(0);
        // (w piece, x piece), smallest products first
        constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, q[s % (PF + 1)][WP[t]]),
                                                          __builtin_bit_cast(bf16x8_b3, b[i & 1][XP[t]]), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];
        if (L::RELU) v[r] = fmaxf(v[r], 0.f);
      }
      const int c = colb + j * 16 + n;
      store.quad(co0, MPERM ? L::P * c + phase + L::OUT_OFF : c + L::OUT_OFF, v);
    }
  }
}

template <class L, bool MPERM, int C1, int C2, class Store, int PF_ = 3>
__device__ __forceinline__ void conv_b3(const B3Image<C1> i1, const B3Image<C2> i2, const uint4* __restrict__ af3,
                                        const float* __restrict__ bias, const int cols, const Store store, const int wave,
                                        const int nwaves, const int lane) {
  uint4 q[PF_ + 1][3];
  conv_b3_impl<L, MPERM, C1, C2, Store, PF_, false>(i1, i2, af3, bias, cols, store, wave, nwaves, lane, q);
}
// the first PF K-steps of the wave's first item of layer L, requested ahead (same item order as conv_b3: item = wave)
template <class L, int PF_ = 3>
__device__ __forceinline__ void conv_b3_request(const uint4* __restrict__ af3, const int cols, const int wave, const int lane,
                                                uint4 (&q)[PF_ + 1][3]) {
  constexpr int KS = (L::CIN1 + L::CIN2) / 32, STEPS = L::TAPS * KS, MT = L::M / 16;
  const int NT = (cols + 15) >> 4, NBLK = (NT + L::NB - 1) / L::NB, items = MT * NBLK;
  if (wave < items) {
    const uint4* ap = af3 + (long)(wave % MT) * (STEPS * 3 * 64) + lane;
#pragma unroll
    for (int s = 0; s < PF_ && s < STEPS; ++s)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) q[s % (PF_ + 1)][pc] = ap[(s * 3 + pc) * 64];
  }
}
template <class L, bool MPERM, int C1, int C2, class Store, int PF_ = 3>
__device__ __forceinline__ void conv_b3_requested(const B3Image<C1> i1, const B3Image<C2> i2, const uint4* __restrict__ af3,
                                                  const float* __restrict__ bias, const int cols, const Store store, const int wave,
                                                  const int nwaves, const int lane, uint4 (&q)[PF_ + 1][3]) {
  conv_b3_impl<L, MPERM, C1, C2, Store, PF_, true>(i1, i2, af3, bias, cols, store, wave, nwaves, lane, q);
}

// ---- a concat layer whose two inputs cannot rest in LDS as piece images at the same time (PhaseNet up1.same) ------------------
// One item (m-tile mt, NB n-tiles from column colb) per wave, its accumulators carried across calls: the K-steps of ONE
// 32-channel step `KSTEP` of the concatenated input (all taps) from the image `im` that holds those channels.
// (PRE: the first three taps' fragments were requested ahead into q by conv_b3_part_request, in front of a barrier)
template <class L, int KSTEP, int NB, bool PRE>
__device__ __forceinline__ void conv_b3_part_impl(const B3Image<32> im, const uint4* __restrict__ af3, const int mt, const int colb,
                                                  const int lane, f32x4 (&acc)[NB], uint4 (&q)[4][3]) {
  static_assert(L::P == 1 && L::SN == 1 && (L::CIN1 + L::CIN2) % 32 == 0, "plain unit-stride conv, 32-channel K-steps");
  constexpr int KS = (L::CIN1 + L::CIN2) / 32, STEPS = L::TAPS * KS, PF = 3;
  static_assert(KSTEP < KS, "channel step of the concat");
  const int g = lane >> 4, n = lane & 15;
  const uint4* ap = af3 + (long)mt * (STEPS * 3 * 64) + lane;
  auto load_a = [&](const int tap) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) q[tap % (PF + 1)][pc] = ap[((tap * KS + KSTEP) * 3 + pc) * 64];
  };
  if (!PRE) {
#pragma unroll
    for (int tap = 0; tap < PF && tap < L::TAPS; ++tap) load_a(tap);
  }
  uint4 b[2][3];
  auto load_b = [&](uint4 (&bv)[3], const int tap, const int j) {
    const bf16_t* p = im.img + (colb + j * 16 + n + tap + L::IN_OFF + im.c0) * 40 + 8 * g;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) bv[pc] = *reinterpret_cast<const uint4*>(p + pc * im.ps);
  };
  load_b(b[0], 0, 0);
#pragma unroll
  for (int tap = 0; tap < L::TAPS; ++tap) {
    if (tap + PF < L::TAPS) load_a(tap + PF);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = tap * NB + j;
      if (i + 1 < L::TAPS * NB) load_b(b[(i + 1) & 1], (i + 1) / NB, (i + 1) % NB);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, q[tap % (PF + 1)][WP[t]]),
                                                        __builtin_bit_cast(bf16x8_b3, b[i & 1][XP[t]]), acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <class L, int KSTEP, int NB>
__device__ __forceinline__ void conv_b3_part(const B3Image<32> im, const uint4* __restrict__ af3, const int mt, const int colb,
                                             const int lane, f32x4 (&acc)[NB]) {
  uint4 q[4][3];
  conv_b3_part_impl<L, KSTEP, NB, false>(im, af3, mt, colb, lane, acc, q);
}
template <class L, int KSTEP>
__device__ __forceinline__ void conv_b3_part_request(const uint4* __restrict__ af3, const int mt, const int lane, uint4 (&q)[4][3]) {
  constexpr int KS = (L::CIN1 + L::CIN2) / 32, STEPS = L::TAPS * KS, PF = 3;
  const uint4* ap = af3 + (long)mt * (STEPS * 3 * 64) + lane;
#pragma unroll
  for (int tap = 0; tap < PF && tap < L::TAPS; ++tap)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) q[tap % (PF + 1)][pc] = ap[((tap * KS + KSTEP) * 3 + pc) * 64];
}
template <class L, int KSTEP, int NB>
__device__ __forceinline__ void conv_b3_part_requested(const B3Image<32> im, const uint4* __restrict__ af3, const int mt, const int colb,
                                                       const int lane, f32x4 (&acc)[NB], uint4 (&q)[4][3]) {
  conv_b3_part_impl<L, KSTEP, NB, true>(im, af3, mt, colb, lane, acc, q);
}

// 4 x 4 transpose between a lane's four registers and the four 16-lane rows of the wavefront: afterwards v[k] of row g holds what
// v[g] of row k held (v_permlane32_swap: upper half of the first operand <-> lower half of the second; v_permlane16_swap: odd
// rows of the first <-> even rows of the second).
__device__ __forceinline__ void b3_transpose_rows(float (&v)[4]) {
  const auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
  const auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
  const auto q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
  const auto q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
  v[0] = __uint_as_float(q01[0]);
  v[1] = __uint_as_float(q01[1]);
  v[2] = __uint_as_float(q23[0]);
  v[3] = __uint_as_float(q23[1]);
}

// lds_epilogue hook (conv_lds.h) of a FOUR-PHASE fp32-MFMA layer (transposed conv, rows (channel, phase)) whose output feeds
// the bf16 chain: a lane's four values are four consecutive samples of ONE channel (channel 4 mt + row g of the lane); the
// transpose above turns them into the m-tile's four channels at ONE sample (phase g), which leave as one 8-byte store per piece.
// Only samples [0, L) are written: zero the other columns first (B3Store::zero_rest with the written range).
template <int C>
struct B3PhaseStore {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  int ps, c0, L;
  template <class LY>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[LY::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(LY::P == 4, "four phases per channel");
#pragma unroll
    for (int j = 0; j < LY::NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];  // biasv: the lane's channel, the same for its four phases
        if (LY::RELU) v[r] = fmaxf(v[r], 0.f);
      }
      b3_transpose_rows(v);  // v[k] = channel 4 mt + k at phase g
      const int t = 4 * (colb + j * 16 + n) + g + LY::OUT_OFF;
      if ((unsigned)t < (unsigned)L) b3_store4(img, ps, C + 8, t + c0, 4 * mt, v);
    }
  }
};

// an fp32 image [32 channels][S], sample t at column B + t, -> the three-piece image (samples [t_lo, t_hi): whatever the fp32
// image holds there, its zero margins included)
template <int S, int B>
__device__ __forceinline__ void b3_from_f32(const float* src, const B3Image<32> im, const int t_lo, const int t_hi, const int tid,
                                            const int nth) {
  const int n = t_hi - t_lo;
  for (int i = tid; i < 8 * n; i += nth) {
    const int cq = i / n, t = t_lo + (i - cq * n);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = src[(4 * cq + r) * S + B + t];
    b3_store4(im.img, im.ps, 40, t + im.c0, 4 * cq, v);
  }
}

// ---- one item per wave, the operand in registers (time-tiled kernels whose weights do not change from tile to tile) --------
// K-steps of a layer with C input channels: C >= 32: (tap, 32-channel step); C = 16 / 8: all channels of 2 / 4 consecutive
// taps per step (net.hip: b3_operand pads the filter with zero taps), lane group g = lane / 16 supplies tap `tap_of_lane`,
// channels `ch_of_lane` .. + 7 of it.
template <int C, int TAPS>
struct B3Steps {
  static constexpr int KS = C >= 32 ? C / 32 : 1, TPK = C >= 32 ? 1 : 32 / C, LPT = C >= 32 ? 4 : C / 8;
  static constexpr int STEPS = C >= 32 ? TAPS * KS : (TAPS + TPK - 1) / TPK;
  __device__ static __forceinline__ int tap_of_lane(const int g) { return C >= 32 ? 0 : g / LPT; }
  __device__ static __forceinline__ int ch_of_lane(const int g) { return C >= 32 ? 8 * g : 8 * (g % LPT); }
  // element offset of K-step s relative to the lane's pointer
  static constexpr int step_off(const int s) { return C >= 32 ? (s / KS) * (C + 8) + (s % KS) * 32 : s * TPK * (C + 8); }
};
template <int C, int TAPS>
__device__ __forceinline__ void b3_load_a(const uint4* __restrict__ af3, const int mt, const int lane,
                                          uint4 (&a)[B3Steps<C, TAPS>::STEPS * 3]) {
  constexpr int N = B3Steps<C, TAPS>::STEPS * 3;
  const uint4* p = af3 + (long)mt * (N * 64) + lane;
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = p[i * 64];
}
// ---- chunk-plane images: [piece][8-channel chunk][column][8 channels] --------------------------------------------------------
// The image layout of the time-tiled decoder tail (eqt_tail_b3.hip).  A fragment piece is one ds_read_b128: 16 lanes read 256
// consecutive bytes, and with the chunk planes a multiple of 256 bytes apart the mixed lane groups of the instruction
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) fall on disjoint banks -- no padding channels, 6 bytes per value.
// The 8-byte stores of a lane's four channels run 4-way conflicted (16 columns two apart, 32-byte pitch).  Measured on
// that kernel (tools/tail_clock.py), against [column][C + 8 channels] (reads 2-way, stores 4-way conflicted) and against
// [quad][column][4 channels] (two ds_read_b64 per piece, conflict-free; stores 2-way): the stage times agree within 3 %, and
// so do one and two (K-step, n-tile) pairs of read-ahead.  What a stage costs beyond its MFMA issue is the vector
// issue of the epilogues (an MFMA holds the SIMD's vector issue for 8 of its 16 cycles; split + store of an n-tile is
// about 40 VALU instructions) and the fragment reads themselves, not their bank conflicts.
template <int C, int NC>
struct B3Chunk {
  static_assert(NC % 2 == 0, "16-byte columns; the bank property above holds for NC % 16 == 0 (PhaseNet's up1.same output: 194)");
  static constexpr int CHS = NC * 8, PS = (C / 8) * CHS;  // bf16 per chunk plane / per piece
};
// four consecutive channels 4 quad .. 4 quad + 3 of column col, split into the three pieces (pairs at a time: v_cvt_pk_bf16_f32)
// B3_EXP (timing probes of tools/, never in the product build; results are WRONG with any of them): 4 = b3c_store4 computes the
// pieces and stores nothing, 16 = it does not compute them either, 32 = it stores them where sixteen lanes fill 128 contiguous bytes
#ifndef B3_EXP
#define B3_EXP 0
#endif
template <int C, int NC>
__device__ __forceinline__ void b3c_store4(bf16_t* img, const int col, const int quad, const float (&v)[4]) {
  using Q = B3Chunk<C, NC>;
  if (B3_EXP & 16) {
    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return;
  }
  const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
  const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
  const unsigned m0 = pack_bf16x2(r0, r1), m1 = pack_bf16x2(r2, r3);
  const unsigned l0 = pack_bf16x2(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), l1 = pack_bf16x2(r2 - bf16_lo(m1), r3 - bf16_hi(m1));
  bf16_t* p = img + (quad >> 1) * Q::CHS + col * 8 + (quad & 1) * 4;
  if (B3_EXP & 32) p = img + (quad >> 1) * Q::CHS + ((col >> 5) * 16 + (threadIdx.x & 15)) * 4;  // 16 lanes -> 128 contiguous bytes: no bank conflict
  if (B3_EXP & 4) {
    asm volatile("" ::"v"(h0), "v"(h1), "v"(m0), "v"(m1), "v"(l0), "v"(l1), "v"(p));
    return;
  }
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + Q::PS) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2 * Q::PS) = make_uint2(l0, l1);
}
// the lane's pointer for b3c_mac_tiles: piece 0, its chunk, column col0 + lane % 16 + tap_of_lane
template <int C, int NC, int TAPS>
__device__ __forceinline__ const bf16_t* b3c_lane_ptr(const bf16_t* img, const int col0, const int lane) {
  using G = B3Steps<C, TAPS>;
  const int g = lane >> 4;
  return img + (G::ch_of_lane(g) / 8) * B3Chunk<C, NC>::CHS + (col0 + (lane & 15) + G::tap_of_lane(g)) * 8;
}
// One m-tile (operand `a` in registers) x NB n-tiles, n-tile after n-tile: the accumulator of n-tile j goes to
// finish(j, acc) -- branch-free code: bias, activation, split, stores -- while the fragments of n-tile j + 1 are on their
// way.  Fragments are read two (K-step, n-tile) pairs ahead of their MFMAs.
// (slot(i), i = 0 .. STEPS * NB - 1: code issued between the K-steps, e.g. the parts of a request that would otherwise stand as
// one burst in front of the stage's MFMAs -- a wave issues in order)
struct B3NoSlot {
  __device__ __forceinline__ void operator()(int) const {}
};
template <int C, int NC, int TAPS, int NB, class Finish, class Slot = B3NoSlot>
__device__ __forceinline__ void b3c_mac_tiles(const bf16_t* p, const uint4 (&a)[B3Steps<C, TAPS>::STEPS * 3], Finish finish, Slot slot = Slot()) {
  using G = B3Steps<C, TAPS>;
  using Q = B3Chunk<C, NC>;
  constexpr int STEPS = G::STEPS;
  constexpr int AHEAD = 2, NBUF = AHEAD + 1, PAIRS = STEPS * NB;
  uint4 b[NBUF][3];
  auto load_b = [&](const int i) {
    const int s = i % STEPS, j = i / STEPS;
    // C >= 32: step = (tap, 32-channel step ks): four chunk planes further per ks; C < 32: TPK taps per step
    const int off = ((C >= 32 ? s / G::KS : s * G::TPK) + j * 16) * 8 + (C >= 32 ? (s % G::KS) * 4 * Q::CHS : 0);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) b[i % NBUF][pc] = *reinterpret_cast<const uint4*>(p + pc * Q::PS + off);
  };
#pragma unroll
  for (int i = 0; i < AHEAD && i < PAIRS; ++i) load_b(i);
  f32x4 prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    if (j > 0) finish(j - 1, prev);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int i = j * STEPS + s;
      if (i + AHEAD < PAIRS) load_b(i + AHEAD);
      slot(i);
      constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};  // (w piece, x piece), smallest products first
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, a[s * 3 + WP[t]]),
                                                     __builtin_bit_cast(bf16x8_b3, b[i % NBUF][XP[t]]), acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    prev = acc;
  }
  finish(NB - 1, prev);
}

// The same with the n-tiles taken in PAIRS whose MFMAs alternate: consecutive instructions never share an accumulator, so one
// wave issues them back to back (a chain into one accumulator issues every 20-26 cycles from one wave, LOG.md).  NB even.
template <int C, int NC, int TAPS, int NB, class Finish, class Slot = B3NoSlot>
__device__ __forceinline__ void b3c_mac_tile_pairs(const bf16_t* p, const uint4 (&a)[B3Steps<C, TAPS>::STEPS * 3], Finish finish, Slot slot = Slot()) {
  using G = B3Steps<C, TAPS>;
  using Q = B3Chunk<C, NC>;
  static_assert(NB % 2 == 0, "pairs of n-tiles");
  constexpr int STEPS = G::STEPS, PAIRS = STEPS * (NB / 2);
  uint4 b[2][2][3];  // [buffer][tile of the pair][piece]
  auto load_b = [&](const int i) {  // i = pair index * STEPS + step
    const int s = i % STEPS, jp = i / STEPS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int off = ((C >= 32 ? s / G::KS : s * G::TPK) + (2 * jp + h) * 16) * 8 + (C >= 32 ? (s % G::KS) * 4 * Q::CHS : 0);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) b[i & 1][h][pc] = *reinterpret_cast<const uint4*>(p + pc * Q::PS + off);
    }
  };
  load_b(0);
  f32x4 prev0 = {0.f, 0.f, 0.f, 0.f}, prev1 = prev0;
#pragma unroll
  for (int jp = 0; jp < NB / 2; ++jp) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    __builtin_amdgcn_sched_barrier(0);
    if (jp > 0) {
      finish(2 * jp - 2, prev0);
      finish(2 * jp - 1, prev1);
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int i = jp * STEPS + s;
      if (i + 1 < PAIRS) load_b(i + 1);
      slot(i);
      constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, a[s * 3 + WP[t]]),
                                                      __builtin_bit_cast(bf16x8_b3, b[i & 1][0][XP[t]]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, a[s * 3 + WP[t]]),
                                                      __builtin_bit_cast(bf16x8_b3, b[i & 1][1][XP[t]]), acc1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    prev0 = acc0;
    prev1 = acc1;
  }
  finish(NB - 2, prev0);
  finish(NB - 1, prev1);
}

// The same with the accumulators of all NB n-tiles kept: acc[j] += operand x fragments of n-tile j.  For a layer whose K is walked
// in two passes over ONE image that is refilled in between (PhaseNet up2.same).
template <int C, int NC, int TAPS, int NB>
__device__ __forceinline__ void b3c_mac_tiles_acc(const bf16_t* p, const uint4 (&a)[B3Steps<C, TAPS>::STEPS * 3], f32x4 (&acc)[NB]) {
  using G = B3Steps<C, TAPS>;
  using Q = B3Chunk<C, NC>;
  constexpr int STEPS = G::STEPS;
  constexpr int AHEAD = 1, NBUF = AHEAD + 1, PAIRS = STEPS * NB;
  uint4 b[NBUF][3];
  auto load_b = [&](const int i) {
    const int s = i % STEPS, j = i / STEPS;
    const int off = ((C >= 32 ? s / G::KS : s * G::TPK) + j * 16) * 8 + (C >= 32 ? (s % G::KS) * 4 * Q::CHS : 0);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) b[i % NBUF][pc] = *reinterpret_cast<const uint4*>(p + pc * Q::PS + off);
  };
  load_b(0);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int i = j * STEPS + s;
      if (i + AHEAD < PAIRS) load_b(i + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, a[s * 3 + WP[t]]),
                                                        __builtin_bit_cast(bf16x8_b3, b[i % NBUF][XP[t]]), acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// B3PhaseStore for a chunk-plane image (C = 16: two chunks of eight channels)
template <int C, int NC>
struct B3PhaseStoreC {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  int c0, L;
  template <class LY>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[LY::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(LY::P == 4, "four phases per channel");
#pragma unroll
    for (int j = 0; j < LY::NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];
        if (LY::RELU) v[r] = fmaxf(v[r], 0.f);
      }
      b3_transpose_rows(v);  // v[k] = channel 4 mt + k at phase g
      const int t = 4 * (colb + j * 16 + n) + g + LY::OUT_OFF;
      if ((unsigned)t < (unsigned)L) b3c_store4<C, NC>(img, t + c0, mt, v);
    }
  }
};
// lds_epilogue hooks (conv_lds.h) of fp32-MFMA layers whose output feeds a bf16 layer with 8 or 16 input channels: chunk-plane
// image, sample t at column t + c0; every computed column inside the image is written (zeros outside [0, L)).
// P = 1: a lane's four rows are four consecutive channels.
template <int C, int NC>
struct B3BlockStoreC {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  int c0, L;
  template <class LY>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[LY::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(LY::P == 1 && LY::OUT_OFF == 0, "plain conv");
#pragma unroll
    for (int j = 0; j < LY::NB; ++j) {
      const int t = colb + j * 16 + n;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];
        if (LY::RELU) v[r] = fmaxf(v[r], 0.f);
        if (t >= L) v[r] = 0.f;
      }
      if (t + c0 < NC) b3c_store4<C, NC>(img, t + c0, (mt * 16 + 4 * g) >> 2, v);
    }
  }
};
// P = 2, rows (channel, phase): a lane holds (channel 2 g, phases 0 1), (channel 2 g + 1, phases 0 1); two v_permlane16_swap
// (odd rows of the first operand <-> even rows of the second) leave rows 0 / 1 with channels 0-3 at phase 0 / 1 and rows
// 2 / 3 with channels 4-7.
template <int C, int NC>
struct B3PairStoreC {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  int c0, L;
  template <class LY>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[LY::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(LY::P == 2 && LY::M == 16, "two phases x eight channels = one m-tile");
#pragma unroll
    for (int j = 0; j < LY::NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][r] + biasv[r];
        if (LY::RELU) v[r] = fmaxf(v[r], 0.f);
      }
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
      const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
      float w[4] = {__uint_as_float(a[0]), __uint_as_float(b[0]), __uint_as_float(a[1]), __uint_as_float(b[1])};
      const int t = 2 * (colb + j * 16 + n) + (g & 1) + LY::OUT_OFF;
      if ((unsigned)t >= (unsigned)L) w[0] = w[1] = w[2] = w[3] = 0.f;
      if ((unsigned)(t + c0) < (unsigned)NC) b3c_store4<C, NC>(img, t + c0, g >> 1, w);
    }
  }
};

// zero the columns of a chunk-plane image outside [col_lo, col_hi)
template <int C, int NC>
__device__ __forceinline__ void b3c_zero_rest(bf16_t* img, const int col_lo, const int col_hi, const int tid, const int nth) {
  using Q = B3Chunk<C, NC>;
  const int nz = NC - (col_hi - col_lo);
  for (int i = tid; i < 3 * (C / 8) * nz; i += nth) {
    const int cp = i / nz, k = i - cp * nz, col = k < col_lo ? k : col_hi + (k - col_lo);
    *reinterpret_cast<uint4*>(img + (cp / (C / 8)) * Q::PS + (cp % (C / 8)) * Q::CHS + col * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
}
// an fp32 image [C channels][S], sample t at column B + t, -> the chunk-plane image, columns [0, NC) <-> samples [-c0, NC - c0)
template <int C, int NC, int S, int B>
__device__ __forceinline__ void b3c_from_f32(const float* src, bf16_t* img, const int c0, const int tid, const int nth) {
  for (int i = tid; i < (C / 4) * NC; i += nth) {
    const int cq = i / NC, col = i - cq * NC;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (B + col - c0 < S) ? src[(4 * cq + r) * S + B + col - c0] : 0.f;
    b3c_store4<C, NC>(img, col, cq, v);
  }
}

}  // namespace vp
