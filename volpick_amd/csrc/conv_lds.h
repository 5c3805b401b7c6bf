// LDS-to-LDS form of the polyphase MFMA convolution (conv_mfma.h): the building block of the
// fused PhaseNet kernels, where whole layer chains run inside one workgroup and activations
// never leave the CU.
//
// An "image" is a [C][S] float array in LDS; logical sample t of a row sits at column BASE + t.
// Columns outside a layer's valid range hold zeros (they are the convolution padding) or
// don't-care values that only feed masked output columns.
//
//   out[co][P*n + p + OUT_OFF] = act(bias[co] + sum_tap sum_ci A[(co,p)][(tap,ci)] * in[ci][SN*n + tap + IN_OFF])
//
// Work is split into items = (m-tile, block of NB n-tiles); wave w takes items w, w+nwaves, ...
// Each item keeps NB accumulators (NB*4 VGPRs), loads one A fragment per K-step from L2/L1 and
// NB B fragments from LDS (per-lane base + immediate offsets).
#pragma once
#include <type_traits>

#include "vp_common.h"

namespace vp {

template <int CIN1_, int CIN2_, int COUT_, int P_, int TAPS_, int SN_, int IN_OFF_, int OUT_OFF_, int NB_, int RELU_>
struct LdsLayer {
  static constexpr int CIN1 = CIN1_, CIN2 = CIN2_, CIN = CIN1_ + CIN2_;
  static constexpr int CB1 = (CIN1 + 3) / 4, CB2 = (CIN2 + 3) / 4, CB = CB1 + CB2, CINP = 4 * CB;
  static constexpr int COUT = COUT_, P = P_, TAPS = TAPS_, SN = SN_, IN_OFF = IN_OFF_, OUT_OFF = OUT_OFF_;
  static constexpr int NB = NB_, RELU = RELU_;
  static constexpr int M = COUT * P, MT = M / 16;
  static_assert(M % 16 == 0, "M must be a multiple of the 16-row MFMA tile");
  static_assert(CIN2 == 0 || CIN1 % 4 == 0, "concat boundary must fall on a 4-channel block");
};

// Strided layers (SN = 4 or 8: Conv1d stride 4, and its two-phase form) read the taps of an output column as ONE run
// of consecutive floats: fetched tap by tap with ds_read_b32, the 16 columns of a fragment are 4 or 8 floats apart and
// collide 4- to 8-way on the 32 LDS banks (measured: the level-0 strided conv was LDS-bound at twice its MFMA time).
// BRun fetches the run as aligned 16-byte chunks instead (lanes 16-32 bytes apart) and hands the taps out of registers.
template <class L, int B>
struct BRun {
  static constexpr bool use = (L::SN % 4 == 0);
  static constexpr int OFF = (((B + L::IN_OFF) % 4) + 4) % 4;  // position of tap 0 inside its 16-byte chunk
  static constexpr int NQ = (OFF + L::TAPS + 3) / 4;
  f32x4 q[NQ];
  // bp = address of tap 0 (16-byte aligned after subtracting OFF: image rows and SN are multiples of 4 floats)
  __device__ __forceinline__ void load(const float* bp) {
    const f32x4* p = reinterpret_cast<const f32x4*>(bp - OFF);
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = p[k];
  }
  __device__ __forceinline__ float tap(int t) const { return q[(OFF + t) / 4][(OFF + t) % 4]; }
};

// Epilogue shared by conv_lds and conv_lds_areg.  Row m = 16*mt + 4*g + r of the D tile is (co, p) = (m / P, m % P);
// with P in {1,2,4} both split into a per-lane part and a compile-time part of r, so every store address is a
// per-lane base plus immediates.  If the whole block of columns is inside the store's unconditional range
// (wave-uniform test) the per-element range checks are skipped.  With P = 4 the four registers of a lane are four
// CONSECUTIVE samples of one channel: a store functor that declares `vec4` takes them as one 16-byte value
// (scalar stores of a four-phase layer are 16 bytes apart across the lanes and collide 8-way on the LDS banks —
// the epilogue of the level-0 transposed conv took as long as its MFMAs).
template <class S, class = void>
struct has_vec4 : std::false_type {};
template <class S>
struct has_vec4<S, std::void_t<decltype(&S::vec4)>> : std::true_type {};

// A store functor may bring its own epilogue for a whole block (`block_epilogue` member template): output layouts whose
// addresses are cheap per block but not per element (eqt_tail.hip: the heads' transposed staging of stage 6).
template <class S, class = void>
struct has_block_epilogue : std::false_type {};
template <class S>
struct has_block_epilogue<S, std::void_t<decltype(S::custom_block_epilogue)>> : std::true_type {};

template <class L, class Store>
__device__ __forceinline__ void lds_epilogue_default(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt,
                                                     const int colb, const int g, const int n, Store& store);

template <class L, class Store>
__device__ __forceinline__ void lds_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                             const int g, const int n, Store& store) {
  if constexpr (has_block_epilogue<Store>::value) {
    store.template block_epilogue<L>(acc, biasv, mt, colb, g, n);
  } else {
    lds_epilogue_default<L>(acc, biasv, mt, colb, g, n, store);
  }
}

template <class L, class Store>
__device__ __forceinline__ void lds_epilogue_default(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt,
                                                     const int colb, const int g, const int n, Store& store) {
  static_assert(L::P == 1 || L::P == 2 || L::P == 4, "P must divide the 4-row register group");
  const int co_lane = mt * (16 / L::P) + (4 * g) / L::P;
  const int t_first = L::P * colb + L::OUT_OFF, t_last = L::P * (colb + L::NB * 16) - 1 + L::OUT_OFF;
  const bool fast = store.all_valid(t_first, t_last);
  if constexpr (L::P == 4 && has_vec4<Store>::value) {
    if (fast) {
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[j][r] + biasv[r];
          if (L::RELU) v[r] = fmaxf(v[r], 0.f);
        }
        store.vec4(co_lane, 4 * (colb + j * 16 + n) + L::OUT_OFF, v);
      }
      return;
    }
  }
  // two copies of the loop nest, not one nest with the test inside: per element hipcc otherwise emits a branch around
  // the range checks plus an exec-masked store (12 instructions and a taken branch per value, ~2 k cycles per block
  // of 16 values with the matrix pipe idle -- tools/tail_clock.py)
  if (fast) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co_lane + r / L::P, p = r % L::P;
      const float b = biasv[r];
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        float v = acc[j][r] + b;
        if (L::RELU) v = fmaxf(v, 0.f);
        store.unchecked(co, L::P * (colb + j * 16 + n) + p + L::OUT_OFF, v);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = co_lane + r / L::P, p = r % L::P;
    const float b = biasv[r];
#pragma unroll
    for (int j = 0; j < L::NB; ++j) {
      float v = acc[j][r] + b;
      if (L::RELU) v = fmaxf(v, 0.f);
      store(co, L::P * (colb + j * 16 + n) + p + L::OUT_OFF, v);
    }
  }
}

// in1/in2: LDS images (row strides S1/S2, logical 0 at column B1/B2).  afrag: packed A
// fragments [MT][CB][TAPS][64] in global memory.  store(co, t, v) consumes every output
// (t = P*col + p + OUT_OFF in the caller's local coordinates) and applies its own masks.
// ADEEP: the A fragments are fetched THREE channel blocks ahead (four register buffers) and the B fragments come out of
// LDS tap by tap — for the layers whose weights do not fit the L1: one block (TAPS * NB MFMAs = 200-1300 cycles) of
// lead does not cover an L2 round trip under load, and every block then opens with a stall.
template <class L, int S1, int B1, int S2, int B2, bool PIPE = true, bool BDB = true, bool ADEEP = false, class Store>
__device__ __forceinline__ void conv_lds(const float* in1, const float* in2, const float* __restrict__ afrag,
                                         const float* __restrict__ bias, const int cols, Store store, const int wave,
                                         const int nwaves, const int lane) {
  const int NT = (cols + 15) >> 4;
  const int NBLK = (NT + L::NB - 1) / L::NB;
  const int items = L::MT * NBLK;
  const int g = lane >> 4, n = lane & 15;
  for (int item = wave; item < items; item += nwaves) {
    const int mt = item % L::MT, nblk = item / L::MT;
    const int colb = nblk * L::NB * 16;
    f32x4 acc[L::NB];
#pragma unroll
    for (int j = 0; j < L::NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = afrag + (long)mt * L::CB * L::TAPS * 64 + lane;
    float biasv[4];  // fetched now, consumed after the K loop (no exposed L2 round trip in the epilogue)
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = bias[mt * (16 / L::P) + (4 * g) / L::P + r / L::P];
    const float* bp1 = in1 + g * S1 + B1 + (colb + n) * L::SN + L::IN_OFF;
    const float* bp2 = in2 + g * S2 + B2 + (colb + n) * L::SN + L::IN_OFF;
    // K loop over 4-channel blocks, software-pipelined by hand: the operands of block cb+1 (TAPS A
    // fragments from L2, TAPS*NB B fragments from LDS) are issued BEFORE block cb's TAPS*NB MFMAs;
    // sched_barrier keeps hipcc from sinking the loads back down to their first use (it otherwise
    // emits load -> s_waitcnt 0 -> mfma for every K-step and the wave stalls on every L2 round trip).
    float a0[L::TAPS], a1[L::TAPS];
    if constexpr (ADEEP) {
      static_assert(!BRun<L, B1>::use, "deep A prefetch is written for the stride-1 layers");
      float a[4][L::TAPS];
      auto load_a = [&](float (&av)[L::TAPS], int cb) {
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap) av[tap] = ap[(cb * L::TAPS + tap) * 64];
      };
      // B fragments one K-step ahead, two register sets, order pinned: the reads of (cb, tap + 1) — or of (cb + 1, 0) —
      // are issued before the MFMAs of (cb, tap).  Left to itself hipcc issues most reads of a block right in front of
      // their MFMAs and waits out the LDS round trip six times per block (tools/micro/micro_u0same.hip: 14.2 -> 12.1 us
      // per pass of up0.same).
      float bA[L::NB], bB[L::NB];
      auto bptr = [&](int cb) { return (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2; };
      auto load_b = [&](float (&bv)[L::NB], const float* bp, int tap) {
#pragma unroll
        for (int j = 0; j < L::NB; ++j) bv[j] = bp[j * 16 * L::SN + tap];
      };
      // one channel block; EVEN: its tap 0 sits in bA (the sets alternate per K-step: block cb starts in set (cb * TAPS) & 1)
      auto mac = [&](const float (&av)[L::TAPS], int cb, auto even) {
        const float* bp = bptr(cb);
        const float* bn = bptr(cb + 1 < L::CB ? cb + 1 : cb);
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap) {
          const bool cur_is_a = ((tap & 1) == 0) == decltype(even)::value;
          if (tap + 1 < L::TAPS) {
            if (cur_is_a) load_b(bB, bp, tap + 1); else load_b(bA, bp, tap + 1);
          } else {
            if (cur_is_a) load_b(bB, bn, 0); else load_b(bA, bn, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < L::NB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], cur_is_a ? bA[j] : bB[j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      load_a(a[0], 0);
      load_b(bA, bptr(0), 0);
      if (L::CB > 1) load_a(a[1], 1);
      if (L::CB > 2) load_a(a[2], 2);
#pragma unroll 1
      for (int cb = 0; cb < L::CB; cb += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cb + u < L::CB) {
            if (cb + u + 3 < L::CB) load_a(a[(u + 3) & 3], cb + u + 3);
            if (((u * L::TAPS) & 1) == 0) {
              mac(a[u], cb + u, std::true_type{});
            } else {
              mac(a[u], cb + u, std::false_type{});
            }
          }
        }
      }
    } else if constexpr (BDB) {
      float b0[L::TAPS][L::NB], b1[L::TAPS][L::NB];
      auto load_ab = [&](float (&av)[L::TAPS], float (&bv)[L::TAPS][L::NB], int cb) {
        const float* bp = (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2;
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap) av[tap] = ap[(cb * L::TAPS + tap) * 64];
        if constexpr (BRun<L, B1>::use) {
          static_assert(BRun<L, B1>::OFF == BRun<L, B2>::OFF, "both images of a strided layer share the tap alignment");
#pragma unroll
          for (int j = 0; j < L::NB; ++j) {
            BRun<L, B1> run;
            run.load(bp + j * 16 * L::SN);
#pragma unroll
            for (int tap = 0; tap < L::TAPS; ++tap) bv[tap][j] = run.tap(tap);
          }
        } else {
#pragma unroll
          for (int tap = 0; tap < L::TAPS; ++tap)
#pragma unroll
            for (int j = 0; j < L::NB; ++j) bv[tap][j] = bp[j * 16 * L::SN + tap];
        }
      };
      auto mac = [&](const float (&av)[L::TAPS], const float (&bv)[L::TAPS][L::NB]) {
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap)
#pragma unroll
          for (int j = 0; j < L::NB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], bv[tap][j], acc[j], 0, 0, 0);
      };
      load_ab(a0, b0, 0);
#pragma unroll 1
      for (int cb = 0; cb < L::CB; cb += 2) {
        if (cb + 1 < L::CB) load_ab(a1, b1, cb + 1);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        mac(a0, b0);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        if (cb + 1 < L::CB) {
          if (cb + 2 < L::CB) load_ab(a0, b0, cb + 2);
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
          mac(a1, b1);
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // wide layers (NB = 6): only the weight fragments are fetched a block ahead; the B fragments come out of
      // LDS tap by tap right before their MFMAs, so that the 24 accumulators + operands fit the 128 registers
      // a 1024-thread workgroup leaves each wave (the double-buffered form spilled 24 dwords per lane)
      auto load_a = [&](float (&av)[L::TAPS], int cb) {
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap) av[tap] = ap[(cb * L::TAPS + tap) * 64];
      };
      auto mac = [&](const float (&av)[L::TAPS], int cb) {
        const float* bp = (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2;
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap) {
          float bv[L::NB];
#pragma unroll
          for (int j = 0; j < L::NB; ++j) bv[j] = bp[j * 16 * L::SN + tap];
#pragma unroll
          for (int j = 0; j < L::NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], bv[j], acc[j], 0, 0, 0);
        }
      };
      load_a(a0, 0);
#pragma unroll 1
      for (int cb = 0; cb < L::CB; cb += 2) {
        if (cb + 1 < L::CB) load_a(a1, cb + 1);
        mac(a0, cb);
        if (cb + 1 < L::CB) {
          if (cb + 2 < L::CB) load_a(a0, cb + 2);
          mac(a1, cb + 1);
        }
      }
    }
    lds_epilogue<L>(acc, biasv, mt, colb, g, n, store);
  }
}

// ---- weights as 16-byte loads ---------------------------------------------------------------------------------------
// Every CU pulls the same weights out of L2.  With one dword per lane per fragment (afrag layout) the weight-heavy layers
// of the PhaseNet core get 16 B/clk/CU out of that path; the same bytes as one dwordx4 per lane stream at 52-56 B/clk/CU
// (tools/micro/micro_stream.hip: 256 CUs reading one 606 KB buffer).  afrag4 is the same A operand with the K-steps
// s = cb * TAPS + tap grouped in fours: [MT][CB * TAPS / 4][64 lanes][4].  A "superblock" is four channel blocks
// (4 * TAPS K-steps = TAPS loads per lane); the loads of superblock sb + 1 are issued before the MFMAs of sb, and the B
// fragments run one K-step ahead in two register sets under pinned order (as the deep path of conv_lds).
// PRE: superblock 0 of the wave's FIRST item is already on its way into qa (conv_lds_q4_request, called in front of the barrier
// that precedes this layer: the layer's first fragments otherwise make their trip to L2 with every wave of the workgroup waiting).
template <class L, int S1, int B1, int S2, int B2, class Store, bool PRE>
__device__ __forceinline__ void conv_lds_q4_impl(const float* in1, const float* in2, const float* __restrict__ afrag4,
                                                 const float* __restrict__ bias, const int cols, Store store, const int wave,
                                                 const int nwaves, const int lane, f32x4 (&qa)[L::TAPS]) {
  static_assert(L::CB % 4 == 0 && L::CB1 % 4 == 0, "superblocks of four channel blocks");
  constexpr int NQ = L::TAPS, SB_STEPS = 4 * L::TAPS, NSB = L::CB / 4;
  const int NT = (cols + 15) >> 4;
  const int NBLK = (NT + L::NB - 1) / L::NB;
  const int items = L::MT * NBLK;
  const int g = lane >> 4, n = lane & 15;
  for (int item = wave; item < items; item += nwaves) {
    const int mt = item % L::MT, nblk = item / L::MT;
    const int colb = nblk * L::NB * 16;
    f32x4 acc[L::NB];
#pragma unroll
    for (int j = 0; j < L::NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ap = reinterpret_cast<const f32x4*>(afrag4) + (long)mt * (L::CB * L::TAPS / 4) * 64 + lane;
    float biasv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = bias[mt * (16 / L::P) + (4 * g) / L::P + r / L::P];
    const float* bp1 = in1 + g * S1 + B1 + (colb + n) * L::SN + L::IN_OFF;
    const float* bp2 = in2 + g * S2 + B2 + (colb + n) * L::SN + L::IN_OFF;
    auto bptr = [&](int cb) { return (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2; };
    auto load_q = [&](f32x4 (&q)[NQ], int sb) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = ap[(sb * NQ + k) * 64];
    };
    auto load_b = [&](float (&bv)[L::NB], const float* bp, int tap) {
#pragma unroll
      for (int j = 0; j < L::NB; ++j) bv[j] = bp[j * 16 * L::SN + tap];
    };
    float bA[L::NB], bB[L::NB];
    auto superblock = [&](const f32x4 (&q)[NQ], int sb) {
      const float* b0 = bptr(4 * sb);                         // the four channel blocks of a superblock lie in one image
      const float* bnext = bptr(4 * (sb + 1 < NSB ? sb + 1 : sb));
      constexpr int CBS = 4 * ((L::CB1 > 0) ? S1 : S2);       // row step between channel blocks (S1 == S2 asserted below)
#pragma unroll
      for (int s = 0; s < SB_STEPS; ++s) {
        const int cbo = s / L::TAPS, tap = s - cbo * L::TAPS;
        if (s + 1 < SB_STEPS) {
          const int cbo1 = (s + 1) / L::TAPS, tap1 = (s + 1) - cbo1 * L::TAPS;
          if (s & 1) load_b(bA, b0 + cbo1 * CBS, tap1); else load_b(bB, b0 + cbo1 * CBS, tap1);
        } else {
          if (s & 1) load_b(bA, bnext, 0); else load_b(bB, bnext, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < L::NB; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[s >> 2][s & 3], (s & 1) ? bB[j] : bA[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        (void)cbo;
        (void)tap;
      }
    };
    static_assert(L::CIN2 == 0 || S1 == S2, "both images share the row stride");
    static_assert(SB_STEPS % 2 == 0, "the B sets alternate per K-step");
    f32x4 qb[NQ];
    if (!(PRE && item == wave)) load_q(qa, 0);  // (uniform)
    load_b(bA, bptr(0), 0);
#pragma unroll 1
    for (int sb = 0; sb < NSB; sb += 2) {
      if (sb + 1 < NSB) load_q(qb, sb + 1);
      superblock(qa, sb);
      if (sb + 1 < NSB) {
        if (sb + 2 < NSB) load_q(qa, sb + 2);
        superblock(qb, sb + 1);
      }
    }
    lds_epilogue<L>(acc, biasv, mt, colb, g, n, store);
  }
}
template <class L, int S1, int B1, int S2, int B2, class Store>
__device__ __forceinline__ void conv_lds_q4(const float* in1, const float* in2, const float* __restrict__ afrag4,
                                            const float* __restrict__ bias, const int cols, Store store, const int wave,
                                            const int nwaves, const int lane) {
  f32x4 qa[L::TAPS];
  conv_lds_q4_impl<L, S1, B1, S2, B2, Store, false>(in1, in2, afrag4, bias, cols, store, wave, nwaves, lane, qa);
}
// superblock 0 of the wave's first item of layer L, requested ahead (same item order as conv_lds_q4: item = wave)
template <class L>
__device__ __forceinline__ void conv_lds_q4_request(const float* __restrict__ afrag4, const int cols, const int wave, const int lane,
                                                    f32x4 (&qa)[L::TAPS]) {
  const int NT = (cols + 15) >> 4, NBLK = (NT + L::NB - 1) / L::NB, items = L::MT * NBLK;
  if (wave < items) {
    const f32x4* ap = reinterpret_cast<const f32x4*>(afrag4) + (long)(wave % L::MT) * (L::CB * L::TAPS / 4) * 64 + lane;
#pragma unroll
    for (int k = 0; k < L::TAPS; ++k) qa[k] = ap[k * 64];
  }
}
template <class L, int S1, int B1, int S2, int B2, class Store>
__device__ __forceinline__ void conv_lds_q4_requested(const float* in1, const float* in2, const float* __restrict__ afrag4,
                                                      const float* __restrict__ bias, const int cols, Store store, const int wave,
                                                      const int nwaves, const int lane, f32x4 (&qa)[L::TAPS]) {
  conv_lds_q4_impl<L, S1, B1, S2, B2, Store, true>(in1, in2, afrag4, bias, cols, store, wave, nwaves, lane, qa);
}

// Store functor: LDS image with a valid range [lo, hi) in the caller's local coordinates and a
// "signal" range [sig_lo, sig_hi): positions inside the image but outside the signal are written
// as zero (they are the next layer's zero padding, not activations of zero-padded input).
template <int S, int B>
struct ImageStore {
  float* img;
  int lo, hi;          // columns of the image this layer may write (local coords)
  int sig_lo, sig_hi;  // local coords whose global position lies inside the signal [0, L)
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if (t >= lo && t < hi) img[co * S + B + t] = (t >= sig_lo && t < sig_hi) ? v : 0.f;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const {
    return t0 >= lo && t1 < hi && t0 >= sig_lo && t1 < sig_hi;
  }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + B + t] = v; }
};

template <int C, int S>
__device__ __forceinline__ void zero_image(float* img, int tid, int nthreads) {
  for (int i = tid; i < C * S / 4; i += nthreads) reinterpret_cast<float4*>(img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace vp

namespace vp {

// ---- register-resident A variant -----------------------------------------------------------------
// For short-K layers run many times by one (persistent) workgroup the A fragments of the wave's
// m-tile are loaded ONCE into registers; the K loop is fully unrolled and only touches LDS.
template <class L>
__device__ __forceinline__ void load_areg(const float* __restrict__ afrag, const int mt, const int lane,
                                          float (&areg)[L::CB * L::TAPS]) {
  const float* ap = afrag + (long)mt * L::CB * L::TAPS * 64 + lane;
#pragma unroll
  for (int k = 0; k < L::CB * L::TAPS; ++k) areg[k] = ap[k * 64];
}
// the same from the regrouped operand (regroup_afrag4): one 16-byte load per four K-steps
template <class L>
__device__ __forceinline__ void load_areg4(const float* __restrict__ afrag4, const int mt, const int lane,
                                           float (&areg)[L::CB * L::TAPS]) {
  static_assert((L::CB * L::TAPS) % 4 == 0, "K-steps in groups of four");
  const f32x4* ap = reinterpret_cast<const f32x4*>(afrag4) + (long)mt * (L::CB * L::TAPS / 4) * 64 + lane;
#pragma unroll
  for (int k = 0; k < L::CB * L::TAPS / 4; ++k) {
    const f32x4 q = ap[k * 64];
    areg[4 * k] = q.x, areg[4 * k + 1] = q.y, areg[4 * k + 2] = q.z, areg[4 * k + 3] = q.w;
  }
}
template <class L>
__device__ __forceinline__ void load_biasreg(const float* __restrict__ bias, const int mt, const int lane,
                                             float (&biasv)[4]) {
  const int g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) biasv[r] = bias[mt * (16 / L::P) + (4 * g) / L::P + r / L::P];
}

// Blocks nblk = first_block, first_block + block_step, ... of m-tile `mt` (the caller maps waves to
// (mt, block) pairs; all blocks of a wave share areg / biasv).
template <class L, int S1, int B1, int S2, int B2, class Store>
__device__ __forceinline__ void conv_lds_areg(const float* in1, const float* in2, const float (&areg)[L::CB * L::TAPS],
                                              const float (&biasv)[4], const int mt, const int cols, Store store,
                                              const int first_block, const int block_step, const int lane) {
  const int NT = (cols + 15) >> 4;
  const int NBLK = (NT + L::NB - 1) / L::NB;
  const int g = lane >> 4, n = lane & 15;
  for (int nblk = first_block; nblk < NBLK; nblk += block_step) {
    const int colb = nblk * L::NB * 16;
    f32x4 acc[L::NB];
#pragma unroll
    for (int j = 0; j < L::NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bp1 = in1 + g * S1 + B1 + (colb + n) * L::SN + L::IN_OFF;
    const float* bp2 = in2 + g * S2 + B2 + (colb + n) * L::SN + L::IN_OFF;
    if constexpr (BRun<L, B1>::use) {
#pragma unroll
      for (int cb = 0; cb < L::CB; ++cb) {
        const float* bp = (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2;
        BRun<L, B1> run[L::NB];
#pragma unroll
        for (int j = 0; j < L::NB; ++j) run[j].load(bp + j * 16 * L::SN);
#pragma unroll
        for (int tap = 0; tap < L::TAPS; ++tap)
#pragma unroll
          for (int j = 0; j < L::NB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[cb * L::TAPS + tap], run[j].tap(tap), acc[j], 0, 0, 0);
      }
    } else {
      // K-steps (cb, tap) flattened; the B fragments of step s + 1 are read before the MFMAs of step s (order pinned,
      // two register sets) — see the deep path of conv_lds
      constexpr int STEPS = L::CB * L::TAPS;
      float bA[L::NB], bB[L::NB];
      auto load_b = [&](float (&bv)[L::NB], int s) {
        const int cb = s / L::TAPS, tap = s - cb * L::TAPS;
        const float* bp = (cb < L::CB1) ? bp1 + cb * 4 * S1 : bp2 + (cb - L::CB1) * 4 * S2;
#pragma unroll
        for (int j = 0; j < L::NB; ++j) bv[j] = bp[j * 16 * L::SN + tap];
      };
      load_b(bA, 0);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (s + 1 < STEPS) {
          if (s & 1) load_b(bA, s + 1); else load_b(bB, s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < L::NB; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s], (s & 1) ? bB[j] : bA[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_epilogue<L>(acc, biasv, mt, colb, g, n, store);
  }
}

}  // namespace vp
