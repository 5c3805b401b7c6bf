// Direct convolution on the vector ALU for the 8-channel stride-1 level-0 layers of PhaseNet (inc, down0.same,
// up3.same; the strided and the transposed conv use every weight once per output and stay on the MFMA): these layers have too few channels for the 16-row MFMA tile (the
// polyphase forms pad 3 -> 4 input channels and 7 -> 8 / 11 taps, 1.1-1.6x the algorithmic MACs) and their
// short K loops are bound by LDS operand traffic, not by the matrix pipe.  Here every lane owns four
// consecutive output samples and all eight output channels: the input window of one channel (12 floats, three
// aligned ds_read_b128) lives in registers, the weights of a (channel, tap) come as four channel PAIRS out of
// SGPRs, and every multiply-add is a v_pk_fma_f32 over a channel pair with the input sample broadcast by
// op_sel — 224 algorithmic MACs per LDS dword read, no padding, no bank conflicts.
// Measured (tools/micro/micro_valu.hip, micro_peak.hip): 93-98 TFLOP/s algorithmic in a tiled loop; the
// packed-FMA issue peak is 125 TFLOP/s and it does NOT overlap with MFMA issue (the two share the SIMD).
#pragma once
#include "vp_common.h"

namespace vp {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned vector (dense output rows)

// Weights are read through the constant address space: a wave-uniform load from it is always a scalar load
// (s_load_dwordx2..16 into SGPRs).  Through a plain global pointer hipcc falls back to per-lane global_load as soon as
// the kernel has stored anything to memory before the load (it cannot prove the weights unclobbered).
typedef const f32x2 __attribute__((address_space(4))) * wptr_t;
__device__ __forceinline__ wptr_t as_weights(const f32x2* p) { return (wptr_t)(p); }
typedef const float __attribute__((address_space(4))) * sptr_t;
__device__ __forceinline__ sptr_t as_scalars(const float* p) { return (sptr_t)(p); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load and
// store (s_waitcnt vmcnt(0)), which defeats register prefetches that are meant to stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS image convention of the VALU kernels: [C][S] floats, local sample l of a row at column l + 4; S % 4 == 0.
//
// acc[c][r] += sum_ci sum_k w2[ci][k][c] * img[ci][t0 + r + k - 3]      (c = channel pair, r = 0..3, t0 % 4 == 0)
// w2: [CIN][7][4] channel-pair weights, wave-uniform address (scalar loads).
//
// One input channel per trip (fully unrolled, hipcc hoists every scalar load to the top and spills hundreds of SGPRs),
// software-pipelined by hand in two halves, because LDS and scalar loads share lgkmcnt and scalar loads return out of
// order, so every wait is a wait for everything in flight:
//   tap 0            <- waits for {window(ci), taps 0-3(ci)}, issued during the previous trip
//   issue taps 4-6(ci);  taps 1-3 cover the load
//   tap 4            <- waits for taps 4-6
//   issue window(ci+1), taps 0-3(ci+1);  taps 5-6 cover the loads
struct ValuWin {
  float v[12];
};
template <int S>
__device__ __forceinline__ void valu_load_win(ValuWin& w, const float* img, int ci, int t0) {
  const f32x4* p = reinterpret_cast<const f32x4*>(img + ci * S + t0);  // column t0 <-> local t0 - 4
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    const f32x4 q = p[v];
    w.v[4 * v] = q.x, w.v[4 * v + 1] = q.y, w.v[4 * v + 2] = q.z, w.v[4 * v + 3] = q.w;
  }
}
// taps [K0, K1) of the window; w holds NC channel-pair weights of each of the taps KB, KB + 1, ...
template <int K0, int K1, int KB, int NC, int NW>
__device__ __forceinline__ void valu_taps(const ValuWin& win, const f32x2 (&w)[NW], f32x2 (&acc)[NC][4]) {
#pragma unroll
  for (int k = K0; k < K1; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 x = {win.v[r + k + 1], win.v[r + k + 1]};
        acc[c][r] = __builtin_elementwise_fma(x, w[(k - KB) * NC + c], acc[c][r]);
      }
}
// NT taps x NC channel pairs starting at pair C0 of tap 0 of p ([tap][4] pairs in memory)
template <int NT, int NC, int C0>
__device__ __forceinline__ void valu_load_w(f32x2 (&w)[NT * NC], const wptr_t p) {
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c) w[k * NC + c] = p[k * 4 + C0 + c];
}

// NC = 4: all eight output channels; NC = 2, C0 in {0, 2}: channels 4 C0/2 .. (one half), so that a kernel can hand
// the first half to memory while it computes the second.
template <int CIN, int S, int NC = 4, int C0 = 0>
__device__ __forceinline__ void valu_conv7_r4(const float* img, const wptr_t w2, const int t0, f32x2 (&acc)[NC][4]) {
  ValuWin winA, winB;
  f32x2 waA[4 * NC], waB[4 * NC], wb[3 * NC];
  valu_load_win<S>(winA, img, 0, t0);
  valu_load_w<4, NC, C0>(waA, w2);
  // one trip on (win, wa); prefetches channel `nxt` into (winN, waN)
#define VALU_TRIP(win, wa, winN, waN, ci, nxt)                    \
  valu_taps<0, 1, 0, NC>(win, wa, acc);                           \
  __builtin_amdgcn_sched_barrier(0);                              \
  valu_load_w<3, NC, C0>(wb, w2 + (ci) * 28 + 16);                \
  __builtin_amdgcn_sched_barrier(0);                              \
  valu_taps<1, 4, 0, NC>(win, wa, acc);                           \
  __builtin_amdgcn_sched_barrier(0);                              \
  valu_taps<4, 5, 4, NC>(win, wb, acc);                           \
  __builtin_amdgcn_sched_barrier(0);                              \
  valu_load_win<S>(winN, img, nxt, t0);                           \
  valu_load_w<4, NC, C0>(waN, w2 + (nxt) * 28);                   \
  __builtin_amdgcn_sched_barrier(0);                              \
  valu_taps<5, 7, 4, NC>(win, wb, acc);                           \
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int ci = 0; ci < CIN; ci += 2) {
    const int n1 = (ci + 1 < CIN) ? ci + 1 : CIN - 1, n2 = (ci + 2 < CIN) ? ci + 2 : CIN - 1;  // clamped prefetch: harmless reload
    VALU_TRIP(winA, waA, winB, waB, ci, n1)
    if (ci + 1 < CIN) {
      VALU_TRIP(winB, waB, winA, waA, ci + 1, n2)
    }
  }
#undef VALU_TRIP
}

}  // namespace vp
