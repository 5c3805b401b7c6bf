// One LSTM direction (H = 16, T = 47) on ONE wavefront: what a recurrent step of eqt_mid_kernel costs, per form of the
// step.  Lane l = 4 * unit + gate (i, f, g, o) owns gate row (gate, unit) of W_hh (16 weights as 8 register pairs); gx[t][l] is
// the input projection of the step, already scaled for v_exp_f32 (rows i, f, o by -log2 e, row g by -2 log2 e).
//   0  the step as eqt_kernels.hip lstm_recur<GS, SCALED = true> has it (C++ with one DPP asm block)
//   1  the whole step as one hand-scheduled asm block: cell state kept in units of 2 log2 e (no multiply in front of the
//      second exponential), gate activation as one fma with per-lane constants (no select), h = o - 2 o r as one fma
// Every form runs `reps` x 47 steps per wavefront, one wavefront per workgroup, 256 workgroups; prints cycles per step and
// the distance of its h sequence to a double-precision host reference.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/micro_lstm.hip -o /tmp/micro_lstm && /tmp/micro_lstm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int T = 47, H = 16, GS = 65;

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- form 0: the kernel's step ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void recur0(const float* gx, const f32x2 (&whh)[H / 2], float* hout, const int hs) {
  const int lane = threadIdx.x & 63;
  const bool is_g = (lane & 3) == 2;
  float h = 0.f, c = 0.f;
  float gnext = gx[lane];
  for (int s = 0; s < T; ++s) {
    f32x2 ga = {gnext, 0.f}, gb = {0.f, 0.f};
    {
      const int sn = s + 1 < T ? s + 1 : s;
      gnext = gx[sn * GS + lane];
    }
#pragma unroll
    for (int j = 0; j < H / 2; j += 2) {
      const f32x2 ha = {lane_bcast(h, 8 * j), lane_bcast(h, 8 * j + 4)};
      const f32x2 hb = {lane_bcast(h, 8 * j + 8), lane_bcast(h, 8 * j + 12)};
      ga = __builtin_elementwise_fma(whh[j], ha, ga);
      gb = __builtin_elementwise_fma(whh[j + 1], hb, gb);
    }
    const float g = (ga.x + ga.y) + (gb.x + gb.y);
    const float sg = rcp_fast(1.f + __builtin_amdgcn_exp2f(g));
    const float act = is_g ? fmaf(sg, 2.f, -1.f) : sg;
    float cn, og;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %3 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %3, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(cn), "=&v"(og)
        : "v"(c), "v"(act));
    c = cn;
    h = og * fmaf(rcp_fast(__builtin_amdgcn_exp2f(c * 2.885390082f) + 1.f), -2.f, 1.f);
    if ((lane & 3) == 0) hout[(lane >> 2) * hs + s] = h;
  }
}

// ---- form 1: one asm block per step ------------------------------------------------------------------------------------------
// State: h (every lane of a quad holds its unit's h), C = c * 2 log2 e.  Per-lane constants: act = sg * A + B with (A, B) =
// (2 K, -K) in the g lane (K = 2 log2 e: the g gate arrives as tanh * K, so that i * g accumulates straight into C) and (1, 0)
// elsewhere.  The step:
//   16 v_readlane (h of the 16 units -> 8 SGPR pairs) interleaved with the 8 packed FMAs of the two accumulator chains,
//   pk_add + add, exp2, +1, rcp, fma (A, B), [2 wait states] G = act(lane 2), O = act(lane 3) by DPP, C' = i * G (DPP) +
//   f * C (DPP), exp2, +1, rcp, h = O - 2 O r as one fma with O2 = -2 O prepared beside the exponential.
// The LDS traffic of the step (next gx row in, h out under a quad-lane-0 exec mask) is issued from inside the block; the
// caller waits for the gx value before the next step (one s_waitcnt lgkmcnt(0), which the h store of the step shares).
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_float*)p; }  // LDS byte offset

template <int V>
__device__ __forceinline__ void recur1(const float* gx, const f32x2 (&whh)[H / 2], float* hout, const int hs) {
  const int lane = threadIdx.x & 63;
  const bool is_g = (lane & 3) == 2;
  constexpr float K = 2.885390082f;
  const float A = is_g ? 2.f * K : 1.f, Bc = is_g ? -K : 0.f;
  float h = 0.f, C = 0.f;
  f32x2 g0 = {gx[lane], 0.f};
  unsigned gaddr = lds_addr(gx + GS + lane);          // the next step's row (the read behind the last row is harmless: LDS that exists)
  unsigned haddr = lds_addr(hout + (lane >> 2) * hs);
  const unsigned long long quad0 = 0x1111111111111111ull;
  for (int s = 0; s < T; ++s) {
    f32x2 acc1, t2;
    float t0, t1, O, O2, gn;
    if constexpr (V == 1) {
    asm volatile(
          "v_readlane_b32 s90, %[h], 0\n\t"
          "v_readlane_b32 s91, %[h], 4\n\t"
          "v_readlane_b32 s92, %[h], 8\n\t"
          "v_readlane_b32 s93, %[h], 12\n\t"
          "ds_read_b32 %[gn], %[gaddr]\n\t"
          "v_pk_fma_f32 %[g0], %[w0], s[90:91], %[g0]\n\t"
          "v_readlane_b32 s94, %[h], 16\n\t"
          "v_readlane_b32 s95, %[h], 20\n\t"
          "v_pk_mul_f32 %[acc1], %[w1], s[92:93]\n\t"
          "v_readlane_b32 s90, %[h], 24\n\t"
          "v_readlane_b32 s91, %[h], 28\n\t"
          "v_pk_fma_f32 %[g0], %[w2], s[94:95], %[g0]\n\t"
          "v_readlane_b32 s92, %[h], 32\n\t"
          "v_readlane_b32 s93, %[h], 36\n\t"
          "v_pk_fma_f32 %[acc1], %[w3], s[90:91], %[acc1]\n\t"
          "v_readlane_b32 s94, %[h], 40\n\t"
          "v_readlane_b32 s95, %[h], 44\n\t"
          "v_pk_fma_f32 %[g0], %[w4], s[92:93], %[g0]\n\t"
          "v_readlane_b32 s90, %[h], 48\n\t"
          "v_readlane_b32 s91, %[h], 52\n\t"
          "v_pk_fma_f32 %[acc1], %[w5], s[94:95], %[acc1]\n\t"
          "v_readlane_b32 s92, %[h], 56\n\t"
          "v_readlane_b32 s93, %[h], 60\n\t"
          "v_pk_fma_f32 %[g0], %[w6], s[90:91], %[g0]\n\t"
          "v_pk_fma_f32 %[acc1], %[w7], s[92:93], %[acc1]\n\t"
          "v_add_u32 %[gaddr], %[gstep], %[gaddr]\n\t"
          "v_pk_add_f32 %[t2], %[g0], %[acc1]"
          : [g0] "+v"(g0), [acc1] "=&v"(acc1), [t2] "=&v"(t2), [gn] "=&v"(gn), [gaddr] "+v"(gaddr)
          : [h] "v"(h), [w0] "v"(whh[0]), [w1] "v"(whh[1]), [w2] "v"(whh[2]), [w3] "v"(whh[3]), [w4] "v"(whh[4]), [w5] "v"(whh[5]),
            [w6] "v"(whh[6]), [w7] "v"(whh[7]), [gstep] "s"(GS * 4)
          : "s90", "s91", "s92", "s93", "s94", "s95", "memory");
    } else if constexpr (V == 2 || V == 3) {  // all sixteen broadcasts first, then the eight packed FMAs
      asm volatile(
          "v_readlane_b32 s80, %[h], 0\n\t"
          "v_readlane_b32 s81, %[h], 4\n\t"
          "v_readlane_b32 s82, %[h], 8\n\t"
          "v_readlane_b32 s83, %[h], 12\n\t"
          "v_readlane_b32 s84, %[h], 16\n\t"
          "v_readlane_b32 s85, %[h], 20\n\t"
          "v_readlane_b32 s86, %[h], 24\n\t"
          "v_readlane_b32 s87, %[h], 28\n\t"
          "v_readlane_b32 s88, %[h], 32\n\t"
          "v_readlane_b32 s89, %[h], 36\n\t"
          "v_readlane_b32 s90, %[h], 40\n\t"
          "v_readlane_b32 s91, %[h], 44\n\t"
          "v_readlane_b32 s92, %[h], 48\n\t"
          "v_readlane_b32 s93, %[h], 52\n\t"
          "v_readlane_b32 s94, %[h], 56\n\t"
          "v_readlane_b32 s95, %[h], 60\n\t"
          "ds_read_b32 %[gn], %[gaddr]\n\t"
          "v_pk_fma_f32 %[g0], %[w0], s[80:81], %[g0]\n\t"
          "v_pk_mul_f32 %[acc1], %[w1], s[82:83]\n\t"
          "v_pk_fma_f32 %[g0], %[w2], s[84:85], %[g0]\n\t"
          "v_pk_fma_f32 %[acc1], %[w3], s[86:87], %[acc1]\n\t"
          "v_pk_fma_f32 %[g0], %[w4], s[88:89], %[g0]\n\t"
          "v_pk_fma_f32 %[acc1], %[w5], s[90:91], %[acc1]\n\t"
          "v_pk_fma_f32 %[g0], %[w6], s[92:93], %[g0]\n\t"
          "v_pk_fma_f32 %[acc1], %[w7], s[94:95], %[acc1]\n\t"
          "v_add_u32 %[gaddr], %[gstep], %[gaddr]\n\t"
          "v_pk_add_f32 %[t2], %[g0], %[acc1]"
          : [g0] "+v"(g0), [acc1] "=&v"(acc1), [t2] "=&v"(t2), [gn] "=&v"(gn), [gaddr] "+v"(gaddr)
          : [h] "v"(h), [w0] "v"(whh[0]), [w1] "v"(whh[1]), [w2] "v"(whh[2]), [w3] "v"(whh[3]), [w4] "v"(whh[4]), [w5] "v"(whh[5]),
            [w6] "v"(whh[6]), [w7] "v"(whh[7]), [gstep] "s"(GS * 4)
          : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "memory");
    } else {  // V == 4: diagnostic, no matrix-vector product (the activation chain alone)
      asm volatile(
          "ds_read_b32 %[gn], %[gaddr]\n\t"
          "v_add_u32 %[gaddr], %[gstep], %[gaddr]\n\t"
          "v_pk_mul_f32 %[t2], %[g0], %[w0]"
          : [t2] "=&v"(t2), [gn] "=&v"(gn), [gaddr] "+v"(gaddr)
          : [g0] "v"(g0), [w0] "v"(whh[0]), [gstep] "s"(GS * 4)
          : "memory");
      t2.x += h * 0.25f;
    }
    t0 = t2.x + t2.y;
    if constexpr (V == 3) {  // diagnostic: the matrix-vector product alone
      h = t0 * 0.03f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      g0 = f32x2{gn, 0.f};
      continue;
    }
    asm volatile(
        "v_exp_f32 %[t0], %[t0]\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %[t0], 1.0, %[t0]\n\t"
        "v_rcp_f32 %[t0], %[t0]\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[t0], %[t0], %[A], %[B]\n\t"
        "s_nop 1\n\t"
        "v_mov_b32_dpp %[t1], %[t0] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[O], %[t0] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %[t1], %[t0], %[t1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[t1], %[t0], %[C] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32 %[O2], -2.0, %[O]\n\t"
        "v_exp_f32 %[t0], %[t1]\n\t"
        "v_mov_b32 %[C], %[t1]\n\t"
        "v_add_f32 %[t0], 1.0, %[t0]\n\t"
        "v_rcp_f32 %[t0], %[t0]\n\t"
        "s_nop 0\n\t"
        "v_fma_f32 %[h], %[t0], %[O2], %[O]\n\t"
        "s_mov_b64 exec, %[quad0]\n\t"
        "ds_write_b32 %[haddr], %[h]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_add_u32 %[haddr], 4, %[haddr]\n\t"
        "s_waitcnt lgkmcnt(1)"  // LDS operations return in order: the gx row (older) has arrived, the h store may still be on its way
        : [h] "=&v"(h), [C] "+v"(C), [t0] "+v"(t0), [t1] "=&v"(t1), [O] "=&v"(O), [O2] "=&v"(O2), [haddr] "+v"(haddr)
        : [A] "v"(A), [B] "v"(Bc), [quad0] "s"(quad0)
        : "memory");
    g0 = f32x2{gn, 0.f};
  }
}

// ---- form 5: lane = 16 * gate + unit; W_hh h as 16 v_fmac_f32_dpp row_newbcast (no SGPR round trip) ----------------------------
// h[u] is kept in lane (gate, u) of ALL four rows, so `row_newbcast:u` hands unit u's h to every lane of a row.  The four
// gates of a unit sit in four different rows: v_permlane32_swap + 2 x v_permlane16_swap bring (i, f, g, o) of unit u to every
// lane (*, u), and all four rows then update c and h redundantly.  gx here: [t][GS] indexed 16 * gate + unit.
#define LSTM5_MATVEC \
  "v_fmac_f32_dpp %[g0], %[h], %[w0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
  "v_mul_f32_dpp %[acc1], %[h], %[w1] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_mul_f32_dpp %[acc2], %[h], %[w2] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
  "v_mul_f32_dpp %[acc3], %[h], %[w3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w4] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w6] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w7] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w8] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w9] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w10] row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w11] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w12] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w13] row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w14] row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w15] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define LSTM5_MATVEC2 \
  "v_fmac_f32_dpp %[g0], %[h], %[w0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w1] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w2] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w3] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w4] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w5] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w6] row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w7] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w8] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w9] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w10] row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w11] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[g0], %[h], %[w12] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc1], %[h], %[w13] row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc2], %[h], %[w14] row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %[acc3], %[h], %[w15] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define LSTM5_NOMAT "v_mov_b32 %[acc1], 0\n\tv_mov_b32 %[acc2], 0\n\tv_mov_b32 %[acc3], 0\n\t"
#define LSTM5_STEP(MV) \
    asm volatile( \
        "ds_read_b32 %[gn], %[gaddr]\n\t" \
        "s_nop 0\n\t" \
        MV \
        "v_add_u32 %[gaddr], %[gstep], %[gaddr]\n\t" \
        "v_add_f32 %[g0], %[g0], %[acc1]\n\t" \
        "v_add_f32 %[acc2], %[acc2], %[acc3]\n\t" \
        "v_add_f32 %[t0], %[g0], %[acc2]\n\t" \
        "v_exp_f32 %[t0], %[t0]\n\t" \
        "s_nop 0\n\t" \
        "v_add_f32 %[t0], 1.0, %[t0]\n\t" \
        "v_rcp_f32 %[t0], %[t0]\n\t" \
        "s_nop 0\n\t" \
        "v_fma_f32 %[b], %[t0], %[A], %[B]\n\t" \
        "v_fma_f32 %[t0], %[t0], %[A], %[B]\n\t" \
        "s_nop 1\n\t" \
        "v_permlane32_swap_b32 %[t0], %[b]\n\t" \
        "v_mov_b32 %[a2], %[t0]\n\t" \
        "v_mov_b32 %[b2], %[b]\n\t" \
        "s_nop 0\n\t" \
        "v_permlane16_swap_b32 %[t0], %[a2]\n\t" \
        "v_permlane16_swap_b32 %[b], %[b2]\n\t" \
        "v_mul_f32 %[t1], %[t0], %[b]\n\t" \
        "v_fmac_f32 %[t1], %[a2], %[C]\n\t" \
        "v_mul_f32 %[a2], -2.0, %[b2]\n\t" \
        "v_exp_f32 %[t0], %[t1]\n\t" \
        "v_mov_b32 %[C], %[t1]\n\t" \
        "v_add_f32 %[t0], 1.0, %[t0]\n\t" \
        "v_rcp_f32 %[t0], %[t0]\n\t" \
        "s_nop 0\n\t" \
        "v_fma_f32 %[h], %[t0], %[a2], %[b2]\n\t" \
        "s_mov_b64 exec, %[row0]\n\t" \
        "ds_write_b32 %[haddr], %[h]\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "v_add_u32 %[haddr], 4, %[haddr]\n\t" \
        "s_waitcnt lgkmcnt(1)" \
        : [h] "+v"(h), [C] "+v"(C), [g0] "+v"(g0), [acc1] "=&v"(acc1), [acc2] "=&v"(acc2), [acc3] "=&v"(acc3), [t0] "=&v"(t0), \
          [t1] "=&v"(t1), [a2] "=&v"(a2), [b] "=&v"(b), [b2] "=&v"(b2), [gn] "=&v"(gn), [gaddr] "+v"(gaddr), [haddr] "+v"(haddr) \
        : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), \
          [w7] "v"(w[7]), [w8] "v"(w[8]), [w9] "v"(w[9]), [w10] "v"(w[10]), [w11] "v"(w[11]), [w12] "v"(w[12]), [w13] "v"(w[13]), \
          [w14] "v"(w[14]), [w15] "v"(w[15]), [A] "v"(A), [B] "v"(Bc), [gstep] "s"(GS * 4), [row0] "s"(row0) \
        : "memory");
template <int DIAG>
__device__ __forceinline__ void recur5(const float* gx, const float (&w)[H], float* hout, const int hs) {
  const int lane = threadIdx.x & 63;
  const bool is_g = (lane >> 4) == 2;
  constexpr float K = 2.885390082f;
  const float A = is_g ? 2.f * K : 1.f, Bc = is_g ? -K : 0.f;
  float h = 0.f, C = 0.f;
  float g0 = gx[lane];
  unsigned gaddr = lds_addr(gx + GS + lane);
  unsigned haddr = lds_addr(hout + (lane & 15) * hs);
  const unsigned long long row0 = 0xffffull;
  for (int s = 0; s < T; ++s) {
    float acc1, acc2, acc3, t0, t1, a2, b, b2, gn;
    if constexpr (DIAG == 0) {
      LSTM5_STEP(LSTM5_MATVEC)
    } else if constexpr (DIAG == 1) {
      LSTM5_STEP(LSTM5_NOMAT)
    } else {
      LSTM5_STEP(LSTM5_MATVEC LSTM5_MATVEC2)
    }
    g0 = gn;
  }
}


template <int FORM, int DIAG5 = 0>
__global__ __launch_bounds__(64) void k(const float* __restrict__ gx_g, const float* __restrict__ whh_g, float* out, int reps,
                                        unsigned long long* clk) {
  __shared__ float gx[T * GS + 64];
  __shared__ float hout[H * 48];
  const int lane = threadIdx.x;
  for (int i = lane; i < T * GS; i += 64) gx[i] = gx_g[i];
  if (FORM == 5) {  // gx rows re-ordered lane = 16 * gate + unit (the quad layout has lane = 4 * unit + gate)
    __syncthreads();
    float tmp[T];
    for (int t = 0; t < T; ++t) tmp[t] = gx[t * GS + 4 * (lane & 15) + (lane >> 4)];
    __syncthreads();
    for (int t = 0; t < T; ++t) gx[t * GS + lane] = tmp[t];
    __syncthreads();
    float w[H];
    const int r5 = (lane >> 4) * 16 + (lane & 15);
    const float sc5 = (lane >> 4) == 2 ? -2.885390082f : -1.442695041f;
    for (int j = 0; j < H; ++j) w[j] = whh_g[r5 * H + j] * sc5;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
      recur5<DIAG5>(gx, w, hout, 48);
      __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0) {
      for (int i = lane; i < H * 48; i += 64) out[i] = hout[i];
      if (lane == 0) clk[0] = c1 - c0;
    }
    return;
  }
  const int row = (lane & 3) * 16 + (lane >> 2);
  f32x2 whh[H / 2];
  const float sc = (lane & 3) == 2 ? -2.885390082f : -1.442695041f;
  for (int j = 0; j < H / 2; ++j) whh[j] = f32x2{whh_g[row * H + 2 * j] * sc, whh_g[row * H + 2 * j + 1] * sc};
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if constexpr (FORM == 0) recur0(gx, whh, hout, 48);
    else if constexpr (FORM != 5) recur1<FORM>(gx, whh, hout, 48);
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0) {
    for (int i = lane; i < H * 48; i += 64) out[i] = hout[i];
    if (lane == 0) clk[0] = c1 - c0;
  }
}

int main() {
  std::mt19937 rng(5);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> gx(T * GS, 0.f), pre(T * 64), whh(64 * H);
  for (auto& w : whh) w = 0.35f * nd(rng);
  for (int t = 0; t < T; ++t)
    for (int row = 0; row < 64; ++row) pre[t * 64 + row] = 1.2f * nd(rng);  // torch row order: gate * 16 + unit
  for (int t = 0; t < T; ++t)
    for (int l = 0; l < 64; ++l) {
      const int gate = l & 3, unit = l >> 2;
      gx[t * GS + l] = pre[t * 64 + gate * 16 + unit] * (gate == 2 ? -2.885390082f : -1.442695041f);
    }
  // host reference in double
  std::vector<double> href(H * T), h(H, 0.0), c(H, 0.0);
  for (int t = 0; t < T; ++t) {
    std::vector<double> hn(H);
    for (int u = 0; u < H; ++u) {
      double g[4];
      for (int q = 0; q < 4; ++q) {
        double a = pre[t * 64 + q * 16 + u];
        for (int j = 0; j < H; ++j) a += (double)whh[(q * 16 + u) * H + j] * h[j];
        g[q] = a;
      }
      const double i = 1 / (1 + std::exp(-g[0])), f = 1 / (1 + std::exp(-g[1])), gg = std::tanh(g[2]), o = 1 / (1 + std::exp(-g[3]));
      c[u] = f * c[u] + i * gg;
      hn[u] = o * std::tanh(c[u]);
      href[u * T + t] = hn[u];
    }
    h = hn;
  }
  float *d_gx, *d_whh, *d_out;
  unsigned long long* d_clk;
  hipMalloc(&d_gx, gx.size() * 4), hipMalloc(&d_whh, whh.size() * 4), hipMalloc(&d_out, H * 48 * 4), hipMalloc(&d_clk, 64);
  hipMemcpy(d_gx, gx.data(), gx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_whh, whh.data(), whh.size() * 4, hipMemcpyHostToDevice);
  const int reps = 400;
  for (int form = 0; form < 8; ++form) {
    for (int rep = 0; rep < 2; ++rep) {
      if (form == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 5) hipLaunchKernelGGL((k<5, 0>), dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else if (form == 6) hipLaunchKernelGGL((k<5, 1>), dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      else hipLaunchKernelGGL((k<5, 2>), dim3(256), dim3(64), 0, 0, d_gx, d_whh, d_out, reps, d_clk);
      hipDeviceSynchronize();
    }
    std::vector<float> out(H * 48);
    unsigned long long cyc = 0;
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&cyc, d_clk, 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (int u = 0; u < H; ++u)
      for (int t = 0; t < T; ++t) err = std::max(err, std::fabs(out[u * 48 + t] - href[u * T + t]));
    printf("form %d: %7.1f cycles per step   max |h - double reference| = %.3e\n", form, (double)cyc / (reps * T), err);
  }
  return 0;
}
