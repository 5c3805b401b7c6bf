// Device helpers shared by the EQTransformer bottleneck kernels (eqt_kernels.hip: teams of eight waves per window;
// eqt_mid4.hip: teams of four).  The includer defines, BEFORE this header, how a workgroup is cut into per-window teams:
//   MID_TID   thread index within the window's team      MID_NT   threads of a team      MID_TEAM   team index
// Everything below is written against those three only.
#pragma once
#include "eqt_kernels.h"
#include "prepost.h"
#include "conv_valu.h"
#if !defined(MID_TID) || !defined(MID_NT) || !defined(MID_TEAM)
#error "define MID_TID / MID_NT / MID_TEAM before including eqt_mid_parts.h"
#endif

// The two-window form shares its barriers, not its decisions: a vote (which form a window's attention scores take) is ONE
// workgroup-wide OR reduction in which every team sets its own bit and reads only that bit back, so what a window computes
// never depends on the window it shares a workgroup with (same barrier count on both teams, no extra LDS).
__device__ __forceinline__ bool team_vote_or(bool mine) {
  const int team = __builtin_amdgcn_readfirstlane(MID_TEAM);  // wave-uniform: the result steers scalar branches
  return __builtin_amdgcn_readfirstlane((__ockl_wgred_or_i32(mine ? (1 << team) : 0) >> team) & 1) != 0;
}

namespace vp {

namespace {

constexpr int T = EQT_T;

__device__ inline float wave_sum64(float v) { return wave_sum(v); }  // DPP reductions of prepost.h
__device__ inline float wave_max64(float v) { return wave_max(v); }
__device__ inline float lane_bcast(float v, int lane) {  // lane is a compile-time constant
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal: hipcc expands it to the full division sequence
// (v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup) — ten instructions on the critical path of every LSTM step
// and of every element of the attention loop.
__device__ inline float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
// recurrent gates: v_exp_f32 + v_rcp_f32 (abs error ~1e-7); 47 dependent steps make this the critical path
__device__ inline float sigmoid_fast(float x) { return rcp_fast(1.f + __expf(-x)); }
// tanh via one exp; abs error ~1e-7, saturates correctly at +-inf
__device__ inline float tanh_fast(float z) {
  const float t = __expf(2.f * z);
  return 1.f - 2.f * rcp_fast(t + 1.f);
}

// One LSTM direction on ONE wavefront.  Every lane owns one gate row (layout below): its W_ih / W_hh rows sit in
// registers.  h_{t-1} is broadcast with v_readlane, the four gates of a unit are gathered inside their quad, c/h are
// kept redundantly in the four lanes of a quad.  xs: LDS [T][CIN]; gx: LDS [T][64] per-lane
// scratch for the input projection; hout: LDS rows [16][hs].
// LSTM lane layout: lane = 4 * unit + gate (torch gate order i, f, g, o), i.e. lane l owns gate row
// (l & 3) * 16 + (l >> 2) of W_ih / W_hh / b.  The four gates of a unit sit in one quad, so gathering them is four DPP
// quad broadcasts instead of four ds_bpermute round trips through the LDS hardware on every one of the 47 dependent steps.
__device__ inline int lstm_row(int lane) { return (lane & 3) * 16 + (lane >> 2); }
template <int SEL>
__device__ inline float quad_bcast(float v) {  // value of lane SEL of the quad, in all four lanes
  const int x = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(x, x, SEL * 0x55, 0xF, 0xF, false));  // quad_perm [SEL,SEL,SEL,SEL]
}

// Input projection of one direction for the time steps t0, t0 + tstep, ... (any number of waves may share it).
template <int CIN>
__device__ void lstm_project(const float* xs, float* gx, const LstmWeights w, const int t0, const int tstep) {
  const int lane = MID_TID & 63, row = lstm_row(lane);
  float wih[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) wih[c] = w.w_ih[row * CIN + c];
  const float b = w.b[row];
  for (int t = t0; t < T; t += tstep) {
    float a0 = b, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; c += 4) {
      const float4 x = *reinterpret_cast<const float4*>(xs + t * CIN + c);  // same address in every lane
      a0 = fmaf(wih[c], x.x, a0);
      a1 = fmaf(wih[c + 1], x.y, a1);
      a2 = fmaf(wih[c + 2], x.z, a2);
      a3 = fmaf(wih[c + 3], x.w, a3);
    }
    gx[t * 64 + lane] = (a0 + a1) + (a2 + a3);
  }
}

// The 47 sequential steps of one direction on ONE wavefront (gx: its input projection, same lane layout).
// W_hh of the lane's gate row as the eight unit PAIRS the packed FMAs of the recurrence consume, in memory order:
// the registers of the 16-byte loads are used as they arrive.  (As sixteen scalars hipcc re-paired them with v_mov
// right behind the loads, i.e. waited for them at the point of issue -- which defeats requesting them a stage ahead.)
__device__ inline void lstm_load_whh(f32x2 (&whh)[EQT_H / 2], const LstmWeights w) {
  const int row = lstm_row(MID_TID & 63);
  const f32x2* p = reinterpret_cast<const f32x2*>(w.w_hh + row * EQT_H);
#pragma unroll
  for (int j = 0; j < EQT_H / 2; ++j) whh[j] = p[j];
}
// Scale of a gate row's pre-activation in the SCALED form of the recurrence: sigmoid(x) = 1 / (1 + 2^(-x log2 e)) and
// tanh(x) = 2 sigmoid(2x) - 1, so with W_hh, W_ih and b of the row multiplied by -log2(e) (i, f, o) or -2 log2(e) (g) the
// step feeds v_exp_f32 directly: three instructions fewer on the serial path of each of the 47 steps.
__device__ __forceinline__ float lstm_gate_scale(const int gate) { return gate == 2 ? -2.885390082f : -1.442695041f; }
// GS: row stride of gx (64, or 65 where the projection writes it with lane = time step).
// SCALED: gx and whh already carry lstm_gate_scale (eqt_mid_kernel); the gate combination then runs as four fused DPP
// instructions (v_mul_f32_dpp / v_fmac_f32_dpp read the quad's i and f gates in place).
template <int GS = 64, bool SCALED = false>
__device__ void lstm_recur(const float* gx, const f32x2 (&whh)[EQT_H / 2], const bool reverse, float* hout, const int hs) {
  const int lane = MID_TID & 63;
  const bool is_g = (lane & 3) == 2;
  float h = 0.f, c = 0.f;
  float gnext = gx[(reverse ? T - 1 : 0) * GS + lane];
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? T - 1 - s : s;
    f32x2 ga = {gnext, 0.f}, gb = {0.f, 0.f};  // partial sums over the units = 0, 1 | 2, 3 (mod 4)
    {  // the next step's input projection is on its way while this step computes (clamped: a harmless re-read at the end)
      const int sn = s + 1 < T ? s + 1 : s;
      gnext = gx[(reverse ? T - 1 - sn : sn) * GS + lane];
    }
#pragma unroll
    for (int j = 0; j < EQT_H / 2; j += 2) {  // h of unit u lives in the quad 4u .. 4u + 3
      const f32x2 ha = {lane_bcast(h, 8 * j), lane_bcast(h, 8 * j + 4)};
      const f32x2 hb = {lane_bcast(h, 8 * j + 8), lane_bcast(h, 8 * j + 12)};
      // four scalar FMAs, not two v_pk_fma_f32: on gfx950 the packed form is no faster (the fp32 rate needs no packing) and its
      // operand pairs cost moves; measured 27.7 k -> 26.2 k cycles per 47 steps with two recurrence waves per SIMD (LOG.md, round 6)
      ga.x = fmaf(whh[j].x, ha.x, ga.x), ga.y = fmaf(whh[j].y, ha.y, ga.y);
      gb.x = fmaf(whh[j + 1].x, hb.x, gb.x), gb.y = fmaf(whh[j + 1].y, hb.y, gb.y);
    }
    const float g0 = ga.x, g1 = ga.y, g2 = gb.x, g3 = gb.y;
    const float g = (g0 + g1) + (g2 + g3);
    if constexpr (SCALED) {
      const float sg = rcp_fast(1.f + __builtin_amdgcn_exp2f(g));
      const float act = is_g ? fmaf(sg, 2.f, -1.f) : sg;
      float cn, og;
      asm volatile(
          "s_nop 1\n\t"
          "v_mov_b32_dpp %0, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"  // g gate
          "v_mov_b32_dpp %1, %3 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"  // o gate
          "v_mul_f32_dpp %0, %3, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"   // i * g
          "v_fmac_f32_dpp %0, %3, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"        // + f * c
          : "=&v"(cn), "=&v"(og)
          : "v"(c), "v"(act));
      c = cn;
      h = og * fmaf(rcp_fast(__builtin_amdgcn_exp2f(c * 2.885390082f) + 1.f), -2.f, 1.f);  // o * tanh(c)
    } else {
      // tanh(g) = 2*sigmoid(2g) - 1: one exp + one rcp for every gate lane, no divergence
      const float sg = sigmoid_fast(is_g ? 2.f * g : g);
      const float act = is_g ? 2.f * sg - 1.f : sg;
      const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
      c = fmaf(fg, c, ig * gg);
      h = og * tanh_fast(c);
    }
    if ((lane & 3) == 0) hout[(lane >> 2) * hs + t] = h;
  }
}

template <int CIN>
__device__ void lstm_direction(const float* xs, float* gx, const LstmWeights w, const bool reverse, float* hout,
                               const int hs) {
  f32x2 whh[EQT_H / 2];
  lstm_load_whh(whh, w);
  lstm_project<CIN>(xs, gx, w, 0, 1);
  lstm_recur(gx, whh, reverse, hout, hs);
}

// ---- small dense products of the middle stages on the matrix cores -------------------------------------------------
// out[16 rows][16 cols] += A[16][K] B[K][16] with v_mfma_f32_16x16x4_f32: lane l holds A(row l % 16, k = 4 ks + l / 16),
// B(k = 4 ks + l / 16, col l % 16) and, of the result, rows 4 (l / 16) + 0..3 of column l % 16.  The activations of
// a window sit in LDS as [channel][48] rows (columns = time steps; column 47 is padding and only ever feeds output
// column 47, which nobody stores), the weights come per lane from memory, four loads per 16 input channels.
// The K index a lane group g = l / 16 covers in step ks is g * KS + ks (any bijection serves, A and B only have to
// agree): the KS weights of a lane are then contiguous in a row-major matrix and come as 16-byte loads -- 4x fewer
// memory instructions and cache-line touches than the canonical 4 ks + g, which matters because a stage requests
// ~40 registers of operands per lane in one burst.
template <int KS>
__device__ __forceinline__ void mfma_load_a(float (&a)[KS], const float* w, const int row_stride) {  // K contiguous
  static_assert(KS % 4 == 0, "16-byte loads");
  const int lane = MID_TID & 63;
  const float4* p = reinterpret_cast<const float4*>(w + (lane & 15) * row_stride + (lane >> 4) * KS);
#pragma unroll
  for (int i = 0; i < KS / 4; ++i) {
    const float4 v = p[i];
    a[4 * i] = v.x, a[4 * i + 1] = v.y, a[4 * i + 2] = v.z, a[4 * i + 3] = v.w;
  }
}
template <int KS>
__device__ __forceinline__ void mfma_load_a_t(float (&a)[KS], const float* w, const int k_stride) {  // rows contiguous
  const int lane = MID_TID & 63;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = w[(lane & 15) + ((lane >> 4) * KS + ks) * k_stride];
}
__device__ __forceinline__ void load4(float (&v)[4], const float* p) {  // p 16-byte aligned
  const float4 q = *reinterpret_cast<const float4*>(p);
  v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
}
template <int KS>
__device__ __forceinline__ f32x4 mfma_tile(const float (&a)[KS], const float* b, const int bs, const int n0, f32x4 acc) {
  const int lane = MID_TID & 63;
  const float* bp = b + (lane >> 4) * KS * bs + n0 + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bp[ks * bs], acc, 0, 0, 0);
  return acc;
}

// Input projection of one gate block (16 torch rows q * 16 .. of W_ih, i.e. gate q of every unit) for all time steps:
// gx[t][lane of (unit, gate q) in lstm_recur] = b + W_ih x_t.  xs: [CIN][48] in LDS.
constexpr int GXS = 65;  // gx row stride: the stores of a tile spread over the banks
template <int CIN>
struct ProjFrag {
  float a[CIN / 4];
  float bias[4];
};
template <int CIN>
__device__ __forceinline__ void lstm_project_load(ProjFrag<CIN>& f, const LstmWeights w, const int q) {
  const int lane = MID_TID & 63;
  mfma_load_a<CIN / 4>(f.a, w.w_ih + q * 16 * CIN, CIN);
  load4(f.bias, w.b + q * 16 + 4 * (lane >> 4));  // lstm_project_mfma applies lstm_gate_scale(q) to the result
}
template <int CIN>
__device__ __forceinline__ void lstm_project_mfma(const ProjFrag<CIN>& f, const float* xs, float* gx, const int q) {
  const int lane = MID_TID & 63;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    f32x4 acc = {f.bias[0], f.bias[1], f.bias[2], f.bias[3]};
    acc = mfma_tile<CIN / 4>(f.a, xs, 48, 16 * nt, acc);
    const int t = 16 * nt + (lane & 15);
    if (t < T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gx[t * GXS + 4 * (4 * (lane >> 4) + r) + q] = acc[r] * lstm_gate_scale(q);
    }
  }
}

// Additive self-attention on one window held in LDS (SeisBench SeqSelfAttention):
//   e[i][j] = Wa . tanh(x_i Wt + x_j Wx + bh)   (+ ba, which cancels in e - max_j e)
//   a = exp(e - rowmax) [band mask] / (sum + eps),  v = a x.
// The row max is taken over the FULL row before the band mask, as upstream does.
constexpr int KP = 34;  // padded row of q/k (even: the score loop reads channel PAIRS as 8-byte words)
// wa_lane = Wa[lane & 31], requested long before: the 32 weights become wave-uniform operands
__device__ __forceinline__ void attn_wa(float (&wa)[32], const float wa_lane) {
#pragma unroll
  for (int u = 0; u < 32; ++u) wa[u] = lane_bcast(wa_lane, u);
}
// e[i * ES + j] = sum_u Wa[u] tanh(q_i[u] + k_j[u]) up to a constant per row (it cancels in e - rowmax).
// tanh(q + k) = 1 - 2 / (exp(2q) exp(2k) + 1): with E_q = exp(2q), E_k = exp(2k) stored instead of q and k (plain = false) the
// 47 x 47 x 32 inner loop needs ONE transcendental (v_rcp) per element instead of two (they issue at quarter rate and
// were 60 % of its cycles), and the constant sum_u Wa[u] drops out.  plain = true: q, k hold the raw projections.
// UH: channel pairs whose q / k words are requested together (16: all of an element's 64 registers at once; 8: in two halves,
// for the kernels that cannot spare the registers -- the order of the sums is the same).
template <int ES, int UH = 16>
__device__ __forceinline__ void attn_scores(const float (*q)[KP], const float (*k)[KP], float* e, const float (&wa)[32], const bool plain) {
  const int tid = MID_TID, nt = MID_NT;
  if (plain) {
    for (int idx = tid; idx < T * T; idx += nt) {
      const int i = idx / T, j = idx - i * T;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int u = 0; u < 32; u += 2) {
        s0 = fmaf(wa[u], tanh_fast(q[i][u] + k[j][u]), s0);
        s1 = fmaf(wa[u + 1], tanh_fast(q[i][u + 1] + k[j][u + 1]), s1);
      }
      e[i * ES + j] = s0 + s1;
    }
  } else {
    // channel pairs through the packed FMA: per pair 2 v_pk_fma_f32 + 2 v_rcp_f32 instead of 4 FMAs + 2 reciprocals
    // (the loop is bound by VALU issue), same two partial sums (even / odd channels) as the scalar form
    for (int idx = tid; idx < T * T; idx += nt) {
      const int i = idx / T, j = idx - i * T;
      const f32x2* qi = reinterpret_cast<const f32x2*>(&q[i][0]);
      const f32x2* kj = reinterpret_cast<const f32x2*>(&k[j][0]);
      f32x2 s = {0.f, 0.f};
#pragma unroll
      for (int u0 = 0; u0 < 16; u0 += UH) {
        if (UH < 16 && u0 > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = u0; u < u0 + UH; ++u) {
          const f32x2 d = __builtin_elementwise_fma(qi[u], kj[u], f32x2{1.f, 1.f});
          const f32x2 r = {rcp_fast(d.x), rcp_fast(d.y)};
          s = __builtin_elementwise_fma(f32x2{wa[2 * u], wa[2 * u + 1]}, r, s);
        }
      }
      e[i * ES + j] = -2.f * (s.x + s.y);  // = sum_u Wa[u] tanh(q + k) - sum_u Wa[u]
    }
  }
}

// a = exp(e - rowmax) [band mask] / (sum + eps), in place; the row max is taken over the FULL row before the band
// mask, as upstream does.  ZERO_PAD: column 47 of every row is set to 0 (the K padding of the matrix-core a.x).
template <int ES, bool ZERO_PAD>
__device__ __forceinline__ void attn_softmax(float* e, const float eps, const int width) {
  // three rows per trip (wave_max3 / wave_sum3)
  const int tid = MID_TID, lane = tid & 63, wave = tid >> 6, nw = MID_NT >> 6;
  constexpr int R = 3;
  const int lower = lane - width / 2;  // mask[i][j] = lower_j <= i < lower_j + width
  for (int i0 = wave; i0 < T; i0 += R * nw) {
    float x[R], m[R], ex[R], sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r * nw;
      x[r] = (lane < T && i < T) ? e[i * ES + lane] : -INFINITY;
    }
    m[0] = x[0], m[1] = x[1], m[2] = x[2];
    wave_max3(m[0], m[1], m[2]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r * nw;
      ex[r] = (lane < T && i < T) ? __expf(x[r] - m[r]) : 0.f;
      if (width > 0 && !(lower <= i && i < lower + width)) ex[r] = 0.f;
    }
    sum[0] = ex[0], sum[1] = ex[1], sum[2] = ex[2];
    wave_sum3(sum[0], sum[1], sum[2]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r * nw;
      if (lane < (ZERO_PAD ? T + 1 : T) && i < T) e[i * ES + lane] = ex[r] * rcp_fast(sum[r] + eps);  // ex = 0 in lane 47
    }
  }
}

__device__ void attention_core(const float (*xs)[EQT_H], float (*q)[KP], float (*k)[KP], float (*e)[48],
                               float (*v)[EQT_H], const AttnWeights w, const float eps, const int width) {
  const int tid = MID_TID, nt = MID_NT;
  const float wa_lane = w.Wa[tid & 31];
  // Guard of the E_q E_k form: |q|, |k| <= 30 (E within 1e+-26, no inf x 0); a window beyond that takes the plain form.
  bool big = false;
  for (int idx = tid; idx < T * 32; idx += nt) {
    const int t = idx >> 5, u = idx & 31;
    float aq = 0.f, ak = w.bh[u];
#pragma unroll
    for (int c = 0; c < EQT_H; ++c) {
      aq = fmaf(xs[t][c], w.Wt[c * 32 + u], aq);
      ak = fmaf(xs[t][c], w.Wx[c * 32 + u], ak);
    }
    q[t][u] = aq;
    k[t][u] = ak;
    big |= !(fabsf(aq) <= 30.f) || !(fabsf(ak) <= 30.f);  // also catches NaN
  }
  const bool plain = team_vote_or(big);  // barrier: q / k complete
  if (!plain) {
    for (int idx = tid; idx < T * 32; idx += nt) {
      const int t = idx >> 5, u = idx & 31;
      q[t][u] = __expf(2.f * q[t][u]);
      k[t][u] = __expf(2.f * k[t][u]);
    }
    __syncthreads();
  }
  float wa[32];
  attn_wa(wa, wa_lane);
  attn_scores<48>(q, k, &e[0][0], wa, plain);
  __syncthreads();
  attn_softmax<48, false>(&e[0][0], eps, width);
  __syncthreads();
  for (int idx = tid; idx < T * EQT_H; idx += nt) {
    const int i = idx >> 4, c = idx & 15;
    float acc = 0.f;
    // banded attention: a[i][j] is exactly 0 outside j in [i - (width - 1 - width / 2), i + width / 2]
    const int j0 = width > 0 ? max(0, i - (width - 1 - width / 2)) : 0, j1 = width > 0 ? min(T - 1, i + width / 2) : T - 1;
    for (int j = j0; j <= j1; ++j) acc = fmaf(e[i][j], xs[j][c], acc);
    v[i][c] = acc;
  }
  __syncthreads();
}

// The attention of the fused kernel: every dense product on the matrix cores, activations as [channel][48] rows.
//   x:  [16][48] input rows (column 47 must be finite: it meets the zero column of a in the a.x product)
//   vT: [16][48] result rows,  q, k: [T][KP],  e: [T][AES]
constexpr int AES = 49;  // row stride of e: the B fragments of a.x (lane -> row) spread over the banks
struct AttnFrag {
  float a[4], bias[4];
};
// fragments of the q / k projection of this wave: m-tile mt = wave & 3 (q rows 0-15, 16-31, k rows 0-15, 16-31)
__device__ __forceinline__ void attn_load(AttnFrag& f, const AttnWeights w) {
  const int lane = MID_TID & 63, mt = (MID_TID >> 6) & 3;
  mfma_load_a_t<4>(f.a, (mt < 2 ? w.Wt : w.Wx) + 16 * (mt & 1), 32);
  // loaded unconditionally (q rows ignore it at use): a select on a loaded value would make the wave wait for the
  // load right here, at the point that exists to leave it in flight
  load4(f.bias, w.bh + 16 * (mt & 1) + 4 * (lane >> 4));
}
struct NoPrefetch {
  __device__ void operator()() const {}
};

__device__ inline void load_window_transposed(const float* src, int ls, float (*xs)[EQT_H]) {
  for (int idx = MID_TID; idx < EQT_H * T; idx += MID_NT) {
    const int c = idx / T, t = idx - c * T;
    xs[t][c] = src[(long)c * ls + HALO + t];
  }
}

// LayerNormalization over the channel axis of one time step (eps under the sqrt).
__device__ inline void layer_norm16(const float* z, const float* gamma, const float* beta, float eps, float* out) {
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < EQT_H; ++c) mean += z[c];
  mean *= (1.f / EQT_H);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < EQT_H; ++c) var = fmaf(z[c] - mean, z[c] - mean, var);
  var = var * (1.f / EQT_H) + eps;
  const float inv = 1.f / sqrtf(var);  // one division instead of sixteen
#pragma unroll
  for (int c = 0; c < EQT_H; ++c) out[c] = (z[c] - mean) * inv * gamma[c] + beta[c];
}

// LayerNormalization of a column whose 16 channels sit as z[r] = channel 4 (l / 16) + r in the lanes l, l ^ 16, l ^ 32,
// l ^ 48 (the result layout of the matrix-core products); gamma at lane G0 + c of `par`, beta at G0 + 16 + c.
template <int G0>
__device__ __forceinline__ void layer_norm_mfma(float (&z)[4], const float par, const float eps) {
  const int g = (MID_TID & 63) >> 4;
  float s = (z[0] + z[1]) + (z[2] + z[3]);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float mean = s * (1.f / EQT_H);
  float d[4], v = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r] = z[r] - mean, v = fmaf(d[r], d[r], v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  const float inv = 1.f / sqrtf(v * (1.f / EQT_H) + eps);
#pragma unroll
  for (int r = 0; r < 4; ++r) z[r] = d[r] * inv * __shfl(par, G0 + 4 * g + r) + __shfl(par, G0 + 16 + 4 * g + r);
}

}  // namespace
}  // namespace vp
