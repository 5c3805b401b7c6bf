// EQTransformer bottleneck kernels; see eqt_kernels.h.
#include "eqt_kernels.h"
#include "prepost.h"
#include "conv_valu.h"

// Thread index within a window's 512-thread team.  Every kernel of this file has at most 512 threads per workgroup except the
// two-window form of eqt_mid_kernel (1024 threads: waves 0-7 take one window, waves 8-15 the next).
// The second team's roles are rotated by two waves: the two recurrence waves of a window (roles 0 and 1) then sit on SIMDs 0, 1
// for the first team and on SIMDs 2, 3 for the second (hardware wave h runs on SIMD h % 4).
#define MID_TID ((threadIdx.x + ((threadIdx.x >> 9) << 7)) & 511)
#define MID_NT (blockDim.x > 512u ? 512u : blockDim.x)  // threads of a window's team
// The two-window form shares its barriers, not its decisions: a vote (which form a window's attention scores take) is ONE
// workgroup-wide OR reduction in which every team sets its own bit and reads only that bit back, so what a window computes
// never depends on the window it shares a workgroup with (same barrier count on both teams, no extra LDS).
#define MID_TEAM (threadIdx.x >> 9)
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
      ga = __builtin_elementwise_fma(whh[j], ha, ga);
      gb = __builtin_elementwise_fma(whh[j + 1], hb, gb);
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
template <int ES>
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
      for (int u = 0; u < 16; ++u) {
        const f32x2 d = __builtin_elementwise_fma(qi[u], kj[u], f32x2{1.f, 1.f});
        const f32x2 r = {rcp_fast(d.x), rcp_fast(d.y)};
        s = __builtin_elementwise_fma(f32x2{wa[2 * u], wa[2 * u + 1]}, r, s);
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
// `prefetch` runs at the start of the score loop, the longest stretch of pure arithmetic of the stage: the caller
// requests the NEXT stage's weights there.
// `finish(acc, n0)`: what waves 0-2 do with their tile of the result (rows = channels 4 (l / 16) + r, column n0 + l % 16).
template <class Prefetch, class Finish>
__device__ void mid_attention(const AttnFrag& f, const float wa_lane, const float* x, float (*q)[KP], float (*k)[KP],
                              float* e, const float eps, const int width, Prefetch&& prefetch, Finish&& finish,
                              unsigned long long* sub) {
#define ATT_SUB(slot) \
  if (sub && MID_TID == 0) sub[slot] = __builtin_readcyclecounter();
  const int tid = MID_TID, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {  // q / k projection: waves 0-3 take the column tiles 0 and 1 of their m-tile, waves 4-7 tile 2
    const int mt = wave & 3;
    float(*dstp)[KP] = mt < 2 ? q : k;
    f32x4 acc[2];
    const int ntiles = wave < 4 ? 2 : 1, nt0 = wave < 4 ? 0 : 2;
    bool big = false;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      acc[n] = mt < 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{f.bias[0], f.bias[1], f.bias[2], f.bias[3]};  // bh belongs to k
      if (n < ntiles) {
        acc[n] = mfma_tile<4>(f.a, x, 48, 16 * (nt0 + n), acc[n]);
#pragma unroll
        for (int r = 0; r < 4; ++r) big |= !(fabsf(acc[n][r]) <= 30.f) && (16 * (nt0 + n) + (lane & 15) < T);
      }
    }
    const bool plain = team_vote_or(big);  // only the vote: nothing has been written yet
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int t = 16 * (nt0 + n) + (lane & 15);
      if (n < ntiles && t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dstp[t][16 * (mt & 1) + 4 * (lane >> 4) + r] = plain ? acc[n][r] : __expf(2.f * acc[n][r]);
      }
    }
    __syncthreads();
    ATT_SUB(1)
    // Loads complete in order: whatever the wave still needs from EARLIER requests is taken out of its registers
    // before the new requests go out, or its first use would wait for them as well.
    float wa[32];
    attn_wa(wa, wa_lane);
    __builtin_amdgcn_sched_barrier(0);
    prefetch();
    __builtin_amdgcn_sched_barrier(0);
    ATT_SUB(0)
    attn_scores<AES>(q, k, e, wa, plain);
  }
  __syncthreads();
  ATT_SUB(2)
  attn_softmax<AES, true>(e, eps, width);
  __syncthreads();
  ATT_SUB(3)
  if (wave < 3) {  // vT[c][i] = sum_j x[c][j] a[i][j]: A = x rows out of LDS, B(k = j, n = i) = a[i][j]
    const int n0 = 16 * wave;
    const float* ap = x + (lane & 15) * 48 + (lane >> 4) * 12;       // K index of (lane group g, step ks) = 12 g + ks
    const float* bp = e + (n0 + (lane & 15)) * AES + (lane >> 4) * 12;  // rows 47 .. of the last tile: stale data, column never stored
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], bp[ks], acc, 0, 0, 0);
    finish(acc, n0);
  }
  __syncthreads();
  ATT_SUB(4)
#undef ATT_SUB
}

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

}  // namespace

// ---------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(128) void bilstm_kernel(const BiLstmArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[T * CIN];
  __shared__ float gx[2][T * 64];
  __shared__ float hc[32][48];
  const int tid = MID_TID, b = blockIdx.x;
  const float* src = a.src + (long)b * a.ws_src;
  for (int idx = tid; idx < CIN * T; idx += 128) {
    const int c = idx / T, t = idx - c * T;
    xs[t * CIN + c] = src[(long)c * a.ls_src + HALO + t];
  }
  __syncthreads();
  const int wave = tid >> 6;
  lstm_direction<CIN>(xs, gx[wave], wave == 0 ? a.fwd : a.bwd, wave == 1, &hc[wave * 16][0], 48);
  __syncthreads();
  float* dst = a.dst + (long)b * a.ws_dst;
  for (int idx = tid; idx < EQT_H * T; idx += 128) {  // Conv1d(32,16,1) + BatchNorm, folded
    const int co = idx / T, t = idx - co * T;
    float acc = a.bc[co];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fmaf(a.wc[co * 32 + c], hc[c][t], acc);
    dst[(long)co * a.ls_dst + HALO + t] = acc;
  }
}

int launch_bilstm(const BiLstmArgs& a, int cin, int B, hipStream_t s) {
  if (cin == 64) {
    hipLaunchKernelGGL(bilstm_kernel<64>, dim3(B), dim3(128), 0, s, a);
  } else if (cin == 16) {
    hipLaunchKernelGGL(bilstm_kernel<16>, dim3(B), dim3(128), 0, s, a);
  } else {
    set_error("bilstm: unsupported input size %d", cin);
    return VP_ERR_UNSUPPORTED;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
constexpr int TR_NTH = 512;  // threads per window (1024 ran the kernel itself 3 us faster and the pipeline 1 % slower: a round-1 workgroup-size sweep, LOG.md)
__global__ __launch_bounds__(TR_NTH) void transformer_kernel(const TransformerArgs a) {
  constexpr int NTH = TR_NTH;
  __shared__ float xs[T][EQT_H];
  // q / k / e of the attention and, afterwards, the padded feed-forward weights share one pool
  constexpr int POOL = 2 * T * KP + T * 48;
  constexpr int W1S = 17, W2S = 129;  // row strides that spread the rows over the LDS banks
  static_assert(128 * W1S + EQT_H * W2S <= POOL, "feed-forward weights must fit the attention scratch");
  __shared__ float pool[POOL];
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(pool);
  float(*k)[KP] = reinterpret_cast<float(*)[KP]>(pool + T * KP);
  float(*e)[48] = reinterpret_cast<float(*)[48]>(pool + 2 * T * KP);
  float* w1s = pool;
  float* w2s = pool + 128 * W1S;
  __shared__ float v[T][EQT_H];
  __shared__ float y1[T][EQT_H];
  __shared__ float h1[T][128];
  const int tid = MID_TID, b = blockIdx.x;
  load_window_transposed(a.src + (long)b * a.ws_src, a.ls_src, xs);
  __syncthreads();
  attention_core(xs, q, k, e, v, a.att, a.attn_eps, 0);  // ends with a barrier: q / k / e are dead now
  for (int i = tid; i < 128 * 16; i += NTH) {  // coalesced global reads, padded LDS rows
    w1s[(i >> 4) * W1S + (i & 15)] = a.w1[i];
    w2s[(i >> 7) * W2S + (i & 127)] = a.w2[i];
  }
  if (tid < T) {  // y1 = LN1(x + attention(x))
    float z[EQT_H];
#pragma unroll
    for (int c = 0; c < EQT_H; ++c) z[c] = xs[tid][c] + v[tid][c];
    layer_norm16(z, a.g1, a.b1, a.ln_eps, y1[tid]);
  }
  __syncthreads();
  for (int idx = tid; idx < T * 128; idx += NTH) {  // FF: Linear(16,128) + ReLU
    const int t = idx >> 7, m = idx & 127;
    float acc = a.bb1[m];
#pragma unroll
    for (int c = 0; c < EQT_H; ++c) acc = fmaf(w1s[m * W1S + c], y1[t][c], acc);
    h1[t][m] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  for (int idx = tid; idx < T * EQT_H; idx += NTH) {  // Linear(128,16) + residual
    const int t = idx >> 4, c = idx & 15;
    float a0 = a.bb2[c], a1 = 0.f;
#pragma unroll 8
    for (int m = 0; m < 128; m += 2) {
      a0 = fmaf(w2s[c * W2S + m], h1[t][m], a0);
      a1 = fmaf(w2s[c * W2S + m + 1], h1[t][m + 1], a1);
    }
    v[t][c] = y1[t][c] + (a0 + a1);
  }
  __syncthreads();
  if (tid < T) layer_norm16(v[tid], a.g2, a.b2, a.ln_eps, xs[tid]);
  __syncthreads();
  float* dst = a.dst + (long)b * a.ws_dst;
  float* up = a.up ? a.up + (long)b * a.ws_up : nullptr;
  for (int idx = tid; idx < EQT_H * T; idx += NTH) {
    const int c = idx / T, t = idx - c * T;
    const float val = xs[t][c];
    dst[(long)c * a.ls_dst + HALO + t] = val;
    if (up) up[(long)c * a.ls_up + HALO + t] = val;
  }
}

int launch_transformer(const TransformerArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(transformer_kernel, dim3(B), dim3(TR_NTH), 0, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------
// P / S branch: LSTM(16,16) -> banded additive attention -> x2-upsampled decoder input.
constexpr int PICK_NTH = 256;  // threads per (window, branch): 1024 held every wave slot of the chip through the 47 sequential LSTM steps of ONE wave
__global__ __launch_bounds__(PICK_NTH) void pick_branch_kernel(const PickBranchArgs a) {
  constexpr int NTH = PICK_NTH;
  __shared__ __attribute__((aligned(16))) float xs[T][EQT_H];
  __shared__ float gx[T * 64];
  __shared__ float hl[EQT_H][48];
  __shared__ float x2[T][EQT_H];
  __shared__ __attribute__((aligned(8))) float q[T][KP], k[T][KP];
  __shared__ float e[T][48];
  __shared__ float v[T][EQT_H];
  const int tid = MID_TID, b = blockIdx.x, br = blockIdx.y;
  load_window_transposed(a.src + (long)b * a.ws_src, a.ls_src, xs);
  __syncthreads();
  if (tid < 64) lstm_direction<EQT_H>(&xs[0][0], gx, a.lstm[br], false, &hl[0][0], 48);
  __syncthreads();
  for (int idx = tid; idx < T * EQT_H; idx += NTH) {
    const int t = idx >> 4, c = idx & 15;
    x2[t][c] = hl[c][t];
  }
  __syncthreads();
  attention_core(x2, q, k, e, v, a.att[br], a.attn_eps, a.width);
  float* up = a.up + (long)((1 + br) * a.B + b) * a.ws_up;
  for (int idx = tid; idx < EQT_H * T; idx += NTH) {
    const int c = idx / T, t = idx - c * T;
    const float val = v[t][c];
    up[(long)c * a.ls_up + HALO + t] = val;
  }
}

int launch_pick_branch(const PickBranchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(pick_branch_kernel, dim3(a.B, 2), dim3(PICK_NTH), 0, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------
// One workgroup (8 wavefronts) per window walks bilstm.0-2 -> transformer_d0 -> transformer_d -> pick branches.
// The current activation travels through LDS as [16 channels][48] rows (`cur`; column 47 is the zero K-padding of the
// a.x products); every stage still writes its tensor to memory (decoder inputs; the others for the layer tests).
// Against the separate kernels above (kept as the six-launch plan, plan_flags[2] = 1):
//   * every dense product is a set of 16x16x4 fp32 matrix-core tiles with the time steps as columns (mfma_tile):
//     LSTM input projections, Conv1d(32,16,1), q / k projections, a.x, both feed-forward layers; LayerNorm runs in the
//     epilogues of a.x and of the second feed-forward product (layer_norm_mfma);
//   * the operands of a stage (BiFrags / TrFrags / PickFrags) are requested one stage ahead, under the 47 LSTM steps or
//     the attention score loop of the stage before, as 16-byte loads in fragment order;
//   * the two pick LSTMs run side by side and five kernel boundaries are gone.
// Serial or transcendental work stays on the VALU: the recurrences (one wave per direction), the 47 x 47 x 32 score
// loops, the row softmax.  DESIGN.md section 6 lists what each of these steps was worth in cycles.
constexpr int MID_NTH = 512;
constexpr int MID_POOL = 16000;  // floats; the stages carve it up in turn

// Debug clock stamps inside the stages (slots 8.. of the window's 32; the last caller of a stage wins).
#define MID_SUB(slot) \
  if (sub && MID_TID == 0) sub[slot] = __builtin_readcyclecounter();

// The operands a stage reads from memory, as registers: requested one stage ahead (under the 47 sequential LSTM steps
// or the attention score loop of the stage before), so that no stage but the first waits for memory.
template <int CIN>
struct BiFrags {
  ProjFrag<CIN> f;      // wave = (direction, gate block)
  f32x2 whh[EQT_H / 2];  // waves 0, 1: the recurrence of the forward / backward direction
  float ac[8], bcv[4];  // waves 2-4: the three column tiles of Conv1d(32,16,1)
};
template <int CIN>
__device__ __forceinline__ void bi_load(BiFrags<CIN>& g, const BiLstmArgs& a) {
  const int lane = MID_TID & 63, wave = __builtin_amdgcn_readfirstlane(MID_TID >> 6);
  // no branches (waves that do not need an operand fetch it anyway): in one basic block the reads of the argument
  // block cluster into a single wait; behind wave-dependent branches they serialised, ~1 k cycles apiece
  lstm_project_load<CIN>(g.f, (wave >> 2) ? a.bwd : a.fwd, wave & 3);
  lstm_load_whh(g.whh, (wave & 1) ? a.bwd : a.fwd);
  mfma_load_a<8>(g.ac, a.wc, 32);
  load4(g.bcv, a.bc + 4 * (lane >> 4));
}

template <int CIN, class Prefetch>
__device__ void mid_bilstm(const BiLstmArgs& a, BiFrags<CIN>& g, const int b, float* P, float* cur,
                           const bool from_memory, Prefetch&& prefetch, unsigned long long* sub) {
  const int tid = MID_TID, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* gx = P;                    // [2][T * GXS]
  float* hc = P + 2 * T * GXS;      // [32][48]
  float* xs = hc + 32 * 48;         // [CIN][48], first stage only (later stages read `cur`)
  static_assert(2 * T * GXS + 32 * 48 + 64 * 48 <= MID_POOL, "BiLSTM stage fits the pool");
  static_assert(MID_NTH == 512, "eight waves: one gate block of one direction each");
  const int dir = wave >> 2, q = wave & 3;
  const float* x = cur;
  if (from_memory) {  // first stage: the window's own rows first (loads return in order), then its weights
    constexpr int NX = (CIN * T + MID_NTH - 1) / MID_NTH;
    float xr[NX];
    const float* src = a.src + (long)b * a.ws_src + HALO;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int idx = tid + k * MID_NTH, c = idx / T, t = idx - c * T;
      xr[k] = idx < CIN * T ? src[(long)c * a.ls_src + t] : 0.f;
    }
    bi_load<CIN>(g, a);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int idx = tid + k * MID_NTH, c = idx / T, t = idx - c * T;
      if (idx < CIN * T) xs[c * 48 + t] = xr[k];
    }
    lds_barrier();  // not __syncthreads(): the weight loads stay in flight
    x = xs;
  }
  MID_SUB(8)
  lstm_project_mfma<CIN>(g.f, x, gx + dir * T * GXS, q);
  __syncthreads();
  MID_SUB(9)
  if (wave < 2) {  // W_hh has arrived long ago; say so before the new requests queue up behind it (loads complete in order)
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) asm volatile("" ::"v"(g.whh[j]));
  }
  __builtin_amdgcn_sched_barrier(0);
  prefetch();
  __builtin_amdgcn_sched_barrier(0);
  if (wave < 2) {
    const float sc = lstm_gate_scale(lane & 3);
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) g.whh[j] *= sc;
    lstm_recur<GXS, true>(gx + wave * T * GXS, g.whh, wave == 1, hc + wave * 16 * 48, 48);
  }
  __syncthreads();
  MID_SUB(10)
  if (wave >= 2 && wave < 5) {  // Conv1d(32,16,1) + BatchNorm, folded
    float* dst = a.dst + (long)b * a.ws_dst;
    const int n0 = 16 * (wave - 2), t = n0 + (lane & 15);
    f32x4 acc = {g.bcv[0], g.bcv[1], g.bcv[2], g.bcv[3]};
    acc = mfma_tile<8>(g.ac, hc, 48, n0, acc);
    if (t < T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * (lane >> 4) + r;
        dst[(long)co * a.ls_dst + HALO + t] = acc[r];
        cur[co * 48 + t] = acc[r];
      }
    }
  }
  __syncthreads();
}

struct TrFrags {
  AttnFrag af;
  float wa_lane, ln_par;
  float a1[4], b1v[4];   // Linear(16,128): m-tile = wave
  float a2[16], b2v[4];  // Linear(128,16): waves 0-5 = (column tile, K half)
};
// early: what the stage needs up to its score loop and right after it (requested during the stage before);
// late: the operands of its last phases, requested by the stage itself at the start of its score loop -- fewer
// registers are live across that loop than with everything fetched a stage ahead.
__device__ __forceinline__ void tr_load_early(TrFrags& g, const TransformerArgs& a) {
  const int lane = MID_TID & 63, wave = __builtin_amdgcn_readfirstlane(MID_TID >> 6);
  attn_load(g.af, a.att);
  g.wa_lane = a.att.Wa[lane & 31];
  mfma_load_a<4>(g.a1, a.w1 + wave * 16 * EQT_H, EQT_H);
  load4(g.b1v, a.bb1 + 16 * wave + 4 * (lane >> 4));
}
__device__ __forceinline__ void tr_load_late(TrFrags& g, const TransformerArgs& a) {
  const int lane = MID_TID & 63, wave = __builtin_amdgcn_readfirstlane(MID_TID >> 6);
  g.ln_par = a.g1[lane];  // g1 | b1 | g2 | b2 are one blob (eqt_kernels.h)
  const int half = wave >= 3;
  mfma_load_a<16>(g.a2, a.w2 + 64 * half, 128);  // waves 6, 7 too: no branch, see bi_load
  load4(g.b2v, a.bb2 + 4 * (lane >> 4));           // the second K half ignores it at use
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

template <class Prefetch>
__device__ void mid_transformer(const TransformerArgs& a, TrFrags& g, const int b, float* P, float* cur,
                                Prefetch&& prefetch, unsigned long long* sub) {
  const int tid = MID_TID, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(P);
  float(*k)[KP] = q + T;
  float* e = P + 2 * T * KP;        // [48][AES]
  float* y1T = e + 48 * AES;        // [16][48] LN1 output
  float* h1T = y1T + 16 * 48;       // [128][48] hidden layer
  float* rT = h1T + 128 * 48;       // [16][48] the second K half of the second linear layer
  static_assert(2 * T * KP + 48 * AES + 16 * 48 + 128 * 48 + 16 * 48 <= MID_POOL, "transformer stage fits the pool");
  const int half = wave >= 3, nt2 = wave - 3 * half;
  const float (&a1)[4] = g.a1, (&b1v)[4] = g.b1v, (&a2)[16] = g.a2, (&b2v)[4] = g.b2v;
  MID_SUB(11)
  // y1 = LN1(x + attention(x)) right in the epilogue of the a.x product (three waves, four channels per lane)
  mid_attention(
      g.af, g.wa_lane, cur, q, k, e, a.attn_eps, 0,
      [&] {
        tr_load_late(g, a);
        prefetch();
      },
      [&](const f32x4 acc, const int n0) {
        const int col = n0 + (lane & 15), c0 = 4 * (lane >> 4);
        float z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = cur[(c0 + r) * 48 + col] + acc[r];
        layer_norm_mfma<0>(z, g.ln_par, a.ln_eps);
#pragma unroll
        for (int r = 0; r < 4; ++r) y1T[(c0 + r) * 48 + col] = z[r];  // all 48 columns: column 47 is padding
      },
      sub ? sub + 20 : nullptr);
  const float ln_par = g.ln_par;
  MID_SUB(13)
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {  // FF: Linear(16,128) + ReLU
    f32x4 acc = {b1v[0], b1v[1], b1v[2], b1v[3]};
    acc = mfma_tile<4>(a1, y1T, 48, 16 * nt, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) h1T[(16 * wave + 4 * (lane >> 4) + r) * 48 + 16 * nt + (lane & 15)] = fmaxf(acc[r], 0.f);
  }
  __syncthreads();
  MID_SUB(14)
  f32x4 acc2 = half ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{b2v[0], b2v[1], b2v[2], b2v[3]};
  if (wave < 6) {  // Linear(128,16), K split in two: waves 3-5 hand their half over through LDS
    acc2 = mfma_tile<16>(a2, h1T + 64 * half * 48, 48, 16 * nt2, acc2);
    if (half) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rT[(4 * (lane >> 4) + r) * 48 + 16 * nt2 + (lane & 15)] = acc2[r];
    }
  }
  __syncthreads();
  MID_SUB(15)
  if (wave < 3) {  // LN2(y1 + FF(y1)) and the stage's outputs, from the accumulators of the first K half
    const int col = 16 * nt2 + (lane & 15), c0 = 4 * (lane >> 4);
    float z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = y1T[(c0 + r) * 48 + col] + (acc2[r] + rT[(c0 + r) * 48 + col]);
    layer_norm_mfma<32>(z, ln_par, a.ln_eps);
    if (col < T) {
      float* dst = a.dst + (long)b * a.ws_dst + HALO + col;
      float* up = a.up ? a.up + (long)b * a.ws_up + HALO + col : nullptr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dst[(long)(c0 + r) * a.ls_dst] = z[r];
        if (up) up[(long)(c0 + r) * a.ls_up] = z[r];
        cur[(c0 + r) * 48 + col] = z[r];
      }
    }
  }
  MID_SUB(16)
  __syncthreads();
}

struct PickFrags {
  ProjFrag<EQT_H> f;  // waves 0-3: P branch, 4-7: S branch
  f32x2 whh[EQT_H / 2];  // waves 0, 1
  AttnFrag af[2];
  float wa_lane[2];
};
__device__ __forceinline__ void pick_load(PickFrags& g, const PickBranchArgs& a) {
  const int lane = MID_TID & 63, wave = __builtin_amdgcn_readfirstlane(MID_TID >> 6);
  // static offsets into the argument block + selects (a wave-dependent index would be a second, dependent read)
  lstm_project_load<EQT_H>(g.f, (wave >> 2) ? a.lstm[1] : a.lstm[0], wave & 3);
  lstm_load_whh(g.whh, (wave & 1) ? a.lstm[1] : a.lstm[0]);  // every wave: no branch, see bi_load
  attn_load(g.af[0], a.att[0]);  // the S branch's attention operands follow under the stage's own recurrence
  g.wa_lane[0] = a.att[0].Wa[lane & 31];
}

__device__ void mid_pick(const PickBranchArgs& a, PickFrags& g, const int b, float* P, const float* cur,
                         unsigned long long* sub) {
  const int tid = MID_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* gx = P;                                   // [2][T * GXS]
  float* hl = gx + 2 * T * GXS;                    // [2][16][48] LSTM outputs of the P and the S branch
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(hl + 2 * 16 * 48);
  float(*k)[KP] = q + T;
  float* e = hl + 2 * 16 * 48 + 2 * T * KP;        // [48][AES]
  float* vT = e + 48 * AES;                        // [16][48]
  static_assert(2 * T * GXS + 2 * 16 * 48 + 2 * T * KP + 48 * AES + 16 * 48 <= MID_POOL, "pick stage fits the pool");
  MID_SUB(17)
  const int br_w = wave >> 2, gq = wave & 3;
  if (tid < 32) hl[tid * 48 + 47] = 0.f;  // K padding of the a.x products
  lstm_project_mfma<EQT_H>(g.f, cur, gx + br_w * T * GXS, gq);
  __syncthreads();
  MID_SUB(18)
  if (wave < 2) {
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) asm volatile("" ::"v"(g.whh[j]));  // see mid_bilstm
  }
  __builtin_amdgcn_sched_barrier(0);
  attn_load(g.af[1], a.att[1]);
  g.wa_lane[1] = a.att[1].Wa[tid & 31];
  __builtin_amdgcn_sched_barrier(0);
  if (wave < 2) {
    const float sc = lstm_gate_scale(tid & 3);
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) g.whh[j] *= sc;
    lstm_recur<GXS, true>(gx + wave * T * GXS, g.whh, false, hl + wave * 16 * 48, 48);
  }
  __syncthreads();
  MID_SUB(19)
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    mid_attention(
        g.af[br], g.wa_lane[br], hl + br * 16 * 48, q, k, e, a.attn_eps, a.width, NoPrefetch(),
        [&](const f32x4 acc, const int n0) {
          const int i = n0 + (tid & 15);
          if (i < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) vT[(4 * ((tid & 63) >> 4) + r) * 48 + i] = acc[r];
          }
        },
        sub && br ? sub + 26 : nullptr);
    float* up = a.up + (long)((1 + br) * a.B + b) * a.ws_up + HALO;
    for (int idx = tid; idx < EQT_H * 48; idx += MID_NTH) {
      const int c = idx / 48, t = idx - c * 48;
      if (t < T) up[(long)c * a.ls_up + t] = vT[c * 48 + t];
    }
    __syncthreads();
  }
}

// WPB windows per workgroup.  The stages are latency-bound (two of a window's eight waves run its recurrences, the chip's other
// wave slots idle) and a window needs 67 KB of LDS and 127 registers: two windows fit a CU.  With WPB = 2 the 256 windows of a
// batch take 128 CUs instead of 256 for the same 66 us, and the other device contexts' kernels run on the CUs left free.
// The two teams execute the same barriers in the same order (same code, same trip counts); an odd batch's last workgroup
// computes its last window twice (identical stores).
template <int WPB>
__global__ __launch_bounds__(WPB * MID_NTH) void eqt_mid_kernel(const MidArgs a) {
  __shared__ __attribute__((aligned(16))) float P_all[WPB * MID_POOL];
  __shared__ float cur_all[WPB * 16 * 48];
  const int team = WPB > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 9) : 0;
  float* P = P_all + team * MID_POOL;
  float* cur = cur_all + team * 16 * 48;
  const int b = min((int)blockIdx.x * WPB + team, a.B - 1);
  int stamp = 0;
  unsigned long long* sub = a.clk ? a.clk + (long)b * 32 : nullptr;
  if (MID_TID < 16) cur[MID_TID * 48 + 47] = 0.f;  // K padding of the a.x products; no stage writes column 47
  {  // The argument block is 13 cache lines and every stage reads its own part of it when it starts: a cold line
     // costs ~3.5 k cycles (measured: the stage that first touched another stage's arguments grew by that much).
     // All lines are requested here at once, so the later reads hit the scalar cache.
    typedef const unsigned __attribute__((address_space(4))) * uptr_t;
    const uptr_t ka = (uptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (unsigned i = 0; i < (sizeof(MidArgs) + 63) / 64; ++i) acc |= ka[16 * i];
    asm volatile("" ::"s"(acc));
  }
#define MID_STAMP()                                                                                   \
  if (a.clk && MID_TID == 0) a.clk[(long)b * 32 + stamp] = __builtin_readcyclecounter();         \
  ++stamp;
  MID_STAMP()
  // each stage requests the next one's weights under its own longest arithmetic phase (see BiFrags)
  BiFrags<64> g0;
  BiFrags<EQT_H> g1, g2;
  TrFrags t0, t1;
  PickFrags pf;
  mid_bilstm<64>(a.lstm[0], g0, b, P, cur, true, [&] { bi_load<EQT_H>(g1, a.lstm[1]); }, sub);
  MID_STAMP()
  mid_bilstm<EQT_H>(a.lstm[1], g1, b, P, cur, false, [&] { bi_load<EQT_H>(g2, a.lstm[2]); }, nullptr);
  MID_STAMP()
  mid_bilstm<EQT_H>(a.lstm[2], g2, b, P, cur, false, [&] { tr_load_early(t0, a.tr[0]); }, nullptr);
  MID_STAMP()
  mid_transformer(a.tr[0], t0, b, P, cur, [&] { tr_load_early(t1, a.tr[1]); }, nullptr);
  MID_STAMP()
  mid_transformer(a.tr[1], t1, b, P, cur, [&] { pick_load(pf, a.pick); }, sub);
  MID_STAMP()
  mid_pick(a.pick, pf, b, P, cur, sub);
  MID_STAMP()
#undef MID_STAMP
}

int launch_eqt_mid(const MidArgs& a, int B, hipStream_t s, bool one_window_per_workgroup) {
  if (one_window_per_workgroup) {
    hipLaunchKernelGGL(eqt_mid_kernel<1>, dim3(B), dim3(MID_NTH), 0, s, a);
  } else {
    hipLaunchKernelGGL(eqt_mid_kernel<2>, dim3((B + 1) / 2), dim3(2 * MID_NTH), 0, s, a);
  }
  return 0;
}

}  // namespace vp
