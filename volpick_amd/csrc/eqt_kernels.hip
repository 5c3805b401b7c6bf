// EQTransformer bottleneck kernels; see eqt_kernels.h.
#include "eqt_kernels.h"
#include "prepost.h"

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
  const int lane = threadIdx.x & 63, row = lstm_row(lane);
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
__device__ void lstm_recur(const float* gx, const LstmWeights w, const bool reverse, float* hout, const int hs) {
  const int lane = threadIdx.x & 63, row = lstm_row(lane);
  float whh[EQT_H];
#pragma unroll
  for (int u = 0; u < EQT_H; ++u) whh[u] = w.w_hh[row * EQT_H + u];
  const bool is_g = (lane & 3) == 2;
  float h = 0.f, c = 0.f;
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? T - 1 - s : s;
    float g0 = gx[t * 64 + lane], g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
    for (int k = 0; k < EQT_H; k += 4) {  // h of unit k lives in the quad 4k .. 4k + 3
      g0 = fmaf(whh[k], lane_bcast(h, 4 * k), g0);
      g1 = fmaf(whh[k + 1], lane_bcast(h, 4 * (k + 1)), g1);
      g2 = fmaf(whh[k + 2], lane_bcast(h, 4 * (k + 2)), g2);
      g3 = fmaf(whh[k + 3], lane_bcast(h, 4 * (k + 3)), g3);
    }
    const float g = (g0 + g1) + (g2 + g3);
    // tanh(g) = 2*sigmoid(2g) - 1: one exp + one rcp for every gate lane, no divergence
    const float sg = sigmoid_fast(is_g ? 2.f * g : g);
    const float act = is_g ? 2.f * sg - 1.f : sg;
    const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
    c = fmaf(fg, c, ig * gg);
    h = og * tanh_fast(c);
    if ((lane & 3) == 0) hout[(lane >> 2) * hs + t] = h;
  }
}

template <int CIN>
__device__ void lstm_direction(const float* xs, float* gx, const LstmWeights w, const bool reverse, float* hout,
                               const int hs) {
  lstm_project<CIN>(xs, gx, w, 0, 1);
  lstm_recur(gx, w, reverse, hout, hs);
}

// Additive self-attention on one window held in LDS (SeisBench SeqSelfAttention):
//   e[i][j] = Wa . tanh(x_i Wt + x_j Wx + bh)   (+ ba, which cancels in e - max_j e)
//   a = exp(e - rowmax) [band mask] / (sum + eps),  v = a x.
// The row max is taken over the FULL row before the band mask, as upstream does.
constexpr int KP = 33;  // padded row of q/k: consecutive rows hit consecutive LDS banks
__device__ void attention_core(const float (*xs)[EQT_H], float (*q)[KP], float (*k)[KP], float (*e)[48],
                               float (*v)[EQT_H], const AttnWeights w, const float eps, const int width,
                               unsigned long long* sub = nullptr) {
#define ATT_SUB(slot) \
  if (sub && threadIdx.x == 0) sub[slot] = __builtin_readcyclecounter();
  const int tid = threadIdx.x, nt = blockDim.x;
  // tanh(q + k) = 1 - 2 / (exp(2q) exp(2k) + 1): with E_q = exp(2q), E_k = exp(2k) stored instead of q and k, the 47 x 47 x 32
  // inner loop needs ONE transcendental (v_rcp) per element instead of two (they issue at quarter rate and were 60 % of
  // its cycles), and the constant sum_u Wa[u] drops out of e - rowmax.  Guard: |q|, |k| <= 30 (E within 1e+-26, no
  // inf x 0); a window beyond that takes the plain form.
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
  const bool plain = __syncthreads_or(big);  // barrier: q / k complete
  ATT_SUB(0)
  if (!plain) {
    for (int idx = tid; idx < T * 32; idx += nt) {
      const int t = idx >> 5, u = idx & 31;
      q[t][u] = __expf(2.f * q[t][u]);
      k[t][u] = __expf(2.f * k[t][u]);
    }
    __syncthreads();
  }
  ATT_SUB(1)
  {
    float wa[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wa[u] = w.Wa[u];
    if (plain) {
      for (int idx = tid; idx < T * T; idx += nt) {
        const int i = idx / T, j = idx - i * T;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
          s0 = fmaf(wa[u], tanh_fast(q[i][u] + k[j][u]), s0);
          s1 = fmaf(wa[u + 1], tanh_fast(q[i][u + 1] + k[j][u + 1]), s1);
        }
        e[i][j] = s0 + s1;
      }
    } else {
      for (int idx = tid; idx < T * T; idx += nt) {
        const int i = idx / T, j = idx - i * T;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
          s0 = fmaf(wa[u], rcp_fast(fmaf(q[i][u], k[j][u], 1.f)), s0);
          s1 = fmaf(wa[u + 1], rcp_fast(fmaf(q[i][u + 1], k[j][u + 1], 1.f)), s1);
        }
        e[i][j] = -2.f * (s0 + s1);  // = sum_u Wa[u] tanh(q + k) - sum_u Wa[u]
      }
    }
  }
  __syncthreads();
  ATT_SUB(2)
  {
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int i = wave; i < T; i += nw) {
      const float x = (lane < T) ? e[i][lane] : -INFINITY;
      const float m = wave_max64(x);
      float ex = (lane < T) ? __expf(x - m) : 0.f;
      if (width > 0) {
        const int lower = lane - width / 2;  // mask[i][j] = lower_j <= i < lower_j + width
        if (!(lower <= i && i < lower + width)) ex = 0.f;
      }
      const float sum = wave_sum64(ex);
      if (lane < T) e[i][lane] = ex * rcp_fast(sum + eps);
    }
  }
  __syncthreads();
  ATT_SUB(3)
  for (int idx = tid; idx < T * EQT_H; idx += nt) {
    const int i = idx >> 4, c = idx & 15;
    float acc = 0.f;
    for (int j = 0; j < T; ++j) acc = fmaf(e[i][j], xs[j][c], acc);
    v[i][c] = acc;
  }
  __syncthreads();
  ATT_SUB(4)
#undef ATT_SUB
}

__device__ inline void load_window_transposed(const float* src, int ls, float (*xs)[EQT_H]) {
  for (int idx = threadIdx.x; idx < EQT_H * T; idx += blockDim.x) {
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
  const int tid = threadIdx.x, b = blockIdx.x;
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
constexpr int TR_NTH = 512;  // threads per window (1024 ran the kernel itself 3 us faster and the pipeline 1 % slower: tools/sweep_pick.sh)
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
  const int tid = threadIdx.x, b = blockIdx.x;
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
  __shared__ float q[T][KP], k[T][KP];
  __shared__ float e[T][48];
  __shared__ float v[T][EQT_H];
  const int tid = threadIdx.x, b = blockIdx.x, br = blockIdx.y;
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
// The current activation (16 x 47) travels through LDS (`cur`); every stage still writes its tensor to memory (decoder
// inputs; the others for the layer-by-layer tests).  Against the separate kernels: the input projections of the LSTMs
// are spread over all waves instead of running in front of the recurrence on its one wave, the two pick LSTMs run
// side by side, and five kernel boundaries are gone.
constexpr int MID_NTH = 512;
constexpr int MID_WPD = MID_NTH / 128;  // waves per LSTM direction in the input projections
constexpr int MID_POOL = 16000;  // floats; the stages carve it up in turn

// Debug clock stamps inside the stages (slots 8.. of the window's 32; the last caller of a stage wins).
#define MID_SUB(slot) \
  if (sub && threadIdx.x == 0) sub[slot] = __builtin_readcyclecounter();

template <int CIN>
__device__ void mid_bilstm(const BiLstmArgs& a, const int b, float* P, float* cur, const bool from_memory,
                           unsigned long long* sub) {
  const int tid = threadIdx.x, wave = tid >> 6;
  float* xs = P;                    // [T][CIN]
  float* gx = P + T * 64;           // [2][T * 64]
  float* hc = P + 3 * T * 64;       // [32][48]
  if (from_memory) {
    const float* src = a.src + (long)b * a.ws_src;
    for (int idx = tid; idx < CIN * T; idx += MID_NTH) {
      const int c = idx / T, t = idx - c * T;
      xs[t * CIN + c] = src[(long)c * a.ls_src + HALO + t];
    }
  } else {
    for (int idx = tid; idx < CIN * T; idx += MID_NTH) {
      const int c = idx / T, t = idx - c * T;
      xs[t * CIN + c] = cur[c * 48 + t];
    }
  }
  __syncthreads();
  MID_SUB(8)
  lstm_project<CIN>(xs, gx + (wave / MID_WPD) * T * 64, (wave / MID_WPD) ? a.bwd : a.fwd, wave % MID_WPD, MID_WPD);  // first half of the waves: fwd
  __syncthreads();
  MID_SUB(9)
  if (wave < 2) lstm_recur(gx + wave * T * 64, wave ? a.bwd : a.fwd, wave == 1, hc + wave * 16 * 48, 48);
  __syncthreads();
  MID_SUB(10)
  float* dst = a.dst + (long)b * a.ws_dst;
  for (int idx = tid; idx < EQT_H * T; idx += MID_NTH) {  // Conv1d(32,16,1) + BatchNorm, folded
    const int co = idx / T, t = idx - co * T;
    float acc = a.bc[co];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fmaf(a.wc[co * 32 + c], hc[c * 48 + t], acc);
    dst[(long)co * a.ls_dst + HALO + t] = acc;
    cur[co * 48 + t] = acc;
  }
  __syncthreads();
}

__device__ void mid_transformer(const TransformerArgs& a, const int b, float* P, float* cur, unsigned long long* sub) {
  constexpr int NTH = MID_NTH;
  const int tid = threadIdx.x;
  float(*xs)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(P);
  constexpr int POOL = 2 * T * KP + T * 48;
  constexpr int W1S = 17, W2S = 129;
  static_assert(128 * W1S + EQT_H * W2S <= POOL, "feed-forward weights must fit the attention scratch");
  float* pool = P + T * EQT_H;
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(pool);
  float(*k)[KP] = reinterpret_cast<float(*)[KP]>(pool + T * KP);
  float(*e)[48] = reinterpret_cast<float(*)[48]>(pool + 2 * T * KP);
  float* w1s = pool;
  float* w2s = pool + 128 * W1S;
  float(*v)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(pool + POOL);
  float(*y1)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(pool + POOL + T * EQT_H);
  float(*h1)[128] = reinterpret_cast<float(*)[128]>(pool + POOL + 2 * T * EQT_H);
  static_assert(T * EQT_H + POOL + 2 * T * EQT_H + T * 128 <= MID_POOL, "transformer stage fits the pool");
  for (int idx = tid; idx < EQT_H * T; idx += NTH) {
    const int c = idx / T, t = idx - c * T;
    xs[t][c] = cur[c * 48 + t];
  }
  __syncthreads();
  MID_SUB(11)
  attention_core(xs, q, k, e, v, a.att, a.attn_eps, 0, sub ? sub + 20 : nullptr);  // ends with a barrier: q / k / e are dead now
  MID_SUB(12)
  for (int i = tid; i < 128 * 16; i += NTH) {
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
  MID_SUB(13)
  for (int idx = tid; idx < T * 128; idx += NTH) {  // FF: Linear(16,128) + ReLU
    const int t = idx >> 7, m = idx & 127;
    float acc = a.bb1[m];
#pragma unroll
    for (int c = 0; c < EQT_H; ++c) acc = fmaf(w1s[m * W1S + c], y1[t][c], acc);
    h1[t][m] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  MID_SUB(14)
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
  MID_SUB(15)
  if (tid < T) layer_norm16(v[tid], a.g2, a.b2, a.ln_eps, xs[tid]);
  __syncthreads();
  MID_SUB(16)
  float* dst = a.dst + (long)b * a.ws_dst;
  float* up = a.up ? a.up + (long)b * a.ws_up : nullptr;
  for (int idx = tid; idx < EQT_H * T; idx += NTH) {
    const int c = idx / T, t = idx - c * T;
    const float val = xs[t][c];
    dst[(long)c * a.ls_dst + HALO + t] = val;
    if (up) up[(long)c * a.ls_up + HALO + t] = val;
    cur[c * 48 + t] = val;
  }
  __syncthreads();
}

__device__ void mid_pick(const PickBranchArgs& a, const int b, float* P, const float* cur, unsigned long long* sub) {
  constexpr int NTH = MID_NTH;
  const int tid = threadIdx.x, wave = tid >> 6;
  float(*xs)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(P);          // transformer output, [T][16]
  float* gx = P + T * EQT_H;                                           // [2][T * 64]
  float* hl = gx + 2 * T * 64;                                         // [2][16][48]
  float(*x2)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(hl + 2 * 16 * 48);
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(hl + 2 * 16 * 48 + T * EQT_H);
  float(*k)[KP] = q + T;
  float(*e)[48] = reinterpret_cast<float(*)[48]>(reinterpret_cast<float*>(k + T));
  float(*v)[EQT_H] = reinterpret_cast<float(*)[EQT_H]>(reinterpret_cast<float*>(e + T));
  static_assert(T * EQT_H + 2 * T * 64 + 2 * 16 * 48 + T * EQT_H + 2 * T * KP + T * 48 + T * EQT_H <= MID_POOL, "pick stage fits the pool");
  for (int idx = tid; idx < EQT_H * T; idx += NTH) {
    const int c = idx / T, t = idx - c * T;
    xs[t][c] = cur[c * 48 + t];
  }
  __syncthreads();
  MID_SUB(17)
  lstm_project<EQT_H>(&xs[0][0], gx + (wave / MID_WPD) * T * 64, a.lstm[wave / MID_WPD], wave % MID_WPD, MID_WPD);  // first half of the waves: P
  __syncthreads();
  MID_SUB(18)
  if (wave < 2) lstm_recur(gx + wave * T * 64, a.lstm[wave], false, hl + wave * 16 * 48, 48);
  __syncthreads();
  MID_SUB(19)
  for (int br = 0; br < 2; ++br) {
    for (int idx = tid; idx < T * EQT_H; idx += NTH) {
      const int t = idx >> 4, c = idx & 15;
      x2[t][c] = hl[(br * 16 + c) * 48 + t];
    }
    __syncthreads();
    attention_core(x2, q, k, e, v, a.att[br], a.attn_eps, a.width, sub ? sub + 26 : nullptr);
    float* up = a.up + (long)((1 + br) * a.B + b) * a.ws_up;
    for (int idx = tid; idx < EQT_H * T; idx += NTH) {
      const int c = idx / T, t = idx - c * T;
      up[(long)c * a.ls_up + HALO + t] = v[t][c];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(MID_NTH) void eqt_mid_kernel(const MidArgs a) {
  __shared__ __attribute__((aligned(16))) float P[MID_POOL];
  __shared__ float cur[16 * 48];
  const int b = blockIdx.x;
  int stamp = 0;
  unsigned long long* sub = a.clk ? a.clk + (long)b * 32 : nullptr;
#define MID_STAMP()                                                                                   \
  if (a.clk && threadIdx.x == 0) a.clk[(long)b * 32 + stamp] = __builtin_readcyclecounter();         \
  ++stamp;
  MID_STAMP()
  mid_bilstm<64>(a.lstm[0], b, P, cur, true, nullptr);
  MID_STAMP()
  mid_bilstm<EQT_H>(a.lstm[1], b, P, cur, false, nullptr);
  MID_STAMP()
  mid_bilstm<EQT_H>(a.lstm[2], b, P, cur, false, sub);
  MID_STAMP()
  mid_transformer(a.tr[0], b, P, cur, nullptr);
  MID_STAMP()
  mid_transformer(a.tr[1], b, P, cur, sub);
  MID_STAMP()
  mid_pick(a.pick, b, P, cur, sub);
  MID_STAMP()
#undef MID_STAMP
}

int launch_eqt_mid(const MidArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(eqt_mid_kernel, dim3(B), dim3(MID_NTH), 0, s, a);
  return 0;
}

}  // namespace vp
