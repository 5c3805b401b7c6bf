// EQTransformer bottleneck kernels; see eqt_kernels.h.
// The device helpers live in eqt_mid_parts.h (shared with the four-wave-team form, eqt_mid4.hip).
// Thread index within a window's 512-thread team.  Every kernel of this file has at most 512 threads per workgroup except the
// two-window form of eqt_mid_kernel (1024 threads: waves 0-7 take one window, waves 8-15 the next).
// The second team's roles are rotated by two waves: the two recurrence waves of a window (roles 0 and 1) then sit on SIMDs 0, 1
// for the first team and on SIMDs 2, 3 for the second (hardware wave h runs on SIMD h % 4).
#define MID_TID ((threadIdx.x + ((threadIdx.x >> 9) << 7)) & 511)
#define MID_NT (blockDim.x > 512u ? 512u : blockDim.x)  // threads of a window's team
#define MID_TEAM (threadIdx.x >> 9)
#include "eqt_mid_parts.h"

namespace vp {

namespace {

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
