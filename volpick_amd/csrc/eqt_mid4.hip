// EQTransformer bottleneck (3 BiLSTM + 2 transformer blocks + P / S pick branches), FOUR windows per 1024-thread workgroup:
// a window's team is four waves (eqt_kernels.hip: eight), its LDS 39 KB (67), so a 256-window batch holds 64 CUs (128).
//
// Why.  The chain is latency-bound where it is serial (4 x 47 LSTM steps on ONE wave per direction) and bound by the quarter-rate
// reciprocals where it is parallel (the 47 x 47 x 32 score loops: 22 k CU-cycles per window whatever the team size).  A CU that
// holds one of these workgroups holds nothing else (16 waves x 128 registers, > 130 KB of LDS), so what the launch costs the
// pipeline is CUs x time.  With four windows per CU the serial phases cost a quarter of a CU each instead of half, the parallel
// phases the same as before: CUs x time per batch falls by ~40 % (profiles/r06_*), and the CUs left free run the other device
// contexts' kernels.
//
// What changes against the eight-wave team (same arithmetic, same summation order in every product: results bit-identical):
//   * a wave owns a gate block of BOTH directions of a BiLSTM input projection (of both pick branches), one m-tile of the q / k
//     projection with all three column tiles, two m-tiles of Linear(16,128), one K half of Linear(128,16) for one or two column
//     tiles (the halves are added as the eight-wave form adds them);
//   * the stage pools alias what is dead: the first stage's input rows lie where the recurrence writes its outputs, the hidden
//     layer of the feed-forward lies over q / k / e, the pick branches' q / k / e over the input projections;
//   * the odd teams' roles are rotated by two waves, so that the eight recurrence waves of a workgroup sit two per SIMD.
// A vote (which form a window's attention scores take) stays per team (team_vote_or): what a window computes never depends on
// the windows it shares a workgroup with (tests/test_gpu_round4.py).
#define MID_TEAM (threadIdx.x >> 8)
#define MID_TID ((threadIdx.x + (((threadIdx.x >> 8) & 1u) << 7)) & 255u)
#define MID_NT 256u
#include "eqt_mid_parts.h"

namespace vp {

namespace {

constexpr int M4_NTH = 256;    // threads of a window's team
constexpr int M4_WPB = 4;      // windows per workgroup
constexpr int M4_POOL = 9184;  // floats per window; the stages carve it up in turn
static_assert((M4_POOL * 4) % 16 == 0 && M4_WPB * (M4_POOL + 16 * 48) * 4 <= 160 * 1024 - 256, "four windows fit a CU's LDS");

// Debug clock stamps inside the stages (slots 8.. of the window's 32, the slots of eqt_mid_kernel: tools/mid_clock.py)
#define M4_SUB(slot) \
  if (sub && MID_TID == 0) sub[slot] = __builtin_readcyclecounter();

__device__ __forceinline__ int m4_wave() { return __builtin_amdgcn_readfirstlane(MID_TID >> 6); }

template <int CIN>
struct Bi4Frags {
  ProjFrag<CIN> f[2];    // wave = gate block; [direction]
  f32x2 whh[EQT_H / 2];  // waves 0, 1: the recurrence of the forward / backward direction
  float ac[8], bcv[4];   // waves 1-3: the three column tiles of Conv1d(32,16,1)
};
template <int CIN>
__device__ __forceinline__ void bi4_load(Bi4Frags<CIN>& g, const BiLstmArgs& a) {
  const int lane = MID_TID & 63, wave = m4_wave();
  // no branches (waves that do not need an operand fetch it anyway): see bi_load in eqt_kernels.hip
  lstm_project_load<CIN>(g.f[0], a.fwd, wave);
  lstm_project_load<CIN>(g.f[1], a.bwd, wave);
  lstm_load_whh(g.whh, (wave & 1) ? a.bwd : a.fwd);
  mfma_load_a<8>(g.ac, a.wc, 32);
  load4(g.bcv, a.bc + 4 * (lane >> 4));
}

template <int CIN, class Prefetch>
__device__ void mid4_bilstm(const BiLstmArgs& a, Bi4Frags<CIN>& g, const int b, float* P, float* cur, const bool from_memory,
                            Prefetch&& prefetch, unsigned long long* sub) {
  const int tid = MID_TID, lane = tid & 63, wave = m4_wave();
  float* gx = P;                // [2][T * GXS]
  float* hc = P + 2 * T * GXS;  // [32][48] recurrence outputs ...
  float* xs = hc;               // ... over the first stage's input rows [CIN][48], which the projection has read by then
  static_assert(2 * T * GXS + 64 * 48 <= M4_POOL, "BiLSTM stage fits the pool");
  const float* x = cur;
  if (from_memory) {  // first stage: the window's own rows first (loads return in order), then its weights
    constexpr int NX = (CIN * T + M4_NTH - 1) / M4_NTH;
    float xr[NX];
    const float* src = a.src + (long)b * a.ws_src + HALO;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int idx = tid + k * M4_NTH, c = idx / T, t = idx - c * T;
      xr[k] = idx < CIN * T ? src[(long)c * a.ls_src + t] : 0.f;
    }
    bi4_load<CIN>(g, a);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int idx = tid + k * M4_NTH, c = idx / T, t = idx - c * T;
      if (idx < CIN * T) xs[c * 48 + t] = xr[k];
    }
    lds_barrier();  // not __syncthreads(): the weight loads stay in flight
    x = xs;
  }
  M4_SUB(8)
  lstm_project_mfma<CIN>(g.f[0], x, gx, wave);
  lstm_project_mfma<CIN>(g.f[1], x, gx + T * GXS, wave);
  __syncthreads();  // xs is dead from here
  M4_SUB(9)
  if (wave < 2) {   // W_hh has arrived long ago; say so before the new requests queue up behind it (loads complete in order)
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
  M4_SUB(10)
  if (wave >= 1) {  // Conv1d(32,16,1) + BatchNorm, folded
    float* dst = a.dst + (long)b * a.ws_dst;
    const int n0 = 16 * (wave - 1), t = n0 + (lane & 15);
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

// The attention of a four-wave team (mid_attention of eqt_kernels.hip with the q / k projection as one m-tile per wave).
//   x:  [16][48] input rows (column 47 must be finite: it meets the zero column of a in the a.x product)
//   q, k: [T][KP],  e: [48][AES];  `finish(acc, n0)`: what waves 0-2 do with their tile of a.x
template <class Prefetch, class Finish>
__device__ void mid4_attention(const AttnFrag& f, const float wa_lane, const float* x, float (*q)[KP], float (*k)[KP], float* e,
                               const float eps, const int width, Prefetch&& prefetch, Finish&& finish, unsigned long long* sub) {
  const int tid = MID_TID, lane = tid & 63, wave = m4_wave();
  {
    const int mt = wave;  // q rows 0-15, 16-31, k rows 0-15, 16-31
    float(*dstp)[KP] = mt < 2 ? q : k;
    f32x4 acc[3];
    bool big = false;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      acc[n] = mt < 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{f.bias[0], f.bias[1], f.bias[2], f.bias[3]};  // bh belongs to k
      acc[n] = mfma_tile<4>(f.a, x, 48, 16 * n, acc[n]);
#pragma unroll
      for (int r = 0; r < 4; ++r) big |= !(fabsf(acc[n][r]) <= 30.f) && (16 * n + (lane & 15) < T);
    }
    const bool plain = team_vote_or(big);  // only the vote: nothing has been written yet
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int t = 16 * n + (lane & 15);
      if (t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dstp[t][16 * (mt & 1) + 4 * (lane >> 4) + r] = plain ? acc[n][r] : __expf(2.f * acc[n][r]);
      }
    }
    __syncthreads();
    M4_SUB(1)
    // Loads complete in order: whatever the wave still needs from EARLIER requests is taken out of its registers before the
    // new requests go out, or its first use would wait for them as well.
    float wa[32];
    attn_wa(wa, wa_lane);
    __builtin_amdgcn_sched_barrier(0);
    prefetch();
    __builtin_amdgcn_sched_barrier(0);
    M4_SUB(0)
    attn_scores<AES, 8>(q, k, e, wa, plain);
  }
  __syncthreads();
  M4_SUB(2)
  attn_softmax<AES, true>(e, eps, width);
  __syncthreads();
  M4_SUB(3)
  if (wave < 3) {  // vT[c][i] = sum_j x[c][j] a[i][j]: A = x rows out of LDS, B(k = j, n = i) = a[i][j]
    const int n0 = 16 * wave;
    const float* ap = x + (lane & 15) * 48 + (lane >> 4) * 12;          // K index of (lane group g, step ks) = 12 g + ks
    const float* bp = e + (n0 + (lane & 15)) * AES + (lane >> 4) * 12;  // rows 47 .. of the last tile: stale data, column never stored
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ks], bp[ks], acc, 0, 0, 0);
    finish(acc, n0);
  }
  __syncthreads();
  M4_SUB(4)
}

struct Tr4Frags {
  AttnFrag af;
  float wa_lane, ln_par;
  float a1[2][4], b1v[2][4];  // Linear(16,128): m-tiles wave and wave + 4
  float a2[16], b2v[4];       // Linear(128,16): K half = wave / 2; waves 0, 2 the column tiles 0 and 1, waves 1, 3 tile 2
};
// early: what the stage needs up to its score loop (requested during the stage before); late: everything else, requested by
// the stage itself at the start of its score loop (11 k cycles of pure arithmetic) -- a four-wave team holds twice the operands
// of an eight-wave one per wave, and the score loop wants 64 registers of q / k rows in flight
__device__ __forceinline__ void tr4_load_early(Tr4Frags& g, const TransformerArgs& a) {
  const int lane = MID_TID & 63;
  attn_load(g.af, a.att);
  g.wa_lane = a.att.Wa[lane & 31];
}
__device__ __forceinline__ void tr4_load_late(Tr4Frags& g, const TransformerArgs& a) {
  const int lane = MID_TID & 63, wave = m4_wave();
  g.ln_par = a.g1[lane];  // g1 | b1 | g2 | b2 are one blob (eqt_kernels.h)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    mfma_load_a<4>(g.a1[h], a.w1 + (wave + 4 * h) * 16 * EQT_H, EQT_H);
    load4(g.b1v[h], a.bb1 + 16 * (wave + 4 * h) + 4 * (lane >> 4));
  }
  mfma_load_a<16>(g.a2, a.w2 + 64 * (wave >> 1), 128);
  load4(g.b2v, a.bb2 + 4 * (lane >> 4));  // the second K half ignores it at use
}

template <class Prefetch>
__device__ void mid4_transformer(const TransformerArgs& a, Tr4Frags& g, const int b, float* P, float* cur, Prefetch&& prefetch,
                                 unsigned long long* sub) {
  const int tid = MID_TID, lane = tid & 63, wave = m4_wave();
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(P);
  float(*k)[KP] = q + T;
  float* e = P + 2 * T * KP;  // [48][AES]
  float* h1T = P;             // [128][48] hidden layer: over q / k / e once the a.x product has read them
  float* y1T = P + 128 * 48;  // [16][48] LN1 output
  float* rT = y1T + 16 * 48;  // [16][48] the second K half of the second linear layer
  static_assert(2 * T * KP + 48 * AES <= 128 * 48 && 128 * 48 + 2 * 16 * 48 <= M4_POOL, "transformer stage fits the pool");
  M4_SUB(11)
  // y1 = LN1(x + attention(x)) right in the epilogue of the a.x product (three waves, four channels per lane)
  mid4_attention(
      g.af, g.wa_lane, cur, q, k, e, a.attn_eps, 0,
      [&] {
        tr4_load_late(g, a);
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
  M4_SUB(13)
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // FF: Linear(16,128) + ReLU
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      f32x4 acc = {g.b1v[h][0], g.b1v[h][1], g.b1v[h][2], g.b1v[h][3]};
      acc = mfma_tile<4>(g.a1[h], y1T, 48, 16 * nt, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        h1T[(16 * (wave + 4 * h) + 4 * (lane >> 4) + r) * 48 + 16 * nt + (lane & 15)] = fmaxf(acc[r], 0.f);
    }
  }
  __syncthreads();
  M4_SUB(14)
  // Linear(128,16) in two K halves, as the eight-wave team sums them: bias + first half (waves 0, 1), plus the second half
  // (waves 2, 3, handed over through LDS)
  const int half = wave >> 1, nt0 = (wave & 1) ? 2 : 0, ntn = (wave & 1) ? 1 : 2;
  f32x4 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    acc2[j] = half ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{g.b2v[0], g.b2v[1], g.b2v[2], g.b2v[3]};
    if (j < ntn) {
      acc2[j] = mfma_tile<16>(g.a2, h1T + 64 * half * 48, 48, 16 * (nt0 + j), acc2[j]);
      if (half) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rT[(4 * (lane >> 4) + r) * 48 + 16 * (nt0 + j) + (lane & 15)] = acc2[j][r];
      }
    }
  }
  __syncthreads();
  M4_SUB(15)
  if (!half) {  // LN2(y1 + FF(y1)) and the stage's outputs, from the accumulators of the first K half
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j < ntn) {
        const int col = 16 * (nt0 + j) + (lane & 15), c0 = 4 * (lane >> 4);
        float z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = y1T[(c0 + r) * 48 + col] + (acc2[j][r] + rT[(c0 + r) * 48 + col]);
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
    }
  }
  M4_SUB(16)
  __syncthreads();
}

struct Pick4Frags {
  ProjFrag<EQT_H> f[2];  // wave = gate block; [branch]
  f32x2 whh[EQT_H / 2];  // wave 0: P, wave 1: S
  AttnFrag af[2];
  float wa_lane[2];
};
__device__ __forceinline__ void pick4_load(Pick4Frags& g, const PickBranchArgs& a) {
  const int lane = MID_TID & 63, wave = m4_wave();
  lstm_project_load<EQT_H>(g.f[0], a.lstm[0], wave);
  lstm_project_load<EQT_H>(g.f[1], a.lstm[1], wave);
  (void)lane;  // W_hh follows under the stage's projection, both branches' attention operands under its recurrence: the
               // score loop this runs under has no registers to spare
}

__device__ void mid4_pick(const PickBranchArgs& a, Pick4Frags& g, const int b, float* P, const float* cur, unsigned long long* sub) {
  const int tid = MID_TID, wave = m4_wave();
  float* gx = P;                 // [2][T * GXS] input projections of the P and the S branch ...
  float(*q)[KP] = reinterpret_cast<float(*)[KP]>(P);  // ... and, once the recurrences have read them, q / k / e
  float(*k)[KP] = q + T;
  float* e = P + 2 * T * KP;     // [48][AES]
  float* hl = P + 2 * T * GXS;   // [2][16][48] LSTM outputs of the P and the S branch
  float* vT = hl + 2 * 16 * 48;  // [16][48]
  static_assert(2 * T * KP + 48 * AES <= 2 * T * GXS && 2 * T * GXS + 3 * 16 * 48 <= M4_POOL, "pick stage fits the pool");
  M4_SUB(17)
  if (tid < 32) hl[tid * 48 + 47] = 0.f;  // K padding of the a.x products
  lstm_load_whh(g.whh, (wave & 1) ? a.lstm[1] : a.lstm[0]);  // every wave: no branch, see bi_load
  lstm_project_mfma<EQT_H>(g.f[0], cur, gx, wave);
  lstm_project_mfma<EQT_H>(g.f[1], cur, gx + T * GXS, wave);
  __syncthreads();
  M4_SUB(18)
  if (wave < 2) {
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) asm volatile("" ::"v"(g.whh[j]));  // see mid4_bilstm
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    attn_load(g.af[br], a.att[br]);
    g.wa_lane[br] = a.att[br].Wa[tid & 31];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (wave < 2) {
    const float sc = lstm_gate_scale(tid & 3);
#pragma unroll
    for (int j = 0; j < EQT_H / 2; ++j) g.whh[j] *= sc;
    lstm_recur<GXS, true>(gx + wave * T * GXS, g.whh, false, hl + wave * 16 * 48, 48);
  }
  __syncthreads();  // gx is dead from here
  M4_SUB(19)
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    mid4_attention(g.af[br], g.wa_lane[br], hl + br * 16 * 48, q, k, e, a.attn_eps, a.width, NoPrefetch(),
                   [&](const f32x4 acc, const int n0) {
                     const int i = n0 + (tid & 15);
                     if (i < T) {
#pragma unroll
                       for (int r = 0; r < 4; ++r) vT[(4 * ((tid & 63) >> 4) + r) * 48 + i] = acc[r];
                     }
                   },
                   sub && br ? sub + 26 : nullptr);
    float* up = a.up + (long)((1 + br) * a.B + b) * a.ws_up + HALO;
    for (int idx = tid; idx < EQT_H * 48; idx += M4_NTH) {
      const int c = idx / 48, t = idx - c * 48;
      if (t < T) up[(long)c * a.ls_up + t] = vT[c * 48 + t];
    }
    __syncthreads();
  }
}

// The four teams execute the same barriers in the same order (same code, same trip counts); the last workgroup of a batch that
// is not a multiple of four computes its last window more than once (identical stores).
__global__ __launch_bounds__(M4_WPB* M4_NTH) void eqt_mid4_kernel(const MidArgs a) {
  __shared__ __attribute__((aligned(16))) float P_all[M4_WPB * M4_POOL];
  __shared__ float cur_all[M4_WPB * 16 * 48];
  const int team = __builtin_amdgcn_readfirstlane(MID_TEAM);
  float* P = P_all + team * M4_POOL;
  float* cur = cur_all + team * 16 * 48;
  const int b = min((int)blockIdx.x * M4_WPB + team, a.B - 1);
  int stamp = 0;
  unsigned long long* sub = a.clk ? a.clk + (long)b * 32 : nullptr;
  if (MID_TID < 16) cur[MID_TID * 48 + 47] = 0.f;  // K padding of the a.x products; no stage writes column 47
  {  // every cache line of the argument block requested at once (see eqt_mid_kernel)
    typedef const unsigned __attribute__((address_space(4))) * uptr_t;
    const uptr_t ka = (uptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (unsigned i = 0; i < (sizeof(MidArgs) + 63) / 64; ++i) acc |= ka[16 * i];
    asm volatile("" ::"s"(acc));
  }
#define M4_STAMP()                                                                          \
  if (a.clk && MID_TID == 0) a.clk[(long)b * 32 + stamp] = __builtin_readcyclecounter(); \
  ++stamp;
  M4_STAMP()
  // each stage requests the next one's weights under its own longest arithmetic phase
  Bi4Frags<64> g0;
  Bi4Frags<EQT_H> g1, g2;
  Tr4Frags t0, t1;
  Pick4Frags pf;
  mid4_bilstm<64>(a.lstm[0], g0, b, P, cur, true, [&] { bi4_load<EQT_H>(g1, a.lstm[1]); }, sub);
  M4_STAMP()
  mid4_bilstm<EQT_H>(a.lstm[1], g1, b, P, cur, false, [&] { bi4_load<EQT_H>(g2, a.lstm[2]); }, nullptr);
  M4_STAMP()
  mid4_bilstm<EQT_H>(a.lstm[2], g2, b, P, cur, false, [&] { tr4_load_early(t0, a.tr[0]); }, nullptr);
  M4_STAMP()
  mid4_transformer(a.tr[0], t0, b, P, cur, [&] { tr4_load_early(t1, a.tr[1]); }, nullptr);
  M4_STAMP()
  mid4_transformer(a.tr[1], t1, b, P, cur, [&] { pick4_load(pf, a.pick); }, sub);
  M4_STAMP()
  mid4_pick(a.pick, pf, b, P, cur, sub);
  M4_STAMP()
#undef M4_STAMP
}

}  // namespace

int launch_eqt_mid4(const MidArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(eqt_mid4_kernel, dim3((B + M4_WPB - 1) / M4_WPB), dim3(M4_WPB * M4_NTH), 0, s, a);
  return 0;
}

}  // namespace vp
