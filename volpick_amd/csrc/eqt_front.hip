// EQTransformer encoder stages 0-2 (Conv1d + ReLU + MaxPool1d(2) each: 3 x 6000 -> 8 x 3000 -> 16 x 1500 -> 16 x 750)
// as ONE launch per TIME TILE of a window.
//
// As three conv_mfma_kernel launches these stages took 67 us per 256 windows for 28 us of MFMA issue: their K is short
// (48 / 72 / 112), so a launch is mostly its load phase, its pooled store and the kernel boundary, and the 8 x 3000 and
// 16 x 1500 rows made a round trip through memory (0.1 GB per step).  Here a workgroup reads 3 x 2060 input samples,
// runs the three stages LDS to LDS with the halo each needs recomputed (1-2 % extra MFMA work) and writes 16 x 250
// pooled samples of stage 2.
//
//   tile of stage-2 pooled outputs [j0, j0 + 250) of the 750-sample row, j0 = 250 i:
//     stage 2 columns  c2 = 2 j0      + [0,  500 /  512)   reads stage-1 pooled [c2 - 3, c2 + 3]
//     stage 1 columns  c1 = 4 j0 - 6  + [0, 1012 / 1024)   reads stage-0 pooled [c1 - 4, c1 + 4]
//     stage 0 columns  n0 = 4 j0 - 10 + [0, 1020 / 1024)   (two conv outputs 2 n0, 2 n0 + 1 per column = one pooled
//                                                           sample) reads input samples [2 n0 - 5, 2 n0 + 6]
//   Conv outputs outside a row's signal are zeroed before pooling: ReLU outputs are >= 0, so that is MaxPool's -1e10 pad
//   for the odd tail and at the same time the next stage's zero padding.
//
// One 512-thread workgroup (8 wavefronts) per CU, persistent over the tiles.  All three stages have M = 16 (one m-tile):
// the 12 + 18 + 28 A fragments stay in registers for the whole kernel and every wave takes one block of 8 / 8 / 4
// n-tiles per stage.  MaxPool: stage 0 pools the two output phases of a column (two registers of a lane); stages 1 and 2
// pool neighbouring columns = neighbouring lanes (one DPP move) and the even lane stores.
// Same packed fragments, same K order, same max / ReLU arithmetic: bit-identical to the launches it replaces
// (plan flag plan_flags[7] & 4 keeps them).
#include "conv_b3.h"
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"
#include "prepost.h"

namespace vp {

namespace {

constexpr int FR_NTH = 512, FR_WAVES = 8;
constexpr int FW = 250, FR_TILES = 3;          // stage-2 pooled samples per tile; tiles per window
constexpr int T_IN = 6000, L0P = 3000, L1P = 1500;  // row lengths: input, pooled stage 0 (= conv length of stage 1), pooled stage 1
//                       CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF NB RELU
using F_e0 = LdsLayer<3, 0, 8, 2, 12, 2, -5, 0, 8, 1>;
using F_e1 = LdsLayer<8, 0, 16, 1, 9, 1, -4, 0, 8, 1>;
using F_e2 = LdsLayer<16, 0, 16, 1, 7, 1, -3, 0, 4, 1>;
constexpr int C0 = 1024, C1 = 1024, C2 = 512;  // MFMA columns per stage
// images: the input with SN = 2 wants a row stride == 17 mod 32 (lanes read every second word: even banks for one
// channel row of a half-wave, odd banks for the other), the others == 16 mod 32
constexpr int SI = 2065, BII = 8, S0P = 1040, S1P = 528, BI = 4;
static_assert(SI % 32 == 17 && S0P % 32 == 16 && S1P % 32 == 16, "bank-conflict-free strides");
static_assert(SI >= BII + 2 * (C0 - 1) + 12 - 5 && S0P >= BI + C1 + 4 && S0P >= C0 && S1P >= BI + C2 + 3 && S1P >= 1 + C1 / 2,
              "image widths");
constexpr int OFFI = 0, OFF0P = 4 * SI + 4 - (4 * SI) % 4, OFF1P = OFF0P + 8 * S0P, FR_LDS_FLOATS = OFF1P + 16 * S1P;
static_assert(FR_LDS_FLOATS * 4 <= 160 * 1024 && OFF0P % 4 == 0 && OFF1P % 4 == 0, "LDS budget");
constexpr int PRE = (3 * SI + FR_NTH - 1) / FR_NTH;  // input samples a thread carries for the next tile
// B3 instantiation (default): stages 1 and 2 (73 % of the tile's MFMA issue) on the bf16 matrix cores with exact three-piece
// operands (conv_b3.h).  Stage 0 (fp32 MFMA: 3 input channels) writes its pooled output as a chunk-plane piece image
// [piece][column][8 channels] (column = its MFMA column), stage 1 packs four taps of its 8 channels into a K = 32 step
// (9 taps -> 3 steps) and writes [piece][2 chunks][column][8 channels] (column = pooled index), stage 2 packs two taps of
// 16 channels (7 -> 4 steps) and writes the pooled row to memory.  The columns behind what a stage writes stay zero from
// the one fill at kernel start (nothing is aliased); every column a kept output touches, padded taps included, is written
// in the same tile.
constexpr int NC0 = 1040, NC1 = 528;
using QE0 = B3Chunk<8, NC0>;
using QE1 = B3Chunk<16, NC1>;
static_assert(NC0 >= C1 + 12 && NC1 >= C2 + 8, "every column stages 1 / 2 read has a place");
static_assert(1012 + 11 <= C0 && 500 + 7 <= C1 / 2, "kept outputs touch only columns written in the same tile");
constexpr int FB_OFF_E0 = (4 * SI + 4) * 4 + 16 - ((4 * SI + 4) * 4) % 16, FB_OFF_E1 = FB_OFF_E0 + 3 * QE0::PS * 2;
constexpr int FB_LDS_BYTES = FB_OFF_E1 + 3 * QE1::PS * 2;
static_assert(FB_LDS_BYTES <= 160 * 1024 && FB_OFF_E0 % 16 == 0 && FB_OFF_E1 % 16 == 0, "LDS budget of the bf16-piece variant");

struct FrontArgs {
  const float* x;  // input rows [B][3][ls]
  int ls_x;
  long ws_x;
  float* y;        // encoder.2 rows [B][16][ls]
  int ls_y;
  long ws_y;
  const float* af[3];  // packed A fragments [CB][TAPS][64]
  const float* bs[3];
  const uint4* af3[2];  // B3: three-piece operands of stages 1 and 2 [steps][piece][64] (net.hip: b3_operand)
  int n_tiles;
  PreArgs pre;  // has_pre: the kernel cuts its windows out of the raw stream and normalises them itself (annotate_batch_pre,
  int has_pre;  // arithmetic and reduction order of gather_normalize_kernel: bitwise the same rows); x is then unused
  int B;
};

// annotate_batch_pre statistics of one window by a 512-thread workgroup, in the arithmetic AND reduction order of
// gather_normalize_kernel (prepost.hip; 1024 threads, thread t sums samples t, t + 1024, ...; one partial per wavefront,
// the 16 partials added in order): a thread stands in for the "virtual" threads tid and tid + 512, its wavefront w for the
// virtual wavefronts w and w + 8.  -> mean[c], den[c] = amplitude + eps, bad = a non-finite sample in the window.
// Results in LDS: out[0..2] = mean, out[3..5] = den, out[6] = 1.0 if bad (read back by the threads that park a tile).
__device__ __forceinline__ void front_window_stats(const PreArgs& p, const float* src, const long cs, float* red, float* out) {
  constexpr int VT = 1024, NWV = 16, MAXE = 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = p.T;
  float* stat = red + 3 * NWV;
  if (p.norm == VP_NORM_PEAK) {
    // ONE pass: max_k |v_k - mean| = max(vmax - mean, mean - vmin) bit for bit (rounding is monotonic and symmetric), so
    // the sums, maxima and minima are gathered together.  (A NaN / Inf sample makes the mean non-finite: the window is
    // flagged and its predictions become NaN whatever the amplitude says.)
    float s[2][3], hi[3] = {-INFINITY, -INFINITY, -INFINITY}, lo[3] = {INFINITY, INFINITY, INFINITY};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v[MAXE];
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
          const int t = tid + 512 * h + k * VT;
          v[k] = t < T ? src[c * cs + t] : 0.f;
        }
        s[h][c] = 0.f;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
          s[h][c] += v[k];
          if (tid + 512 * h + k * VT < T) hi[c] = fmaxf(hi[c], v[k]), lo[c] = fminf(lo[c], v[k]);
        }
      }
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float r = wave_sum(s[h][c]);
        if (lane == 0) red[c * NWV + wave + 8 * h] = r;
      }
      const float rh = wave_max(hi[c]), rl = -wave_max(-lo[c]);
      if (lane == 0) red[3 * NWV + 8 + c * 16 + wave] = rh, red[3 * NWV + 8 + c * 16 + 8 + wave] = rl;
    }
    __syncthreads();
    if (tid == 0) {
      bool bad = false;
      float amp[3], mean[3];
      for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int i = 0; i < NWV; ++i) acc += red[c * NWV + i];
        mean[c] = acc / (float)T;
        float vh = red[3 * NWV + 8 + c * 16], vl = red[3 * NWV + 8 + c * 16 + 8];
        for (int i = 1; i < 8; ++i) vh = fmaxf(vh, red[3 * NWV + 8 + c * 16 + i]), vl = fminf(vl, red[3 * NWV + 8 + c * 16 + 8 + i]);
        amp[c] = fmaxf(vh - mean[c], mean[c] - vl);
        bad |= !isfinite(mean[c]) || !isfinite(amp[c]);
      }
      if (!p.per_comp) amp[0] = amp[1] = amp[2] = fmaxf(amp[0], fmaxf(amp[1], amp[2]));
      for (int c = 0; c < 3; ++c) {
        out[c] = mean[c];
        out[3 + c] = amp[c] + p.norm_eps;
      }
      out[6] = bad ? 1.f : 0.f;
    }
    __syncthreads();
    return;
  }
  // norm = std: the window is read twice, for the mean and for the sum of squares (36 samples per thread held across the
  // barriers would not fit beside the kernel's resident weights; the second read hits L2)
  float s[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v[MAXE];
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int t = tid + 512 * h + k * VT;
        v[k] = t < T ? src[c * cs + t] : 0.f;
      }
      s[h][c] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) s[h][c] += v[k];
    }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    for (int c = 0; c < 3; ++c) {
      const float r = wave_sum(s[h][c]);
      if (lane == 0) red[c * NWV + wave + 8 * h] = r;
    }
  __syncthreads();
  if (tid < 3) {
    float acc = 0.f;
    for (int i = 0; i < NWV; ++i) acc += red[tid * NWV + i];
    stat[tid * 2] = acc / (float)T;
  }
  __syncthreads();
  const float mean[3] = {stat[0], stat[2], stat[4]};
  float m[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      m[h][c] = 0.f;
      float v[MAXE];
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int t = tid + 512 * h + k * VT;
        v[k] = t < T ? src[c * cs + t] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int t = tid + 512 * h + k * VT;
        if (t < T) {
          const float d = v[k] - mean[c];
          if (p.norm == VP_NORM_PEAK) {
            m[h][c] = fmaxf(m[h][c], fabsf(d));
            if (d != d) m[h][c] = d;  // propagate NaN like torch.max
          } else {
            m[h][c] += d * d;
          }
        }
      }
    }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h)
    for (int c = 0; c < 3; ++c) {
      const float r = (p.norm == VP_NORM_PEAK) ? wave_max(m[h][c]) : wave_sum(m[h][c]);
      if (lane == 0) red[c * NWV + wave + 8 * h] = r;
    }
  __syncthreads();
  if (tid < 3) {
    const float* r = red + tid * NWV;
    float acc = r[0];
    for (int i = 1; i < NWV; ++i) acc = (p.norm == VP_NORM_PEAK) ? fmaxf(acc, r[i]) : acc + r[i];
    stat[tid * 2 + 1] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    bool bad = false;
    for (int c = 0; c < 3; ++c) bad |= !isfinite(stat[2 * c]) || !isfinite(stat[2 * c + 1]);
    float amp[3];
    if (p.per_comp) {
      for (int c = 0; c < 3; ++c) amp[c] = (p.norm == VP_NORM_PEAK) ? stat[2 * c + 1] : sqrtf(stat[2 * c + 1] / (float)(T - 1));
    } else {
      const float g = (p.norm == VP_NORM_PEAK) ? fmaxf(stat[1], fmaxf(stat[3], stat[5]))
                                               : sqrtf((stat[1] + stat[3] + stat[5]) / (float)(3 * T - 1));
      amp[0] = amp[1] = amp[2] = g;
    }
    for (int c = 0; c < 3; ++c) {
      out[c] = mean[c];
      out[3 + c] = amp[c] + p.norm_eps;
    }
    out[6] = bad ? 1.f : 0.f;
  }
  __syncthreads();  // the statistics are visible; `red` is free for the next window
}

__device__ __forceinline__ float lane_xor1(float v) { return dpp_move<0xB1, 0xF>(v); }  // quad_perm [1,0,3,2]

// Stage 0: the two phases of a column (registers r, r + 1 of a lane) are the conv outputs 2 n0, 2 n0 + 1 = one pooled sample.
struct Pool0Store {
  static constexpr bool custom_block_epilogue = true;
  float* img;      // image + shift: pooled sample of column c at img[co * S0P + c]
  int s_lo;        // pooled sample index of column 0
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 2 && L::MT == 1 && L::RELU == 1, "stage 0 of the encoder");
    (void)mt;
    const bool fast = (unsigned)(s_lo + colb) < (unsigned)L0P && (unsigned)(s_lo + colb + L::NB * 16 - 1) < (unsigned)L0P;
#pragma unroll
    for (int rr = 0; rr < 4; rr += 2) {
      float* row = img + (2 * g + rr / 2) * S0P + colb + n;
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        float v = fmaxf(fmaxf(acc[j][rr] + biasv[rr], 0.f), fmaxf(acc[j][rr + 1] + biasv[rr + 1], 0.f));
        if (!fast) v = ((unsigned)(s_lo + colb + j * 16 + n) < (unsigned)L0P) ? v : 0.f;
        row[j * 16] = v;
      }
    }
  }
};

// B3, stage 0: the same pooling, the pooled sample of column c of channels 2 g, 2 g + 1 as a pair of bfloat16 per piece
struct Pool0Pieces {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  int s_lo;
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 2 && L::MT == 1 && L::RELU == 1, "stage 0 of the encoder");
    (void)mt;
    const bool fast = (unsigned)(s_lo + colb) < (unsigned)L0P && (unsigned)(s_lo + colb + L::NB * 16 - 1) < (unsigned)L0P;
    bf16_t* q = img + (colb + n) * 8 + 2 * g;
#pragma unroll
    for (int j = 0; j < L::NB; ++j) {
      float v0 = fmaxf(fmaxf(acc[j][0] + biasv[0], 0.f), fmaxf(acc[j][1] + biasv[1], 0.f));
      float v1 = fmaxf(fmaxf(acc[j][2] + biasv[2], 0.f), fmaxf(acc[j][3] + biasv[3], 0.f));
      if (!fast && !((unsigned)(s_lo + colb + j * 16 + n) < (unsigned)L0P)) v0 = v1 = 0.f;
      const unsigned h = pack_bf16x2(v0, v1);
      const float r0 = v0 - bf16_lo(h), r1 = v1 - bf16_hi(h);
      const unsigned m = pack_bf16x2(r0, r1);
      const unsigned l = pack_bf16x2(r0 - bf16_lo(m), r1 - bf16_hi(m));
      *reinterpret_cast<unsigned*>(q + j * 128) = h;
      *reinterpret_cast<unsigned*>(q + j * 128 + QE0::PS) = m;
      *reinterpret_cast<unsigned*>(q + j * 128 + 2 * QE0::PS) = l;
    }
  }
};

// Stages 1 / 2: neighbouring columns (lanes n, n ^ 1) pool; conv outputs outside the signal count as zero.
template <bool TO_MEMORY>
struct Pool1Store {
  static constexpr bool custom_block_epilogue = true;
  float* img;      // LDS: pooled sample of columns (c, c + 1) at img[co * stride + c / 2]; memory: row base of channel 0
  int stride;
  int cs_lo;       // conv sample index of column 0
  unsigned len;    // conv row length
  int cols;        // columns that exist (TO_MEMORY: the tile's share)
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 1 && L::MT == 1 && L::RELU == 1, "stages 1 / 2 of the encoder");
    (void)mt;
    const bool fast = (unsigned)(cs_lo + colb) < len && (unsigned)(cs_lo + colb + L::NB * 16 - 1) < len &&
                      colb + L::NB * 16 <= cols;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* row = img + (long)(4 * g + r) * stride + ((colb + n) >> 1);
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        float v = fmaxf(acc[j][r] + biasv[r], 0.f);
        if (!fast) v = ((unsigned)(cs_lo + colb + j * 16 + n) < len) ? v : 0.f;
        const float m = fmaxf(v, lane_xor1(v));
        if (!(n & 1) && (fast || colb + j * 16 + n < cols)) row[j * 8] = m;
      }
    }
  }
};

// CUT: the kernel cuts its windows out of the raw stream and normalises them itself (FrontArgs::pre).
template <bool B3, bool CUT>
__global__ __launch_bounds__(FR_NTH) void eqt_front_kernel(const FrontArgs a) {
  extern __shared__ float4 fr_lds_raw[];
  float* lds = reinterpret_cast<float*>(fr_lds_raw);
  int off0 = OFF0P / 4, off1 = OFF1P / 4;  // opaque image offsets (eqt_tail.hip)
  asm volatile("" : "+v"(off0), "+v"(off1));
  float* XI = lds + OFFI;
  float* E0P = lds + 4 * off0;
  float* E1P = lds + 4 * off1;
  bf16_t* E0 = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(fr_lds_raw) + FB_OFF_E0);  // B3
  bf16_t* E1 = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(fr_lds_raw) + FB_OFF_E1);
  __shared__ float fr_red[3 * 16 + 8 + 3 * 16];  // CUT: partial sums (+ maxima / minima) of front_window_stats
  __shared__ float fr_stat[8];          //      the window's mean[3], den[3], non-finite flag
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Tile order.  Reading the input tensor: tiles blockIdx.x, + gridDim.x, ... (any tile to any workgroup).  Cutting the
  // windows itself (has_pre): window-major -- workgroup b takes the FR_TILES tiles of windows b, b + gridDim.x, ... and
  // computes a window's statistics once, in front of its first tile.
  constexpr bool has_pre = CUT;
  int seq = 0;
  auto tile_at = [&](const int q) {
    return has_pre ? ((int)blockIdx.x + (q / FR_TILES) * (int)gridDim.x) * FR_TILES + q % FR_TILES : (int)blockIdx.x + q * (int)gridDim.x;
  };
  int tile = tile_at(0);
  if (tile >= a.n_tiles) return;
  // the fourth channel row of the input image pads K to the 4-channel MFMA step: zero, once
  for (int i = tid; i < SI + 4; i += FR_NTH) XI[3 * SI + i] = 0.f;
  if constexpr (B3) {  // the piece images: the columns no stage writes are read (zero weights, unkept outputs) and must be finite
    uint4* z = reinterpret_cast<uint4*>(reinterpret_cast<char*>(fr_lds_raw) + FB_OFF_E0);
    for (int i = tid; i < (FB_LDS_BYTES - FB_OFF_E0) / 16; i += FR_NTH) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }

  // input samples of a tile: 3 rows x the SI physical columns of the image; column p <-> sample 8 j0 - 28 + p of the row,
  // zero outside [0, 6000)
  float pre[PRE];
  // has_pre: where window `win` of this launch starts in the raw stream (PreArgs, as gather_normalize_kernel) and its row stride
  auto raw_window = [&](const int win, long& cs) -> const float* {
    const PreArgs& p = a.pre;
    long start = p.dense ? 0 : (long)(p.first_window + win) * p.step;
    if (!p.dense && start > p.N - T_IN) start = p.N - T_IN;  // tail window flush with the end
    const float* src = p.src + (p.dense ? (long)win * 3 * T_IN : start);
    cs = p.dense ? T_IN : p.N;
    if (p.table) {
      const long* e = p.table + 3 * (p.first_window + win);
      src = p.src + e[0] + e[2];
      cs = e[1];
    }
    return src;
  };
  auto request = [&](int t) {
    const int win = t / FR_TILES, j0 = (t - win * FR_TILES) * FW;
    long cs = a.ls_x;
    const float* src = a.x + (long)win * a.ws_x + HALO;
    if constexpr (has_pre) src = raw_window(win, cs);
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * FR_NTH, c = idx / SI, p = idx - c * SI;
      const int smp = 8 * j0 - 28 + p;
      pre[k] = (idx < 3 * SI && (unsigned)smp < (unsigned)T_IN) ? src[(long)c * cs + smp] : 0.f;
    }
  };
  auto park = [&](const int j0) {
    if constexpr (has_pre) {  // (v - mean) / (amp + eps), tapered: the expression of gather_normalize_kernel, sample by sample
      const int taper = a.pre.taper;
      const float m0 = fr_stat[0], m1 = fr_stat[1], m2 = fr_stat[2];
      const NormDiv n0 = norm_div_prepare(fr_stat[3]), n1 = norm_div_prepare(fr_stat[4]), n2 = norm_div_prepare(fr_stat[5]);
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const int idx = tid + k * FR_NTH, c = idx / SI, p = idx - c * SI;
        const int t = 8 * j0 - 28 + p;
        if (idx < 3 * SI) {
          float o = 0.f;
          if ((unsigned)t < (unsigned)T_IN) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
            const NormDiv den = c == 0 ? n0 : (c == 1 ? n1 : n2);
            o = norm_div(pre[k] - mean, den);
            if (taper > 0) {
              const int e = (t < taper) ? t : ((T_IN - 1 - t < taper) ? T_IN - 1 - t : -1);
              if (e >= 0) {  // 0.5 * (1 + cos(linspace(pi, 2 pi, taper)[e]))
                const float ang = 3.14159265358979323846f * (1.f + (float)e / (float)(taper - 1));
                o *= 0.5f * (1.f + cosf(ang));
              }
            }
          }
          XI[idx] = o;
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int idx = tid + k * FR_NTH;
      if (idx < 3 * SI) XI[idx] = pre[k];  // rows are contiguous: image index == idx
    }
  };
  request(tile);

  // one m-tile per stage, the same weights for every window: all A fragments live in registers for the whole kernel
  float areg0[F_e0::CB * F_e0::TAPS], areg1[F_e1::CB * F_e1::TAPS], areg2[F_e2::CB * F_e2::TAPS];
  float bias0[4], bias1[4], bias2[4];
  load_areg<F_e0>(a.af[0], 0, lane, areg0);
  load_areg<F_e1>(a.af[1], 0, lane, areg1);
  load_areg<F_e2>(a.af[2], 0, lane, areg2);
  load_biasreg<F_e0>(a.bs[0], 0, lane, bias0);
  load_biasreg<F_e1>(a.bs[1], 0, lane, bias1);
  load_biasreg<F_e2>(a.bs[2], 0, lane, bias2);
  uint4 a1[B3Steps<8, 9>::STEPS * 3], a2[B3Steps<16, 7>::STEPS * 3];  // B3: the operands of stages 1 and 2, for the whole kernel
  float b1v[4], b2v[4];
  if constexpr (B3) {
    b3_load_a<8, 9>(a.af3[0], 0, lane, a1);
    b3_load_a<16, 7>(a.af3[1], 0, lane, a2);
#pragma unroll
    for (int r = 0; r < 4; ++r) b1v[r] = a.bs[1][4 * (lane >> 4) + r], b2v[r] = a.bs[2][4 * (lane >> 4) + r];
    __syncthreads();  // the zero fill of the piece images
  }

  while (true) {
    const int next = tile_at(++seq);
    const bool more = next < a.n_tiles;
    const int win = tile / FR_TILES, j0 = (tile - win * FR_TILES) * FW;
    if (has_pre && j0 == 0) {  // (CUT) first tile of a window: its statistics (uniform branch; ends with a barrier)
      long cs;
      const float* src = raw_window(win, cs);
      front_window_stats(a.pre, src, cs, fr_red, fr_stat);
      if (tid == 0 && a.pre.flags) a.pre.flags[win] = fr_stat[6];  // the tail kernel turns the window's predictions into NaN
    }
    park(j0);
    __syncthreads();
    if constexpr (B3) {
      const int n = lane & 15, g = lane >> 4;
      {  // stage 0 (fp32 MFMA): column c <-> pooled sample 4 j0 - 10 + c -> column c of the stage-1 piece image
        Pool0Pieces st{E0, 4 * j0 - 10};
        conv_lds_areg<F_e0, SI, BII, SI, BII>(XI, XI, areg0, bias0, 0, C0, st, wave_u, FR_WAVES, lane);
      }
      __syncthreads();
      if (more) request(next);  // travels under stages 1 and 2
      __builtin_amdgcn_sched_barrier(0);
      {  // stage 1: column c <-> conv sample 4 j0 - 6 + c reads image columns c .. c + 8; pooled -> column c / 2 of the stage-2 image
        const int colb = wave_u * 128, s_lo = 4 * j0 - 6;
        b3c_mac_tiles<8, NC0, 9, 8>(b3c_lane_ptr<8, NC0, 9>(E0, colb, lane), a1, [&](const int j, const f32x4 acc) {
          const int c = colb + j * 16 + n;
          const bool in = (unsigned)(s_lo + c) < (unsigned)L0P;
          float m[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = in ? fmaxf(acc[r] + b1v[r], 0.f) : 0.f;
            m[r] = fmaxf(v, lane_xor1(v));
          }
          if (!(n & 1)) b3c_store4<16, NC1>(E1, c >> 1, g, m);
        });
      }
      __syncthreads();
      {  // stage 2: column c <-> conv sample 2 j0 + c reads image columns c .. c + 6; pooled sample j0 + c / 2 of the encoder.2 row
        const int colb = wave_u * 64;
        float* y = a.y + (long)win * a.ws_y + HALO + j0 + (long)(4 * g) * a.ls_y;
        b3c_mac_tiles<16, NC1, 7, 4>(b3c_lane_ptr<16, NC1, 7>(E1, colb, lane), a2, [&](const int j, const f32x4 acc) {
          const int c = colb + j * 16 + n;
          const bool in = (unsigned)(2 * j0 + c) < (unsigned)L1P;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = in ? fmaxf(acc[r] + b2v[r], 0.f) : 0.f;
            const float m = fmaxf(v, lane_xor1(v));
            if (!(n & 1) && c < 2 * FW) y[(long)r * a.ls_y + (c >> 1)] = m;
          }
        });
      }
    } else {
    {  // stage 0: column c <-> pooled sample 4 j0 - 10 + c = stage-1 logical column c - 4
      Pool0Store st{E0P + BI - 4, 4 * j0 - 10};
      conv_lds_areg<F_e0, SI, BII, SI, BII>(XI, XI, areg0, bias0, 0, C0, st, wave_u, FR_WAVES, lane);
    }
    __syncthreads();
    if (more) request(next);  // travels under stages 1 and 2
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 1: column c <-> conv sample 4 j0 - 6 + c; pooled sample 2 j0 - 3 + c / 2 = stage-2 logical column c / 2 - 3
      Pool1Store<false> st{E1P + BI - 3, S1P, 4 * j0 - 6, (unsigned)L0P, C1};
      conv_lds_areg<F_e1, S0P, BI, S0P, BI>(E0P, E0P, areg1, bias1, 0, C1, st, wave_u, FR_WAVES, lane);
    }
    __syncthreads();
    {  // stage 2: column c <-> conv sample 2 j0 + c; pooled sample j0 + c / 2 of the window's encoder.2 row
      Pool1Store<true> st{a.y + (long)win * a.ws_y + HALO + j0, a.ls_y, 2 * j0, (unsigned)L1P, 2 * FW};
      conv_lds_areg<F_e2, S1P, BI, S1P, BI>(E1P, E1P, areg2, bias2, 0, C2, st, wave_u, FR_WAVES, lane);
    }
    }
    if (!more) break;
    tile = next;
    // no barrier: the next park writes the input image, whose last readers (stage 0) are two barriers back
  }
}

}  // namespace

// Replaces the steps "encoder.0", "encoder.1", "encoder.2" of the plan by one fused step.
int plan_eqt_fuse_front(Net& net, bool b3) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "encoder.0") first = (int)i;
  if (first < 0 || first + 3 > (int)net.steps.size() || net.steps[first + 2].name != "encoder.2") {
    set_error("fused encoder front: layer plan not found");
    return VP_ERR_INVALID;
  }
  ConvLayer* c[3] = {nullptr, nullptr, nullptr};
  for (auto& l : net.convs)
    for (int i = 0; i < 3; ++i)
      if (l->name == "encoder." + std::to_string(i)) c[i] = l.get();
  if (!c[0] || !c[1] || !c[2]) {
    set_error("fused encoder front: conv layers missing");
    return VP_ERR_INVALID;
  }
  HostBlob* p3[2] = {nullptr, nullptr};
  if (b3) {
    if (c[1]->g.cinp() != 8 || c[1]->g.taps != 9 || c[1]->g.M() != 16 || c[2]->g.cinp() != 16 || c[2]->g.taps != 7 || c[2]->g.M() != 16) {
      set_error("fused encoder front: unexpected layer shape");
      return VP_ERR_INVALID;
    }
    for (int i = 0; i < 2; ++i) p3[i] = net.add_blob(b3_operand(*c[1 + i], false));
  }
  const int x_in = c[0]->src1, y_out = c[2]->dst;
  net.tensor_sets[c[0]->dst] = 0;  // encoder.0 / .1 live in LDS under this plan
  net.tensor_sets[c[1]->dst] = 0;
  Step st;
  st.name = "fused.front (encoder.0-2, time-tiled)";
  st.flops_per_window = 0;
  for (int i = 0; i < 3; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  // per tile: stage 0 64 n-tiles x 12 K-steps of fp32 MFMAs; stages 1 / 2 64 x 18 and 32 x 28 fp32 MFMAs, or (bf16 pieces) 64 n-tiles x
  // 3 K-steps and 32 x 4 K-steps of six-MFMA groups
  if (b3)
    st.set_issued(FR_TILES * 64.0 * 12 * 2048.0, FR_TILES * (64.0 * B3Steps<8, 9>::STEPS + 32.0 * B3Steps<16, 7>::STEPS) * 6 * 16384.0, 0.0);
  else
    st.set_issued(FR_TILES * (64.0 * 12 + 64.0 * 18 + 32.0 * 28) * 2048.0, 0.0, 0.0);
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    FrontArgs a{};
    const Tensor &tx = n.tensors[x_in], &ty = n.tensors[y_out];
    a.x = tx.p;
    a.ls_x = tx.ls;
    a.ws_x = (long)tx.win_stride();
    a.y = ty.p;
    a.ls_y = ty.ls;
    a.ws_y = (long)ty.win_stride();
    for (int i = 0; i < 3; ++i) {
      a.af[i] = c[i]->afrag.d;
      a.bs[i] = c[i]->bias.d;
    }
    a.n_tiles = B * FR_TILES;
    a.B = B;
    if (n.pre) {
      a.pre = *n.pre;
      a.has_pre = 1;
    }
    const int grid = a.has_pre ? (B < 256 ? B : 256) : (a.n_tiles < 256 ? a.n_tiles : 256);
    if (b3) {
      for (int i = 0; i < 2; ++i) a.af3[i] = reinterpret_cast<const uint4*>(p3[i]->d);
      if (a.has_pre)
        hipLaunchKernelGGL((eqt_front_kernel<true, true>), dim3(grid), dim3(FR_NTH), FB_LDS_BYTES, s, a);
      else
        hipLaunchKernelGGL((eqt_front_kernel<true, false>), dim3(grid), dim3(FR_NTH), FB_LDS_BYTES, s, a);
    } else {
      if (a.has_pre)
        hipLaunchKernelGGL((eqt_front_kernel<false, true>), dim3(grid), dim3(FR_NTH), FR_LDS_FLOATS * sizeof(float), s, a);
      else
        hipLaunchKernelGGL((eqt_front_kernel<false, false>), dim3(grid), dim3(FR_NTH), FR_LDS_FLOATS * sizeof(float), s, a);
    }
    return 0;
  };
  if (b3) {
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_front_kernel<true, true>), (size_t)FB_LDS_BYTES});
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_front_kernel<true, false>), (size_t)FB_LDS_BYTES});
  } else {
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_front_kernel<false, true>), FR_LDS_FLOATS * sizeof(float)});
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_front_kernel<false, false>), FR_LDS_FLOATS * sizeof(float)});
  }
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 3);
  net.steps.insert(net.steps.begin() + first, std::move(st));
  // plan_flags[6] = 2: the kernel cuts its windows out of the raw stream and normalises them itself (no gather_normalize launch,
  // no input tensor: -13 us of small kernels and 37 MB per step).  NOT the default: the statistics of a window (one pass over
  // its 72 KB, two barriers) and the 13 divisions per thread and tile stand in front of the first MFMA of every workgroup and
  // cost the kernel 8.6 us (34.3 -> 42.9), while the small launches they replace mostly hide in the tails of the big
  // kernels: 728 k against 736 k windows/s end to end (profiles/r03_eqt_front_cuts_its_windows_ab.txt).  Kept for that A/B
  // and for the bitwise test of the in-kernel arithmetic.
  net.fused_pre = first == 0 && net.cfg.plan_flags[6] == 2;
  return VP_OK;
}

}  // namespace vp
