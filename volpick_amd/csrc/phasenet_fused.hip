// PhaseNet forward in ONE launch (default) or three (debug dumps, A/B timing) instead of 18 (same packed weights):
//
//   pn_window_kernel one 1024-thread workgroup per window: window cut + annotate_batch_pre, the level-0 down path,
//                    the 13 core layers and the level-0 up path back to back out of one 158 KB LDS arena
//                    (DESIGN.md 4a); only the skip tensor of level 0 makes a round trip through memory.
//
//   pn_down0v_kernel / pn_down0_kernel   inc -> down0.same -> down0.down, time-tiled (stride-1 convs on the VALU /
//                    all MFMA); only the skip tensor (down0.same) and the 751-sample down0.down rows go to memory.
//   pn_core_kernel   ONE workgroup per window: levels 1-4 down and up0..up2 (13 layers, 71 % of the
//                    model's FLOPs) run back to back out of a 158 KB LDS arena.
//   pn_up3v_kernel / pn_up3p_kernel      up3.convT -> concat(skip0) -> up3.same -> 1x1 conv + softmax, time-tiled.
//
// The MFMA layers are conv_lds<> (conv_lds.h): LDS image -> MFMA -> LDS image; the 8-channel stride-1 layers of
// level 0 are direct convolutions on the VALU (conv_valu.h).  Coordinates inside a tiled kernel are local to the
// tile; ImageStore writes explicit zeros where the global position falls outside the signal so that the next layer
// sees the reference's zero padding.
#include "conv_b3.h"
#include "conv_lds.h"
#include "conv_valu.h"
#include "net.h"

namespace vp {

namespace {

constexpr int T0 = 3001, T1 = 751, T2 = 188, T3 = 47, T4 = 12;
constexpr int IB = 4;  // column of logical sample 0 in every LDS image

constexpr int img_stride(int L) { return ((4 + L + 20 - 16 + 31) / 32) * 32 + 16; }
constexpr int S1_ = img_stride(T1), S2_ = img_stride(T2), S3_ = img_stride(T3), S4_ = img_stride(T4);
static_assert(S1_ == 784 && S2_ == 240 && S3_ == 80 && S4_ == 48, "image strides");

template <int C, int S, int L, int B = IB>
__device__ __forceinline__ void zero_halo(float* img, int tid, int nth) {
  constexpr int RW = S - L;
  for (int i = tid; i < C * RW; i += nth) {
    const int c = i / RW, k = i - c * RW;
    img[c * S + (k < B ? k : L + k)] = 0.f;
  }
}

template <int S, int B>
struct RangeStore {  // LDS image store, valid t in [0, L)
  float* img;
  int L;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)L) img[co * S + B + t] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < L; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + B + t] = v; }
};

// The images written by the four-phase transposed convs of the core keep sample 0 at column TB = 5: with OUT_OFF = -1
// a lane's four consecutive outputs 4c - 1 .. 4c + 2 then start on a 16-byte boundary and leave as one ds_write_b128.
constexpr int TB = 5;
template <int S, int B>
struct RangeStoreS : RangeStore<S, B> {};
template <int S, int B>
struct RangeStoreV : RangeStore<S, B> {
  __device__ __forceinline__ void vec4(int co, int t, f32x4 v) const { *reinterpret_cast<f32x4*>(this->img + co * S + B + t) = v; }
};

struct GlobalRowStore {  // haloed activation tensor row store with a valid range
  float* p;              // window base + HALO
  int ls, L, t_add;      // global t = t_local + t_add
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    const int tg = t + t_add;
    if (t >= 0 && tg >= 0 && tg < L) p[(long)co * ls + tg] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t0 + t_add >= 0 && t1 + t_add < L; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { p[(long)co * ls + t + t_add] = v; }
};

// ---------------------------------------------------------------------------------------------
// Core: one workgroup (16 waves) per window.
// ---------------------------------------------------------------------------------------------
using C_d1same = LdsLayer<8, 0, 16, 1, 7, 1, -3, 0, 6, 1>;
using C_d1down = LdsLayer<16, 0, 16, 1, 7, 4, -2, 0, 3, 1>;
using C_d2same = LdsLayer<16, 0, 32, 1, 7, 1, -3, 0, 3, 1>;
using C_d2down = LdsLayer<32, 0, 32, 1, 7, 4, -1, 0, 1, 1>;
using C_d3same = LdsLayer<32, 0, 64, 1, 7, 1, -3, 0, 3, 1>;
using C_d3down = LdsLayer<64, 0, 64, 1, 7, 4, -2, 0, 1, 1>;
using C_d4same = LdsLayer<64, 0, 128, 1, 7, 1, -3, 0, 1, 1>;
using C_u0T = LdsLayer<128, 0, 64, 4, 2, 1, -1, -1, 1, 1>;
using C_u0same = LdsLayer<64, 64, 64, 1, 7, 1, -3, 0, 3, 1>;
using C_u1T = LdsLayer<64, 0, 32, 4, 2, 1, -1, -1, 3, 1>;
using C_u1same = LdsLayer<32, 32, 32, 1, 7, 1, -3, 0, 6, 1>;
using C_u2T = LdsLayer<32, 0, 16, 4, 2, 1, -1, -1, 3, 1>;
using C_u2same = LdsLayer<16, 16, 16, 1, 7, 1, -3, 0, 6, 1>;

// layers with at least this many n-tiles per item read their B fragments tap by tap instead of double-buffering a
// whole channel block of them (register budget of a 1024-thread workgroup: 128 per wave)
constexpr int BDB_MAX_NB = 6;

// LDS arena (floats); lifetimes in the header comment of pn_core_kernel
constexpr int A_SKIP1 = 0;                       // 16 x 784
constexpr int A_SKIP2 = A_SKIP1 + 16 * S1_;      // 32 x 240
constexpr int A_Q = A_SKIP2 + 32 * S2_;          // scratch region Q
constexpr int A_SKIP3 = A_Q;                     // 64 x 80
constexpr int A_R = A_Q + 64 * S3_;
constexpr int A_D0 = A_R;                        // 8 x 784
constexpr int A_D1 = A_R;                        // 16 x 240
constexpr int A_D2 = A_R;                        // 32 x 80
constexpr int A_D3 = A_R;                        // 64 x 48
constexpr int A_BOT = A_R + 64 * S4_;            // 128 x 48
constexpr int A_U0T = A_BOT + 128 * S4_;         // 64 x 80
constexpr int A_U0S = A_R;                       // 64 x 80
constexpr int A_U1T = A_U0S + 64 * S3_;          // 32 x 240
constexpr int A_U1S = A_Q;                       // 32 x 240
constexpr int A_U2T = A_U1S + 32 * S2_;          // 16 x 784
constexpr int CORE_LDS_FLOATS = A_U2T + 16 * S1_;
static_assert(A_U0T + 64 * S3_ <= CORE_LDS_FLOATS && A_U1T + 32 * S2_ <= CORE_LDS_FLOATS, "arena overflow");
static_assert(CORE_LDS_FLOATS * 4 <= 160 * 1024, "core arena must fit the 160 KiB LDS");

struct CoreArgs {
  const float* d0;  // [B][8][ls]   (down0.down)
  int ls_d0;
  long ws_d0;
  float* u2s;       // [B][16][ls]  (up2.same)
  int ls_u2s;
  long ws_u2s;
  const float* af[13];
  const float* bs[13];
  // optional debug dumps of every intermediate (null = off): order skip1,d1,skip2,d2,skip3,d3,bottom,u0T,u0s,u1T,u1s,u2T
  float* dbg[12];
  int dbg_ls[12];
  long dbg_ws[12];
  unsigned long long* clk;  // optional [B][32]: [0..14] shader-clock stamps (start, load, 13 layers), [16],[17] 100 MHz wall clock
  int warm;                 // 1: the first workgroup of each XCD pre-touches the weights (pn_core_kernel: every launch -- the
                            // level-0 launches of the three-launch plan run in between; pn_window_kernel: a plan's first launch)
};

template <int C, int S, int B = IB>
__device__ void dump_image(const float* img, int L, float* dst, int ls, long ws, int win, int tid, int nth) {
  if (!dst) return;
  float* d = dst + (long)win * ws + HALO;
  for (int i = tid; i < C * L; i += nth) {
    const int c = i / L, t = i - c * L;
    d[(long)c * ls + t] = img[c * S + B + t];
  }
}

template <bool PIPE>
__global__ __launch_bounds__(1024) void pn_core_kernel(const CoreArgs a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  // (the wave index as a scalar: item loops, block indices and the epilogues' "whole block in range" tests become
  // scalar code instead of per-lane predicates)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), win = blockIdx.x;
  constexpr int NTH = 1024, NWV = 16;
  int stamp = 0;
  if (a.clk && tid == 0) a.clk[(long)win * 32 + 16] = wall_clock64();  // 100 MHz constant clock
#define CORE_STAMP()                                                              \
  if (a.clk && tid == 0) a.clk[(long)win * 32 + stamp] = __builtin_readcyclecounter(); \
  ++stamp;
  CORE_STAMP()

  // The first workgroup of each XCD (consecutive workgroups go to consecutive XCDs) touches one word of every
  // 128-byte line of the 1.06 MB of packed weights: kernels of the other contexts have pushed them out of L2 since
  // the last launch, and the layer chain below would otherwise meet the misses one channel block at a time.
  if (win < 8 && a.warm) {
    float sink = 0.f;
#define CORE_WARM(IDX, LAYER)                                                                          \
  for (int l = tid; l < LAYER::MT * LAYER::CB * LAYER::TAPS * 2; l += NTH) sink += a.af[IDX][l * 32];
    CORE_WARM(0, C_d1same) CORE_WARM(1, C_d1down) CORE_WARM(2, C_d2same) CORE_WARM(3, C_d2down) CORE_WARM(4, C_d3same)
    CORE_WARM(5, C_d3down) CORE_WARM(6, C_d4same) CORE_WARM(7, C_u0T) CORE_WARM(8, C_u0same) CORE_WARM(9, C_u1T)
    CORE_WARM(10, C_u1same) CORE_WARM(11, C_u2T) CORE_WARM(12, C_u2same)
#undef CORE_WARM
    if (sink == 1.2345678e-30f) a.u2s[0] = sink;  // never true: keeps the loads alive
  }

  // ---- load down0.down (8 x 751) -------------------------------------------------------
  {
    const float* src = a.d0 + (long)win * a.ws_d0 + HALO;
    float* img = lds + A_D0;
    for (int i = tid; i < 8 * T1; i += NTH) {
      const int c = i / T1, t = i - c * T1;
      img[c * S1_ + IB + t] = src[(long)c * a.ls_d0 + t];
    }
    zero_halo<8, S1_, T1>(img, tid, NTH);
  }
  __syncthreads();
  CORE_STAMP()

#define CORE_LAYER(IDX, LAYER, IN1, SI1, IN2, SI2, B2, OUT, SO, OB, STORE, CO, COLS, LOUT, DBG)                    \
  {                                                                                                                \
    STORE<SO, OB> st{{lds + (OUT), (LOUT)}};                                                                      \
    zero_halo<CO, SO, LOUT, OB>(lds + (OUT), tid, NTH);                                                          \
    conv_lds<LAYER, SI1, IB, SI2, B2, PIPE, (LAYER::NB < BDB_MAX_NB)>(lds + (IN1), lds + (IN2), a.af[IDX], a.bs[IDX], (COLS), st, wave, NWV, lane); \
    __syncthreads();                                                                                               \
    CORE_STAMP()                                                                                                   \
    if (a.dbg[DBG]) dump_image<CO, SO, OB>(lds + (OUT), (LOUT), a.dbg[DBG], a.dbg_ls[DBG], a.dbg_ws[DBG], win, tid, NTH); \
  }
#define CORE_LAYER_AREG(IDX, LAYER, IN1, SI1, OUT, SO, OB, STORE, CO, COLS, LOUT, WMT, WFIRST, WSTEP, DBG)                    \
  {                                                                                                                \
    STORE<SO, OB> st{{lds + (OUT), (LOUT)}};                                                                      \
    zero_halo<CO, SO, LOUT, OB>(lds + (OUT), tid, NTH);                                                          \
    if ((WMT) < LAYER::MT && (WFIRST) < ((((COLS) + 15) >> 4) + LAYER::NB - 1) / LAYER::NB) {                    \
      float ar[LAYER::CB * LAYER::TAPS], br[4];                                                                    \
      load_areg<LAYER>(a.af[IDX], (WMT), lane, ar);                                                                \
      load_biasreg<LAYER>(a.bs[IDX], (WMT), lane, br);                                                             \
      conv_lds_areg<LAYER, SI1, IB, SI1, IB>(lds + (IN1), lds + (IN1), ar, br, (WMT), (COLS), st, (WFIRST), (WSTEP), lane); \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    CORE_STAMP()                                                                                                   \
    if (a.dbg[DBG]) dump_image<CO, SO, OB>(lds + (OUT), (LOUT), a.dbg[DBG], a.dbg_ls[DBG], a.dbg_ws[DBG], win, tid, NTH); \
  }
  //          idx layer      in1      S    in2      S    b2  out      S    ob  store        C    cols     Lout dbg
  CORE_LAYER(0, C_d1same, A_D0, S1_, A_D0, S1_, IB, A_SKIP1, S1_, IB, RangeStoreS, 16, T1, T1, 0)
  CORE_LAYER(1, C_d1down, A_SKIP1, S1_, A_SKIP1, S1_, IB, A_D1, S2_, IB, RangeStoreS, 16, T2, T2, 1)
  CORE_LAYER(2, C_d2same, A_D1, S2_, A_D1, S2_, IB, A_SKIP2, S2_, IB, RangeStoreS, 32, T2, T2, 2)
  CORE_LAYER(3, C_d2down, A_SKIP2, S2_, A_SKIP2, S2_, IB, A_D2, S3_, IB, RangeStoreS, 32, T3, T3, 3)
  CORE_LAYER(4, C_d3same, A_D2, S3_, A_D2, S3_, IB, A_SKIP3, S3_, IB, RangeStoreS, 64, T3, T3, 4)
  CORE_LAYER(5, C_d3down, A_SKIP3, S3_, A_SKIP3, S3_, IB, A_D3, S4_, IB, RangeStoreS, 64, T4, T4, 5)
  CORE_LAYER(6, C_d4same, A_D3, S4_, A_D3, S4_, IB, A_BOT, S4_, IB, RangeStoreS, 128, T4, T4, 6)
  CORE_LAYER(7, C_u0T, A_BOT, S4_, A_BOT, S4_, IB, A_U0T, S3_, TB, RangeStoreV, 64, T4 + 1, T3, 7)
  CORE_LAYER(8, C_u0same, A_SKIP3, S3_, A_U0T, S3_, TB, A_U0S, S3_, IB, RangeStoreS, 64, T3, T3, 8)
  // Two-tap layers with few items: the wave's whole A operand (16-32 fragments) is fetched up front — two fragments per
  // channel block in flight (the double buffer of conv_lds) left these layers waiting on L2 at every block.
  CORE_LAYER_AREG(9, C_u1T, A_U0S, S3_, A_U1T, S2_, TB, RangeStoreV, 32, T3 + 1, T2, wave, 0, 1, 9)        // 8 m-tiles x 1 block
  CORE_LAYER(10, C_u1same, A_SKIP2, S2_, A_U1T, S2_, TB, A_U1S, S2_, IB, RangeStoreS, 32, T2, T2, 10)
  CORE_LAYER_AREG(11, C_u2T, A_U1S, S2_, A_U2T, S1_, TB, RangeStoreV, 16, T2 + 1, T1, wave & 3, wave >> 2, 4, 11)  // 4 m-tiles x 4 blocks
#undef CORE_LAYER_AREG
#undef CORE_LAYER
  {
    GlobalRowStore st{a.u2s + (long)win * a.ws_u2s + HALO, a.ls_u2s, T1, 0};
    conv_lds<C_u2same, S1_, IB, S1_, TB, PIPE, (C_u2same::NB < BDB_MAX_NB)>(lds + A_SKIP1, lds + A_U2T, a.af[12], a.bs[12], T1, st, wave, NWV, lane);
  }
  __syncthreads();
  CORE_STAMP()
  if (a.clk && tid == 0) a.clk[(long)win * 32 + 17] = wall_clock64();
#undef CORE_STAMP
}

// ---------------------------------------------------------------------------------------------
// Level-0 down path, time-tiled: inc -> down0.same -> down0.down.
// Local coordinate l <-> global level-0 sample (t0 - 12) + l, t0 = tile * TT.
// ---------------------------------------------------------------------------------------------
constexpr int TT = 512;                   // level-0 samples per tile
constexpr int D0_S = 560;                 // image stride (== 16 mod 32), covers local [-4, 556)
using D_inc = LdsLayer<3, 0, 8, 2, 8, 2, -3, 0, 5, 1>;
using D_same = LdsLayer<8, 0, 8, 2, 8, 2, -3, 0, 5, 1>;
using D_down = LdsLayer<8, 0, 8, 2, 11, 8, 9, 0, 1, 1>;  // reads skip0 local 8n + tap + 9 (= global 8(c0+n) + tap - 3)
constexpr int D0_X = 0, D0_H = 4 * D0_S, D0_K = 12 * D0_S, D0_LDS_FLOATS = 20 * D0_S;

struct Down0Args {
  const float* x;   // [B][3][ls] normalised input
  int ls_x;
  long ws_x;
  float* skip0;     // [B][8][ls]  (down0.same)
  int ls_s;
  long ws_s;
  float* d0;        // [B][8][ls]  (down0.down)
  int ls_d;
  long ws_d;
  float* h0_dbg;    // optional [B][8][ls] (inc)
  int ls_h;
  long ws_h;
  const float *af_inc, *bs_inc, *af_same, *bs_same, *af_down, *bs_down;
};

template <bool PIPE>
__global__ __launch_bounds__(256) void pn_down0_kernel(const Down0Args a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, win = blockIdx.y;
  const int t0 = tile * TT, o = t0 - 12;
  constexpr int NTH = 256, NWV = 4;

  // x image: local [-4, TT + 24) as float4s; physical index = HALO + o - 4 + 4q = t0 - 8 + 4q
  {
    const float* src = a.x + (long)win * a.ws_x;
    constexpr int NQ = (TT + 28) / 4;  // 135 float4 per row
    for (int i = tid; i < 3 * NQ; i += NTH) {
      const int c = i / NQ, q = i - c * NQ;
      const int p = t0 - 8 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p >= 0 && p + 3 < a.ls_x) v = *reinterpret_cast<const float4*>(src + (long)c * a.ls_x + p);
      *reinterpret_cast<float4*>(lds + D0_X + c * D0_S + 4 * q) = v;
    }
    for (int i = tid; i < D0_S / 4; i += NTH)  // 4th (padding) channel must be true zeros
      *reinterpret_cast<float4*>(lds + D0_X + 3 * D0_S + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int sig_lo = -o, sig_hi = T0 - o;
  {
    ImageStore<D0_S, IB> st{lds + D0_H, 0, TT + 16, sig_lo, sig_hi};
    conv_lds<D_inc, D0_S, IB, D0_S, IB, PIPE>(lds + D0_X, lds + D0_X, a.af_inc, a.bs_inc, (TT + 16) / 2, st, wave, NWV, lane);
  }
  __syncthreads();
  if (a.h0_dbg) {
    float* d = a.h0_dbg + (long)win * a.ws_h + HALO;
    for (int i = tid; i < 8 * TT; i += NTH) {
      const int c = i / TT, l = 12 + (i - c * TT);
      if (o + l < T0) d[(long)c * a.ls_h + o + l] = lds[D0_H + c * D0_S + IB + l];
    }
  }
  {
    ImageStore<D0_S, IB> st{lds + D0_K, 0, TT + 16, sig_lo, sig_hi};
    conv_lds<D_same, D0_S, IB, D0_S, IB, PIPE>(lds + D0_H, lds + D0_H, a.af_same, a.bs_same, (TT + 16) / 2, st, wave, NWV, lane);
  }
  __syncthreads();
  {  // skip tensor rows [t0, t0 + TT) -> memory, 16-byte coalesced (local 12 <-> column 16)
    float* d = a.skip0 + (long)win * a.ws_s + HALO + t0;
    for (int i = tid; i < 8 * (TT / 4); i += NTH) {
      const int c = i / (TT / 4), q = i - c * (TT / 4);
      const int t = t0 + 4 * q;
      if (t < T0) {
        float4 v = *reinterpret_cast<const float4*>(lds + D0_K + c * D0_S + IB + 12 + 4 * q);  // zeros beyond the signal
        *reinterpret_cast<float4*>(d + (long)c * a.ls_s + 4 * q) = v;
      }
    }
  }
  {
    GlobalRowStore st{a.d0 + (long)win * a.ws_d + HALO, a.ls_d, T1, tile * (TT / 4)};
    conv_lds<D_down, D0_S, IB, D0_S, IB, PIPE>(lds + D0_K, lds + D0_K, a.af_down, a.bs_down, TT / 8, st, wave, NWV, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// Level-0 up path, time-tiled: up3.convT -> cat(skip0, .) -> up3.same -> 1x1 conv + softmax.
// Level-1 local n <-> global (t0/4 - 4) + n; level-0 local l <-> global (t0 - 16) + l.
// ---------------------------------------------------------------------------------------------
constexpr int U_S1 = 176;                 // u2s image stride: local1 [-4, 172)
constexpr int U_S0 = 592;                 // level-0 image stride: local0 [-4, 588)
constexpr int U_SO = 516;                 // staged up3.same tile
using U_T = LdsLayer<16, 0, 8, 4, 2, 1, -1, -2, 5, 1>;
using U_same = LdsLayer<8, 8, 8, 2, 8, 2, 13, 0, 4, 1>;  // out local'' 2n+p <-> level-0 local 16 + 2n + p
constexpr int U_U2S = 0, U_SKIP = 16 * U_S1, U_UT = U_SKIP + 8 * U_S0, U_OUT = U_UT + 8 * U_S0;
constexpr int UP3_LDS_FLOATS = U_OUT + 8 * U_SO;

struct Up3Args {
  const float* u2s;    // [B][16][ls] (up2.same)
  int ls_u;
  long ws_u;
  const float* skip0;  // [B][8][ls]
  int ls_s;
  long ws_s;
  float* y;            // dense [B][3][T0]
  float* ut_dbg;       // optional [B][8][ls] (up3.convT)
  int ls_t;
  long ws_t;
  const float *af_t, *bs_t, *af_same, *bs_same;
  const float* w_out;  // [3][8]
  const float* b_out;  // [3]
  unsigned long long* clk;  // optional debug stamps (tile 2 of each window): slots 18..23 of the core's [B][32] block
};

template <bool PIPE>
__global__ __launch_bounds__(256) void pn_up3_kernel(const Up3Args a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, win = blockIdx.y;
  const int t0 = tile * TT, o1 = t0 / 4 - 4, o0 = t0 - 16;
  constexpr int NTH = 256, NWV = 4;
  int stamp = 18;
#define UP3_STAMP()                                                                                   \
  if (a.clk && tid == 0 && tile == 2) a.clk[(long)win * 32 + stamp] = __builtin_readcyclecounter(); \
  ++stamp;
  UP3_STAMP()

  {  // up2.same rows: local1 [0, 144); physical = HALO + o1 + 4q
    const float* src = a.u2s + (long)win * a.ws_u + HALO + o1;
    for (int i = tid; i < 16 * 36; i += NTH) {
      const int c = i / 36, q = i - c * 36;
      *reinterpret_cast<float4*>(lds + U_U2S + c * U_S1 + IB + 4 * q) =
          *reinterpret_cast<const float4*>(src + (long)c * a.ls_u + 4 * q);
    }
    if (tid < 16) *reinterpret_cast<float4*>(lds + U_U2S + tid * U_S1) = make_float4(0.f, 0.f, 0.f, 0.f);  // local1 -4..-1
  }
  {  // skip rows: local0 [12, 532); physical = HALO + o0 + 12 + 4q = t0 + 4 + 4q
    const float* src = a.skip0 + (long)win * a.ws_s + HALO + o0 + 12;
    for (int i = tid; i < 8 * 130; i += NTH) {
      const int c = i / 130, q = i - c * 130;
      *reinterpret_cast<float4*>(lds + U_SKIP + c * U_S0 + IB + 12 + 4 * q) =
          *reinterpret_cast<const float4*>(src + (long)c * a.ls_s + 4 * q);
    }
  }
  __syncthreads();
  UP3_STAMP()
  {
    ImageStore<U_S0, IB> st{lds + U_UT, 0, U_S0 - IB, -o0, T0 - o0};
    conv_lds<U_T, U_S1, IB, U_S1, IB, PIPE>(lds + U_U2S, lds + U_U2S, a.af_t, a.bs_t, 144, st, wave, NWV, lane);
  }
  __syncthreads();
  UP3_STAMP()
  if (a.ut_dbg) {
    float* d = a.ut_dbg + (long)win * a.ws_t + HALO;
    for (int i = tid; i < 8 * TT; i += NTH) {
      const int c = i / TT, l = 16 + (i - c * TT);
      if (o0 + l < T0) d[(long)c * a.ls_t + o0 + l] = lds[U_UT + c * U_S0 + IB + l];
    }
  }
  {
    ImageStore<U_SO, 0> st{lds + U_OUT, 0, TT, 0, TT};
    conv_lds<U_same, U_S0, IB, U_S0, IB, PIPE>(lds + U_SKIP, lds + U_UT, a.af_same, a.bs_same, TT / 2, st, wave, NWV, lane);
  }
  __syncthreads();
  UP3_STAMP()
  {  // 1x1 conv (8 -> 3) + softmax over channels, dense output
    float w[3][8], bb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bb[c] = a.b_out[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) w[c][k] = a.w_out[c * 8 + k];
    }
    float* y = a.y + (long)win * 3 * T0;
    for (int l = tid; l < TT; l += NTH) {
      const int t = t0 + l;
      if (t < T0) {
        float z[3] = {bb[0], bb[1], bb[2]};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = lds[U_OUT + k * U_SO + l];
#pragma unroll
          for (int c = 0; c < 3; ++c) z[c] = fmaf(w[c][k], v, z[c]);
        }
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
        const float inv = 1.f / (e0 + e1 + e2);
        y[t] = e0 * inv;
        y[T0 + t] = e1 * inv;
        y[2 * T0 + t] = e2 * inv;
      }
    }
  }
  __syncthreads();
  UP3_STAMP()
#undef UP3_STAMP
}

// ---------------------------------------------------------------------------------------------
// Persistent forms of the two tiled kernels.  One workgroup walks TPS consecutive tiles of a window:
// the A fragments of its waves' m-tiles are loaded ONCE into registers (the layers here have K of
// only 8-32 steps, so an exposed L2 round trip per tile costs as much as the MFMAs), and the next
// tile's input rows are fetched into registers while the current tile computes.
// ---------------------------------------------------------------------------------------------
constexpr int N_TILES = (T0 + TT - 1) / TT;     // 6
constexpr int NSPLIT_U = 2, TPS_U = (N_TILES + NSPLIT_U - 1) / NSPLIT_U;  // up3:   66 KB LDS -> 2 workgroups / CU

// (pn_down0p_kernel, the persistent form of the level-0 down kernel -- plan_flags[3] = 2, measured 4-15 % slower than one workgroup
// per tile -- was removed in round 6.)

__global__ __launch_bounds__(256) void pn_up3p_kernel(const Up3Args a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int win = blockIdx.y;
  const int tile_lo = blockIdx.x * TPS_U, tile_hi = (tile_lo + TPS_U < N_TILES) ? tile_lo + TPS_U : N_TILES;
  constexpr int NTH = 256;
  float aT[U_T::CB * U_T::TAPS], aS[U_same::CB * U_same::TAPS], bT[4], bS[4];
  const int mtT = wave & 1;  // up3.convT: M = 32 -> two m-tiles; waves (0,2) take m-tile 0, (1,3) m-tile 1
  load_areg<U_T>(a.af_t, mtT, lane, aT);
  load_biasreg<U_T>(a.bs_t, mtT, lane, bT);
  load_areg<U_same>(a.af_same, 0, lane, aS);
  load_biasreg<U_same>(a.bs_same, 0, lane, bS);
  float w[3][8], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bb[c] = a.b_out[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[c][k] = a.w_out[c * 8 + k];
  }
  if (tid < 16) *reinterpret_cast<float4*>(lds + U_U2S + tid * U_S1) = make_float4(0.f, 0.f, 0.f, 0.f);  // local1 -4..-1

  const float* usrc = a.u2s + (long)win * a.ws_u + HALO;
  const float* ssrc = a.skip0 + (long)win * a.ws_s + HALO;
  float4 pu[3], ps[5];  // 16 x 36 = 576 and 8 x 130 = 1040 float4
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int o1 = tile * (TT / 4) - 4, o0 = tile * TT - 16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = tid + k * NTH;
      pu[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 16 * 36) {
        const int c = i / 36, q = i - c * 36;
        pu[k] = *reinterpret_cast<const float4*>(usrc + o1 + (long)c * a.ls_u + 4 * q);
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = tid + k * NTH;
      ps[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 8 * 130) {
        const int c = i / 130, q = i - c * 130;
        ps[k] = *reinterpret_cast<const float4*>(ssrc + o0 + 12 + (long)c * a.ls_s + 4 * q);
      }
    }
  };
  fetch(tile_lo);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int t0 = tile * TT, o0 = t0 - 16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = tid + k * NTH;
      if (i < 16 * 36) {
        const int c = i / 36, q = i - c * 36;
        *reinterpret_cast<float4*>(lds + U_U2S + c * U_S1 + IB + 4 * q) = pu[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = tid + k * NTH;
      if (i < 8 * 130) {
        const int c = i / 130, q = i - c * 130;
        *reinterpret_cast<float4*>(lds + U_SKIP + c * U_S0 + IB + 12 + 4 * q) = ps[k];
      }
    }
    __syncthreads();
    if (tile + 1 < tile_hi) fetch(tile + 1);
    {
      ImageStore<U_S0, IB> st{lds + U_UT, 0, U_S0 - IB, -o0, T0 - o0};
      conv_lds_areg<U_T, U_S1, IB, U_S1, IB>(lds + U_U2S, lds + U_U2S, aT, bT, mtT, 144, st, wave >> 1, 2, lane);
    }
    __syncthreads();
    {
      ImageStore<U_SO, 0> st{lds + U_OUT, 0, TT, 0, TT};
      conv_lds_areg<U_same, U_S0, IB, U_S0, IB>(lds + U_SKIP, lds + U_UT, aS, bS, 0, TT / 2, st, wave, 4, lane);
    }
    __syncthreads();
    float* y = a.y + (long)win * 3 * T0;
    for (int l = tid; l < TT; l += NTH) {  // 1x1 conv (8 -> 3) + softmax over channels
      const int t = t0 + l;
      if (t < T0) {
        float z[3] = {bb[0], bb[1], bb[2]};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = lds[U_OUT + k * U_SO + l];
#pragma unroll
          for (int c = 0; c < 3; ++c) z[c] = fmaf(w[c][k], v, z[c]);
        }
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
        const float inv = 1.f / (e0 + e1 + e2);
        y[t] = e0 * inv;
        y[T0 + t] = e1 * inv;
        y[2 * T0 + t] = e2 * inv;
      }
    }
    // the next iteration's first barrier orders this read of the staged tile before its rewrite
  }
}

// ---------------------------------------------------------------------------------------------
// VALU forms of the two level-0 kernels (conv_valu.h): direct convolution, four consecutive samples and all
// eight output channels per lane, 256 lanes = a 1024-sample span per workgroup.  Layer chains shrink the valid
// span by 3 samples per side per k7 layer, so consecutive tiles advance by 1008 (down path) / 1016 (up path).
// ---------------------------------------------------------------------------------------------
constexpr int VT = 1024, VS = VT + 8;           // lanes x 4 samples; image row stride (local l at column l + 4)
constexpr int VD_TS = 1008, VD_TILES = (T0 + VD_TS - 1) / VD_TS;  // down0: local 0 <-> global VD_TS * tile - 8
constexpr int VU_TS = 1016, VU_TILES = (T0 + VU_TS - 1) / VU_TS;  // up3:   local 0 <-> global VU_TS * tile - 4
constexpr int VU_SX = 272;                      // up2.same image row stride (258 level-1 samples per tile; == 16 mod 32)
// 50 KB and 50 KB of LDS: the three tiles of a window are resident on one CU together (256 windows on 256 CUs = one round)
constexpr int VD_LDS_FLOATS = 12 * VS + 64, VU_LDS_FLOATS = 8 * VS + 16 * VU_SX + 64;  // + margin for masked MFMA columns
// the strided and the transposed conv of the two kernels stay on the MFMA (weights used once per output sample:
// on the VALU they are bound by scalar-load latency, measured 15 k cycles for 448 packed FMAs per lane)
using VD_down = LdsLayer<4, 4, 8, 2, 11, 8, 5, 0, 1, 1>;   // out n' = 2n + p reads local 8n + tap + 5; channels 0-3 / 4-7 in two images
using VU_T = LdsLayer<16, 0, 8, 4, 2, 1, -1, -2, 5, 1>;    // out local 4m + p - 2 reads level-1 local m + tap - 1
struct TileRowStore {  // haloed activation row store of one tile: local t in [0, t_hi), global t + t_add in [0, L)
  float* p;
  int ls, L, t_add, t_hi;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)t_hi && t + t_add < L) p[(long)co * ls + t + t_add] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < t_hi && t1 + t_add < L; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { p[(long)co * ls + t + t_add] = v; }
};
constexpr int VX_Q = VS / 4;                    // float4 per image row
static_assert(VD_TILES == 3 && VU_TILES == 3, "three tiles per window");

struct Down0VArgs {
  Down0Args t;  // tensors as in the MFMA form (af_* / bs_* unused)
  const f32x2 *w_inc, *b_inc, *w_same, *b_same, *w_down, *b_down;  // [cin][7][4] channel pairs, [4] bias pairs
  int n_windows;
};

// relu + zero outside the signal, channel pair c of acc -> two float4 rows
template <int NC>
__device__ __forceinline__ void valu_finish(const f32x2 (&acc)[NC][4], int c, int tg, f32x4* lo, f32x4* hi) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool in = (unsigned)(tg + r) < (unsigned)T0;
    (*lo)[r] = in ? fmaxf(acc[c][r].x, 0.f) : 0.f;
    (*hi)[r] = in ? fmaxf(acc[c][r].y, 0.f) : 0.f;
  }
}
template <int NC>
__device__ __forceinline__ void valu_bias(f32x2 (&acc)[NC][4], const f32x2* b) {
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[c][r] = as_weights(b)[c];
}

__global__ __launch_bounds__(256) void pn_down0v_kernel(const Down0VArgs a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  float *X = lds, *H = lds + 4 * VS;  // X: 3 input rows, later rows 0-3 of down0.same; H: inc, later rows 4-7 of down0.same
  const int tid = threadIdx.x;
  // Workgroup -> (window, tile): consecutive workgroups go to consecutive XCDs, and the core kernel runs window w on
  // XCD w % 8 — with this mapping the rows a window hands from kernel to kernel stay in one XCD's L2.
  int win, tile;
  {
    const int id = blockIdx.x, B = a.n_windows;
    if ((B & 7) == 0) {
      const int slot = id >> 3;
      win = (slot / 3) * 8 + (id & 7);
      tile = slot % 3;
    } else {
      win = id / 3;
      tile = id % 3;
    }
  }
  const int g0 = VD_TS * tile - 8;  // global sample of local 0
  {  // x image: local [-4, 1028); physical index of local -4 + 4q = HALO + g0 - 4 + 4q (16-byte aligned)
    const float* src = a.t.x + (long)win * a.t.ws_x;
    for (int i = tid; i < 3 * VX_Q; i += 256) {
      const int c = i / VX_Q, q = i - c * VX_Q;
      const int p = HALO + g0 - 4 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p >= 0 && p + 3 < a.t.ls_x) v = *reinterpret_cast<const float4*>(src + (long)c * a.t.ls_x + p);
      *reinterpret_cast<float4*>(X + c * VS + 4 * q) = v;
    }
  }
  __syncthreads();
  const int t0 = 4 * tid, tg = g0 + t0;
  const bool own = t0 >= 8 && t0 < 8 + VD_TS && tg < T0;  // samples this tile hands to memory
  f32x2 acc[4][4];
  {  // inc: Conv1d(3, 8, 7, same, bias) + BN + ReLU
    valu_bias(acc, a.b_inc);
    valu_conv7_r4<3, VS>(X, as_weights(a.w_inc), t0, acc);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 lo, hi;
      valu_finish(acc, c, tg, &lo, &hi);
      *reinterpret_cast<f32x4*>(H + (2 * c) * VS + 4 + t0) = lo;
      *reinterpret_cast<f32x4*>(H + (2 * c + 1) * VS + 4 + t0) = hi;
      if (a.t.h0_dbg && own) {
        float* d = a.t.h0_dbg + (long)win * a.t.ws_h + HALO + tg;
        *reinterpret_cast<f32x4*>(d + (long)(2 * c) * a.t.ls_h) = lo;
        *reinterpret_cast<f32x4*>(d + (long)(2 * c + 1) * a.t.ls_h) = hi;
      }
    }
  }
  __syncthreads();
  {  // down0.same: Conv1d(8, 8, 7, same) + BN + ReLU -> skip tensor + image for the strided conv.  Two passes of four
     // output channels: the skip rows of the first pass drain to memory under the FMAs of the second (in one pass the
     // whole 25 MB of a launch left the chip in one burst after the last FMA, with nothing to hide it).
    float* d = a.t.skip0 + (long)win * a.t.ws_s + HALO + tg;
    f32x2 acc2[2][4];
    f32x4 lo[2], hi[2];
    valu_bias(acc2, a.b_same);
    valu_conv7_r4<8, VS, 2, 0>(H, as_weights(a.w_same), t0, acc2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // rows 0-3 take the place of the x image (its last reader finished two barriers ago)
      valu_finish(acc2, c, tg, &lo[c], &hi[c]);
      *reinterpret_cast<f32x4*>(X + (2 * c) * VS + 4 + t0) = lo[c];
      *reinterpret_cast<f32x4*>(X + (2 * c + 1) * VS + 4 + t0) = hi[c];
      if (own) {  // the float4 holding sample T0 - 1 also rewrites up to three zeros of the row's right margin
        __builtin_nontemporal_store(lo[c], reinterpret_cast<f32x4*>(d + (long)(2 * c) * a.t.ls_s));
        __builtin_nontemporal_store(hi[c], reinterpret_cast<f32x4*>(d + (long)(2 * c + 1) * a.t.ls_s));
      }
    }
    valu_bias(acc2, a.b_same + 2);
    valu_conv7_r4<8, VS, 2, 2>(H, as_weights(a.w_same), t0, acc2);
    lds_barrier();  // every lane has read its inc window: rows 4-7 overwrite the first rows of that image
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      valu_finish(acc2, c, tg, &lo[c], &hi[c]);
      *reinterpret_cast<f32x4*>(H + (2 * c) * VS + 4 + t0) = lo[c];
      *reinterpret_cast<f32x4*>(H + (2 * c + 1) * VS + 4 + t0) = hi[c];
    }
    lds_barrier();
    if (own) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        __builtin_nontemporal_store(lo[c], reinterpret_cast<f32x4*>(d + (long)(4 + 2 * c) * a.t.ls_s));
        __builtin_nontemporal_store(hi[c], reinterpret_cast<f32x4*>(d + (long)(5 + 2 * c) * a.t.ls_s));
      }
    }
  }
  {  // down0.down: Conv1d(8, 8, 7, stride 4, pad 3) + BN + ReLU on the MFMA; output n = 252 * tile + n' reads local 4 n' + 5 + k
    TileRowStore st{a.t.d0 + (long)win * a.t.ws_d + HALO, a.t.ls_d, T1, (VD_TS / 4) * tile, VD_TS / 4};
    conv_lds<VD_down, VS, 4, VS, 4, false>(X, H, a.t.af_down, a.t.bs_down, VD_TS / 8, st, tid >> 6, 4, tid & 63);
  }
}

struct Up3VArgs {
  Up3Args t;  // tensors as in the MFMA form
  const f32x2 *w_t, *b_t, *w_same, *b_same;  // up3.convT [16][7][4], up3.same [16][7][4] (skip channels first)
  int n_windows;
};

__global__ __launch_bounds__(256) void pn_up3v_kernel(const Up3VArgs a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  float *A = lds, *U = lds + 8 * VS;  // A: skip rows, later the up3.convT rows; U: up2.same rows
  const int tid = threadIdx.x;
  // Workgroup -> (window, tile): consecutive workgroups go to consecutive XCDs, and the core kernel runs window w on
  // XCD w % 8 — with this mapping the rows a window hands from kernel to kernel stay in one XCD's L2.
  int win, tile;
  {
    const int id = blockIdx.x, B = a.n_windows;
    if ((B & 7) == 0) {
      const int slot = id >> 3;
      win = (slot / 3) * 8 + (id & 7);
      tile = slot % 3;
    } else {
      win = id / 3;
      tile = id % 3;
    }
  }
  const int g0 = VU_TS * tile - 4;  // global sample of local 0
  int stamp = 18;  // debug clock stamps of tile 1 (slots 18..25 of the core's [B][32] block)
#define UP3V_STAMP()                                                                                  \
  if (a.t.clk && tid == 0 && tile == 1) a.t.clk[(long)win * 32 + stamp] = __builtin_readcyclecounter(); \
  ++stamp;
  UP3V_STAMP()
  if (a.t.clk && tid == 0 && tile == 1) a.t.clk[(long)win * 32 + 26] = wall_clock64();
  const int lane = tid & 63, wave = tid >> 6;
  // skip rows: local [-4, 1028); physical index HALO + g0 - 4 + 4q = VU_TS * tile + 4q
  constexpr int NSK = (8 * VX_Q + 255) / 256;
  float4 sk[NSK];
  {
    const float* src = a.t.skip0 + (long)win * a.t.ws_s + VU_TS * tile;
#pragma unroll
    for (int k = 0; k < NSK; ++k) {
      const int i = tid + k * 256, c = i / VX_Q, q = i - c * VX_Q;
      sk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 8 * VX_Q && VU_TS * tile + 4 * q + 3 < a.t.ls_s) sk[k] = *reinterpret_cast<const float4*>(src + (long)c * a.t.ls_s + 4 * q);
    }
  }
  // up2.same rows: U[ci][j] = x[ci][(g0 >> 2) - 1 + j], j in [0, 258); physical index HALO + 254 tile - 2 + j >= 6.
  // Fetched into registers behind the skip rows and written to their image after the skip half of up3.same
  // (first use: the transposed conv): they stay in flight across the first barrier.
  constexpr int NU = (16 * 258 + 255) / 256;
  float u[NU];
  {
    const float* us = a.t.u2s + (long)win * a.t.ws_u + HALO + (VU_TS / 4) * tile - 2;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      const int i = tid + k * 256, c = i / 258, j = i - c * 258;
      u[k] = (i < 16 * 258) ? us[(long)c * a.t.ls_u + j] : 0.f;
    }
  }
  // A fragments of the transposed conv (M = 32: waves 0, 2 take m-tile 0, waves 1, 3 m-tile 1), resident in registers
  float aT[VU_T::CB * VU_T::TAPS], bT[4];
  load_areg<VU_T>(a.t.af_t, wave & 1, lane, aT);
  load_biasreg<VU_T>(a.t.bs_t, wave & 1, lane, bT);
#pragma unroll
  for (int k = 0; k < NSK; ++k) {
    const int i = tid + k * 256, c = i / VX_Q, q = i - c * VX_Q;
    if (i < 8 * VX_Q) *reinterpret_cast<float4*>(A + c * VS + 4 * q) = sk[k];
  }
  lds_barrier();  // not __syncthreads(): the up2.same rows and the A fragments are still in flight
  UP3V_STAMP()
  const int t0 = 4 * tid, tg = g0 + t0;
  const bool own = t0 >= 4 && t0 < 4 + VU_TS && tg < T0;
  f32x2 acc[4][4];
  // up3.same on cat([skip0, up3.convT]): the skip half first, then the convT rows take the skip image's place
  valu_bias(acc, a.b_same);
  valu_conv7_r4<8, VS>(A, as_weights(a.w_same), t0, acc);
  UP3V_STAMP()
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int i = tid + k * 256, c = i / 258, j = i - c * 258;
    if (i < 16 * 258) U[c * VU_SX + j] = u[k];
  }
  __syncthreads();
  UP3V_STAMP()
  {  // up3.convT: ConvTranspose1d(16, 8, 7, stride 4) + BN + ReLU, crop [1:-2] and centre crop (t = o - 2), on the MFMA
    ImageStore<VS, 4> st{A, 0, VT, -g0, T0 - g0};
    conv_lds_areg<VU_T, VU_SX, 1, VU_SX, 1>(U, U, aT, bT, wave & 1, VT / 4 + 1, st, wave >> 1, 2, lane);
  }
  UP3V_STAMP()
  __syncthreads();
  if (a.t.ut_dbg && own) {
    float* d = a.t.ut_dbg + (long)win * a.t.ws_t + HALO + tg;
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(d + (long)c * a.t.ls_t) = *reinterpret_cast<const f32x4*>(A + c * VS + 4 + t0);
  }
  UP3V_STAMP()
  valu_conv7_r4<8, VS>(A, as_weights(a.w_same + 8 * 28), t0, acc);
  UP3V_STAMP()
  if (own) {  // BN + ReLU -> Conv1d(8, 3, 1) -> softmax over channels
    float z[3][4];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[o][r] = as_scalars(a.t.b_out)[o];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v0 = fmaxf(acc[c][r].x, 0.f), v1 = fmaxf(acc[c][r].y, 0.f);
#pragma unroll
        for (int o = 0; o < 3; ++o)
          z[o][r] = fmaf(as_scalars(a.t.w_out)[o * 8 + 2 * c + 1], v1, fmaf(as_scalars(a.t.w_out)[o * 8 + 2 * c], v0, z[o][r]));
      }
    f32x4 y0, y1, y2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mx = fmaxf(z[0][r], fmaxf(z[1][r], z[2][r]));
      const float e0 = __expf(z[0][r] - mx), e1 = __expf(z[1][r] - mx), e2 = __expf(z[2][r] - mx);
      const float inv = 1.f / (e0 + e1 + e2);
      y0[r] = e0 * inv, y1[r] = e1 * inv, y2[r] = e2 * inv;
    }
    float* y = a.t.y + (long)win * 3 * T0 + tg;
    if (tg + 3 < T0) {  // dense rows of odd length: 4-byte aligned vector stores
      *reinterpret_cast<f32x4u*>(y) = y0;
      *reinterpret_cast<f32x4u*>(y + T0) = y1;
      *reinterpret_cast<f32x4u*>(y + 2 * T0) = y2;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (tg + r < T0) y[r] = y0[r], y[T0 + r] = y1[r], y[2 * T0 + r] = y2[r];
    }
  }
  UP3V_STAMP()
  if (a.t.clk && tid == 0 && tile == 1) a.t.clk[(long)win * 32 + 27] = wall_clock64();
#undef UP3V_STAMP
}

// ---------------------------------------------------------------------------------------------
// The whole network in ONE launch: one 1024-thread workgroup per window runs the level-0 down path (VALU + MFMA),
// the 13 core layers and the level-0 up path back to back out of the same 158 KB LDS arena.  Against the three
// launches above: down0.down and up2.same never leave LDS (18 MB + 24 MB of traffic per 256 windows gone), the skip
// tensor is written and read back by the SAME CU inside one kernel (no end-of-kernel L2 write-back and invalidate
// between producer and consumer: it is served from the XCD's L2), there are no tile halos to recompute, and the two
// memory-bound phases that every workgroup of a level-0 launch entered in lock step (all load, then all compute)
// shrink to one 36 KB read per window at the start.
// ---------------------------------------------------------------------------------------------
constexpr int W0_S = 3024;                 // level-0 image row stride: sample t at column t + 4, t in [-4, 3020); == 16 mod 32
constexpr int W0_Q = W0_S / 4;             // float4 per row = lanes that store into a row
constexpr int W_LANES = (T0 + 3) / 4;      // lanes that own signal samples (four each)
constexpr int W_WAVES = (W0_Q + 63) / 64;  // waves that run the VALU convs (the others only load, store and do MFMA items)
// down phase: inc (8 rows; down0.same later overwrites it in place) | x (3 rows)
constexpr int WD_H = 0, WD_X = 8 * W0_S;
static_assert(11 * W0_S <= CORE_LDS_FLOATS, "level-0 down images must fit the core arena");
static_assert(WD_X <= A_D0 && A_D0 + 8 * S1_ <= 11 * W0_S, "down0.down lands on the dead x rows");
// up phase: up2.same (16 x S1_) in the middle of the arena (dead while up2.same is computed), the eight level-0 rows
// (skip, then up3.convT) in two groups of four around it
constexpr int WU_U = A_SKIP2, WU_G0 = 0, WU_G1 = WU_U + 16 * S1_;
static_assert(WU_G0 + 4 * W0_S <= WU_U && WU_U + 16 * S1_ <= A_U2T && WU_G1 + 4 * W0_S <= CORE_LDS_FLOATS, "up-phase regions");
using W_down = LdsLayer<8, 0, 8, 2, 11, 8, -3, 0, 1, 1>;   // out n' = 2n + p reads sample 8n + tap - 3
using W_upT = LdsLayer<16, 0, 8, 4, 2, 1, -1, -2, 3, 1>;    // out sample 4m + p - 2 reads level-1 sample m + tap - 1

struct WindowArgs {
  CoreArgs c;       // d0 / u2s unused (they live in LDS)
  const float* af4[13];  // weights of the core layers regrouped for 16-byte loads (conv_lds_q4), null where unused
  const uint4* af3[6];   // down3.same .. up0.same (+ up1.same, U1B) as three-piece bf16 operands (conv_b3.h), B3 instantiation
  int af3_lines[6];      // their sizes in 128-byte lines (L2 warm-up)
  const uint4* af3_u2[2];  // U2B: up2.same's operand per input half (skip 1 | up2.convT), 16-channel K-steps (B3Steps<16, 7>)
  const uint4* af3_uT[2];  // U3B: up1.convT / up2.convT, rows (phase, channel)
  const uint4* af3_d12[2]; // D12B: down1.same (B3Steps<8, 7>), down2.same (B3Steps<16, 7>)
  const uint4* af3_inc;    // D0T: inc, rows (phase, channel), ONE K-step of eight taps x four channels (three + a zero one)
  const uint4* af3_d0s;    // D0T: down0.same, rows (phase, channel), two K-steps of four taps x eight channels (B3Steps<8, 8>)
  const float *bs_inc8, *bs_d0s;  // D0T: their biases [8] (BatchNorm folded)
  const uint4* af3_u3t;    // U3T: up3.convT, rows (phase, channel), ONE K-step of two taps x 16 channels (B3Steps<16, 2>), two m-tiles
  const uint4* af3_u3s;    // U3T: up3.same, rows (phase, channel), four K-steps of two taps x 16 channels (skip 0 | up3.convT) (B3Steps<16, 8>)
  const float *bs_u3t, *bs_u3s;  // U3T: their biases [8]
  const float* x;   // [B][3][ls] normalised input
  int ls_x;
  long ws_x;
  float* skip0;     // [B][8][ls] (down0.same): written in the down phase, read back in the up phase
  int ls_s;
  long ws_s;
  float* y;         // dense [B][3][T0]
  const f32x2 *w_inc, *b_inc, *w_same, *b_same, *w_up, *b_up;  // VALU weights: [cin][7][4] channel pairs, [4] bias pairs
  const float *af_down, *bs_down, *af_t, *bs_t;                // MFMA fragments of down0.down and up3.convT
  const float *w_out, *b_out;                                  // 1x1 output conv
  PreArgs pre;                                                 // has_pre: the kernel cuts and normalises its window itself
  int has_pre;                                                 // (annotate_batch_pre, as gather_normalize_kernel); else it reads x
};

struct SplitRowStore {  // up3.convT -> level-0 rows 0-3 (g0) and 4-7 (g1); zero outside the signal
  float *g0, *g1;
  __device__ __forceinline__ float* row(int co) const { return (co < 4) ? g0 + co * W0_S : g1 + (co - 4) * W0_S; }
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)(W0_S - 4)) row(co)[4 + t] = (t < T0) ? v : 0.f;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < T0; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { row(co)[4 + t] = v; }
  __device__ __forceinline__ void vec4(int co, int t, f32x4 v) const {  // OUT_OFF = -2: two 8-byte aligned halves
    float* d = row(co) + 4 + t;
    *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[1]};
    *reinterpret_cast<f32x2*>(d + 2) = f32x2{v[2], v[3]};
  }
};

// layers of the whole-network kernel that fetch their weights three channel blocks ahead (conv_lds ADEEP): the three
// up-path "same" convs (measured: up0.same 31.6 -> 30.4 k cycles, up1.same 29.3 -> 27.5 k, +1.7 % end to end; the
// down-path layers lose a little)
#define ADEEP_LAYER(LAYER) (LAYER::SN == 1 && LAYER::TAPS == 7 && LAYER::NB >= 3)
// layers of the whole-network kernel whose weights come as 16-byte loads (conv_lds_q4): the weight-heavy ones
// (the six-tile layers keep their dword path: 14 float4 of weights on top of 24 accumulators spill at 128 registers)
#define Q4_LAYER(LAYER) (LAYER::CB % 4 == 0 && LAYER::NB <= 3)
constexpr bool q4_layer_index(int i) { return (i >= 1 && i <= 9) || i == 11; }  // d1down .. up0.same, and the two register-resident two-tap layers
// B3: the five deepest layers (down3.same .. up0.same: 28 % of the kernel's cycles, 40 % of its fp32 MFMA issue) run on the
// bf16 matrix cores with exact three-piece operands (conv_b3.h); their images are the three-piece kind, placed in the
// same arena: down2.down's output / down3.down's output / up0.convT's output at A_R one after the other, the skip-3
// image at the end of the arena, the bottom image in the old skip-3 slot (A_Q), which then takes up0.same's fp32 output.
constexpr int B3_D2_NC = 54, B3_SK3_NC = 68, B3_D3_NC = 22, B3_BOT_NC = 18, B3_U0T_NC = 54;  // columns per image
constexpr int B3_D2_PS = B3_D2_NC * 40, B3_SK3_PS = B3_SK3_NC * 72, B3_D3_PS = B3_D3_NC * 72, B3_BOT_PS = B3_BOT_NC * 136,
              B3_U0T_PS = B3_U0T_NC * 72;                                                      // elements per piece
constexpr int B3_SK3_OFF = CORE_LDS_FLOATS * 2 - 3 * B3_SK3_PS;  // bf16 elements from the arena start
static_assert(B3_SK3_OFF % 8 == 0 && (A_R * 2) % 8 == 0 && (A_Q * 2) % 8 == 0, "16-byte aligned images");
static_assert(A_R * 2 + 3 * B3_U0T_PS <= B3_SK3_OFF && A_R * 2 + 3 * B3_D2_PS <= B3_SK3_OFF && 3 * B3_BOT_PS <= (A_R - A_Q) * 2,
              "three-piece images of the deep layers fit their slots");
// U1B (with B3): up1.same on the bf16 matrix cores as well.  Its two inputs (skip 2, up1.convT's output: 32 channels x 188 each)
// do not fit the arena as piece images side by side, so it runs in two K halves over ONE 48 KB image at the end of the arena
// (up2.convT's output slot): up1.convT writes its output there as pieces, eight waves (m-tile x four blocks of three n-tiles)
// take its taps, the image is refilled from the fp32 skip-2 rows, the same waves add the other half and store.
constexpr int B3_U1_NC = 200, B3_U1_PS = B3_U1_NC * 40;  // columns (sample t at column t + 3) / elements per piece
static_assert(A_U2T * 4 % 16 == 0 && A_U2T * 2 + 3 * B3_U1_PS <= CORE_LDS_FLOATS * 2 && B3_U1_NC >= 192 + 6, "up1.same piece image");
// U2B (with U1B): up2.same the same way.  Its inputs are 16 channels x 751 each: 74 KB as a chunk-plane piece image, one at a
// time.  up1.same's output moves to the skip-2 slot (dead once its pieces are made), up2.convT writes pieces into the 74 KB behind
// it, waves 0-7 (six n-tiles each, accumulators kept) take that half, the image is refilled from the fp32 skip-1 rows, the
// same waves add the other half and store into the skip-1 slot -- the up phase then finds up2.same at the start of the arena
// and its two groups of level-0 rows behind it.
constexpr int B3_U2_NC = 784, B3_U2_OFF = A_Q;  // columns (sample t at column t + 3) / float offset of the image
static_assert(B3_U2_OFF % 4 == 0 && B3_U2_OFF * 4 + 3 * B3Chunk<16, B3_U2_NC>::PS * 2 <= CORE_LDS_FLOATS * 4 &&
                  B3_U2_NC >= 48 * 16 + 8 && B3_U2_OFF >= A_SKIP2 + 32 * S2_,
              "up2.same piece image: behind up1.same's relocated output, inside the arena");
// U3B (with U2B): the two transposed convs in front of them on the bf16 matrix cores as well.  up0.same then writes ITS output
// as pieces (21.6 KB in the old skip-3 slot; up0.convT's image moves 1.1 KB up to make room), up1.convT reads them and
// writes up1.same's first image as before; up1.same writes its output as a chunk-plane piece image into the skip-2 slot
// (37 KB: it reaches 6.4 KB into the slot behind), up2.convT reads that and writes up2.same's first image, which therefore
// starts 6.4 KB later and has 768 columns instead of 784 (the 48th n-tile of up2.same, whose outputs nobody keeps, then reads a
// few columns of the neighbouring plane).
constexpr int B3_U0S_NC = 50, B3_U0S_PS = B3_U0S_NC * 72;                    // up0.same's output: sample t at column t + 1
constexpr int B3_U0T_SHIFT = ((A_Q * 2 + 3 * B3_U0S_PS - A_R * 2 + 7) / 8) * 8;  // bf16 elements: up0.convT's image starts this much later
constexpr int B3_U1S_NC = 194;                                                // up1.same's output (chunk planes): sample t at column t + 1
constexpr int B3_U2_NC3 = 768, B3_U2_OFF3 = (A_SKIP2 * 4 + 3 * B3Chunk<32, B3_U1S_NC>::PS * 2 + 15) / 16 * 4;  // floats
static_assert(B3_U0T_SHIFT >= 0 && A_R * 2 + B3_U0T_SHIFT + 3 * B3_U0T_PS <= B3_SK3_OFF && (A_R * 2 + B3_U0T_SHIFT) % 8 == 0,
              "up0.convT's image between up0.same's output pieces and the skip-3 image");
static_assert(B3_U2_OFF3 * 4 + 3 * B3Chunk<16, B3_U2_NC3>::PS * 2 <= CORE_LDS_FLOATS * 4 && B3_U2_NC3 >= 47 * 16 + 8 + 3 &&
                  B3_U1S_NC >= 192 + 2,
              "up1.same's output pieces and up2.same's image behind them fit the arena");
// D12B: down1.same and down2.same on the bf16 matrix cores (8 / 16 input channels: K-steps of four / two taps).  Their inputs
// arrive as chunk-plane piece images written by the fp32-MFMA strided convs in front of them (down0.down: two phases per
// m-tile, two v_permlane16_swap bring four channels of one sample to a lane; down1.down: plain), their outputs are the fp32
// skip rows as before.
constexpr int B3_D0_NC = 760, B3_D1_NC = 200;  // sample t at column t + 3
static_assert(A_D0 * 4 + 3 * B3Chunk<8, B3_D0_NC>::PS * 2 <= CORE_LDS_FLOATS * 4 && B3_D0_NC >= 47 * 16 + 7 && B3_D1_NC >= 192 + 7,
              "down0.down / down1.down as piece images");
// D0T (with D12B): inc and down0.same on the bf16 matrix cores, TIME-TILED.  Neither layer's input exists in fp32 form: the
// normalised window goes from the registers it was read into straight into bf16 pieces that rest inside the rows down0.same fills
// later (pn_window_kernel), inc's epilogue writes PIECES into a 1040-column ring [piece][parity][column / 2][8 channels] (sample s at
// column s mod 1040; it keeps the previous tile, whose tail down0.same's taps reach back into), down0.same reads the ring and
// writes its fp32 rows (the image of the strided conv behind it and the skip tensor).  Both GEMMs are M = 16 rows (output phase,
// channel), columns = sample pairs: inc K = 8 taps x 4 channels = ONE K-step, down0.same K = 8 taps x 8 channels = two K-steps.
// Six tiles of 512 samples = sixteen n-tiles per layer; in phase j every wave runs one n-tile of inc on tile j and one of
// down0.same on tile j - 1, eight samples behind (it never needs a sample inc has not produced), one barrier per phase:
// 6 x 16 x (6 + 12) = 1,728 MFMAs in place of 1,792 packed FMAs per lane.  plan_flags[5] = 8 keeps the VALU forms.
// (Round 4's slice-by-slice attempt converted inc's fp32 rows on the fly and lost to the packed FMAs.)
constexpr int D0T_TS = 512, D0T_TILES = 6, D0T_RING = 1040, D0T_HPS = B3Chunk<8, D0T_RING>::PS;
constexpr int D0T_RED = CORE_LDS_FLOATS - 256;  // the reduction scratch of the normalisation (floats): behind the ring
static_assert(D0T_TILES * D0T_TS >= W0_S - 4 + 8 && D0T_RING >= 2 * D0T_TS + 11 + 4 && D0T_RING % 2 == 0 &&
                  WD_X * 4 + 3 * D0T_HPS * 2 <= D0T_RED * 4 && (9 * 16 + 8) <= 256,
              "level-0 tiles: the ring behind the eight fp32 rows, the scratch behind the ring, inside the arena");
// U3T (with D0T): the level-0 UP path on the bf16 matrix cores, time-tiled the same way.  up2.same writes its output as a piece
// image (16 channels x 751 samples, chunk planes, at the start of the arena); behind it a 528-column ring holds the 16 input
// channels of up3.same as pieces -- chunk 0 the skip tensor (read back from memory tile by tile and split), chunk 1 the output of
// up3.convT (bf16 MFMA: rows (phase, channel) = two m-tiles, K = two taps x 16 channels = ONE K-step; its epilogue writes pieces)
// -- as even / odd column planes.  Twelve tiles of 256 samples: in phase j waves 8-15 produce tile j (samples 256 j - 2 ..:
// one (m-tile, n-tile) of the transposed conv and one (sample, channel quad) of the skip tensor per lane), waves 0-7 run up3.same
// (rows (phase, channel), K = 8 taps x 16 channels = four K-steps) on tile j - 1, eight samples behind, and finish it in
// registers: the two lanes that hold a sample's eight channels exchange their halves of the 1 x 1 conv (v_permlane16_swap),
// softmax, store.  No fp32 level-0 row exists in the up path any more.  plan_flags[5] = 9 keeps the VALU / fp32-MFMA form.
constexpr int U3T_TS = 256, U3T_TILES = 12, U3T_RING = 528, U3T_NCU = 768;
using U3T_QU = B3Chunk<16, U3T_NCU>;                      // up2.same's output: sample t at column t + 1
constexpr int U3T_PLN = U3T_RING / 2, U3T_MIR = 4;  // entries of a parity plane; its first four entries are repeated behind it, so
                                                    // that the four K-steps of a fragment (two columns apart) never wrap
constexpr int U3T_PL = (U3T_PLN + U3T_MIR) * 8, U3T_CH = 2 * U3T_PL, U3T_PS = 2 * U3T_CH;  // ring: bf16 per parity plane / chunk / piece
constexpr int U3T_RING_OFF = 3 * U3T_QU::PS;              // bf16 elements from the arena start: behind the U image
static_assert(U3T_TILES * U3T_TS >= T0 + 8 && U3T_TILES * U3T_TS < 6 * U3T_RING && U3T_RING >= 2 * U3T_TS + 11 + 4 && U3T_NCU >= T1 + 2 &&
                  (U3T_RING_OFF + 3 * U3T_PS) * 2 <= CORE_LDS_FLOATS * 4 && U3T_RING_OFF % 8 == 0,
              "level-0 up tiles: up2.same's piece image and the ring behind it fit the arena");
template <bool PIPE, bool B3, bool U1B = false, bool U2B = false, bool U3B = false, bool D12B = false, bool D0T = false, bool U3T = false>
// amdgpu_num_vgpr counts the VGPR half of the unified file on gfx90a+ (LLVM doubles it): 60 -> at most 120 registers per lane, so that
// four forward waves leave each SIMD the 32 registers the post-processing kernels need to run beside them (prepost.hip; a dozen
// one-off spills per window in the D0T form, none inside a loop)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(60))) void pn_window_kernel(const WindowArgs a) {
  static_assert(!D12B || B3, "D12B is a form of the B3 kernel");
  static_assert(!D0T || D12B, "D0T is a form of the D12B kernel");
  static_assert(!U3T || (D0T && U3B), "U3T is a form of the D0T kernel");
  static_assert(!U2B || U1B, "U2B relocates up1.same's output: needs the U1B form");
  static_assert(!U3B || U2B, "U3B builds on the U2B layout");
  constexpr int U2_NC = U3B ? B3_U2_NC3 : B3_U2_NC, U2_OFF = U3B ? B3_U2_OFF3 : B3_U2_OFF;
  // up phase: up2.same | level-0 rows 0-3 | level-0 rows 4-7
  constexpr int XU_U = U2B ? A_SKIP1 : WU_U, XU_G0 = U2B ? A_SKIP2 : WU_G0, XU_G1 = U2B ? A_SKIP2 + 4 * W0_S : WU_G1;
  static_assert(XU_G1 + 4 * W0_S <= CORE_LDS_FLOATS && XU_U + 16 * S1_ <= (U2B ? XU_G0 : A_U2T), "up-phase regions");
  constexpr int X_U1S = U2B ? A_SKIP2 : A_U1S;  // up1.same's output
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  // (the wave index as a scalar: item loops, block indices and the epilogues' "whole block in range" tests become
  // scalar code instead of per-lane predicates)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), win = blockIdx.x;
  constexpr int NTH = 1024, NWV = 16;
  unsigned long long* clk = a.c.clk;
  if (clk && tid == 0) clk[(long)win * 32 + 16] = wall_clock64();
#define WIN_STAMP(slot) \
  if (clk && tid == 0) clk[(long)win * 32 + (slot)] = __builtin_readcyclecounter();
// -DD0T_PROBE / -DU3T_PROBE (investigation builds only): the phases of the tiled level-0 down / up path stamp slots 2 .. instead of the core layers
#if defined(D0T_PROBE) || defined(U3T_PROBE)
#define CORE_WIN_STAMP(slot)
#else
#define CORE_WIN_STAMP(slot) WIN_STAMP(slot)
#endif
  WIN_STAMP(0)
  WIN_STAMP(18)
  // first workgroup of each XCD: touch one word per 128-byte line of the core weights (pn_core_kernel) -- on the FIRST launch of a
  // plan only (Net::warm_launches): from then on the weights are L2-resident from launch to launch (nothing but this kernel
  // runs on the chip), and pulling 2 MB through one CU's L1 made those eight workgroups, hence the launch, 9 us longer
  // (107.4 -> 97.8 us back to back, tools/ab_steps.py phasenet "0" "0,0,0,0,1")
  if (win < 8 && a.c.warm) {
    float sink = 0.f;
#define CORE_WARM(IDX, LAYER)                                                                          \
  for (int l = tid; l < LAYER::MT * LAYER::CB * LAYER::TAPS * 2; l += NTH) sink += a.c.af[IDX][l * 32];
    CORE_WARM(0, C_d1same) CORE_WARM(1, C_d1down) CORE_WARM(2, C_d2same) CORE_WARM(3, C_d2down)
    if constexpr (B3) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        for (int l = tid; l < a.af3_lines[i]; l += NTH) sink += __uint_as_float(reinterpret_cast<const unsigned*>(a.af3[i])[l * 32]);
    } else {
      CORE_WARM(4, C_d3same) CORE_WARM(5, C_d3down) CORE_WARM(6, C_d4same) CORE_WARM(7, C_u0T) CORE_WARM(8, C_u0same)
    }
    CORE_WARM(9, C_u1T) CORE_WARM(10, C_u1same) CORE_WARM(11, C_u2T) CORE_WARM(12, C_u2same)
#undef CORE_WARM
    if (sink == 1.2345678e-30f) a.y[0] = sink;  // never true: keeps the loads alive
  }
  // Waves without VALU work pull the weights of the NEXT VALU phase through the scalar cache (one dword per 64-byte
  // line): otherwise the twelve conv waves, in lock step, miss on every line together and each trip of the conv waits
  // out an L2 round trip (up3.same took 13.6 k cycles for its skip half and 8.5 k for the identical second half).
#define WIN_WARM_SCALAR(PTR, N_FLOATS)                                                      \
  {                                                                                         \
    float warm_ = 0.f;                                                                      \
    for (int l_ = 0; l_ < (N_FLOATS); l_ += 16) warm_ += as_scalars(reinterpret_cast<const float*>(PTR))[l_]; \
    asm volatile("" ::"s"(warm_));                                                          \
  }
  bool poisoned = false;                     // the window holds a NaN / Inf: its predictions are NaN (prepost.h)
  const int t0 = 4 * tid;                    // this lane's level-0 samples t0 .. t0 + 3 (VALU phases)
  const bool vconv = wave < W_WAVES;         // wave-uniform: runs the VALU convs
  const bool vstore = tid < W0_Q;            // lanes whose float4 lies inside an image row (751..755 store the zero margin)
  const bool own = tid < W_LANES;            // lanes holding signal samples

  // Round 6: every layer's first fragments are requested in front of the barrier BEFORE the layer (a layer's operand request
  // otherwise makes its trip to L2 with all sixteen waves waiting for it: 1.1 k cycles in front of down2.same, 2.9 k in front of
  // up0.convT -- tools/core_clock.py, profiles/r06_j_*)
  [[maybe_unused]] uint4 aw1[B3Steps<8, 7>::STEPS * 3];  // down1.same's operand
  // ================= level-0 down path: inc -> down0.same -> down0.down =================
  {
    float *H = lds + WD_H, *X = lds + WD_X;
    constexpr int MAXE = (T0 + NTH - 1) / NTH;
    float v[3][MAXE];  // the window: samples tid, tid + 1024, tid + 2048 of the three channels (D0T: the normalised ones, kept)
    // D0T: the A operands of inc and down0.same, the same 9 KB for every wave: fetched FIRST, so that their trip through the CU's
    // L1 (16 waves x 9 KB at 64 B per clock) passes under the window's trip from memory instead of in front of the first tile
    [[maybe_unused]] uint4 aI[3], aS[B3Steps<8, 8>::STEPS * 3];
    [[maybe_unused]] f32x4 bI, bS;
    if constexpr (D0T) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) aI[pc] = a.af3_inc[pc * 64 + lane];
      b3_load_a<8, 8>(a.af3_d0s, 0, lane, aS);
#pragma unroll
      for (int r = 0; r < 4; ++r) bI[r] = a.bs_inc8[4 * ((lane >> 4) & 1) + r], bS[r] = a.bs_d0s[4 * ((lane >> 4) & 1) + r];
    }
    if (a.has_pre) {
      // SeisBench annotate_batch_pre inside the kernel, arithmetic and reduction order of gather_normalize_kernel
      // (prepost.hip): window cut from the stream, per-channel mean, peak / std amplitude, scale — the window is read
      // once into registers and the normalised rows go straight into the x image (no input tensor in memory).
      const PreArgs& p = a.pre;
      float* red = lds + (D0T ? D0T_RED : 11 * W0_S);  // [9][NWV] partials (sums, maxima, minima), then stat[3][2] (free arena space behind the x rows / the ring)
      float* stat = red + 9 * NWV;
      long start = p.dense ? 0 : (long)(p.first_window + win) * p.step;
      if (!p.dense && start > p.N - T0) start = p.N - T0;  // tail window flush with the end
      const float* src = p.src + (p.dense ? (long)win * 3 * T0 : start);
      long cs = p.dense ? T0 : p.N;
      if (p.table) {
        const long* e = p.table + 3 * (p.first_window + win);
        src = p.src + e[0] + e[2];
        cs = e[1];
      }
      float sum[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
          const int t = tid + k * NTH;
          v[c][k] = t < T0 ? src[c * cs + t] : 0.f;
          sum[c] += v[c][k];
        }
      const bool one_pass = p.norm == VP_NORM_PEAK;  // uniform
      // norm = peak in ONE reduction round: max_k |v_k - mean| = max(vmax - mean, mean - vmin) bit for bit (rounding is
      // monotonic and symmetric), so the maxima and minima travel with the sums (two barriers and one reduction round
      // fewer in front of every window's first convolution; a NaN / Inf sample makes the mean non-finite: the window is
      // flagged and its predictions become NaN whatever the amplitude says)
      float vhi[3] = {-INFINITY, -INFINITY, -INFINITY}, vlo[3] = {INFINITY, INFINITY, INFINITY};
      if (one_pass) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < MAXE; ++k)
            if (tid + k * NTH < T0) vhi[c] = fmaxf(vhi[c], v[c][k]), vlo[c] = fminf(vlo[c], v[c][k]);
      }
      wave_sum3(sum[0], sum[1], sum[2]);  // (the DPP tree of wave_sum, three rows interleaved by hand: prepost.h)
      if (one_pass) {
        float nlo[3] = {-vlo[0], -vlo[1], -vlo[2]};
        wave_max3(vhi[0], vhi[1], vhi[2]);
        wave_max3(nlo[0], nlo[1], nlo[2]);
        if (lane == 0)
          for (int c = 0; c < 3; ++c) red[(3 + c) * NWV + wave] = vhi[c], red[(6 + c) * NWV + wave] = -nlo[c];
      }
      if (lane == 0)
        for (int c = 0; c < 3; ++c) red[c * NWV + wave] = sum[c];
      WIN_STAMP(29)
      if constexpr (!D0T) {
        for (int i = tid; i < 3 * (W0_S - T0); i += NTH) {  // zero margins of the x rows: samples -4 .. -1 and T0 .. 3019
          const int c = i / (W0_S - T0), k = i - c * (W0_S - T0);
          X[c * W0_S + (k < 4 ? k : T0 + k)] = 0.f;
        }
      }
      __syncthreads();
      if (tid < 3) {
        float acc = 0.f;
        for (int i = 0; i < NWV; ++i) acc += red[tid * NWV + i];
        const float mu = acc / (float)T0;
        stat[tid * 2] = mu;
        if (one_pass) {
          float h = red[(3 + tid) * NWV], l = red[(6 + tid) * NWV];
          for (int i = 1; i < NWV; ++i) h = fmaxf(h, red[(3 + tid) * NWV + i]), l = fminf(l, red[(6 + tid) * NWV + i]);
          stat[tid * 2 + 1] = fmaxf(h - mu, mu - l);
        }
      }
      __syncthreads();
      WIN_STAMP(30)
      const float mean[3] = {stat[0], stat[2], stat[4]};
      if (!one_pass) {
      float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
          const int t = tid + k * NTH;
          if (t < T0) {
            const float d = v[c][k] - mean[c];
            m[c] += d * d;
          }
        }
      __syncthreads();
      for (int c = 0; c < 3; ++c) {
        const float r = wave_sum(m[c]);
        if (lane == 0) red[c * NWV + wave] = r;
      }
      __syncthreads();
      if (tid < 3) {
        const float* r = red + tid * NWV;
        float acc = r[0];
        for (int i = 1; i < NWV; ++i) acc = acc + r[i];
        stat[tid * 2 + 1] = acc;
      }
      __syncthreads();
      }
      for (int c = 0; c < 3; ++c) poisoned |= !isfinite(stat[2 * c]) || !isfinite(stat[2 * c + 1]);
      float amp[3];
      if (p.per_comp) {
        for (int c = 0; c < 3; ++c) amp[c] = (p.norm == VP_NORM_PEAK) ? stat[2 * c + 1] : sqrtf(stat[2 * c + 1] / (float)(T0 - 1));
      } else {
        const float g = (p.norm == VP_NORM_PEAK) ? fmaxf(stat[1], fmaxf(stat[3], stat[5]))
                                                 : sqrtf((stat[1] + stat[3] + stat[5]) / (float)(3 * T0 - 1));
        amp[0] = amp[1] = amp[2] = g;
      }
      const NormDiv den[3] = {norm_div_prepare(amp[0] + p.norm_eps), norm_div_prepare(amp[1] + p.norm_eps),
                              norm_div_prepare(amp[2] + p.norm_eps)};
      if (p.taper > 0) {  // (uniform; PhaseNet's default is no taper: the plain loop below then carries no branch per sample)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < MAXE; ++k) {
            const int t = tid + k * NTH;
            if (t < T0) {
              float o = norm_div(v[c][k] - mean[c], den[c]);
              const int e = (t < p.taper) ? t : ((T0 - 1 - t < p.taper) ? T0 - 1 - t : -1);
              if (e >= 0) o *= 0.5f * (1.f + cosf(3.14159265358979323846f * (1.f + (float)e / (float)(p.taper - 1))));
              if constexpr (D0T) v[c][k] = o;
              else X[c * W0_S + 4 + t] = o;
            }
          }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int k = 0; k < MAXE; ++k) {
            const int t = tid + k * NTH;
            if constexpr (D0T) v[c][k] = norm_div(v[c][k] - mean[c], den[c]);
            else if (k + 1 < MAXE || t < T0) X[c * W0_S + 4 + t] = norm_div(v[c][k] - mean[c], den[c]);
          }
      }
    } else if constexpr (D0T) {  // the normalised rows of the input tensor, into the same registers
      const float* src = a.x + (long)win * a.ws_x + HALO;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
          const int t = tid + k * NTH;
          v[c][k] = t < T0 ? src[(long)c * a.ls_x + t] : 0.f;
        }
    } else {  // x rows: sample 4q - 4 .. 4q - 1 at float4 q; physical index HALO + 4q - 4 (16-byte aligned)
      const float* src = a.x + (long)win * a.ws_x;
      for (int i = tid; i < 3 * W0_Q; i += NTH) {
        const int c = i / W0_Q, q = i - c * W0_Q;
        const int p = 4 * q + HALO - 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p + 3 < a.ls_x) v = *reinterpret_cast<const float4*>(src + (long)c * a.ls_x + p);
        *reinterpret_cast<float4*>(X + c * W0_S + 4 * q) = v;
      }
    }
    if constexpr (!D0T) {
      if (tid < 8) *reinterpret_cast<float4*>(H + tid * W0_S) = make_float4(0.f, 0.f, 0.f, 0.f);  // samples -4 .. -1: left padding
    } else if (tid < 8) {  // (word 3 of rows 0-5 takes x's first sample below: zeroed behind the tile loop)
      H[tid * W0_S] = H[tid * W0_S + 1] = H[tid * W0_S + 2] = 0.f;
      if (tid >= 6) H[tid * W0_S + 3] = 0.f;
    }
    if constexpr (D0T) {
      bf16_t* const l16 = reinterpret_cast<bf16_t*>(lds);
      bf16_t* const HP = l16 + WD_X * 2;  // inc's output: the ring, behind the eight fp32 rows
      unsigned* const HU = reinterpret_cast<unsigned*>(H);
      const int g = lane >> 4, n = lane & 15, ph = g >> 1, quad = g & 1;  // GEMM rows 4 g .. 4 g + 3 = (phase ph, channels 4 quad ..)
      // The normalised window as bf16 pieces INSIDE the rows that down0.same fills later: piece pc of sample t rests in rows
      // 2 pc (channels 0, 1) and 2 pc + 1 (channel 2 and a zero) at word 3 + t -- one word below the place of down0.same's sample
      // t, which is written a tile (512 samples) behind inc's reads.  Stored once, straight from the registers the window was
      // read into: no fp32 x image, no per-tile copies.
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int t = tid + k * NTH;
        if (k + 1 < MAXE || t < T0) {
          const float q0 = v[0][k], q1 = v[1][k], q2 = v[2][k];
          const unsigned h0 = pack_bf16x2(q0, q1), h1 = pack_bf16x2(q2, 0.f);
          const float r0 = q0 - bf16_lo(h0), r1 = q1 - bf16_hi(h0), r2 = q2 - bf16_lo(h1);
          const unsigned m0 = pack_bf16x2(r0, r1), m1 = pack_bf16x2(r2, 0.f);
          unsigned* const xp = HU + 3 + t;
          xp[0] = h0;
          xp[W0_S] = h1;
          xp[2 * W0_S] = m0;
          xp[3 * W0_S] = m1;
          xp[4 * W0_S] = pack_bf16x2(r0 - bf16_lo(m0), r1 - bf16_hi(m0));
          xp[5 * W0_S] = pack_bf16x2(r2 - bf16_lo(m1), 0.f);
        }
      }
      if (tid < 6 * (W0_S - 3 - T0)) {  // zeros behind the signal: words 3 + T0 .. of the six rows
        const int r = tid / (W0_S - 3 - T0), c = tid - r * (W0_S - 3 - T0);
        HU[r * W0_S + 3 + T0 + c] = 0u;
      }
      // the ring as two planes per piece, even columns | odd columns (sample s at column s mod 1040): the sixteen lanes of a
      // fragment step two columns at a time and so read consecutive 16-byte chunks of ONE plane
      auto ring_at = [](const int c) { return (c & 1) * (D0T_RING / 2 * 8) + (c >> 1) * 8; };
      if (tid < 48)  // ring columns 1024 .. 1039 <-> samples -16 .. -1: zeros
        *reinterpret_cast<uint4*>(HP + (tid >> 4) * D0T_HPS + ring_at(D0T_RING - 16 + (tid & 15))) = make_uint4(0u, 0u, 0u, 0u);
      // Every wave runs one n-tile (32 samples) of inc on tile j AND one of down0.same on tile j - 1 per phase: two independent
      // MFMA chains and epilogues per wave.  (Measured on the way here, tools/d0t_phase_probe.py: a phase costs the SUM of what
      // its waves issue -- scalar instructions and branches included, the CU has one scalar unit -- plus the latency of each
      // wave's one serial chain LDS read -> MFMAs -> epilogue -> barrier; thirteen phases of eight-wave roles with per-tile x
      // copies took 2.1 k cycles each, 870 of them the copies.)
      // the skip tensor leaves tile by tile, two phases behind down0.same (one 16-byte store per lane and phase: all 96 KB of a
      // window behind the last tile made every CU of the chip store at once, 5.3 k cycles)
      const int skip_ch = tid >> 7, skip_q = tid & 127;
      float* const skip_row = a.skip0 + (long)win * a.ws_s + HALO + (long)skip_ch * a.ls_s;
      auto store_skip_tile = [&](const int k) {  // samples 512 k - 8 + 4 q .. + 3
        const int ts = D0T_TS * k - 8 + 4 * skip_q;
        if (ts >= 0 && ts < T0)
          *reinterpret_cast<f32x4*>(skip_row + ts) = *reinterpret_cast<const f32x4*>(H + skip_ch * W0_S + 4 + ts);
      };
      WIN_STAMP(31)
      __syncthreads();
      WIN_STAMP(19)
#define D0T_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, W), __builtin_bit_cast(bf16x8_b3, X), ACC, 0, 0, 0)
      float aD[W_down::CB * W_down::TAPS], bD[4];
      int cinc = 32 * wave + 2 * n + ph;                    // inc: ring column of this lane's sample of tile j
      int csame = D0T_RING - 11 + 32 * wave + 2 * n + g;     // down0.same: ring column of tap g's sample for tile j - 1
      csame = csame >= D0T_RING ? csame - D0T_RING : csame;
      const unsigned* xq = HU + 32 * wave + 2 * n + 2 * g;   // inc: words 3 + (sample - 3 + tap), tap = 2 g, of tile 0
      float* hq = H + 4 * quad * W0_S + 4 - 8 + 32 * wave + 2 * n + ph;  // down0.same: this lane's sample of tile 0
#pragma unroll
      for (int j = 0; j <= D0T_TILES; ++j) {
        uint4 bi[3], bs[2][3];
        if (j < D0T_TILES) {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const uint2 lo = *reinterpret_cast<const uint2*>(xq + 2 * pc * W0_S), hi = *reinterpret_cast<const uint2*>(xq + (2 * pc + 1) * W0_S);
            bi[pc] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
        if (j > 0) {
          int c1 = csame + 4;
          c1 = c1 >= D0T_RING ? c1 - D0T_RING : c1;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            bs[0][pc] = *reinterpret_cast<const uint4*>(HP + pc * D0T_HPS + ring_at(csame));
            bs[1][pc] = *reinterpret_cast<const uint4*>(HP + pc * D0T_HPS + ring_at(c1));
          }
        }
        if (j >= 2) store_skip_tile(j - 2);
        if (j == D0T_TILES) {  // the A operand of down0.down, into the registers inc's operand has left
          load_areg<W_down>(a.af_down, 0, lane, aD);
          load_biasreg<W_down>(a.bs_down, 0, lane, bD);
        }
        f32x4 ia = bI, sa = {0.f, 0.f, 0.f, 0.f}, sb = bS;
        if (j < D0T_TILES) {  // inc, n-tile `wave` of tile j: samples 512 j + 32 wave + 2 n + ph; smallest products first, bias as the accumulator input
          D0T_MFMA(ia, aI[2], bi[0]);
          D0T_MFMA(ia, aI[1], bi[1]);
          D0T_MFMA(ia, aI[0], bi[2]);
          D0T_MFMA(ia, aI[1], bi[0]);
          D0T_MFMA(ia, aI[0], bi[1]);
          D0T_MFMA(ia, aI[0], bi[0]);
        }
        if (j > 0) {  // down0.same, n-tile `wave` of tile j - 1: samples 512 (j - 1) - 8 + 32 wave + 2 n + ph read inc's samples .. - 3 + tap, tap = g + 4 step
          D0T_MFMA(sa, aS[2], bs[0][0]);  // one chain per K-step
          D0T_MFMA(sb, aS[5], bs[1][0]);
          D0T_MFMA(sa, aS[1], bs[0][1]);
          D0T_MFMA(sb, aS[4], bs[1][1]);
          D0T_MFMA(sa, aS[0], bs[0][2]);
          D0T_MFMA(sb, aS[3], bs[1][2]);
          D0T_MFMA(sa, aS[1], bs[0][0]);
          D0T_MFMA(sb, aS[4], bs[1][0]);
          D0T_MFMA(sa, aS[0], bs[0][1]);
          D0T_MFMA(sb, aS[3], bs[1][1]);
          D0T_MFMA(sa, aS[0], bs[0][0]);
          D0T_MFMA(sb, aS[3], bs[1][0]);
        }
        if (j < D0T_TILES) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(ia[r], 0.f);
          if (D0T_TS * (j + 1) > T0) {  // (uniform) the tile that meets the end of the signal: zeros beyond it
            const int s = D0T_TS * j + 32 * wave + 2 * n + ph;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = s < T0 ? o[r] : 0.f;
          }
          b3_store4(HP + ring_at(cinc), D0T_HPS, 0, 0, 4 * quad, o);
          cinc += D0T_TS;
          cinc = cinc >= D0T_RING ? cinc - D0T_RING : cinc;
          xq += D0T_TS;
        }
        if (j > 0) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(sa[r] + sb[r], 0.f);
          if (j == 1 || D0T_TS * j > T0) {  // (uniform) the tiles that meet the ends of the signal
            const int t = D0T_TS * (j - 1) - 8 + 32 * wave + 2 * n + ph;
            if ((unsigned)t < (unsigned)(W0_S - 4)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) hq[r * W0_S] = t < T0 ? o[r] : 0.f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) hq[r * W0_S] = o[r];
          }
          csame += D0T_TS;
          csame = csame >= D0T_RING ? csame - D0T_RING : csame;
          hq += D0T_TS;
        }
        lds_barrier();
        if (j == 0) { WIN_STAMP(20) }
#ifdef D0T_PROBE
        WIN_STAMP(2 + j)
#endif
      }
#undef D0T_MFMA
      // down0.same rests in H (fp32); the ring gives way to down0.down's piece image
      b3c_zero_rest<8, B3_D0_NC>(l16 + A_D0 * 2, 3, B3_D0_NC, tid, NTH);
      if (tid < 6) H[tid * W0_S + 3] = 0.f;  // word 3 = sample -1 of down0.same (padding): x's first sample rested there
      store_skip_tile(D0T_TILES - 1);
      lds_barrier();
      WIN_STAMP(21)
      const B3PairStoreC<8, B3_D0_NC> st{l16 + A_D0 * 2, 3, T1};
      conv_lds_areg<W_down, W0_S, 4, W0_S, 4>(H, H, aD, bD, 0, (T1 + 1) / 2, st, wave, NWV, lane);
    } else {
    WIN_STAMP(31)
    __syncthreads();
    WIN_STAMP(19)
    if (vconv) {  // inc: Conv1d(3, 8, 7, same, bias) + BN + ReLU
      f32x2 acc[4][4];
      valu_bias(acc, a.b_inc);
      valu_conv7_r4<3, W0_S>(X, as_weights(a.w_inc), t0, acc);
      if (vstore) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          f32x4 lo, hi;
          valu_finish(acc, c, t0, &lo, &hi);
          *reinterpret_cast<f32x4*>(H + (2 * c) * W0_S + 4 + t0) = lo;
          *reinterpret_cast<f32x4*>(H + (2 * c + 1) * W0_S + 4 + t0) = hi;
        }
      }
    } else {
      WIN_WARM_SCALAR(a.w_same, 8 * 7 * 8)
    }
    __syncthreads();
    WIN_STAMP(20)
    {  // down0.same: Conv1d(8, 8, 7, same) + BN + ReLU; the result overwrites inc in place (image of the strided conv)
       // and goes to memory as the skip tensor (it stays in this XCD's L2 for the up phase)
      float aD[W_down::CB * W_down::TAPS], bD[4];  // A fragments of down0.down: fetched under the FMAs of down0.same
      load_areg<W_down>(a.af_down, 0, lane, aD);
      load_biasreg<W_down>(a.bs_down, 0, lane, bD);
      f32x2 acc[4][4];
      if (vconv) {
        valu_bias(acc, a.b_same);
        valu_conv7_r4<8, W0_S>(H, as_weights(a.w_same), t0, acc);
      }
      lds_barrier();  // every lane has read its inc window
      f32x4 lo[4], hi[4];
      if (vconv) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          valu_finish(acc, c, t0, &lo[c], &hi[c]);
          if (vstore) {
            *reinterpret_cast<f32x4*>(H + (2 * c) * W0_S + 4 + t0) = lo[c];
            *reinterpret_cast<f32x4*>(H + (2 * c + 1) * W0_S + 4 + t0) = hi[c];
          }
        }
      }
      if constexpr (D12B) {  // the x rows are dead: down0.down takes their place (D12B: as a piece image)
        b3c_zero_rest<8, B3_D0_NC>(reinterpret_cast<bf16_t*>(lds) + A_D0 * 2, 3, B3_D0_NC, tid, NTH);
      } else {
        zero_halo<8, S1_, T1>(lds + A_D0, tid, NTH);
      }
      lds_barrier();
      if (own) {  // the float4 holding sample T0 - 1 also rewrites up to three zeros of the row's right margin
        float* d = a.skip0 + (long)win * a.ws_s + HALO + t0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          *reinterpret_cast<f32x4*>(d + (long)(2 * c) * a.ls_s) = lo[c];
          *reinterpret_cast<f32x4*>(d + (long)(2 * c + 1) * a.ls_s) = hi[c];
        }
      }
      WIN_STAMP(21)
      // down0.down: Conv1d(8, 8, 7, stride 4, pad 3) + BN + ReLU on the MFMA, straight into the core's input image
      if constexpr (D12B) {
        const B3PairStoreC<8, B3_D0_NC> st{reinterpret_cast<bf16_t*>(lds) + A_D0 * 2, 3, T1};
        conv_lds_areg<W_down, W0_S, 4, W0_S, 4>(H, H, aD, bD, 0, (T1 + 1) / 2, st, wave, NWV, lane);
      } else {
        RangeStore<S1_, IB> st{lds + A_D0, T1};
        conv_lds_areg<W_down, W0_S, 4, W0_S, 4>(H, H, aD, bD, 0, (T1 + 1) / 2, st, wave, NWV, lane);
      }
    }
    }
    if constexpr (D12B) {
      b3_load_a<8, 7>(a.af3_d12[0], 0, lane, aw1);
    }
    lds_barrier();  // not __syncthreads(): the skip rows drain to memory under the first core layers
    WIN_STAMP(22)
    WIN_STAMP(1)
  }

  // ================= levels 1-4 down, up0 .. up2 (pn_core_kernel) =================
  int stamp = 2;
#define CORE_LAYER(IDX, LAYER, IN1, SI1, IN2, SI2, B2, OUT, SO, OB, STORE, CO, COLS, LOUT)                         \
  {                                                                                                                \
    STORE<SO, OB> st{{lds + (OUT), (LOUT)}};                                                                      \
    zero_halo<CO, SO, LOUT, OB>(lds + (OUT), tid, NTH);                                                          \
    if constexpr (Q4_LAYER(LAYER)) {                                                                              \
      conv_lds_q4<LAYER, SI1, IB, SI2, B2>(lds + (IN1), lds + (IN2), a.af4[IDX], a.c.bs[IDX], (COLS), st, wave, NWV, lane); \
    } else {                                                                                                       \
      conv_lds<LAYER, SI1, IB, SI2, B2, PIPE, (LAYER::NB < BDB_MAX_NB), ADEEP_LAYER(LAYER)>(lds + (IN1), lds + (IN2), a.c.af[IDX], a.c.bs[IDX], (COLS), st, wave, NWV, lane); \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    CORE_WIN_STAMP(stamp)                                                                                               \
    ++stamp;                                                                                                       \
  }
#define CORE_LAYER_AREG(IDX, LAYER, IN1, SI1, OUT, SO, OB, STORE, CO, COLS, LOUT, WMT, WFIRST, WSTEP)                       \
  {                                                                                                                \
    STORE<SO, OB> st{{lds + (OUT), (LOUT)}};                                                                      \
    zero_halo<CO, SO, LOUT, OB>(lds + (OUT), tid, NTH);                                                          \
    if ((WMT) < LAYER::MT && (WFIRST) < ((((COLS) + 15) >> 4) + LAYER::NB - 1) / LAYER::NB) {                    \
      float ar[LAYER::CB * LAYER::TAPS], br[4];                                                                    \
      load_areg4<LAYER>(a.af4[IDX], (WMT), lane, ar);                                                              \
      load_biasreg<LAYER>(a.c.bs[IDX], (WMT), lane, br);                                                           \
      conv_lds_areg<LAYER, SI1, IB, SI1, IB>(lds + (IN1), lds + (IN1), ar, br, (WMT), (COLS), st, (WFIRST), (WSTEP), lane); \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    CORE_WIN_STAMP(stamp)                                                                                               \
    ++stamp;                                                                                                       \
  }
  // the two fp32 strided layers' first superblock (seven 16-byte fragments per lane), requested a layer ahead as well
  // (conv_lds_q4_request): down1.down 5.6 -> 4.6 k cycles, down2.down 6.2 -> 5.5 k
  [[maybe_unused]] f32x4 qa_d1d[C_d1down::TAPS], qa_d2d[C_d2down::TAPS];
  if constexpr (D12B) {
    bf16_t* const iD0 = reinterpret_cast<bf16_t*>(lds) + A_D0 * 2;  // written by down0.down
    bf16_t* const iD1 = reinterpret_cast<bf16_t*>(lds) + A_D1 * 2;
    const int g = lane >> 4, n = lane & 15;
    {  // down1.same: one m-tile, 47 n-tiles: three per wave
      zero_halo<16, S1_, T1, IB>(lds + A_SKIP1, tid, NTH);
      auto& aw = aw1;
      float biasv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[0][4 * g + r];
      const int colb = wave * 48;
      b3c_mac_tiles<8, B3_D0_NC, 7, 3>(b3c_lane_ptr<8, B3_D0_NC, 7>(iD0, colb, lane), aw, [&](const int j, const f32x4 acc) {
        const int t = colb + j * 16 + n;
        if (t < T1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[A_SKIP1 + (4 * g + r) * S1_ + IB + t] = fmaxf(acc[r] + biasv[r], 0.f);
        }
      });
      if constexpr (Q4_LAYER(C_d1down)) {
        conv_lds_q4_request<C_d1down>(a.af4[1], T2, wave, lane, qa_d1d);
        lds_barrier();
      } else {
        __syncthreads();
      }
      CORE_WIN_STAMP(stamp)
      ++stamp;
    }
    [[maybe_unused]] uint4 aw2[B3Steps<16, 7>::STEPS * 3];  // down2.same's operand (requested a layer ahead, behind down1.down's MFMAs)
    {  // down1.down (fp32 MFMA, strided) -> piece image
      const B3BlockStoreC<16, B3_D1_NC> st{iD1, 3, T2};
      b3c_zero_rest<16, B3_D1_NC>(iD1, 3, 3 + 192, tid, NTH);
      if constexpr (Q4_LAYER(C_d1down)) {
        conv_lds_q4_requested<C_d1down, S1_, IB, S1_, IB>(lds + A_SKIP1, lds + A_SKIP1, a.af4[1], a.c.bs[1], T2, st, wave, NWV, lane, qa_d1d);
      } else {
        conv_lds<C_d1down, S1_, IB, S1_, IB, PIPE, (C_d1down::NB < BDB_MAX_NB), ADEEP_LAYER(C_d1down)>(lds + A_SKIP1, lds + A_SKIP1, a.c.af[1], a.c.bs[1], T2, st, wave, NWV, lane);
      }
      if (wave < 8) b3_load_a<16, 7>(a.af3_d12[1], wave & 1, lane, aw2);  // travels under the barrier wait
      lds_barrier();  // not __syncthreads(): it would wait for the request just made
      CORE_WIN_STAMP(stamp)
      ++stamp;
    }
    {  // down2.same: wave = (m-tile, block of three n-tiles), eight waves
      zero_halo<32, S2_, T2, IB>(lds + A_SKIP2, tid, NTH);
      if (wave < 8) {
        const int mt = wave & 1, colb = (wave >> 1) * 48;
        auto& aw = aw2;
        float biasv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[2][mt * 16 + 4 * g + r];
        b3c_mac_tiles<16, B3_D1_NC, 7, 3>(b3c_lane_ptr<16, B3_D1_NC, 7>(iD1, colb, lane), aw, [&](const int j, const f32x4 acc) {
          const int t = colb + j * 16 + n;
          if (t < T2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[A_SKIP2 + (mt * 16 + 4 * g + r) * S2_ + IB + t] = fmaxf(acc[r] + biasv[r], 0.f);
          }
        });
      }
      if constexpr (B3) {  // (down2.down's conv_lds_q4 call sits in the B3 block below)
        conv_lds_q4_request<C_d2down>(a.af4[3], T3, wave, lane, qa_d2d);
        lds_barrier();
      } else {
        __syncthreads();
      }
      CORE_WIN_STAMP(stamp)
      ++stamp;
    }
  } else {
  CORE_LAYER(0, C_d1same, A_D0, S1_, A_D0, S1_, IB, A_SKIP1, S1_, IB, RangeStoreS, 16, T1, T1)
  CORE_LAYER(1, C_d1down, A_SKIP1, S1_, A_SKIP1, S1_, IB, A_D1, S2_, IB, RangeStoreS, 16, T2, T2)
  CORE_LAYER(2, C_d2same, A_D1, S2_, A_D1, S2_, IB, A_SKIP2, S2_, IB, RangeStoreS, 32, T2, T2)
  }
  [[maybe_unused]] uint4 q_u1t[4][3];  // up1.convT's first fragments (requested under up0.same's closing barrier)
  [[maybe_unused]] uint4 q_u1s[4][3];  // up1.same's, either K half
  [[maybe_unused]] uint4 aT2[B3Steps<32, 2>::STEPS * 3];  // up2.convT's operand
  [[maybe_unused]] uint4 aw_u2[B3Steps<16, 7>::STEPS * 3];  // up2.same's first operand (the K half of up2.convT's channels)
  constexpr int X_U0S = B3 ? A_Q : A_U0S, X_U1T = A_U1T;  // B3: up0.same lands in the old skip-3 slot (up1.same's output slot later on)
  if constexpr (B3) {
    bf16_t* l16 = reinterpret_cast<bf16_t*>(lds);
    const B3Image<32> iD2{l16 + A_R * 2, B3_D2_PS, 3};
    const B3Image<64> iSK3{l16 + B3_SK3_OFF, B3_SK3_PS, 3}, iD3{l16 + A_R * 2, B3_D3_PS, 3}, iU0T{l16 + A_R * 2 + (U3B ? B3_U0T_SHIFT : 0), B3_U0T_PS, 3};
    const B3Image<128> iBOT{l16 + A_Q * 2, B3_BOT_PS, 1};
#define B3_END          \
  lds_barrier(); /* not __syncthreads(): the next layer's first fragments are in flight */ \
  CORE_WIN_STAMP(stamp)      \
  ++stamp;
    // the first K-steps of every layer's first item are requested in front of the barrier BEFORE the layer
    // (conv_b3_request): a layer's first fragments otherwise make their trip to L2 with all sixteen waves waiting
    [[maybe_unused]] uint4 q_d3s[4][3], q_d3d[4][3], q_d4s[4][3], q_u0t[4][3], q_u0s[3][3];
    {  // down2.down (fp32 MFMA, strided) -> three-piece image
      B3BlockStore<32> st{{iD2.img, iD2.ps, iD2.c0, T3, B3_D2_NC}};
      st.zero_rest(3, 3 + 48, tid, NTH);
      if constexpr (D12B) conv_lds_q4_requested<C_d2down, S2_, IB, S2_, IB>(lds + A_SKIP2, lds + A_SKIP2, a.af4[3], a.c.bs[3], T3, st, wave, NWV, lane, qa_d2d);
      else conv_lds_q4<C_d2down, S2_, IB, S2_, IB>(lds + A_SKIP2, lds + A_SKIP2, a.af4[3], a.c.bs[3], T3, st, wave, NWV, lane);
      conv_b3_request<C_d3same>(a.af3[0], T3, wave, lane, q_d3s);
      B3_END
    }
    {  // down3.same
      B3Store<64> st{iSK3.img, iSK3.ps, iSK3.c0, T3, B3_SK3_NC};
      st.zero_rest(3, 3 + 48, tid, NTH);
      conv_b3_requested<C_d3same, false, 32, 32>(iD2, iD2, a.af3[0], a.c.bs[4], T3, st, wave, NWV, lane, q_d3s);
      conv_b3_request<C_d3down>(a.af3[1], T4, wave, lane, q_d3d);
      B3_END
    }
    {  // down3.down
      B3Store<64> st{iD3.img, iD3.ps, iD3.c0, T4, B3_D3_NC};
      st.zero_rest(3, 3 + 16, tid, NTH);
      conv_b3_requested<C_d3down, false, 64, 64>(iSK3, iSK3, a.af3[1], a.c.bs[5], T4, st, wave, NWV, lane, q_d3d);
      conv_b3_request<C_d4same>(a.af3[2], T4, wave, lane, q_d4s);
      B3_END
    }
    {  // down4.same
      B3Store<128> st{iBOT.img, iBOT.ps, iBOT.c0, T4, B3_BOT_NC};
      st.zero_rest(1, 1 + 16, tid, NTH);
      conv_b3_requested<C_d4same, false, 64, 64>(iD3, iD3, a.af3[2], a.c.bs[6], T4, st, wave, NWV, lane, q_d4s);
      conv_b3_request<C_u0T>(a.af3[3], T4 + 1, wave, lane, q_u0t);
      B3_END
    }
    {  // up0.convT: rows ordered (phase, channel); samples 4 c + phase - 1, columns c in [0, 16)
      B3Store<64> st{iU0T.img, iU0T.ps, iU0T.c0, T3, B3_U0T_NC};
      st.zero_rest(2, B3_U0T_NC, tid, NTH);
      conv_b3_requested<C_u0T, true, 128, 128>(iBOT, iBOT, a.af3[3], a.c.bs[7], T4 + 1, st, wave, NWV, lane, q_u0t);
      conv_b3_request<C_u0same, 2>(a.af3[4], T3, wave, lane, q_u0s);
      B3_END
    }
    if constexpr (U3B) {  // up0.same: cat(skip 3, up0.convT) -> three-piece image for up1.convT
      B3Store<64> st{l16 + A_Q * 2, B3_U0S_PS, 1, T3, B3_U0S_NC};
      st.zero_rest(1, 1 + 48, tid, NTH);
      conv_b3_requested<C_u0same, false, 64, 64, decltype(st), 2>(iSK3, iU0T, a.af3[4], a.c.bs[8], T3, st, wave, NWV, lane, q_u0s);
      if (U1B) conv_b3_request<C_u1T>(a.af3_uT[0], T3 + 1, wave, lane, q_u1t);
      B3_END
    } else {  // up0.same: cat(skip 3, up0.convT) -> fp32 image for up1.convT
      F32QuadStore<S3_, IB> st{lds + X_U0S, T3};
      zero_halo<64, S3_, T3, IB>(lds + X_U0S, tid, NTH);
      conv_b3_requested<C_u0same, false, 64, 64, decltype(st), 2>(iSK3, iU0T, a.af3[4], a.c.bs[8], T3, st, wave, NWV, lane, q_u0s);
      B3_END
    }
#undef B3_END
  } else {
  CORE_LAYER(3, C_d2down, A_SKIP2, S2_, A_SKIP2, S2_, IB, A_D2, S3_, IB, RangeStoreS, 32, T3, T3)
  CORE_LAYER(4, C_d3same, A_D2, S3_, A_D2, S3_, IB, A_SKIP3, S3_, IB, RangeStoreS, 64, T3, T3)
  CORE_LAYER(5, C_d3down, A_SKIP3, S3_, A_SKIP3, S3_, IB, A_D3, S4_, IB, RangeStoreS, 64, T4, T4)
  CORE_LAYER(6, C_d4same, A_D3, S4_, A_D3, S4_, IB, A_BOT, S4_, IB, RangeStoreS, 128, T4, T4)
  CORE_LAYER(7, C_u0T, A_BOT, S4_, A_BOT, S4_, IB, A_U0T, S3_, TB, RangeStoreV, 64, T4 + 1, T3)
  CORE_LAYER(8, C_u0same, A_SKIP3, S3_, A_U0T, S3_, TB, A_U0S, S3_, IB, RangeStoreS, 64, T3, T3)
  }
  if constexpr (B3 && U1B) {
    const B3Image<32> iP{reinterpret_cast<bf16_t*>(lds) + A_U2T * 2, B3_U1_PS, 3};
    if constexpr (U3B) {  // up1.convT on the bf16 matrix cores: rows (phase, channel), samples 4 c + phase - 1, columns c in [0, 48)
      const B3Image<64> iU0S{reinterpret_cast<bf16_t*>(lds) + A_Q * 2, B3_U0S_PS, 1};
      B3Store<32> st{iP.img, iP.ps, iP.c0, T2, B3_U1_NC};
      st.zero_rest(2, 2 + 192, tid, NTH);
      conv_b3_requested<C_u1T, true, 64, 64>(iU0S, iU0S, a.af3_uT[0], a.c.bs[9], T3 + 1, st, wave, NWV, lane, q_u1t);
      if (wave < 8) conv_b3_part_request<C_u1same, 1>(a.af3[5], wave & 1, lane, q_u1s);
      lds_barrier();
      CORE_WIN_STAMP(stamp)
      ++stamp;
    } else {  // up1.convT (fp32 MFMA, 8 m-tiles x 1 block) -> three-piece image
      const B3PhaseStore<32> st{iP.img, iP.ps, iP.c0, T2};
      (B3Store<32>{iP.img, iP.ps, iP.c0, T2, B3_U1_NC}).zero_rest(3, 3 + T2, tid, NTH);
      if (wave < C_u1T::MT) {
        float ar[C_u1T::CB * C_u1T::TAPS], br[4];
        load_areg4<C_u1T>(a.af4[9], wave, lane, ar);
        load_biasreg<C_u1T>(a.c.bs[9], wave, lane, br);
        conv_lds_areg<C_u1T, S3_, IB, S3_, IB>(lds + X_U0S, lds + X_U0S, ar, br, wave, T3 + 1, st, 0, 1, lane);
      }
      __syncthreads();
      CORE_WIN_STAMP(stamp)
      ++stamp;
    }
    {  // up1.same: K half of up1.convT's channels, then the half of skip 2
      const F32QuadStore<S2_, IB> st{lds + X_U1S, T2};  // (U2B: skip 2's own slot -- its halo columns are zero already, and the
      if constexpr (!U3B) zero_halo<32, S2_, T2, IB>(lds + X_U1S, tid, NTH);  //  stores come after the rows have been turned into pieces)
      [[maybe_unused]] bf16_t* const iU1S = reinterpret_cast<bf16_t*>(lds) + A_SKIP2 * 2;  // U3B: the output as chunk-plane pieces
      const int mt = wave & 1, colb = (wave >> 1) * 48;
      f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (wave < 8) {
        if (U3B) conv_b3_part_requested<C_u1same, 1, 3>(iP, a.af3[5], mt, colb, lane, acc, q_u1s);
        else conv_b3_part<C_u1same, 1, 3>(iP, a.af3[5], mt, colb, lane, acc);
        conv_b3_part_request<C_u1same, 0>(a.af3[5], mt, lane, q_u1s);  // the second half's first taps: under the refill
      }
      lds_barrier();
      b3_from_f32<S2_, IB>(lds + A_SKIP2, iP, -3, B3_U1_NC - 3, tid, NTH);
      lds_barrier();
      if constexpr (U3B) {
        if (wave >= 8) b3c_zero_rest<32, B3_U1S_NC>(iU1S, 1, 193, tid - 512, NTH - 512);
      }
      if (wave < 8) {
        conv_b3_part_requested<C_u1same, 0, 3>(iP, a.af3[5], mt, colb, lane, acc, q_u1s);
        const int co0 = mt * 16 + 4 * (lane >> 4);
        float biasv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[10][co0 + r];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = C_u1same::RELU ? fmaxf(acc[j][r] + biasv[r], 0.f) : acc[j][r] + biasv[r];
          const int t = colb + j * 16 + (lane & 15);
          if constexpr (U3B) {
            if (t >= T2) v[0] = v[1] = v[2] = v[3] = 0.f;
            b3c_store4<32, B3_U1S_NC>(iU1S, t + 1, co0 >> 2, v);
          } else {
            st.quad(co0, t, v);
          }
        }
      }
      if constexpr (U3B) {
        b3_load_a<32, 2>(a.af3_uT[1], wave & 3, lane, aT2);
      }
      if (U3B) lds_barrier();
      else __syncthreads();
      CORE_WIN_STAMP(stamp)
      ++stamp;
    }
  } else {
  CORE_LAYER_AREG(9, C_u1T, X_U0S, S3_, X_U1T, S2_, TB, RangeStoreV, 32, T3 + 1, T2, wave, 0, 1)        // 8 m-tiles x 1 block (pn_core_kernel)
  CORE_LAYER(10, C_u1same, A_SKIP2, S2_, X_U1T, S2_, TB, A_U1S, S2_, IB, RangeStoreS, 32, T2, T2)
  }
  [[maybe_unused]] bf16_t* const P2 = reinterpret_cast<bf16_t*>(lds) + U2_OFF * 2;
  if constexpr (U3B) {  // up2.convT on the bf16 matrix cores: wave = (phase m-tile, block of three n-tiles), samples 4 c + phase - 1
    const bf16_t* iU1S = reinterpret_cast<const bf16_t*>(lds) + A_SKIP2 * 2;
    b3c_zero_rest<16, U2_NC>(P2, 3, 3 + T1, tid, NTH);
    const int mt = wave & 3, colb = (wave >> 2) * 48, g = lane >> 4, n = lane & 15;
    auto& aT = aT2;
    if (!(B3 && U1B)) b3_load_a<32, 2>(a.af3_uT[1], mt, lane, aT);
    float biasv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[11][4 * g + r];
    b3c_mac_tiles<32, B3_U1S_NC, 2, 3>(b3c_lane_ptr<32, B3_U1S_NC, 2>(iU1S, colb, lane), aT, [&](const int j, const f32x4 acc) {
      const int t = 4 * (colb + j * 16 + n) + mt - 1;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = C_u2T::RELU ? fmaxf(acc[r] + biasv[r], 0.f) : acc[r] + biasv[r];
      if ((unsigned)t < (unsigned)T1) b3c_store4<16, U2_NC>(P2, t + 3, g, v);
    });
    if constexpr (U3T) {
      b3_load_a<16, 7>(a.af3_u2[1], 0, lane, aw_u2);
    }
    if (U3T) lds_barrier();
    else __syncthreads();
    CORE_WIN_STAMP(stamp)
    ++stamp;
  } else if constexpr (U2B) {  // up2.convT (fp32 MFMA, 4 m-tiles x 4 blocks) -> chunk-plane piece image
    const B3PhaseStoreC<16, U2_NC> st{P2, 3, T1};
    b3c_zero_rest<16, U2_NC>(P2, 3, 3 + T1, tid, NTH);
    {
      float ar[C_u2T::CB * C_u2T::TAPS], br[4];
      load_areg4<C_u2T>(a.af4[11], wave & 3, lane, ar);
      load_biasreg<C_u2T>(a.c.bs[11], wave & 3, lane, br);
      conv_lds_areg<C_u2T, S2_, IB, S2_, IB>(lds + X_U1S, lds + X_U1S, ar, br, wave & 3, T2 + 1, st, wave >> 2, 4, lane);
    }
    __syncthreads();
    CORE_WIN_STAMP(stamp)
    ++stamp;
  } else {
  CORE_LAYER_AREG(11, C_u2T, X_U1S, S2_, A_U2T, S1_, TB, RangeStoreV, 16, T2 + 1, T1, wave & 3, wave >> 2, 4)  // 4 m-tiles x 4 blocks
  }
#undef CORE_LAYER_AREG
  // up2.same has eight items: waves 0-7 run it, waves 8-15 meanwhile fetch the eight skip rows of the up phase into
  // registers (their LDS destination is still in use by this layer) and park them in LDS right after the barrier —
  // the read-back of the skip tensor costs the up phase nothing (it was 8 k cycles of exposed memory latency).
  constexpr int NSKQ = (8 * W0_Q + 511) / 512;
  // U3T: the operands of the up path (the consumer waves' 48 registers of up3.same, the producer waves' 12 of up3.convT, the
  // 1 x 1 head, the first skip quads) are requested under up2.same, behind every wave's last MFMA of it
  [[maybe_unused]] uint4 u3_aw[B3Steps<16, 8>::STEPS * 3];
  [[maybe_unused]] f32x4 u3_bv;
  [[maybe_unused]] float u3_w1[3][4], u3_b1[3], u3_sk[4] = {0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] const int u3_pl = tid - 512, u3_skq = u3_pl & 1, u3_sks = u3_pl >> 1;  // producer lane: skip channel quad, sample within the tile
  [[maybe_unused]] const float* const u3_src = a.skip0 + (long)win * a.ws_s + HALO + (long)(4 * u3_skq) * a.ls_s;
  [[maybe_unused]] auto u3_fetch_skip = [&](const int j, const bool edge) {  // samples 256 j - 2 + u3_sks of channels 4 u3_skq ..
    const int ts = U3T_TS * j - 2 + u3_sks;
#pragma unroll
    for (int r = 0; r < 4; ++r) u3_sk[r] = (!edge || (unsigned)ts < (unsigned)T0) ? u3_src[(long)r * a.ls_s + ts] : 0.f;
  };
  [[maybe_unused]] auto u3_load_operands = [&]() {
    const int q = (lane >> 4) & 1;
    if (wave >= 8) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) u3_aw[pc] = a.af3_u3t[(long)(wave & 1) * (3 * 64) + pc * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) u3_bv[r] = a.bs_u3t[4 * q + r];
      u3_fetch_skip(0, true);
    } else {
      b3_load_a<16, 8>(a.af3_u3s, 0, lane, u3_aw);
#pragma unroll
      for (int r = 0; r < 4; ++r) u3_bv[r] = a.bs_u3s[4 * q + r];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        u3_b1[c] = a.b_out[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) u3_w1[c][r] = a.w_out[c * 8 + 4 * q + r];
      }
    }
  };
  {
    RangeStoreS<S1_, IB> st{{lds + XU_U, T1}};
    if constexpr (!U2B) zero_halo<16, S1_, T1, IB>(lds + XU_U, tid, NTH);
    if constexpr (U3T) {
      // U3T: nobody fetches skip rows here, so ALL sixteen waves share up2.same: wave w takes n-tiles 3 w .. 3 w + 2 (the matrix
      // time per SIMD is the same; four waves per SIMD instead of two hide each other's fragment reads and epilogues)
      bf16_t* const UP = reinterpret_cast<bf16_t*>(lds);
      const bf16_t* bp = b3c_lane_ptr<16, U2_NC, 7>(P2, wave * 48, lane);
      f32x4 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto& aw = aw_u2;
      if (!U3B) b3_load_a<16, 7>(a.af3_u2[1], 0, lane, aw);
      b3c_mac_tiles_acc<16, U2_NC, 7, 3>(bp, aw, acc);
      b3_load_a<16, 7>(a.af3_u2[0], 0, lane, aw);  // on its way under the refill
      __syncthreads();
      b3c_from_f32<16, U2_NC, S1_, IB>(lds + A_SKIP1, P2, 3, tid, NTH);
      __syncthreads();  // skip 1 rests in the image: its fp32 rows give way to up2.same's output (pieces, U3T_QU)
      if (tid < 3 * 2) *reinterpret_cast<uint4*>(UP + (tid >> 1) * U3T_QU::PS + (tid & 1) * U3T_QU::CHS) = make_uint4(0u, 0u, 0u, 0u);  // column 0 = sample -1
      b3c_mac_tiles_acc<16, U2_NC, 7, 3>(bp, aw, acc);
      u3_load_operands();
      {
        const int co0 = 4 * (lane >> 4);
        float biasv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[12][co0 + r];
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // pieces, sample t at column t + 1 (zeros behind the signal: columns 752 .. 767)
          const int t = wave * 48 + j * 16 + (lane & 15);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = t < T1 ? fmaxf(acc[j][r] + biasv[r], 0.f) : 0.f;
          if (t + 1 < U3T_NCU) b3c_store4<16, U3T_NCU>(UP, t + 1, co0 >> 2, v);
        }
      }
      __syncthreads();
    } else if (wave >= 8) {
      float4 skq[NSKQ];
      const float* src = a.skip0 + (long)win * a.ws_s;
#pragma unroll
      for (int k = 0; k < NSKQ; ++k) {
        const int i = tid - 512 + k * 512, c = i / W0_Q, q = i - c * W0_Q;
        const int p = 4 * q + HALO - 4;
        skq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 8 * W0_Q && p + 3 < a.ls_s) skq[k] = *reinterpret_cast<const float4*>(src + (long)c * a.ls_s + p);
      }
      WIN_WARM_SCALAR(a.w_up, 16 * 7 * 8)
      if constexpr (U2B) {
        __syncthreads();  // waves 0-7 are through with up2.convT's pieces
        b3c_from_f32<16, U2_NC, S1_, IB>(lds + A_SKIP1, P2, 3, tid, NTH);
        __syncthreads();  // skip 1 rests in the image: its fp32 rows give way to up2.same's output
        zero_halo<16, S1_, T1, IB>(lds + XU_U, tid - 512, NTH - 512);
      }
      __syncthreads();  // up2.same done: its inputs give way to the level-0 rows (0-3 -> G0, 4-7 -> G1)
#pragma unroll
      for (int k = 0; k < NSKQ; ++k) {
        const int i = tid - 512 + k * 512, c = i / W0_Q, q = i - c * W0_Q;
        if (i < 8 * W0_Q)
          *reinterpret_cast<float4*>(((c < 4) ? lds + XU_G0 + c * W0_S : lds + XU_G1 + (c - 4) * W0_S) + 4 * q) = skq[k];
      }
    } else if constexpr (U2B) {
      // wave w: n-tiles 6 w .. 6 w + 5 (48 for the 47 that hold samples), K = 2 halves x 4 steps of two taps x 16 channels
      const bf16_t* bp = b3c_lane_ptr<16, U2_NC, 7>(P2, wave * 96, lane);
      f32x4 acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      uint4 aw[B3Steps<16, 7>::STEPS * 3];
      b3_load_a<16, 7>(a.af3_u2[1], 0, lane, aw);
      b3c_mac_tiles_acc<16, U2_NC, 7, 6>(bp, aw, acc);
      b3_load_a<16, 7>(a.af3_u2[0], 0, lane, aw);  // on its way under the refill
      __syncthreads();
      b3c_from_f32<16, U2_NC, S1_, IB>(lds + A_SKIP1, P2, 3, tid, NTH);
      __syncthreads();
      b3c_mac_tiles_acc<16, U2_NC, 7, 6>(bp, aw, acc);
      {
        const int co0 = 4 * (lane >> 4);
        float biasv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) biasv[r] = a.c.bs[12][co0 + r];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int t = wave * 96 + j * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) st(co0 + r, t, C_u2same::RELU ? fmaxf(acc[j][r] + biasv[r], 0.f) : acc[j][r] + biasv[r]);
        }
      }
      __syncthreads();
    } else {
      conv_lds<C_u2same, S1_, IB, S1_, TB, PIPE, (C_u2same::NB < BDB_MAX_NB), ADEEP_LAYER(C_u2same)>(lds + A_SKIP1, lds + A_U2T, a.c.af[12], a.c.bs[12], T1, st, wave, NWV, lane);
      __syncthreads();
    }
    CORE_WIN_STAMP(stamp)
    ++stamp;
  }
#undef CORE_LAYER

  // ================= level-0 up path: up3.convT -> cat(skip0, .) -> up3.same -> 1x1 -> softmax =================
  if constexpr (U3T) {
    bf16_t* const l16 = reinterpret_cast<bf16_t*>(lds);
    const bf16_t* const UP = l16;
    bf16_t* const RU = l16 + U3T_RING_OFF;
    const int g = lane >> 4, n = lane & 15, ph = g >> 1, quad = g & 1;
    auto ring_at = [](const int c) { return (c & 1) * U3T_PL + (c >> 1) * 8; };
    // a producer's store of four channels of one column, all three pieces (+ the mirror entry behind the plane for columns 0 .. 7)
    auto ring_store = [&](bf16_t* const chunk, const int col, const int q, const float (&v)[4]) {
      const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
      const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
      const unsigned m0 = pack_bf16x2(r0, r1), m1 = pack_bf16x2(r2, r3);
      const unsigned l0 = pack_bf16x2(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), l1 = pack_bf16x2(r2 - bf16_lo(m1), r3 - bf16_hi(m1));
      bf16_t* const p = chunk + ring_at(col) + 4 * q;
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + U3T_PS) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(p + 2 * U3T_PS) = make_uint2(l0, l1);
      if (col < 2 * U3T_MIR) {
        *reinterpret_cast<uint2*>(p + U3T_PLN * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + U3T_PLN * 8 + U3T_PS) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(p + U3T_PLN * 8 + 2 * U3T_PS) = make_uint2(l0, l1);
      }
    };
    WIN_STAMP(23)
    const bool producer = wave >= 8;  // (uniform)
    const int wv = wave & 7;
    auto& aw = u3_aw;  // requested under up2.same (above)
    const f32x4 bv = u3_bv;
    auto& w1 = u3_w1;
    auto& b1 = u3_b1;
    auto& sk = u3_sk;
    const int sk_q = u3_skq, sk_s = u3_sks;
    if (tid < 96) {  // ring columns 512 .. 527 <-> samples -16 .. -1 of both chunks: zeros
      const int cp = tid >> 4, c = U3T_RING - 16 + (tid & 15);  // cp = piece * 2 + chunk
      *reinterpret_cast<uint4*>(RU + (cp >> 1) * U3T_PS + (cp & 1) * U3T_CH + ring_at(c)) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    WIN_STAMP(24)
#define U3T_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, W), __builtin_bit_cast(bf16x8_b3, X), ACC, 0, 0, 0)
    // producer: transposed conv item (m-tile wv & 1 = phases 2 (wv & 1) + ph, n-tile wv >> 1 of the tile's 64 level-1 samples)
    const int mphase = 2 * (wv & 1) + ph;
    int ct = mphase - 2 + 4 * (16 * (wv >> 1) + n);   // sample of this lane's output in tile 0 (tile j: + 256 j), >= -2
    ct = ct < 0 ? ct + U3T_RING : ct;                 // its ring column
    int cs = sk_s - 2;                                // skip sample of tile 0
    cs = cs < 0 ? cs + U3T_RING : cs;
    const bf16_t* up = UP + quad * U3T_QU::CHS + (16 * (wv >> 1) + n + ph) * 8;  // U column m + tap (sample m + tap - 1), tap = ph, chunk quad
    // consumer: n-tile wv of tile j - 1: samples 256 (j - 1) - 8 + 32 wv + 2 n + ph read the ring's samples .. - 3 + tap, tap = 2 step + ph, chunk quad
    int cc = U3T_RING - 11 + 32 * wv + 2 * n + ph;
    cc = cc >= U3T_RING ? cc - U3T_RING : cc;
    float* const yrow = a.y + (long)win * 3 * T0;
#pragma unroll
    for (int j = 0; j <= U3T_TILES; ++j) {
      if (producer) {
        if (j < U3T_TILES) {
          uint4 b[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) b[pc] = *reinterpret_cast<const uint4*>(up + pc * U3T_QU::PS);
          f32x4 acc = bv;
          U3T_MFMA(acc, aw[2], b[0]);
          U3T_MFMA(acc, aw[1], b[1]);
          U3T_MFMA(acc, aw[0], b[2]);
          U3T_MFMA(acc, aw[1], b[0]);
          U3T_MFMA(acc, aw[0], b[1]);
          U3T_MFMA(acc, aw[0], b[0]);
          // the skip quad fetched a phase ago -> pieces, chunk 0
          ring_store(RU, cs, sk_q, sk);
          if (j + 1 < U3T_TILES) u3_fetch_skip(j + 1, U3T_TS * (j + 2) > T0);
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[r], 0.f);
          if (j == 0 || U3T_TS * (j + 1) > T0) {  // (uniform) the tiles that meet the ends of the signal
            const int s = U3T_TS * j + mphase - 2 + 4 * (16 * (wv >> 1) + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (unsigned)s < (unsigned)T0 ? o[r] : 0.f;
          }
          ring_store(RU + U3T_CH, ct, quad, o);
          ct += U3T_TS, cs += U3T_TS;
          ct = ct >= U3T_RING ? ct - U3T_RING : ct;
          cs = cs >= U3T_RING ? cs - U3T_RING : cs;
          up += (U3T_TS / 4) * 8;
        }
      } else if (j > 0) {
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = bv;
        const bf16_t* const rp0 = RU + quad * U3T_CH + ring_at(cc);  // K-step st: two columns = one plane entry further (mirrored: no wrap)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          uint4 b[3];
          const bf16_t* rp = rp0 + st * 8;
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) b[pc] = *reinterpret_cast<const uint4*>(rp + pc * U3T_PS);
          U3T_MFMA(sa, aw[st * 3 + 2], b[0]);
          U3T_MFMA(sb, aw[st * 3 + 1], b[0]);
          U3T_MFMA(sa, aw[st * 3 + 1], b[1]);
          U3T_MFMA(sb, aw[st * 3 + 0], b[1]);
          U3T_MFMA(sa, aw[st * 3 + 0], b[2]);
          U3T_MFMA(sb, aw[st * 3 + 0], b[0]);
        }
        // BN + ReLU -> Conv1d(8, 3, 1): this lane's four channels, the other four from the lane 16 further (the other channel quad)
        float z[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float zz = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) zz = fmaf(w1[k][r], fmaxf(sa[r] + sb[r], 0.f), zz);
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(zz), __float_as_uint(zz), false, false);
          z[k] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + b1[k];  // rows (0, 1) and (2, 3): the pair's sum in both
        }
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
        const float inv = __builtin_amdgcn_rcpf(e0 + e1 + e2);  // (1 ulp; the IEEE division is ten instructions on this issue-bound path)
        float y0 = e0 * inv, y1 = e1 * inv, y2 = e2 * inv;
        if (poisoned) y0 = y1 = y2 = __builtin_nanf("");
        const int t = U3T_TS * (j - 1) - 8 + 32 * wv + 2 * n + ph;
        if (quad == 0 && (unsigned)t < (unsigned)T0) yrow[t] = y0, yrow[T0 + t] = y1, yrow[2 * T0 + t] = y2;
        cc += U3T_TS;
        cc = cc >= U3T_RING ? cc - U3T_RING : cc;
      }
      lds_barrier();
#ifdef U3T_PROBE
      WIN_STAMP(2 + j)
#endif
    }
#undef U3T_MFMA
    WIN_STAMP(28)
  } else {
    float *G0 = lds + XU_G0, *G1 = lds + XU_G1, *U = lds + XU_U;
    WIN_STAMP(23)
    __syncthreads();
    WIN_STAMP(24)
    float aT[W_upT::CB * W_upT::TAPS], bT[4];  // A fragments of up3.convT (waves alternate over its two m-tiles)
    load_areg<W_upT>(a.af_t, wave & 1, lane, aT);
    load_biasreg<W_upT>(a.bs_t, wave & 1, lane, bT);
    f32x2 acc[4][4];
    if (vconv) {  // up3.same on cat([skip0, up3.convT]): the skip half first, then the convT rows take the skip rows' place
      valu_bias(acc, a.b_up);
      valu_conv7_r4<4, W0_S>(G0, as_weights(a.w_up), t0, acc);
      valu_conv7_r4<4, W0_S>(G1, as_weights(a.w_up + 4 * 28), t0, acc);
    }
    __syncthreads();
    WIN_STAMP(25)
    {  // up3.convT: ConvTranspose1d(16, 8, 7, stride 4) + BN + ReLU, crop [1:-2] and centre crop (t = o - 2), on the MFMA
      SplitRowStore st{G0, G1};
      conv_lds_areg<W_upT, S1_, IB, S1_, IB>(U, U, aT, bT, wave & 1, T1 + 1, st, wave >> 1, NWV / 2, lane);
    }
    __syncthreads();
    WIN_STAMP(26)
    if (vconv) {
      valu_conv7_r4<4, W0_S>(G0, as_weights(a.w_up + 8 * 28), t0, acc);
      valu_conv7_r4<4, W0_S>(G1, as_weights(a.w_up + 12 * 28), t0, acc);
    }
    WIN_STAMP(27)
    if (own) {  // BN + ReLU -> Conv1d(8, 3, 1) -> softmax over channels
      float z[3][4];
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[o][r] = as_scalars(a.b_out)[o];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = fmaxf(acc[c][r].x, 0.f), v1 = fmaxf(acc[c][r].y, 0.f);
#pragma unroll
          for (int o = 0; o < 3; ++o)
            z[o][r] = fmaf(as_scalars(a.w_out)[o * 8 + 2 * c + 1], v1, fmaf(as_scalars(a.w_out)[o * 8 + 2 * c], v0, z[o][r]));
        }
      f32x4 y0, y1, y2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mx = fmaxf(z[0][r], fmaxf(z[1][r], z[2][r]));
        const float e0 = __expf(z[0][r] - mx), e1 = __expf(z[1][r] - mx), e2 = __expf(z[2][r] - mx);
        const float inv = 1.f / (e0 + e1 + e2);
        y0[r] = e0 * inv, y1[r] = e1 * inv, y2[r] = e2 * inv;
        if (poisoned) y0[r] = y1[r] = y2[r] = __builtin_nanf("");
      }
      float* y = a.y + (long)win * 3 * T0 + t0;
      if (t0 + 3 < T0) {  // dense rows of odd length: 4-byte aligned vector stores
        *reinterpret_cast<f32x4u*>(y) = y0;
        *reinterpret_cast<f32x4u*>(y + T0) = y1;
        *reinterpret_cast<f32x4u*>(y + 2 * T0) = y2;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t0 + r < T0) y[r] = y0[r], y[T0 + r] = y1[r], y[2 * T0 + r] = y2[r];
      }
    }
    WIN_STAMP(28)
  }
  if (clk && tid == 0) clk[(long)win * 32 + 17] = wall_clock64();
#undef WIN_STAMP
#undef CORE_WIN_STAMP
#undef WIN_WARM_SCALAR
}

int tensor_id(const Net& net, const std::string& name) {
  for (size_t i = 0; i < net.tensors.size(); ++i)
    if (net.tensors[i].name == name) return (int)i;
  return -1;
}

}  // namespace

// Replaces the 18 layer steps planned by plan_phasenet with the three fused launches.  The
// packed weights of the layer plan are reused as they are (same P / taps / channel padding).
namespace {
// [cout = 8][cin][7] (Conv1d) or [cin][cout = 8][7] (ConvTranspose1d) -> [cin][7][8] with the BatchNorm scale folded in:
// channel pairs (2c, 2c + 1) are adjacent, one s_load_dwordx8 fetches a (channel, tap) for all outputs
std::vector<float> pack_valu(const float* W, int cin, bool transposed, const std::vector<float>& scale) {
  std::vector<float> out((size_t)cin * 7 * 8);
  for (int ci = 0; ci < cin; ++ci)
    for (int k = 0; k < 7; ++k)
      for (int co = 0; co < 8; ++co)
        out[((size_t)ci * 7 + k) * 8 + co] = scale[co] * (transposed ? W[((size_t)ci * 8 + co) * 7 + k] : W[((size_t)co * cin + ci) * 7 + k]);
  return out;
}
}  // namespace

int plan_phasenet_fused(Net& net, const ParamView& pv, int debug_flags) {
  const bool debug_dumps = (debug_flags & 1) != 0, debug_clock = (debug_flags & 2) != 0;
  // Pruned in round 6 (kept in source as `if constexpr` branches of the templates, no longer instantiated): the hand-pipelined K
  // loop (plan_flags[2] = 1; the compiler's own schedule measured 0-5 % faster on every layer, tools/micro/micro_layers.hip) and
  // the intermediate forms of pn_window_kernel between its references (plan_flags[5] = 4, 5, 6, 7, 9).
  {
    const int f5 = net.cfg.plan_flags[5];
    if (net.cfg.plan_flags[2] == 1 || f5 == 4 || f5 == 5 || f5 == 6 || f5 == 7 || f5 == 9) {
      set_error("PhaseNet plan_flags[2] = %d / plan_flags[5] = %d: this A/B form was removed in round 6 (kept: plan_flags[5] = 0, 1, 2, 3, 8)",
                net.cfg.plan_flags[2], f5);
      return VP_ERR_UNSUPPORTED;
    }
  }
  const bool persistent = net.cfg.plan_flags[3] != 1;  // plan_flags[3] = 1: one workgroup per tile for up3 too (A/B timing)
  if (net.cfg.plan_flags[3] == 2) {
    set_error("PhaseNet plan_flags[3] = 2 (persistent level-0 down kernel): removed in round 6");
    return VP_ERR_UNSUPPORTED;
  }
  // plan_flags[5] = 1 keeps the MFMA forms of the two level-0 kernels (A/B timing; bit-identical to the layer plan),
  // 2 the three-launch plan with the VALU level-0 kernels; default: the whole network in one launch (pn_window_kernel).
  // The debug dumps of the intermediates exist in the three-launch plans only.
  const bool valu = net.cfg.plan_flags[5] != 1;
  const bool whole = valu && net.cfg.plan_flags[5] != 2 && !debug_dumps;
  const bool b3 = whole && net.cfg.plan_flags[5] != 3;  // plan_flags[5] = 3: the one-launch kernel with all core layers on the fp32 MFMA
  const bool u1b = b3 && net.cfg.plan_flags[5] != 4;    // plan_flags[5] = 4: up1.same and up2.same stay on the fp32 MFMA (the form of round 2)
  const bool u2b = u1b && net.cfg.plan_flags[5] != 5;   // plan_flags[5] = 5: only up2.same does
  const bool u3b = u2b && net.cfg.plan_flags[5] != 6;   // plan_flags[5] = 6: up1.convT / up2.convT stay on the fp32 MFMA
  const bool d12b = u3b && net.cfg.plan_flags[5] != 7;  // plan_flags[5] = 7: down1.same / down2.same stay on the fp32 MFMA
  const bool d0t = d12b && net.cfg.plan_flags[5] != 8;  // plan_flags[5] = 8: inc / down0.same stay on the vector ALUs
  const bool up3t = d0t && net.cfg.plan_flags[5] != 9;   // plan_flags[5] = 9: up3.convT on the fp32 MFMA, up3.same on the vector ALUs
  HostBlob *vw[5] = {}, *vb[5] = {};
  if (valu) {
    const float eps = net.cfg.bn_eps;
    struct {
      const char *conv, *bn;
      int cin;
      bool transposed, bias;
    } spec[5] = {{"inc", "in_bn", 3, false, true},
                 {"down_branch.0.0", "down_branch.0.1", 8, false, false},
                 {"down_branch.0.2", "down_branch.0.3", 8, false, false},
                 {"up_branch.3.0", "up_branch.3.1", 16, true, false},
                 {"up_branch.3.2", "up_branch.3.3", 16, false, false}};
    for (int i = 0; i < 5; ++i) {
      std::vector<float> scale, shift;
      bn_fold(pv, spec[i].bn, 8, eps, spec[i].bias ? pv.get(std::string(spec[i].conv) + ".bias") : nullptr, &scale, &shift);
      vw[i] = net.add_blob(pack_valu(pv.get(std::string(spec[i].conv) + ".weight"), spec[i].cin, spec[i].transposed, scale));
      vb[i] = net.add_blob(shift);
    }
  }
  if (net.convs.size() != 18) {
    set_error("fused PhaseNet plan expects the 18-layer plan");
    return VP_ERR_INVALID;
  }
  const int n_tiles = (T0 + TT - 1) / TT;  // 6
  const int x = net.input, h0 = tensor_id(net, "inc"), skip0 = tensor_id(net, "down0.same");
  const int d0 = tensor_id(net, "down0.down"), u2s = tensor_id(net, "up2.same"), u3t = tensor_id(net, "up3.convT");
  // furthest reads of the tiled loaders
  net.need(x, (n_tiles - 1) * TT - 8 + 4 * ((TT + 28) / 4));
  net.need(skip0, HALO + (n_tiles - 1) * TT - 16 + 12 + 4 * 130);
  net.need(u2s, HALO + (n_tiles - 1) * TT / 4 - 4 + 144);
  if (valu) {
    net.need(x, HALO + VD_TS * (VD_TILES - 1) - 12 + VS);      // x image float4 loads
    net.need(skip0, VU_TS * (VU_TILES - 1) + VS);              // skip image float4 loads
    net.need(skip0, HALO + VD_TS * VD_TILES);                  // skip stores of the last down tile
    net.need(u2s, HALO + (VU_TS / 4) * (VU_TILES - 1) - 2 + 258);  // up2.same image loads
    net.need(x, HALO - 4 + W0_S);      // whole-window float4 loads of pn_window_kernel
    net.need(skip0, HALO - 4 + W0_S);
  }
  std::vector<Step> steps;
  auto flops = [&](int lo, int hi) {
    double f = 0;
    for (int i = lo; i <= hi; ++i) f += net.convs[i]->flops_per_window;
    return f;
  };
  {
    Step st;
    st.name = "fused.down0 (inc+down0.same+down0.down)";
    st.flops_per_window = flops(0, 2);
    st.run = [=](Net& n, int B, hipStream_t s) -> int {
      Down0Args a{};
      const Tensor &tx = n.tensors[x], &ts = n.tensors[skip0], &td = n.tensors[d0], &th = n.tensors[h0];
      a.x = tx.p;
      a.ls_x = tx.ls;
      a.ws_x = (long)tx.win_stride();
      a.skip0 = ts.p;
      a.ls_s = ts.ls;
      a.ws_s = (long)ts.win_stride();
      a.d0 = td.p;
      a.ls_d = td.ls;
      a.ws_d = (long)td.win_stride();
      if (debug_dumps) {
        a.h0_dbg = th.p;
        a.ls_h = th.ls;
        a.ws_h = (long)th.win_stride();
      }
      a.af_inc = n.convs[0]->afrag.d;
      a.bs_inc = n.convs[0]->bias.d;
      a.af_same = n.convs[1]->afrag.d;
      a.bs_same = n.convs[1]->bias.d;
      a.af_down = n.convs[2]->afrag.d;
      a.bs_down = n.convs[2]->bias.d;
      // measured (tools/ab_steps.py, one process): the persistent form is 4-15 % SLOWER here (its 46
      // A registers cost occupancy and one-tile workgroups already stagger well); plan_flags[3] = 2 selects it
      if (valu) {
        Down0VArgs v{};
        v.t = a;
        v.w_inc = reinterpret_cast<const f32x2*>(vw[0]->d);
        v.b_inc = reinterpret_cast<const f32x2*>(vb[0]->d);
        v.w_same = reinterpret_cast<const f32x2*>(vw[1]->d);
        v.b_same = reinterpret_cast<const f32x2*>(vb[1]->d);
        v.w_down = reinterpret_cast<const f32x2*>(vw[2]->d);
        v.b_down = reinterpret_cast<const f32x2*>(vb[2]->d);
        v.n_windows = B;
        hipLaunchKernelGGL(pn_down0v_kernel, dim3(VD_TILES * B), dim3(256), VD_LDS_FLOATS * sizeof(float), s, v);
      } else {
        hipLaunchKernelGGL(pn_down0_kernel<false>, dim3(n_tiles, B), dim3(256), D0_LDS_FLOATS * sizeof(float), s, a);
      }
      return 0;
    };
    steps.push_back(std::move(st));
  }
  {
    Step st;
    st.name = "fused.core (down1..down4, up0..up2)";
    st.flops_per_window = flops(3, 15);
    static const char* dbg_names[12] = {"down1.same", "down1.down", "down2.same", "down2.down", "down3.same", "down3.down",
                                        "down4.same", "up0.convT",  "up0.same",   "up1.convT",  "up1.same",   "up2.convT"};
    std::vector<int> dbg_ids(12);
    for (int i = 0; i < 12; ++i) dbg_ids[i] = tensor_id(net, dbg_names[i]);
    HostBlob* clk = debug_clock ? net.add_blob(std::vector<float>(((size_t)net.max_batch * 32 + 64 * 8) * 2, 0.f)) : nullptr;
    net.debug_clock = clk;
    st.run = [=](Net& n, int B, hipStream_t s) -> int {
      CoreArgs a{};
      const Tensor &td = n.tensors[d0], &tu = n.tensors[u2s];
      a.d0 = td.p;
      a.ls_d0 = td.ls;
      a.ws_d0 = (long)td.win_stride();
      a.u2s = tu.p;
      a.ls_u2s = tu.ls;
      a.ws_u2s = (long)tu.win_stride();
      for (int i = 0; i < 13; ++i) {
        a.af[i] = n.convs[3 + i]->afrag.d;
        a.bs[i] = n.convs[3 + i]->bias.d;
      }
      for (int i = 0; i < 12; ++i) {
        if (debug_dumps && dbg_ids[i] >= 0) {
          const Tensor& t = n.tensors[dbg_ids[i]];
          a.dbg[i] = t.p;
          a.dbg_ls[i] = t.ls;
          a.dbg_ws[i] = (long)t.win_stride();
        }
      }
      a.clk = clk ? reinterpret_cast<unsigned long long*>(clk->d) : nullptr;
      a.warm = n.cfg.plan_flags[4] != 1;
      hipLaunchKernelGGL(pn_core_kernel<false>, dim3(B), dim3(1024), CORE_LDS_FLOATS * sizeof(float), s, a);
      return 0;
    };
    steps.push_back(std::move(st));
  }
  {
    Step st;
    st.name = "fused.up3 (up3.convT+up3.same+out+softmax)";
    st.flops_per_window = flops(16, 17);
    HostBlob* e0 = &net.convs[17]->e0;
    HostBlob* e1 = &net.convs[17]->e1;
    st.run = [=](Net& n, int B, hipStream_t s) -> int {
      Up3Args a{};
      const Tensor &tu = n.tensors[u2s], &ts = n.tensors[skip0], &tt = n.tensors[u3t];
      a.u2s = tu.p;
      a.ls_u = tu.ls;
      a.ws_u = (long)tu.win_stride();
      a.skip0 = ts.p;
      a.ls_s = ts.ls;
      a.ws_s = (long)ts.win_stride();
      a.y = n.y;
      if (debug_dumps) {
        a.ut_dbg = tt.p;
        a.ls_t = tt.ls;
        a.ws_t = (long)tt.win_stride();
      }
      a.af_t = n.convs[16]->afrag.d;
      a.bs_t = n.convs[16]->bias.d;
      a.af_same = n.convs[17]->afrag.d;
      a.bs_same = n.convs[17]->bias.d;
      a.w_out = e0->d;
      a.b_out = e1->d;
      a.clk = n.debug_clock ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) : nullptr;
      if (valu) {
        Up3VArgs v{};
        v.t = a;
        v.w_t = reinterpret_cast<const f32x2*>(vw[3]->d);
        v.b_t = reinterpret_cast<const f32x2*>(vb[3]->d);
        v.w_same = reinterpret_cast<const f32x2*>(vw[4]->d);
        v.b_same = reinterpret_cast<const f32x2*>(vb[4]->d);
        v.n_windows = B;
        hipLaunchKernelGGL(pn_up3v_kernel, dim3(VU_TILES * B), dim3(256), VU_LDS_FLOATS * sizeof(float), s, v);
      } else if (persistent && !debug_dumps) {
        hipLaunchKernelGGL(pn_up3p_kernel, dim3(NSPLIT_U, B), dim3(256), UP3_LDS_FLOATS * sizeof(float), s, a);
      } else {
        hipLaunchKernelGGL(pn_up3_kernel<false>, dim3(n_tiles, B), dim3(256), UP3_LDS_FLOATS * sizeof(float), s, a);
      }
      return 0;
    };
    steps.push_back(std::move(st));
  }
  if (whole) {
    Step st;
    st.name = "fused.window (whole PhaseNet, one workgroup per window)";
    st.flops_per_window = flops(0, 17);
    {  // issued: inc, down0.same, both halves of up3.same and the 1x1 head on the vector ALUs (direct convolutions: no padding);
       // every other layer as whole 16-column tiles of M x (padded channels x taps) on the matrix cores -- the five deepest
       // (convs 7 .. 11) as six-MFMA groups over bf16 pieces when b3
      auto padded = [&](int i) {
        const ConvLayer& L = *net.convs[i];
        return 2.0 * L.g.M() * ((L.cols + 15) / 16 * 16) * L.g.cinp() * L.g.taps;
      };
      double f32 = 0, bf16 = 0;
      for (int i = 2; i <= 16; ++i) {
        if (up3t && i == 16) bf16 += (double)U3T_TILES * 8 * 6 * 16384.0;  // up3.convT: 8 (m-tile, n-tile) items per tile x one K-step
        else if (b3 && i >= 7 && i <= 11) bf16 += 6.0 * padded(i);
        else if (u1b && i == 13) bf16 += 6.0 * 2.0 * 32 * 192 * 64 * 7;  // up1.same: 2 m-tiles x 12 n-tiles x 14 K-steps
        else if (u2b && i == 15) bf16 += 6.0 * 2.0 * 16 * 768 * 32 * 8;  // up2.same: 48 n-tiles x 2 halves x 4 K-steps
        else if (d12b && i == 3) bf16 += 48.0 * 2 * 6 * 16384.0;          // down1.same: 48 n-tiles x 2 K-steps
        else if (d12b && i == 5) bf16 += 2.0 * 12 * 4 * 6 * 16384.0;      // down2.same: 2 m-tiles x 12 n-tiles x 4 K-steps
        else if (u3b && i == 12) bf16 += 8.0 * 3 * 4 * 6 * 16384.0;       // up1.convT: 8 m-tiles x 3 n-tiles x 4 K-steps
        else if (u3b && i == 14) bf16 += 4.0 * 12 * 2 * 6 * 16384.0;      // up2.convT: 4 m-tiles x 12 n-tiles x 2 K-steps
        else f32 += padded(i);
      }
      if (d0t) bf16 += (double)D0T_TILES * 16 * (1 + 2) * 6 * 16384.0;  // inc: 96 n-tiles x 1 K-step; down0.same: 96 x 2; six MFMAs each
      if (up3t) bf16 += (double)U3T_TILES * 8 * 4 * 6 * 16384.0;  // up3.same: 8 n-tiles per tile x four K-steps
      // (the 1 x 1 head: 2 x 3 x 8 FLOP per sample on the vector ALUs in every form)
      st.set_issued(f32, bf16, (d0t ? 0.0 : flops(0, 1)) + (up3t ? 2.0 * 3 * 8 * T0 : flops(17, 17)));
    }
    HostBlob* e0 = &net.convs[17]->e0;
    HostBlob* e1 = &net.convs[17]->e1;
    HostBlob* clk = debug_clock ? net.debug_clock : nullptr;
    HostBlob* q4[13] = {};
    for (int i = 0; i < 13; ++i) {
      if (!q4_layer_index(i)) continue;
      std::vector<float> v = regroup_afrag4(*net.convs[3 + i]);
      q4[i] = net.add_blob(std::move(v));
    }
    HostBlob* p3[6] = {};
    if (b3)
      for (int i = 0; i < 5; ++i) p3[i] = net.add_blob(b3_operand(*net.convs[3 + 4 + i], i == 3));
    if (u1b) p3[5] = net.add_blob(b3_operand(*net.convs[3 + 10], false));
    HostBlob* p3d12[2] = {};
    if (d12b) {
      p3d12[0] = net.add_blob(b3_operand(*net.convs[3 + 0], false));
      p3d12[1] = net.add_blob(b3_operand(*net.convs[3 + 2], false));
    }
    HostBlob *p3inc = nullptr, *p3d0s = nullptr;
    if (d0t) {
      p3inc = net.add_blob(b3_operand(*net.convs[0], true));
      p3d0s = net.add_blob(b3_operand(*net.convs[1], true));
    }
    HostBlob *p3u3t = nullptr, *p3u3s = nullptr;
    if (up3t) {
      p3u3t = net.add_blob(b3_operand(*net.convs[16], true));
      p3u3s = net.add_blob(b3_operand(*net.convs[17], true));
    }
    HostBlob* p3uT[2] = {};
    if (u3b) {
      p3uT[0] = net.add_blob(b3_operand(*net.convs[3 + 9], true));
      p3uT[1] = net.add_blob(b3_operand(*net.convs[3 + 11], true));
    }
    HostBlob* p3u2[2] = {};
    if (u2b) {  // up2.same per input half as 16-channel K-steps (B3Steps<16, 7>: two taps per step, tap 7 = zero weights):
                // [step][piece][lane][8], lane = 16 g + row, tap = 2 step + g / 2, channels 16 half + 8 (g % 2) ..
      const ConvLayer& L = *net.convs[3 + 12];
      const int taps = L.g.taps, CB = L.g.cinp() / 4;
      auto rne = [](float x) -> uint16_t {
        uint32_t u;
        memcpy(&u, &x, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
      };
      auto widen = [](uint16_t h) -> float {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
      };
      for (int half = 0; half < 2; ++half) {
        std::vector<uint16_t> o((size_t)4 * 3 * 64 * 8);
        for (int st = 0; st < 4; ++st)
          for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) {
              const int m = l & 15, g = l >> 4, tap = 2 * st + g / 2, ci = 16 * half + 8 * (g % 2) + i;
              const float w = tap < taps ? L.afrag.h[(((size_t)(ci / 4)) * taps + tap) * 64 + (ci % 4) * 16 + m] : 0.f;
              (void)CB;
              const uint16_t h = rne(w);
              const float r1 = w - widen(h);
              const uint16_t md = rne(r1);
              const size_t base = (((size_t)st * 3) * 64 + l) * 8 + i;
              o[base] = h;
              o[base + 64 * 8] = md;
              o[base + 2 * 64 * 8] = rne(r1 - widen(md));
            }
        std::vector<float> f(o.size() / 2);
        memcpy(f.data(), o.data(), o.size() * 2);
        p3u2[half] = net.add_blob(std::move(f));
      }
    }
    st.run = [=](Net& n, int B, hipStream_t s) -> int {
      WindowArgs a{};
      for (int i = 0; i < 13; ++i) a.af4[i] = q4[i] ? q4[i]->d : nullptr;
      for (int i = 0; i < 2; ++i) a.af3_d12[i] = p3d12[i] ? reinterpret_cast<const uint4*>(p3d12[i]->d) : nullptr;
      for (int i = 0; i < 2; ++i) a.af3_uT[i] = p3uT[i] ? reinterpret_cast<const uint4*>(p3uT[i]->d) : nullptr;
      for (int i = 0; i < 2; ++i) a.af3_u2[i] = p3u2[i] ? reinterpret_cast<const uint4*>(p3u2[i]->d) : nullptr;
      a.af3_inc = p3inc ? reinterpret_cast<const uint4*>(p3inc->d) : nullptr;
      a.af3_d0s = p3d0s ? reinterpret_cast<const uint4*>(p3d0s->d) : nullptr;
      a.bs_inc8 = n.convs[0]->bias.d;
      a.bs_d0s = n.convs[1]->bias.d;
      a.af3_u3t = p3u3t ? reinterpret_cast<const uint4*>(p3u3t->d) : nullptr;
      a.af3_u3s = p3u3s ? reinterpret_cast<const uint4*>(p3u3s->d) : nullptr;
      a.bs_u3t = n.convs[16]->bias.d;
      a.bs_u3s = n.convs[17]->bias.d;
      for (int i = 0; i < 6; ++i) {
        a.af3[i] = p3[i] ? reinterpret_cast<const uint4*>(p3[i]->d) : nullptr;
        a.af3_lines[i] = p3[i] ? (int)(p3[i]->h.size() * 4 / 128) : 0;
      }
      const Tensor &tx = n.tensors[x], &ts = n.tensors[skip0];
      for (int i = 0; i < 13; ++i) {
        a.c.af[i] = n.convs[3 + i]->afrag.d;
        a.c.bs[i] = n.convs[3 + i]->bias.d;
      }
      a.c.clk = clk ? reinterpret_cast<unsigned long long*>(clk->d) : nullptr;
      a.c.warm = n.cfg.plan_flags[4] != 1 && n.warm_launches > 0;
      if (n.warm_launches > 0) --n.warm_launches;
      a.x = tx.p;
      a.ls_x = tx.ls;
      a.ws_x = (long)tx.win_stride();
      a.skip0 = ts.p;
      a.ls_s = ts.ls;
      a.ws_s = (long)ts.win_stride();
      a.y = n.y;
      a.w_inc = reinterpret_cast<const f32x2*>(vw[0]->d);
      a.b_inc = reinterpret_cast<const f32x2*>(vb[0]->d);
      a.w_same = reinterpret_cast<const f32x2*>(vw[1]->d);
      a.b_same = reinterpret_cast<const f32x2*>(vb[1]->d);
      a.w_up = reinterpret_cast<const f32x2*>(vw[4]->d);
      a.b_up = reinterpret_cast<const f32x2*>(vb[4]->d);
      a.af_down = n.convs[2]->afrag.d;
      a.bs_down = n.convs[2]->bias.d;
      a.af_t = n.convs[16]->afrag.d;
      a.bs_t = n.convs[16]->bias.d;
      a.w_out = e0->d;
      a.b_out = e1->d;
      if (n.pre) {
        a.pre = *n.pre;
        a.has_pre = 1;
      }
      // Three forms are kept (round 6 pruned the rest: the intermediate forms of rounds 2-5 -- plan_flags[5] = 4, 5, 6, 7, 9 and the
      // hand-pipelined K loop plan_flags[2] = 1 -- were A/B stations on the way, no test's reference any more): the default, the
      // round-4 form with level 0 on the vector ALUs (plan_flags[5] = 8: the rounding reference of the tiled level-0 layers), and
      // every core layer on the fp32 MFMA (plan_flags[5] = 3: the reference of the bf16-piece layers).
      if (up3t) {
        hipLaunchKernelGGL((pn_window_kernel<false, true, true, true, true, true, true, true>), dim3(B), dim3(1024), CORE_LDS_FLOATS * sizeof(float), s, a);
      } else if (d12b) {
        hipLaunchKernelGGL((pn_window_kernel<false, true, true, true, true, true>), dim3(B), dim3(1024), CORE_LDS_FLOATS * sizeof(float), s, a);
      } else {
        hipLaunchKernelGGL((pn_window_kernel<false, false>), dim3(B), dim3(1024), CORE_LDS_FLOATS * sizeof(float), s, a);
      }
      return 0;
    };
    steps.clear();
    steps.push_back(std::move(st));
    net.fused_pre = net.cfg.plan_flags[6] != 1;  // plan_flags[6] = 1: gather_normalize_kernel fills the input tensor as in the other plans
    net.fused_pre_poisons = true;                // ... and then writes the NaN predictions of a non-finite window itself
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_window_kernel<false, false>), CORE_LDS_FLOATS * sizeof(float)});
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_window_kernel<false, true, true, true, true, true>), CORE_LDS_FLOATS * sizeof(float)});
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_window_kernel<false, true, true, true, true, true, true, true>), CORE_LDS_FLOATS * sizeof(float)});
  }
  net.steps = std::move(steps);
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_core_kernel<false>), CORE_LDS_FLOATS * sizeof(float)});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_down0_kernel<false>), D0_LDS_FLOATS * sizeof(float)});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_up3_kernel<false>), UP3_LDS_FLOATS * sizeof(float)});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_up3p_kernel), UP3_LDS_FLOATS * sizeof(float)});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_down0v_kernel), VD_LDS_FLOATS * sizeof(float)});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&pn_up3v_kernel), VU_LDS_FLOATS * sizeof(float)});
  return VP_OK;
}

}  // namespace vp
