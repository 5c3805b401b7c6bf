// EQTransformer decoder tail as ONE launch: stages 4, 5, 6 of the three decoders (Upsample(2) + Conv1d + ReLU, folded
// into two-phase filters, eqt.hip) and the Conv1d(8,1,11) + sigmoid heads, per TIME TILE of one (decoder, window) row.
//
// Why: as three conv_mfma_kernel launches these stages wrote 16 x 1500 and 16 x 3000 samples per decoder and window
// to memory only to read them straight back (0.44 GB of the 1.16 GB a 256-window step moved, PMC), and every launch
// paid its load phase in lock step (all workgroups fetch, then all compute) plus a kernel boundary.  Here a tile's
// rows never leave the CU: the workgroup reads 32 x 135 samples of stage 3, runs the three stages LDS to LDS with the
// halo each needs recomputed and writes 1000 probabilities.
//
//   tile of outputs [t0, t0 + 1000) of the 6000-sample row, t0 = 1000 j; column counts include the halos of the
//   stages behind them (needed / computed as whole blocks of n-tiles):
//     stage 6 columns n6 = t0/2 - 3 + [0, 506 / 512)  reads stage-5 samples [n6 - 3, n6 + 3]   (K = 11 folded to 7 taps)
//     stage 5 columns n5 = t0/4 - 3 + [0, 256 / 256)  reads stage-4 samples [n5 - 2, n5 + 2]   (K = 9 folded to 5 taps)
//     stage 4 columns n4 = t0/8 - 3 + [0, 131 / 160)  reads stage-3 samples [n4 - 2, n4 + 2]   (K = 7 folded to 5 taps)
//   = 2336 MFMAs per tile against 2208 for the same outputs without halos and padding (+6 %).
//   Samples outside a row's signal are written as zeros: they are the next stage's zero padding.
//
// 256-thread workgroups (4 wavefronts, one per SIMD), 78 KB of LDS: TWO per CU, persistent; each takes two tiles by
// its index and every further one from a ticket counter (one returning atomic per tile, issued a whole tile before its
// answer is needed).  The two workgroups of a CU do not run at the same speed -- the issue arbiter prefers the older
// one -- and with a fixed share of the tiles the favoured workgroups were done after 170 us and the others after
// 206 us (tools/tail_clock.py).  A workgroup alternates phases that keep the matrix pipes busy (the three stages) with
// phases that cannot (the heads: 88 FMAs per output on the VALU; epilogues; the barriers between the phases) -- with one
// big workgroup per CU those were 35 % of the time with the MFMA pipes idle (tools/tail_clock.py); two independent
// workgroups fill each other's gaps.  Every wave keeps one m-tile per stage, so a stage's A operand is 40 / 20 / 28
// fragments per lane held in registers (requested a phase ahead) and the K loops touch nothing but LDS: per 16x16x4
// MFMA one B fragment, fetched in pairs (ds_read2_b32) one K-step ahead (conv_lds_areg).  The next tile's stage-3 rows
// are requested into registers before stage 6 starts and parked in LDS after the heads, so the only exposed memory
// round trip of a workgroup is its first tile's.
// Same packed fragments, same K order and same head arithmetic as the launches it replaces: bit-identical results
// (plan flag reserved[7] & 1 keeps the three launches; tests/test_gpu_eqt.py compares the two).
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

constexpr int TW = 1000;                    // output samples per tile
constexpr int TILES_PER_ROW = 6;
constexpr int T_OUT = 6000;
constexpr int C4 = 144, C5 = 256, C6 = 512;  // MFMA columns per stage handed to conv_lds_areg (9, 16, 32 n-tiles)
constexpr int TAIL_NTH = 256, TAIL_WAVES = 4;
//                       CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF NB RELU
using T_d4 = LdsLayer<32, 0, 16, 2, 5, 1, -2, 0, 5, 1>;  // 2 m-tiles x 2 blocks of 5 n-tiles (the tenth is padding)
using T_d5 = LdsLayer<16, 0, 16, 2, 5, 1, -2, 0, 8, 1>;  // 2 m-tiles x 2 blocks of 8
using T_d6 = LdsLayer<16, 0, 8, 2, 7, 1, -3, 0, 8, 1>;   // 1 m-tile x 4 blocks of 8
constexpr int W4 = 2 * 5 * 16, W5 = 2 * 8 * 16, W6 = 4 * 8 * 16;  // columns the blocks actually compute
// LDS images (row strides == 16 mod 32: the two channel rows of a ds_read_b32 half-wave fall on disjoint banks);
// logical column 0 of an image = the stage's MFMA column 0, at physical column 4.  Every image has a place for every
// column the producing stage computes (no range tests in the epilogues).
constexpr int S4 = 176, S5 = 336, S6 = 528, BI = 4;
constexpr int SO = 1028;                   // staged stage-6 rows: u in [0, 1024) <-> sample t0 - 6 + u (the heads read u < 1012)
constexpr int OFF6 = 0, OFF4 = 16 * S6, OFF5 = OFF4 + 32 * S4, OFFW = OFF5 + 16 * S5, TAIL_LDS_FLOATS = OFFW + 8 * 12 + 4;
constexpr int OFFO = OFF4;                 // the staged stage-6 rows reuse the stage-4 / stage-5 input images
static_assert(8 * SO <= 32 * S4 + 16 * S5, "stage-6 staging fits the dead images");
static_assert(2 * TAIL_LDS_FLOATS * 4 <= 160 * 1024, "two workgroups per CU");
static_assert(S4 >= BI + W4 + 2 && S5 >= BI - 3 + 2 * W4 && S5 >= BI + W5 + 2 && S6 >= BI - 3 + 2 * W5 &&
                  S6 >= BI + W6 + 3 && SO >= 2 * W6,
              "image widths: every column a stage computes or reads has a place");
static_assert(S4 % 32 == 16 && S5 % 32 == 16 && S6 % 32 == 16 && SO % 4 == 0, "bank-conflict-free strides");
static_assert(TILES_PER_ROW * TW == T_OUT && TW % 8 == 0 && (TW / 4) % 2 == 0 && TW / 4 <= TAIL_NTH, "tile grid");
constexpr int PRE_ROWS = 32 / TAIL_WAVES, PRE_P = (S4 + 63) / 64;  // stage-3 samples a lane carries for the next tile

struct TailArgs {
  const float* x3;  // stage-3 rows [3 B][32][ls]
  int ls3;
  long ws3;
  float* y;         // dense (B, 3, 6000)
  const float *af4, *af5, *af6;  // packed A fragments [set][MT][CB][TAPS][64]
  const float *bs4, *bs5, *bs6;  // bias [set][COUT]
  long af4_stride, af5_stride, af6_stride;
  const float* head_w;  // [3][8][11]
  const float* head_b;  // [3]
  int B, n_tiles;
  int* ticket;       // tile dispenser of this launch (starts at 0); ticket_next: the next launch's, zeroed here
  int* ticket_next;
  unsigned long long* clk;  // debug (plan flag reserved[1] & 2): 32 words per workgroup: six shader-clock stamps for each of
                            // its first four tiles (tile start, image parked, after stages 4 / 5 / 6, heads done);
                            // [30], [31] the 100 MHz wall clock at kernel start / end
};

// Store functor of a stage: output t = 2 * column + phase lands at img[co * S + t] (the caller has folded the
// image base and the shift between the two stages' column grids into img).  [sig_lo, sig_lo + sig_w) is inside the
// row's signal, everything else is written as the next stage's zero padding (edge blocks of the first / last tile of
// a row only: a block that lies inside the signal takes the unchecked path of lds_epilogue).
template <int S>
struct TileStore {
  float* img;
  int sig_lo;
  unsigned sig_w;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    img[co * S + t] = ((unsigned)(t - sig_lo) < sig_w) ? v : 0.f;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const {
    return (unsigned)(t0 - sig_lo) < sig_w && (unsigned)(t1 - sig_lo) < sig_w;
  }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + t] = v; }
};

struct TileId {
  int d, win, t0;
};
__device__ __forceinline__ TileId tile_id(int tile, int B) {
  const int row = tile / TILES_PER_ROW, j = tile - row * TILES_PER_ROW;  // row = d * B + b
  return TileId{row / B, row, j * TW};
}

__global__ __launch_bounds__(TAIL_NTH, 2) void eqt_tail_kernel(const TailArgs a) {
  extern __shared__ float4 tail_lds_raw[];
  float* lds = reinterpret_cast<float*>(tail_lds_raw);
  // The image offsets go through an opaque register: with all images addressed off ONE base hipcc folds them into
  // constants beyond the reach of a DS instruction's immediate and pays a v_add per B fragment.
  // (in units of 16 bytes, so that the 16-byte alignment of the images stays visible: the heads read them as b128)
  static_assert(OFF4 % 4 == 0 && OFF5 % 4 == 0 && OFFO % 4 == 0 && OFFW % 4 == 0, "16-byte aligned images");
  int off4 = OFF4 / 4, off5 = OFF5 / 4, offo = OFFO / 4, offw = OFFW / 4;
  asm volatile("" : "+v"(off4), "+v"(off5), "+v"(offo), "+v"(offw));
  float* IN6 = lds + OFF6;
  float* IN4 = lds + 4 * off4;
  float* IN5 = lds + 4 * off5;
  float* OUT6 = lds + 4 * offo;
  float* HW = lds + 4 * offw;  // head weights of the current decoder: [8][12] (11 taps + pad), read as broadcasts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  if (blockIdx.x == 0 && tid == 0) *a.ticket_next = 0;  // launches on one stream are ordered: nobody reads it before the next one
  int next = tile + gridDim.x;  // the second tile is still a fixed one: the first ticket has a whole tile to arrive
  int* slot = reinterpret_cast<int*>(HW + 8 * 12);
  TileId id = tile_id(tile, a.B);
  // (every other workgroup of the grid is stamped: rows 0 .. B - 1 of the debug buffer cover both residencies of a CU)
  unsigned long long* clk =
      (a.clk && tid == 0 && (blockIdx.x & 1) == 0 && (int)(blockIdx.x >> 1) < a.B) ? a.clk + (long)(blockIdx.x >> 1) * 32 : nullptr;
  int n_done = 0;
#define TAIL_STAMP(k) \
  if (clk && n_done < 4) clk[n_done * 6 + (k)] = __builtin_readcyclecounter();
  if (clk) clk[30] = __builtin_amdgcn_s_memrealtime();

  // stage-3 samples of a tile: 32 rows x the S4 physical columns of the image (column c <-> sample t0/8 - 7 + c of the
  // row; only logical [-2, 133) is ever read for a kept output).  Wave w carries rows 8 w .. 8 w + 7, a lane the
  // columns lane + 64 p: uniform row bases + one per-lane offset, 256-byte requests.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float pre[PRE_ROWS][PRE_P];
  auto request = [&](const TileId& t) {
    const float* src = a.x3 + (long)t.win * a.ws3 + (HALO - 7 + t.t0 / 8) + (long)(PRE_ROWS * wave_u) * a.ls3;
#pragma unroll
    for (int rr = 0; rr < PRE_ROWS; ++rr) {
      const float* rowp = src + (long)rr * a.ls3;
#pragma unroll
      for (int p = 0; p < PRE_P; ++p) pre[rr][p] = (64 * p + 64 <= S4 || lane < S4 - 64 * p) ? rowp[lane + 64 * p] : 0.f;
    }
  };
  auto park = [&]() {
    float* dst = IN4 + (PRE_ROWS * wave_u) * S4 + lane;
#pragma unroll
    for (int rr = 0; rr < PRE_ROWS; ++rr)
#pragma unroll
      for (int p = 0; p < PRE_P; ++p)
        if (64 * p + 64 <= S4 || lane < S4 - 64 * p) dst[rr * S4 + 64 * p] = pre[rr][p];
  };
  request(id);

  // Every wave keeps one m-tile per stage; its A fragments (40 / 20 / 28 per lane) sit in registers while the stage
  // runs and are requested ahead: stage 5's before stage 4's MFMAs, stage 6's before stage 5's, the next tile's
  // stage-4 set as soon as stage 4 is through (behind the heads the wait for it was exposed).  88 dwords per lane and
  // tile out of L2 are nothing against the MFMA work of a tile.
  // (block indices from the scalar copy of the wave index: the epilogue's "whole block inside the signal" test must be
  // a scalar branch -- as a per-lane predicate every store of a block became its own exec-masked region)
  const int mt45 = wave_u & 1, blk45 = wave_u >> 1;
  float areg4[T_d4::CB * T_d4::TAPS], areg5[T_d5::CB * T_d5::TAPS], areg6[T_d6::CB * T_d6::TAPS];
  float bias4[4], bias5[4], bias6[4];
  load_areg<T_d4>(a.af4 + id.d * a.af4_stride, mt45, lane, areg4);
  load_biasreg<T_d4>(a.bs4 + id.d * 16, mt45, lane, bias4);
  int d_loaded = -1;

  while (true) {
    if (id.d != d_loaded) {  // head weights of this decoder (the previous tile's heads are behind the loop's closing barrier)
      if (tid < 96) {
        const int ci = tid / 12, k = tid - ci * 12;
        HW[tid] = (k < 11) ? a.head_w[id.d * 88 + ci * 11 + k] : 0.f;
      }
      d_loaded = id.d;
    }
    const bool more = next < a.n_tiles;
    const TileId nid = more ? tile_id(next, a.B) : id;
    int drawn = 0;
    if (tid == 0 && more) drawn = atomicAdd(a.ticket, 1);  // the tile after `next`; consumed behind the heads
    TAIL_STAMP(0)
    park();
    __syncthreads();
    TAIL_STAMP(1)
    const int t0 = id.t0;
    load_areg<T_d5>(a.af5 + id.d * a.af5_stride, mt45, lane, areg5);
    load_biasreg<T_d5>(a.bs5 + id.d * 16, mt45, lane, bias5);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 4: output t = 2 c + p of column c is stage-4 sample 2 (t0/8 - 3) + t = stage-5 logical column t - 3
      TileStore<S5> st{IN5 + BI - 3, 6 - t0 / 4, 1500u};
      conv_lds_areg<T_d4, S4, BI, S4, BI>(IN4, IN4, areg4, bias4, mt45, C4, st, blk45, 2, lane);
    }
    if (more) {  // this stage's fragments are spent: the next tile's travel under stages 5, 6 and the heads
      load_areg<T_d4>(a.af4 + nid.d * a.af4_stride, mt45, lane, areg4);
      load_biasreg<T_d4>(a.bs4 + nid.d * 16, mt45, lane, bias4);
    }
    __syncthreads();
    TAIL_STAMP(2)
    load_areg<T_d6>(a.af6 + id.d * a.af6_stride, 0, lane, areg6);
    load_biasreg<T_d6>(a.bs6 + id.d * 8, 0, lane, bias6);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 5: output t is stage-5 sample 2 (t0/4 - 3) + t = stage-6 logical column t - 3
      TileStore<S6> st{IN6 + BI - 3, 6 - t0 / 2, 3000u};
      conv_lds_areg<T_d5, S5, BI, S5, BI>(IN5, IN5, areg5, bias5, mt45, C5, st, blk45, 2, lane);
    }
    __syncthreads();
    TAIL_STAMP(3)
    // the next tile's stage-3 rows travel while stage 6 and the heads run
    if (more) request(nid);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 6: output t is sample t0 - 6 + t of the row, staged at u = t for the head
      TileStore<SO> st{OUT6, 6 - t0, (unsigned)T_OUT};
      conv_lds_areg<T_d6, S6, BI, S6, BI>(IN6, IN6, areg6, bias6, 0, C6, st, wave_u, TAIL_WAVES, lane);
    }
    __syncthreads();
    TAIL_STAMP(4)
    // heads: y[t0 + q] = sigmoid(b + sum_ci sum_k w[ci][k] stage6[ci][t0 + q + k - 5]); thread i owns q = 4 i .. 4 i + 3
    // and reads the staged samples u = 4 i .. 4 i + 15 of each channel as four aligned 16-byte values
    if (tid < TW / 4) {
      const float bh = a.head_b[id.d];
      float acc[4] = {bh, bh, bh, bh};
      // software-pipelined over the channels (two register sets): the staged values and the three weight quads of
      // channel ci + 1 are requested before the 44 FMAs of channel ci (rolled up, every channel waited out its LDS
      // round trip and a scalar-cache miss on its weights)
      auto fetch = [&](float4 (&x)[4], float4 (&wq)[3], int ci) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) x[q4] = *reinterpret_cast<const float4*>(OUT6 + ci * SO + 4 * tid + 4 * q4);
#pragma unroll
        for (int q4 = 0; q4 < 3; ++q4) wq[q4] = *reinterpret_cast<const float4*>(HW + ci * 12 + 4 * q4);
      };
      auto mac = [&](const float4 (&x)[4], const float4 (&wq)[3]) {
        float v[16], wk[12];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[4 * q4] = x[q4].x, v[4 * q4 + 1] = x[q4].y, v[4 * q4 + 2] = x[q4].z, v[4 * q4 + 3] = x[q4].w;
#pragma unroll
        for (int q4 = 0; q4 < 3; ++q4) wk[4 * q4] = wq[q4].x, wk[4 * q4 + 1] = wq[q4].y, wk[4 * q4 + 2] = wq[q4].z, wk[4 * q4 + 3] = wq[q4].w;
#pragma unroll
        for (int k = 0; k < 11; ++k)
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk[k], v[o + 1 + k], acc[o]);  // u = (4 i + o) + 6 + k - 5
      };
      float4 xa[4], xb[4], wa[3], wb[3];
      fetch(xa, wa, 0);
#pragma unroll
      for (int ci = 0; ci < 8; ci += 2) {
        fetch(xb, wb, ci + 1);
        __builtin_amdgcn_sched_barrier(0);
        mac(xa, wa);
        if (ci + 2 < 8) fetch(xa, wa, ci + 2);
        __builtin_amdgcn_sched_barrier(0);
        mac(xb, wb);
      }
      const int b = id.win - id.d * a.B;
      float4 r;
      r.x = 1.f / (1.f + expf(-acc[0]));
      r.y = 1.f / (1.f + expf(-acc[1]));
      r.z = 1.f / (1.f + expf(-acc[2]));
      r.w = 1.f / (1.f + expf(-acc[3]));
      *reinterpret_cast<float4*>(a.y + ((long)b * 3 + id.d) * T_OUT + t0 + 4 * tid) = r;
    }
    TAIL_STAMP(5)
    ++n_done;
    if (!more) break;
    if (tid == 0) *slot = 2 * (int)gridDim.x + drawn;
    __syncthreads();  // the staged rows are dead: their space takes the next tile's stage-3 image
    tile = next;
    id = nid;
    next = *slot;  // (rewritten at the end of the next tile, four barriers from here)
  }
  if (clk) clk[31] = __builtin_amdgcn_s_memrealtime();
#undef TAIL_STAMP
}

}  // namespace

// Replaces the steps "decoder.4", "decoder.5", "decoder.6+heads" of the plan by one fused step.
int plan_eqt_fuse_tail(Net& net) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "decoder.4") first = (int)i;
  if (first < 0 || first + 3 != (int)net.steps.size() || net.steps[first + 2].name != "decoder.6+heads") {
    set_error("fused decoder tail: layer plan not found");
    return VP_ERR_INVALID;
  }
  ConvLayer *c4 = nullptr, *c5 = nullptr, *c6 = nullptr;
  for (auto& c : net.convs) {
    if (c->name == "decoder.4") c4 = c.get();
    if (c->name == "decoder.5") c5 = c.get();
    if (c->name == "decoder.6") c6 = c.get();
  }
  if (!c4 || !c5 || !c6 || c4->n_sets != 3) {
    set_error("fused decoder tail: conv layers missing");
    return VP_ERR_INVALID;
  }
  const int x3 = c4->src1;
  net.need(x3, HALO - 7 + (TILES_PER_ROW - 1) * TW / 8 + S4);  // the last tile reads past the row: zero margin
  // stages 4 and 5 are never materialised by this plan: their tensors take no memory
  net.tensor_sets[c4->dst] = 0;
  net.tensor_sets[c5->dst] = 0;
  // two tile dispensers used alternately: a launch counts up one and zeroes the other for the launch behind it
  HostBlob* tickets = net.add_blob(std::vector<float>(2, 0.f));
  auto launches = std::make_shared<unsigned>(0u);
  Step st;
  st.name = "fused.tail (decoder.4-6 + heads, time-tiled)";
  st.flops_per_window = 0;
  for (int i = 0; i < 3; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  {  // issued MFMA work per tile: 2 m-tiles x 10 n-tiles x 40 K-steps, 2 x 16 x 20, 1 x 32 x 28 (2048 FLOP each)
    const double mfma = 2.0 * (W4 / 16) * 40 + 2.0 * (W5 / 16) * 20 + 1.0 * (W6 / 16) * 28;
    st.issued_flops_per_window = 3.0 * TILES_PER_ROW * mfma * 2048.0;
  }
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    TailArgs a{};
    const Tensor& t3 = n.tensors[x3];
    a.x3 = t3.p;
    a.ls3 = t3.ls;
    a.ws3 = (long)t3.win_stride();
    a.y = n.y;
    a.af4 = c4->afrag.d, a.af5 = c5->afrag.d, a.af6 = c6->afrag.d;
    a.bs4 = c4->bias.d, a.bs5 = c5->bias.d, a.bs6 = c6->bias.d;
    a.af4_stride = (long)(c4->afrag.h.size() / 3);
    a.af5_stride = (long)(c5->afrag.h.size() / 3);
    a.af6_stride = (long)(c6->afrag.h.size() / 3);
    a.head_w = c6->e0.d;
    a.head_b = c6->e1.d;
    a.B = B;
    a.n_tiles = 3 * B * TILES_PER_ROW;
    a.clk = (n.debug_clock && n.debug_clock->d)
                ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) + (size_t)n.max_batch * 32 + 64 * 8
                : nullptr;
    const int grid = a.n_tiles < 512 ? a.n_tiles : 512;  // two resident workgroups per CU
    int* tk = reinterpret_cast<int*>(tickets->d);
    a.ticket = tk + (*launches & 1);
    a.ticket_next = tk + ((*launches + 1) & 1);
    ++*launches;
    hipLaunchKernelGGL(eqt_tail_kernel, dim3(grid), dim3(TAIL_NTH), TAIL_LDS_FLOATS * sizeof(float), s, a);
    return 0;
  };
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_tail_kernel), TAIL_LDS_FLOATS * sizeof(float)});
  net.steps.erase(net.steps.begin() + first, net.steps.end());
  net.steps.push_back(std::move(st));
  return VP_OK;
}

}  // namespace vp
