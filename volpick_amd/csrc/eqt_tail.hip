// EQTransformer decoder tail as ONE launch: stages 4, 5, 6 of the three decoders (Upsample(2) + Conv1d + ReLU, folded
// into two-phase filters, eqt.hip) and the Conv1d(8,1,11) + sigmoid heads, per TIME TILE of one (decoder, window) row.
//
// Why: as three conv_mfma_kernel launches these stages wrote 16 x 1500 and 16 x 3000 samples per decoder and window
// to memory only to read them straight back (0.44 GB of the 1.16 GB a 256-window step moved, PMC), and every launch
// paid its load phase in lock step (all workgroups fetch, then all compute) plus a kernel boundary.  Here a tile's
// rows never leave the CU: the workgroup reads 32 x 260 samples of stage 3, runs the three stages LDS to LDS with the
// halo each needs recomputed (2 % extra MFMA work at 2000 output samples per tile) and writes 2000 probabilities.
//
//   tile of outputs [t0, t0 + 2000) of the 6000-sample row, t0 = 2000 j; column counts include the halos of the
//   stages behind them (needed / computed as whole blocks of n-tiles):
//     stage 6 columns n6 = t0/2 - 3 + [0, 1006 / 1024)  reads stage-5 samples [n6 - 3, n6 + 3]  (K = 11 folded to 7 taps)
//     stage 5 columns n5 = t0/4 - 3 + [0,  506 /  512)  reads stage-4 samples [n5 - 2, n5 + 2]  (K = 9 folded to 5 taps)
//     stage 4 columns n4 = t0/8 - 3 + [0,  256 /  256)  reads stage-3 samples [n4 - 2, n4 + 2]  (K = 7 folded to 5 taps)
//   Samples outside a row's signal are written as zeros: they are the next stage's zero padding.
//
// One 512-thread workgroup (8 wavefronts, two per SIMD) per CU, persistent over the tiles in (decoder, window, tile)
// order.  Every wave keeps one m-tile per stage, so a stage's A operand is 40 / 20 / 28 fragments per lane held in
// registers (requested a phase ahead) and the K loops touch nothing but LDS: per 16x16x4 MFMA one B fragment,
// fetched in pairs (ds_read2_b32) one K-step ahead (conv_lds_areg; tools/micro/micro_mfma_lds.hip: this loop shape
// sustains 89-94 % of the pipe's issue rate).  The next tile's stage-3 rows are requested into registers before
// stage 6 starts and parked in LDS after the heads, so the only exposed memory round trip of a workgroup is its first
// tile's.
//
// The heads run on the matrix pipes too.  On gfx950 the fp32 MFMA shares the SIMD's FMA lanes with the VALU -- a VALU
// phase of one wave does not hide behind another wave's MFMAs (tools/tail_clock.py: as 88 v_fmac per output the heads
// were 6-8 k of the 53 k cycles of a tile, beside whatever ran with them) -- so the 8 -> 1, k = 11 convolution is
// written as a Toeplitz product: 16 consecutive outputs are the 16 rows of a tile, the columns are 16-sample blocks of
// the row, K = 8 channels x 28 taps (11 of them non-zero per row):  y[16 n + m] = sum_ci sum_tap A[m][ci, tap] x_ci[16 n + tap + 1],
// A[m][ci, tap] = w[ci][tap - m].  56 MFMAs per wave and tile.  Stage 6 stages its output for that read pattern:
// sample u of channel ci sits at [ci][u % 16][u / 16], so that a B fragment (fixed tap, 16 blocks) is 16 consecutive
// words.  Zero products leave an fp32 FMA chain unchanged and the non-zero ones come in the order ci, k of the
// VALU loop: bit-identical to it.
// Same packed fragments, same K order and same head arithmetic as the launches it replaces: bit-identical results
// (plan flag plan_flags[7] & 1 keeps the three launches; tests/test_gpu_eqt.py compares the two).
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

constexpr int TW = 2000;                    // output samples per tile
constexpr int TILES_PER_ROW = 3;
constexpr int T_OUT = 6000;
constexpr int C4 = 256, C5 = 512, C6 = 1024;  // MFMA columns per stage handed to conv_lds_areg (16, 32, 64 n-tiles)
constexpr int TAIL_NTH = 512, TAIL_WAVES = 8;
//                       CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF NB RELU
using T_d4 = LdsLayer<32, 0, 16, 2, 5, 1, -2, 0, 4, 1>;  // 2 m-tiles x 4 blocks of 4 n-tiles: one block per wave
using T_d5 = LdsLayer<16, 0, 16, 2, 5, 1, -2, 0, 8, 1>;  // 2 m-tiles x 4 blocks of 8
using T_d6 = LdsLayer<16, 0, 8, 2, 7, 1, -3, 0, 8, 1>;   // 1 m-tile x 8 blocks of 8
// LDS images (row strides == 16 mod 32: the two channel rows of a ds_read_b32 half-wave fall on disjoint banks);
// logical column 0 of an image = the stage's MFMA column 0, at physical column 4.  Every image has a place for every
// column the producing stage computes (no range tests in the epilogues).
constexpr int S4 = 272, S5 = 528, S6 = 1040, BI = 4;
// staged stage-6 output for the heads: sample u = t0 - 6 + u of channel ci at [ci][u % 16][u / 16]; row stride == 2 mod 32
// keeps stage 6's stores (lanes: u = 2 n + p, 16 rows 2 apart) and the heads' reads (4 taps x 16 blocks) off each other's banks
constexpr int HSB = 130, HCH = 16 * HSB;
constexpr int HEAD_KS = 8 * 7;             // K-steps of the head product: 8 channels x 28 taps / 4
constexpr int OFF6 = 0, OFF4 = 16 * S6, OFF5 = OFF4 + 32 * S4, TAIL_LDS_FLOATS = OFF5 + 16 * S5;
constexpr int OFFO = OFF4;                 // the staged stage-6 rows reuse the stage-4 / stage-5 input images
static_assert(8 * HCH <= 32 * S4 + 16 * S5, "stage-6 staging fits the dead images");
static_assert(TAIL_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
static_assert(S4 >= BI + C4 + 2 && S5 >= BI - 3 + 2 * C4 && S5 >= BI + C5 + 2 && S6 >= BI - 3 + 2 * C5 && S6 >= BI + C6 + 3 &&
                  HSB >= 2 * C6 / 16 + 2,
              "image widths: every column a stage computes or reads has a place");
static_assert(S4 % 32 == 16 && S5 % 32 == 16 && S6 % 32 == 16 && HSB % 32 == 2, "bank-conflict-free strides");
static_assert(TILES_PER_ROW * TW == T_OUT && TW % 16 == 0 && TW / 16 <= 16 * TAIL_WAVES, "tile grid");
constexpr int PRE_ROWS = 32 / TAIL_WAVES, PRE_P = (S4 + 63) / 64;  // stage-3 samples a lane carries for the next tile

struct TailArgs {
  const float* x3;  // stage-3 rows [3 B][32][ls]
  int ls3;
  long ws3;
  float* y;         // dense (B, 3, 6000)
  const float *af4, *af5, *af6;  // packed A fragments regrouped for 16-byte loads [set][MT][CB * TAPS / 4][64][4]
  const float *bs4, *bs5, *bs6;  // bias [set][COUT]
  long af4_stride, af5_stride, af6_stride;
  const float* head_a;  // Toeplitz A fragments of the heads [3][HEAD_KS / 4][64][4]
  const float* head_b;  // [3]
  int B, n_tiles;
  unsigned long long* clk;  // debug (plan flag plan_flags[1] & 2): 32 words per workgroup: six shader-clock stamps for each of
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

// Stage 6 -> the heads' staging layout (file comment): t = 2 (colb + 16 j + n) + p with colb a multiple of 128, so
// t % 16 = (2 n + p) % 16 is a per-lane constant of the block and t / 16 = colb / 8 + 2 j + n / 8: two per-lane bases
// (p = 0, 1) per block, every store an immediate offset from one of them.
struct HeadStage {
  static constexpr bool custom_block_epilogue = true;
  float* img;
  int sig_lo;
  unsigned sig_w;
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 2 && L::MT == 1 && L::RELU == 1 && L::OUT_OFF == 0 && (L::NB * 16) % 8 == 0, "stage 6 of the decoder");
    (void)mt;
    float* base[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) base[p] = img + (2 * g) * HCH + ((2 * n + p) & 15) * HSB + (n >> 3) + (colb >> 3);
    const int t_first = 2 * colb, t_last = 2 * (colb + L::NB * 16) - 1;
    const bool fast = (unsigned)(t_first - sig_lo) < sig_w && (unsigned)(t_last - sig_lo) < sig_w;
    if (fast) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < L::NB; ++j) base[r & 1][(r >> 1) * HCH + 2 * j] = fmaxf(acc[j][r] + biasv[r], 0.f);
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        const int t = 2 * (colb + j * 16 + n) + (r & 1);
        const float v = fmaxf(acc[j][r] + biasv[r], 0.f);
        base[r & 1][(r >> 1) * HCH + 2 * j] = ((unsigned)(t - sig_lo) < sig_w) ? v : 0.f;
      }
  }
};

struct TileId {
  int d, win, t0;
};
__device__ __forceinline__ TileId tile_id(int tile, int B) {
  const int row = tile / TILES_PER_ROW, j = tile - row * TILES_PER_ROW;  // row = d * B + b
  return TileId{row / B, row, j * TW};
}

__global__ __launch_bounds__(TAIL_NTH) void eqt_tail_kernel(const TailArgs a) {
  extern __shared__ float4 tail_lds_raw[];
  float* lds = reinterpret_cast<float*>(tail_lds_raw);
  // The image offsets go through an opaque register: with all images addressed off ONE base hipcc folds them into
  // constants beyond the 64 KB reach of a DS instruction's immediate and pays a v_add per B fragment.
  static_assert(OFF4 % 4 == 0 && OFF5 % 4 == 0 && OFFO % 4 == 0, "16-byte aligned images");
  int off4 = OFF4 / 4, off5 = OFF5 / 4, offo = OFFO / 4;
  asm volatile("" : "+v"(off4), "+v"(off5), "+v"(offo));
  float* IN6 = lds + OFF6;
  float* IN4 = lds + 4 * off4;
  float* IN5 = lds + 4 * off5;
  float* OUT6 = lds + 4 * offo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  TileId id = tile_id(tile, a.B);
  unsigned long long* clk = (a.clk && tid == 0 && (int)blockIdx.x < a.B) ? a.clk + (long)blockIdx.x * 32 : nullptr;
  int n_done = 0;
#define TAIL_STAMP(k) \
  if (clk && n_done < 4) clk[n_done * 6 + (k)] = __builtin_readcyclecounter();
  if (clk) clk[30] = __builtin_amdgcn_s_memrealtime();

  // stage-3 samples of a tile: 32 rows x the S4 physical columns of the image (column c <-> sample t0/8 - 7 + c of the
  // row; only logical [-2, 258) is ever read for a kept output).  Wave w carries rows 4 w .. 4 w + 3, a lane the
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

  // Every wave keeps one m-tile per stage; its A fragments sit in registers while the stage runs and are requested
  // ahead: stage 5's before stage 4's MFMAs, stage 6's before stage 5's, the heads' before stage 6's, the next tile's
  // stage-4 set as soon as stage 4 is through.  144 dwords per lane and tile out of L2 (as 36 16-byte loads) are nothing
  // against the MFMA work of a tile.  (Block indices from the scalar copy of the wave index: the epilogue's "whole block inside the
  // signal" test must be a scalar branch -- as a per-lane predicate every store of a block became its own exec-masked
  // region.)
  const int mt45 = wave_u & 1, blk45 = wave_u >> 1;
  float areg4[T_d4::CB * T_d4::TAPS], areg5[T_d5::CB * T_d5::TAPS], areg6[T_d6::CB * T_d6::TAPS];
  float bias4[4], bias5[4], bias6[4];
  load_areg4<T_d4>(a.af4 + id.d * a.af4_stride, mt45, lane, areg4);
  load_biasreg<T_d4>(a.bs4 + id.d * 16, mt45, lane, bias4);

  while (true) {
    const int next = tile + gridDim.x;
    const bool more = next < a.n_tiles;
    const TileId nid = more ? tile_id(next, a.B) : id;
    TAIL_STAMP(0)
    park();
    __syncthreads();
    TAIL_STAMP(1)
    const int t0 = id.t0;
    load_areg4<T_d5>(a.af5 + id.d * a.af5_stride, mt45, lane, areg5);
    load_biasreg<T_d5>(a.bs5 + id.d * 16, mt45, lane, bias5);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 4: output t = 2 c + p of column c is stage-4 sample 2 (t0/8 - 3) + t = stage-5 logical column t - 3
      TileStore<S5> st{IN5 + BI - 3, 6 - t0 / 4, 1500u};
      conv_lds_areg<T_d4, S4, BI, S4, BI>(IN4, IN4, areg4, bias4, mt45, C4, st, blk45, 4, lane);
    }
    if (more) {  // this stage's fragments are spent: the next tile's travel under stages 5, 6 and the heads
      load_areg4<T_d4>(a.af4 + nid.d * a.af4_stride, mt45, lane, areg4);
      load_biasreg<T_d4>(a.bs4 + nid.d * 16, mt45, lane, bias4);
    }
    __syncthreads();
    TAIL_STAMP(2)
    load_areg4<T_d6>(a.af6 + id.d * a.af6_stride, 0, lane, areg6);
    load_biasreg<T_d6>(a.bs6 + id.d * 8, 0, lane, bias6);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 5: output t is stage-5 sample 2 (t0/4 - 3) + t = stage-6 logical column t - 3
      TileStore<S6> st{IN6 + BI - 3, 6 - t0 / 2, 3000u};
      conv_lds_areg<T_d5, S5, BI, S5, BI>(IN5, IN5, areg5, bias5, mt45, C5, st, blk45, 4, lane);
    }
    __syncthreads();
    TAIL_STAMP(3)
    // the next tile's stage-3 rows and the heads' A fragments (stage 5's registers are free) travel while stage 6 runs
    if (more) request(nid);
    float hreg[HEAD_KS];
    {
      const f32x4* ha = reinterpret_cast<const f32x4*>(a.head_a) + (long)id.d * (HEAD_KS / 4) * 64 + lane;
#pragma unroll
      for (int s4 = 0; s4 < HEAD_KS / 4; ++s4) {
        const f32x4 v = ha[s4 * 64];
        hreg[4 * s4] = v.x, hreg[4 * s4 + 1] = v.y, hreg[4 * s4 + 2] = v.z, hreg[4 * s4 + 3] = v.w;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 6: output t is sample t0 - 6 + t of the row, staged for the heads at [co][t % 16][t / 16]
      HeadStage st{OUT6, 6 - t0, (unsigned)T_OUT};
      conv_lds_areg<T_d6, S6, BI, S6, BI>(IN6, IN6, areg6, bias6, 0, C6, st, wave_u, TAIL_WAVES, lane);
    }
    __syncthreads();
    TAIL_STAMP(4)
    {  // heads (file comment): wave w owns the 16-sample blocks 16 w .. 16 w + 15 of the tile: lane (n, g) accumulates
       // outputs q = 16 (16 w + n) + 4 g + r, r = 0 .. 3, reading x_ci[u = 16 (16 w + n) + 4 tq + g + 1] per K-step (ci, tq)
      const int n = lane & 15, g = lane >> 4;
      const float bh = a.head_b[id.d];
      f32x4 acc = {bh, bh, bh, bh};
      const float* bp = OUT6 + (16 * wave_u + n);
      float bA, bB;
      auto load_b = [&](int s) {
        const int ci = s / 7, tq = s - ci * 7;
        const int eg = 4 * tq + 1 + g;  // u = 16 block + eg, eg in [1, 28]: row eg % 16, column block + eg / 16
        return bp[ci * HCH + (eg & 15) * HSB + (eg >> 4)];
      };
      bA = load_b(0);
#pragma unroll
      for (int s = 0; s < HEAD_KS; ++s) {
        if (s + 1 < HEAD_KS) {
          if (s & 1) bA = load_b(s + 1); else bB = load_b(s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[s], (s & 1) ? bB : bA, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int blk = 16 * wave_u + n;  // 16-sample block of the tile
      if (blk < TW / 16) {
        const int b = id.win - id.d * a.B;
        float4 r;
        r.x = 1.f / (1.f + expf(-acc[0]));
        r.y = 1.f / (1.f + expf(-acc[1]));
        r.z = 1.f / (1.f + expf(-acc[2]));
        r.w = 1.f / (1.f + expf(-acc[3]));
        *reinterpret_cast<float4*>(a.y + ((long)b * 3 + id.d) * T_OUT + t0 + 16 * blk + 4 * g) = r;
      }
    }
    TAIL_STAMP(5)
    ++n_done;
    if (!more) break;
    __syncthreads();  // the staged rows are dead: their space takes the next tile's stage-3 image
    tile = next;
    id = nid;
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
  // every tile fetches its A operands anew: as 16-byte loads (regroup_afrag4), a quarter of the load instructions
  HostBlob* q4 = c4->afrag_q4 ? c4->afrag_q4 : net.add_blob(regroup_afrag4(*c4));
  HostBlob* q5 = c5->afrag_q4 ? c5->afrag_q4 : net.add_blob(regroup_afrag4(*c5));
  HostBlob* q6 = c6->afrag_q4 ? c6->afrag_q4 : net.add_blob(regroup_afrag4(*c6));
  // Toeplitz A operand of the heads (file comment): K-step s = ci * 7 + tq, lane (m = lane & 15, g = lane >> 4) holds
  // A[m][ci, tap = 4 tq + g] = w[ci][tap - m] for 0 <= tap - m <= 10, else 0
  std::vector<float> ha((size_t)3 * HEAD_KS * 64, 0.f);
  for (int d = 0; d < 3; ++d)
    for (int ci = 0; ci < 8; ++ci)
      for (int tq = 0; tq < 7; ++tq)
        for (int l = 0; l < 64; ++l) {
          const int m = l & 15, k = 4 * tq + (l >> 4) - m;
          const int st = ci * 7 + tq;  // regrouped for 16-byte loads: [step / 4][lane][step % 4]
          if (k >= 0 && k <= 10)
            ha[(((size_t)d * (HEAD_KS / 4) + st / 4) * 64 + l) * 4 + (st & 3)] = c6->e0.h[(size_t)d * 88 + ci * 11 + k];
        }
  HostBlob* head_a = net.add_blob(std::move(ha));
  Step st;
  st.name = "fused.tail (decoder.4-6 + heads, time-tiled)";
  st.flops_per_window = 0;
  for (int i = 0; i < 3; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  {  // issued MFMA work per tile: 2 m-tiles x 16 n-tiles x 40 K-steps, 2 x 32 x 20, 1 x 64 x 28, heads 8 x 56 (2048 FLOP each)
    const double mfma = 2.0 * (C4 / 16) * 40 + 2.0 * (C5 / 16) * 20 + 1.0 * (C6 / 16) * 28 + 8.0 * HEAD_KS;
    st.set_issued(3.0 * TILES_PER_ROW * mfma * 2048.0, 0.0, 0.0);
  }
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    TailArgs a{};
    const Tensor& t3 = n.tensors[x3];
    a.x3 = t3.p;
    a.ls3 = t3.ls;
    a.ws3 = (long)t3.win_stride();
    a.y = n.y;
    a.af4 = q4->d, a.af5 = q5->d, a.af6 = q6->d;
    a.bs4 = c4->bias.d, a.bs5 = c5->bias.d, a.bs6 = c6->bias.d;
    a.af4_stride = (long)(c4->afrag.h.size() / 3);
    a.af5_stride = (long)(c5->afrag.h.size() / 3);
    a.af6_stride = (long)(c6->afrag.h.size() / 3);
    a.head_a = head_a->d;
    a.head_b = c6->e1.d;
    a.B = B;
    a.n_tiles = 3 * B * TILES_PER_ROW;
    a.clk = (n.debug_clock && n.debug_clock->d)
                ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) + (size_t)n.max_batch * 32 + 64 * 8
                : nullptr;
    const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
    hipLaunchKernelGGL(eqt_tail_kernel, dim3(grid), dim3(TAIL_NTH), TAIL_LDS_FLOATS * sizeof(float), s, a);
    return 0;
  };
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_tail_kernel), TAIL_LDS_FLOATS * sizeof(float)});
  net.steps.erase(net.steps.begin() + first, net.steps.end());
  net.steps.push_back(std::move(st));
  return VP_OK;
}

}  // namespace vp
