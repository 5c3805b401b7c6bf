// EQTransformer decoder stages 0-3 as ONE launch: per (decoder, window) row the four stages (Upsample(2) + Conv1d + ReLU,
// folded into two-phase filters, eqt.hip) run back to back inside one workgroup with every intermediate row in LDS:
//
//   decoder.in 16 x 47 -> stage 0: 64 x 94 -> stage 1: 64 x 188 -> stage 2: 32 x 375 (+ the cropped edge) -> stage 3: 32 x 750 -> memory
//
// As five launches (four conv_mfma_kernel + decoder2_edge_kernel) these stages took 148 us per 256 windows for 88 us of
// MFMA issue: the rows are short (47 .. 375 columns), every launch is one or two residencies of the chip with its load
// phase in lock step, and every intermediate made a round trip through memory (0.19 GB per step).  Here one 512-thread
// workgroup per CU owns a row: 138 KB of LDS images, A operands of a stage (12 / 48 / 48 / 40 fragments per lane: every
// wave keeps one m-tile) in registers, fetched as 16-byte loads while the stage before runs, B fragments out of LDS one
// K-step ahead (conv_lds_areg).  The two outputs at the cropped right edge of stage 2 (eqt.hip: decoder2_edge_kernel)
// are computed by one wave beside the stage itself, from the definition, with the same pre-summed taps.
// Same packed fragments, same K order: bit-identical to the launches it replaces (plan flag plan_flags[7] & 2 keeps them).
#include "conv_b3.h"
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

constexpr int D03_NTH = 512, D03_WAVES = 8;
//                      CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF NB RELU
using D_0 = LdsLayer<16, 0, 64, 2, 3, 1, -1, 0, 3, 1>;  // 8 m-tiles x 1 block of 3 n-tiles: one block per wave
using D_1 = LdsLayer<64, 0, 64, 2, 3, 1, -1, 0, 6, 1>;  // 8 m-tiles x 1 block of 6
using D_2 = LdsLayer<64, 0, 32, 2, 3, 1, -1, 0, 6, 1>;  // 4 m-tiles x 2 blocks of 6
using D_3 = LdsLayer<32, 0, 32, 2, 5, 1, -2, 0, 6, 1>;  // 4 m-tiles x 4 blocks of 6: two per wave
constexpr int L0 = 47, L1 = 94, L2 = 188, L3 = 375, L4 = 750;  // row lengths: stage inputs and the final output
constexpr int C0 = 48, C1 = 96, C2 = 192, C3 = 384;              // MFMA columns per stage
// images: row strides == 16 mod 32, logical column 0 at physical column 4; wide enough for every column a stage computes
constexpr int S0 = 80, S1 = 112, S2 = 208, S3 = 400, BI = 4;
static_assert(S0 >= BI + C0 + 1 && S1 >= BI + 2 * C0 + 1 && S1 >= BI + C1 + 1 && S2 >= BI + 2 * C1 + 1 && S2 >= BI + C2 + 1 &&
                  S3 >= BI + 2 * C2 + 2 && S3 >= BI + C3 + 2,
              "image widths");
static_assert(S0 % 32 == 16 && S1 % 32 == 16 && S2 % 32 == 16 && S3 % 32 == 16, "bank-conflict-free strides");
constexpr int OFF0 = 0, OFF1 = OFF0 + 16 * S0, OFF2 = OFF1 + 64 * S1, OFF3 = OFF2 + 64 * S2, D03_LDS_FLOATS = OFF3 + 32 * S3;
static_assert(D03_LDS_FLOATS * 4 <= 160 * 1024 && OFF1 % 4 == 0 && OFF2 % 4 == 0 && OFF3 % 4 == 0, "LDS budget");

// B3 instantiation: stages 1, 2 and 3 (97 % of the row's MFMA issue) on the bf16 matrix cores with exact three-piece
// operands (conv_b3.h).  Stage 0 (fp32 MFMA, its GEMM rows regrouped (phase, channel)) writes its output as a three-piece
// image, stages 1 and 2 read and write piece images, stage 3 reads one and writes the row to memory.  LDS: the stage-2
// input image (85 KB, [column][64 + 8 channels]), and behind it the stage-3 input image (77 KB, chunk planes, unpadded),
// which takes the place of the (dead) stage-0 input and stage-1 input images -- one more barrier per row, 158 KB in all.
constexpr int B3_X1_NC = 112, B3_X2_NC = 208;                          // columns: sample t at column t + 1
using QX1 = B3Chunk<64, B3_X1_NC>;                                     // chunk-plane images (conv_b3.h): no padding channels
using QX2 = B3Chunk<64, B3_X2_NC>;
constexpr int B3_X1_PS = QX1::PS, B3_X2_PS = QX2::PS;                  // bf16 elements per piece
static_assert(B3_X1_NC >= C1 + 3 && B3_X2_NC >= C2 + 3, "every column stages 1 / 2 read has a place");
constexpr int B3_OFF_X2 = 0, B3_OFF_R = 3 * B3_X2_PS * 2;              // bytes
constexpr int B3_OFF_X0 = B3_OFF_R, B3_OFF_X1 = B3_OFF_X0 + 16 * S0 * 4, B3_OFF_X3 = B3_OFF_R;
// stage-3 input as a chunk-plane three-piece image (conv_b3.h): sample t at column t + 2, 400 columns, no padding channels
constexpr int B3_X3_NC = 400, B3_X3_C0 = 2;
using QX3 = B3Chunk<32, B3_X3_NC>;
constexpr int B3_OFF_EDGE = B3_OFF_X3 + 3 * QX3::PS * 2;  // [8 waves][64] partial sums of the two edge samples
constexpr int B3_LDS_BYTES = B3_OFF_EDGE + D03_WAVES * 64 * 4;
static_assert(B3_X3_NC >= C3 + 4 + 0 && B3_X3_NC >= L3 + B3_X3_C0, "every column stage 3 reads has a place");
static_assert(B3_OFF_X1 + 3 * B3_X1_PS * 2 <= B3_LDS_BYTES && B3_LDS_BYTES <= 160 * 1024 && B3_OFF_R % 16 == 0 && B3_OFF_X1 % 16 == 0,
              "LDS budget of the bf16-piece variant");

struct Dec03Args {
  const float* x;  // decoder.in rows [3 B][16][ls]
  int ls_x;
  long ws_x;
  float* y;        // decoder.3 rows [3 B][32][ls]
  int ls_y;
  long ws_y;
  const float* af[4];  // A fragments regrouped for 16-byte loads [set][MT][CB * TAPS / 4][64][4]
  const float* bs[4];  // bias [set][COUT]
  long af_stride[4];
  const uint4* af3[3];  // B3: three-piece operands of stages 1, 2 (rows (phase, channel)) and 3 (rows (channel, phase)): [set][MT][steps][piece][64]
  long af3_stride[3];   // uint4 per set
  const float* edge_w;  // [3][64][64][3] pre-summed taps of the two edge outputs of stage 2 (eqt.hip)
  const float* edge_b;  // [3][32]
  int B, n_rows;
  int even_split;  // plan_flags[7] bit 11: stage 3's n-tiles 12 + 12 over the two waves of a SIMD instead of 14 + 10
  unsigned long long* clk;  // -DD3_CLOCK=1 builds (tools/dec03_clock.py): shader-clock stamps of waves 0 and 4 of workgroup 0
};

// Stage output t = 2 * column + phase at img[co * S + t]; [0, len) is the row, beyond it the next stage's zero padding.
template <int S>
struct RowStore {
  float* img;  // image + BI
  unsigned len;
  __device__ __forceinline__ void operator()(int co, int t, float v) const { img[co * S + t] = ((unsigned)t < len) ? v : 0.f; }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return (unsigned)t1 < len; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + t] = v; }
};
// Stage 2: the samples from `hi` on belong to the edge fix (written beside this stage) or to the zero padding
// (never written by anybody): they are skipped, not zeroed.
template <int S>
struct ClipStore {
  float* img;
  int hi;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if (t < hi) img[co * S + t] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t1 < hi; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + t] = v; }
};
// Stage 3 -> memory: registers (0, 1) / (2, 3) of a lane are two consecutive samples of one channel: 8-byte stores,
// 16 lanes = one full 128-byte line per channel row.
struct RowOut {
  static constexpr bool custom_block_epilogue = true;
  float* row0;  // y + win * ws + HALO
  int ls;
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 2 && L::RELU == 1 && L::OUT_OFF == 0 && L4 % 2 == 0, "stage 3 of the decoder");
#pragma unroll
    for (int rr = 0; rr < 4; rr += 2) {
      float* row = row0 + (long)(mt * 8 + 2 * g + rr / 2) * ls;
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        const int t = 2 * (colb + j * 16 + n);
        if (t < L4)
          *reinterpret_cast<float2*>(row + t) =
              make_float2(fmaxf(acc[j][rr] + biasv[rr], 0.f), fmaxf(acc[j][rr + 1] + biasv[rr + 1], 0.f));
      }
    }
  }
};

// B3, stage 0: rows (phase, channel): m-tile mt holds phase mt / 4, channels (mt % 4) * 16 ..; -> chunk-plane three-piece image
struct Stage0Pieces {
  static constexpr bool custom_block_epilogue = true;
  bf16_t* img;
  __device__ __forceinline__ void quad(const int co, const int t, const float (&v)[4]) const {
    const float z[4] = {t < L1 ? v[0] : 0.f, t < L1 ? v[1] : 0.f, t < L1 ? v[2] : 0.f, t < L1 ? v[3] : 0.f};
    b3c_store4<64, B3_X1_NC>(img, t + 1, co >> 2, z);
  }
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 2 && L::RELU == 1 && L::OUT_OFF == 0 && L::COUT == 64, "stage 0 of the decoder");
#pragma unroll
    for (int j = 0; j < L::NB; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[j][r] + biasv[r], 0.f);  // biasv: the caller's (phase, channel) order
      this->quad((mt & 3) * 16 + 4 * g, 2 * (colb + j * 16 + n) + (mt >> 2), v);
    }
  }
};
// B3, stage 2 -> the three-piece image of stage 3; samples from `hi` on are the edge fix's / padding (ClipStore)
struct ClipQuad {
  bf16_t* img;
  int hi;
  __device__ __forceinline__ void quad(const int co, const int t, const float (&v)[4]) const {
    if (t < hi) b3c_store4<32, B3_X3_NC>(img, t + B3_X3_C0, co >> 2, v);
  }
};

template <bool B3>
__global__ __launch_bounds__(D03_NTH) void eqt_dec03_kernel(const Dec03Args a) {
  extern __shared__ float4 d03_lds_raw[];
  float* lds = reinterpret_cast<float*>(d03_lds_raw);
  if constexpr (B3) {
    char* base = reinterpret_cast<char*>(d03_lds_raw);
    int o_r = B3_OFF_R / 16;  // opaque: keeps the fp32 images' addresses inside the DS immediates (as below)
    asm volatile("" : "+v"(o_r));
    float* X0 = reinterpret_cast<float*>(base + 16 * o_r);
    bf16_t* X3 = reinterpret_cast<bf16_t*>(base + 16 * o_r);
    float* EDGE = reinterpret_cast<float*>(base + B3_OFF_EDGE);
    bf16_t* const X1 = reinterpret_cast<bf16_t*>(base + B3_OFF_X1);
    bf16_t* const X2 = reinterpret_cast<bf16_t*>(base + B3_OFF_X2);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int row = blockIdx.x;
    if (row >= a.n_rows) return;
    for (int i = tid; i < B3_LDS_BYTES / 16; i += D03_NTH) d03_lds_raw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pre[3];  // the row's input as a whole 16 x 80 image (zero outside the 47 samples: the image is rewritten every row)
    auto request = [&](int r) {
      const float* src = a.x + (long)r * a.ws_x + HALO;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = tid + k * D03_NTH, c = idx / S0, t = idx - c * S0 - BI;
        pre[k] = (idx < 16 * S0 && (unsigned)t < (unsigned)L0) ? src[(long)c * a.ls_x + t] : 0.f;
      }
    };
    auto park = [&]() {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = tid + k * D03_NTH;
        if (idx < 16 * S0) X0[idx] = pre[k];
      }
    };
    request(row);
    int d = row / a.B;
    float areg0[D_0::CB * D_0::TAPS], bias0[4];
    auto load_stage0 = [&](int dd) {
      load_areg4<D_0>(a.af[0] + dd * a.af_stride[0], wave_u, lane, areg0);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias0[r] = a.bs[0][dd * 64 + (wave_u & 3) * 16 + 4 * g + r];
    };
    load_stage0(d);
    __syncthreads();
    const int mt23 = wave_u & 3, blk23 = wave_u >> 2;
    // (The operands of stages 1-3 are requested as bursts of 18 / 18 / 15 1-KB loads per wave at the stage boundaries.  Spreading
    // them over the K-steps of the stage before, as eqt_res3t_kernel does, was built and measured in round 6: stage 0 4.1 k cycles
    // instead of 5.2 k, stage 3 14.8 k instead of 13.7 k (29 registers spilled), the row 42.6 k either way, 0.8 % SLOWER in the
    // pipeline.  The stages' K loops run at 65-84 % of their matrix time whatever the requests do: tools/dec03_clock.py.)
#ifndef D3_CLOCK
#define D3_CLOCK 0
#endif
#if D3_CLOCK
    unsigned long long* clk = (a.clk && (tid & 255) == 0 && blockIdx.x == 0) ? a.clk + (tid >> 8) * 128 : nullptr;
    int stamp = 0;
#define D3_STAMP() \
  if (clk && stamp < 120) clk[stamp++] = __builtin_readcyclecounter();
#else
#define D3_STAMP()
#endif
    while (true) {
      const int next = row + gridDim.x;
      const bool more = next < a.n_rows;
      const int nd = more ? next / a.B : d;
      D3_STAMP()  // 0: row start
      park();
      __syncthreads();
      D3_STAMP()  // 1: input parked
      uint4 a1[B3Steps<64, 3>::STEPS * 3];  // stage 1's operand: on its way under stage 0
      b3_load_a<64, 3>(a.af3[0] + d * a.af3_stride[0], wave_u, lane, a1);
      {  // stage 0: 16 x 47 -> 64 x 94, three pieces
        Stage0Pieces st{X1};
        b3c_zero_rest<64, B3_X1_NC>(X1, 1, 1 + 2 * C0, tid, D03_NTH);
        conv_lds_areg<D_0, S0, BI, S0, BI>(X0, X0, areg0, bias0, wave_u, C0, st, 0, 1, lane);
      }
      const int n = lane & 15;
      {  // stage 1: 64 x 94 -> 64 x 188 on the bf16 matrix cores: wave = m-tile (phase w / 4, channels 16 (w % 4) ..), six n-tiles,
         // its operand in registers (requested ahead of the barrier)
        const int ph = wave_u >> 2, co0 = (wave_u & 3) * 16 + 4 * g;
        float bias1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias1[r] = a.bs[1][d * 64 + co0 + r];
        D3_STAMP()  // 2: stage 0 done (this wave)
        __syncthreads();
        D3_STAMP()  // 3: barrier
        b3c_mac_tile_pairs<64, B3_X1_NC, 3, 6>(b3c_lane_ptr<64, B3_X1_NC, 3>(X1, 0, lane), a1, [&](const int j, const f32x4 acc) {
          const int t = 2 * (j * 16 + n) + ph;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = t < L2 ? fmaxf(acc[r] + bias1[r], 0.f) : 0.f;
          b3c_store4<64, B3_X2_NC>(X2, t + 1, co0 >> 2, v);
        });
      }
      D3_STAMP()  // 4: stage 1 done (this wave)
      uint4 a2[B3Steps<64, 3>::STEPS * 3];  // stage 2's operand: requested before the barrier, stage 1's registers are free
      b3_load_a<64, 3>(a.af3[1] + d * a.af3_stride[1], wave_u & 3, lane, a2);
      __syncthreads();
      D3_STAMP()  // 5: barrier
      float eb;
      {  // stage 2: 64 x 188 -> 32 x 375 (three-piece image in the place of the dead stage-0 / stage-1 inputs: its padding
         // columns are cleared here, every row) + the two samples at the cropped edge from the definition
        constexpr int PADC = B3_X3_C0 + (B3_X3_NC - L3 - B3_X3_C0);  // columns 0, 1 and 377 .. 399
        if (tid < PADC * 12) {
          const int k = tid % PADC, cp = tid / PADC;  // cp = piece * 4 + chunk
          const int col = k < B3_X3_C0 ? k : L3 + k;
          *reinterpret_cast<uint4*>(X3 + (cp >> 2) * QX3::PS + (cp & 3) * QX3::CHS + col * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
        // The two samples at the cropped edge (decoder2_edge_kernel, eqt.hip): lane = (channel, which sample), 64 input channels x
        // 3 pre-summed taps.  Every wave takes eight input channels -- weights requested here, used behind the conv -- and leaves
        // its partial sum in LDS (one wave doing all 64 kept the other seven at the barrier for 10 k cycles per row).
        constexpr int n0 = (L3 - 2 - 2) >> 1;
        // (Round 6: the 24 weights of a lane as eight 12-byte loads between the K-steps of the conv below -- as 24 4-byte loads in
        // front of it they held its MFMAs up, 192 requests per workgroup; and the edge sum reads its nine fragments as 16-byte
        // LDS words, not 72 2-byte ones.  Same products, same order of the sum: bit-identical.)
        struct F3 {
          float x, y, z;
        };
        F3 ew[8];
        eb = a.edge_b[d * 32 + (lane >> 1)];  // used behind the next barrier: requested here, or its wait there covers the operand loads too
        const F3* e = reinterpret_cast<const F3*>(a.edge_w + ((long)d * 64 * 64 + lane) * 3 + (long)(8 * wave_u) * 64 * 3);
        const ClipQuad st{X3, L3 - 2};
        {  // wave = (m-tile w % 4: phase mt / 2, channels 16 (mt % 2) ..; block w / 4 of six n-tiles), operand in registers
          const int mt2 = wave_u & 3, ph = mt2 >> 1, co0 = (mt2 & 1) * 16 + 4 * g, colb = (wave_u >> 2) * 96;
          float bias2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) bias2[r] = a.bs[2][d * 32 + co0 + r];
          b3c_mac_tile_pairs<64, B3_X2_NC, 3, 6>(b3c_lane_ptr<64, B3_X2_NC, 3>(X2, colb, lane), a2, [&](const int j, const f32x4 acc) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[r] + bias2[r], 0.f);
            st.quad(co0, 2 * (colb + j * 16 + n) + ph, v);
          }, [&](const int i) {  // 18 slots: a weight triple every other one
            if (i % 2 == 0 && i / 2 < 8) ew[i / 2] = e[(i / 2) * 64];
          });
        }
        D3_STAMP()  // 6: stage 2's MFMAs issued (this wave)
        float pacc = 0.f;
        {
          // x[k][u] = channel 8 w + u (chunk w) at column n0 + 1 + k: the sum of its three pieces, exact; a 16-byte word = the eight
          // channels of one piece at one column, channel 2 i in the low half of dword i
          float xk[3][8];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            uint4 q[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) q[pc] = *reinterpret_cast<const uint4*>(X2 + pc * QX2::PS + wave_u * QX2::CHS + (n0 + 1 + k) * 8);
            const unsigned* qh = reinterpret_cast<const unsigned*>(&q[0]);
            const unsigned* qm = reinterpret_cast<const unsigned*>(&q[1]);
            const unsigned* ql = reinterpret_cast<const unsigned*>(&q[2]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              xk[k][2 * i] = (bf16_lo(qh[i]) + bf16_lo(qm[i])) + bf16_lo(ql[i]);
              xk[k][2 * i + 1] = (bf16_hi(qh[i]) + bf16_hi(qm[i])) + bf16_hi(ql[i]);
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pacc = fmaf(ew[u].x, xk[0][u], pacc);
            pacc = fmaf(ew[u].y, xk[1][u], pacc);
            pacc = fmaf(ew[u].z, xk[2][u], pacc);
          }
        }
        EDGE[wave_u * 64 + lane] = pacc;
      }
      D3_STAMP()  // 7: stage 2 done (this wave)
      // stage 3's operand (m-tile mt23 = channels 8 mt23 .. + 7, both phases): 60 registers, so only now that stage 2's are free
      uint4 a3[B3Steps<32, 5>::STEPS * 3];
      float bias3[4];
      b3_load_a<32, 5>(a.af3[2] + d * a.af3_stride[2], mt23, lane, a3);
#pragma unroll
      for (int r = 0; r < 4; ++r) bias3[r] = a.bs[3][d * 32 + mt23 * 8 + 2 * g + (r >> 1)];
      // the next row's input and stage-0 operand: requested unconditionally (the last row asks for its own again) -- behind a
      // branch the number of loads in flight is unknown, and the wait for stage 3's operand would have to be for everything
      request(more ? next : row);
      load_stage0(nd);
      __syncthreads();
      D3_STAMP()  // 8: barrier
      if (blk23 == 1) {  // the four waves whose last n-tiles read the edge samples finish them, each for itself (same values)
        float acc = eb;
#pragma unroll
        for (int w8 = 0; w8 < D03_WAVES; ++w8) acc += EDGE[w8 * 64 + lane];
        unsigned short h, m, l;
        b3_split(fmaxf(acc, 0.f), h, m, l);
        const int ch = lane >> 1;
        bf16_t* q = X3 + (ch >> 3) * QX3::CHS + (L3 - 2 + (lane & 1) + B3_X3_C0) * 8 + (ch & 7);
        q[0] = h, q[QX3::PS] = m, q[2 * QX3::PS] = l;
      }
      {  // stage 3: 32 x 375 -> 32 x 750, straight to memory: rows (channel, phase), so a lane's accumulator pairs are
         // two consecutive samples of one channel (8-byte stores, full 128-byte lines per 16 lanes)
        float* yrow = a.y + (long)row * a.ws_y + HALO + (long)(mt23 * 8 + 2 * g) * a.ls_y;
        // one wait for the bias here, where every path passes: met first inside the stores' exec-masked blocks, it is waited
        // for again at the top of every such block -- together with the stores of the n-tile before
        asm volatile("" ::"v"(bias3[0]), "v"(bias3[1]), "v"(bias3[2]), "v"(bias3[3]));
        // The 24 n-tiles of an m-tile: 14 to the older wave of the SIMD (waves 0-3, whose MFMAs issue first), 10 to the younger
        // one, which also finishes the edge samples: with 12 + 12 the younger waves were 5 k cycles behind at the end of a row.
        // plan_flags[7] bit 11: the even split (A/B).
        auto stage3 = [&](const int colb, auto nb_tag) {
          constexpr int NB = decltype(nb_tag)::value;
          b3c_mac_tiles<32, B3_X3_NC, 5, NB>(b3c_lane_ptr<32, B3_X3_NC, 5>(X3, colb, lane), a3, [&](const int j, const f32x4 acc) {
            const int t = 2 * (colb + j * 16 + (lane & 15));
            if (t < L4) {
              *reinterpret_cast<float2*>(yrow + t) = make_float2(fmaxf(acc[0] + bias3[0], 0.f), fmaxf(acc[1] + bias3[1], 0.f));
              *reinterpret_cast<float2*>(yrow + a.ls_y + t) = make_float2(fmaxf(acc[2] + bias3[2], 0.f), fmaxf(acc[3] + bias3[3], 0.f));
            }
          });
        };
        if (a.even_split) {
          stage3(blk23 * 96, std::integral_constant<int, 6>{});
          stage3((blk23 + 2) * 96, std::integral_constant<int, 6>{});
        } else if (blk23 == 0) {
          stage3(0, std::integral_constant<int, 7>{});
          stage3(112, std::integral_constant<int, 7>{});
        } else {
          stage3(224, std::integral_constant<int, 5>{});
          stage3(304, std::integral_constant<int, 5>{});
        }
      }
      D3_STAMP()  // 9: stage 3 done (this wave)
      if (!more) break;
      row = next;
      d = nd;
      __syncthreads();  // the next row's input and stage-0 output land where stage 3 has just read
    }
#undef D3_STAMP
    return;
  }
  // image offsets through an opaque register (eqt_tail.hip: keeps the B fragments' addresses inside the DS immediates)
  int off1 = OFF1 / 4, off2 = OFF2 / 4, off3 = OFF3 / 4;
  asm volatile("" : "+v"(off1), "+v"(off2), "+v"(off3));
  float* X0 = lds + OFF0;
  float* X1 = lds + 4 * off1;
  float* X2 = lds + 4 * off2;
  float* X3 = lds + 4 * off3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int row = blockIdx.x;  // row = d * B + b
  if (row >= a.n_rows) return;
  // every column no stage ever writes is convolution padding: zero once, for all rows of this workgroup
  for (int i = tid; i < D03_LDS_FLOATS / 4; i += D03_NTH) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // the row's input: 16 x 47 samples, two per thread (rows of 47 + 1 slots), requested a row ahead
  float pre[2];
  auto request = [&](int r) {
    const float* src = a.x + (long)r * a.ws_x + HALO;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * D03_NTH, c = idx / 48, t = idx - c * 48;
      pre[k] = (idx < 16 * 48 && t < L0) ? src[(long)c * a.ls_x + t] : 0.f;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * D03_NTH, c = idx / 48, t = idx - c * 48;
      if (idx < 16 * 48) X0[c * S0 + BI + t] = pre[k];
    }
  };
  request(row);
  int d = row / a.B;
  float areg0[D_0::CB * D_0::TAPS], bias0[4];
  load_areg4<D_0>(a.af[0] + d * a.af_stride[0], wave_u, lane, areg0);
  load_biasreg<D_0>(a.bs[0] + d * 64, wave_u, lane, bias0);
  __syncthreads();  // the zero fill is complete before anybody parks or stores

  const int mt23 = wave_u & 3, blk23 = wave_u >> 2;
  while (true) {
    const int next = row + gridDim.x;
    const bool more = next < a.n_rows;
    const int nd = more ? next / a.B : d;
    park();
    float areg1[D_1::CB * D_1::TAPS], bias1[4];
    load_areg4<D_1>(a.af[1] + d * a.af_stride[1], wave_u, lane, areg1);
    load_biasreg<D_1>(a.bs[1] + d * 64, wave_u, lane, bias1);
    __syncthreads();
    {  // stage 0: 16 x 47 -> 64 x 94
      RowStore<S1> st{X1 + BI, (unsigned)L1};
      conv_lds_areg<D_0, S0, BI, S0, BI>(X0, X0, areg0, bias0, wave_u, C0, st, 0, 1, lane);
    }
    float areg2[D_2::CB * D_2::TAPS], bias2[4];
    load_areg4<D_2>(a.af[2] + d * a.af_stride[2], mt23, lane, areg2);
    load_biasreg<D_2>(a.bs[2] + d * 32, mt23, lane, bias2);
    __syncthreads();
    {  // stage 1: 64 x 94 -> 64 x 188
      RowStore<S2> st{X2 + BI, (unsigned)L2};
      conv_lds_areg<D_1, S1, BI, S1, BI>(X1, X1, areg1, bias1, wave_u, C1, st, 0, 1, lane);
    }
    float areg3[D_3::CB * D_3::TAPS], bias3[4];
    load_areg4<D_3>(a.af[3] + d * a.af_stride[3], mt23, lane, areg3);
    load_biasreg<D_3>(a.bs[3] + d * 32, mt23, lane, bias3);
    __syncthreads();
    {  // stage 2: 64 x 188 -> 32 x 375; the folded filter is wrong for the two samples at the cropped edge (373, 374)
      ClipStore<S3> st{X3 + BI, L3 - 2};
      conv_lds_areg<D_2, S2, BI, S2, BI>(X2, X2, areg2, bias2, mt23, C2, st, blk23, 2, lane);
      if (wave_u == D03_WAVES - 1) {  // ... which one wave computes from the definition (decoder2_edge_kernel, eqt.hip)
        constexpr int n0 = (L3 - 2 - 2) >> 1;  // first stage-1 sample the two outputs see (185)
        const float* e = a.edge_w + ((long)d * 64 * 64 + lane) * 3;
        float acc = a.edge_b[d * 32 + (lane >> 1)];
#pragma unroll 8
        for (int ci = 0; ci < 64; ++ci) {
          const float* ec = e + (long)ci * 64 * 3;
          const float* xs = X2 + ci * S2 + BI + n0;
          acc = fmaf(ec[0], xs[0], acc);
          acc = fmaf(ec[1], xs[1], acc);
          acc = fmaf(ec[2], xs[2], acc);
        }
        X3[(lane >> 1) * S3 + BI + L3 - 2 + (lane & 1)] = fmaxf(acc, 0.f);
      }
    }
    if (more) {
      request(next);
      load_areg4<D_0>(a.af[0] + nd * a.af_stride[0], wave_u, lane, areg0);
      load_biasreg<D_0>(a.bs[0] + nd * 64, wave_u, lane, bias0);
    }
    __syncthreads();
    {  // stage 3: 32 x 375 -> 32 x 750, straight to memory
      RowOut st{a.y + (long)row * a.ws_y + HALO, a.ls_y};
      conv_lds_areg<D_3, S3, BI, S3, BI>(X3, X3, areg3, bias3, mt23, C3, st, blk23, 2, lane);
    }
    if (!more) break;
    row = next;
    d = nd;
    // no barrier here: the next row's park writes X0 and its stage 0 writes X1, which nobody still reads; stage 3's
    // readers of X3 are two barriers ahead of the next writer of X3
  }
}

}  // namespace

// Replaces the steps "decoder.0", "decoder.1", "decoder.2", "decoder.2.edge", "decoder.3" of the plan by one fused step.
int plan_eqt_fuse_dec03(Net& net, bool b3) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "decoder.0") first = (int)i;
  if (first < 0 || first + 5 > (int)net.steps.size() || net.steps[first + 3].name != "decoder.2.edge" ||
      net.steps[first + 4].name != "decoder.3") {
    set_error("fused decoder stages 0-3: layer plan not found");
    return VP_ERR_INVALID;
  }
  ConvLayer* c[4] = {nullptr, nullptr, nullptr, nullptr};
  for (auto& l : net.convs)
    for (int i = 0; i < 4; ++i)
      if (l->name == "decoder." + std::to_string(i)) c[i] = l.get();
  HostBlob* ew = net.named.count("decoder.2.edge.w") ? net.named["decoder.2.edge.w"] : nullptr;
  HostBlob* eb = net.named.count("decoder.2.edge.b") ? net.named["decoder.2.edge.b"] : nullptr;
  if (!c[0] || !c[1] || !c[2] || !c[3] || !ew || !eb || c[0]->n_sets != 3) {
    set_error("fused decoder stages 0-3: conv layers missing");
    return VP_ERR_INVALID;
  }
  HostBlob* q[4];
  for (int i = 0; i < 4; ++i) q[i] = c[i]->afrag_q4 ? c[i]->afrag_q4 : net.add_blob(regroup_afrag4(*c[i]));
  HostBlob* p3[3] = {nullptr, nullptr, nullptr};
  if (b3) {  // stage 0 with its rows regrouped (phase, channel), stages 1-3 as three-piece operands
    q[0] = net.add_blob(regroup_afrag4_phase_major(*c[0]));
    for (int i = 0; i < 3; ++i) p3[i] = net.add_blob(b3_operand(*c[1 + i], i < 2));
  }
  const int x_in = c[0]->src1, y_out = c[3]->dst;
  for (int i = 0; i < 3; ++i) net.tensor_sets[c[i]->dst] = 0;  // stages 0-2 live in LDS under this plan
  Step st;
  st.name = "fused.dec03 (decoder.0-3, one row per workgroup)";
  st.flops_per_window = 0;
  for (int i = 0; i < 5; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  // issued MFMA work per row: 8 m-tiles x 3 n-tiles x 12 K-steps, 8 x 6 x 48, 4 x 12 x 48, 4 x 24 x 40 (2048 FLOP each)
  // (B3: stages 1-3 as six-MFMA groups over K = 32: an eighth of the fp32 K-steps; the two edge samples of stage 2: VALU, not counted)
  if (b3)
    st.set_issued(3.0 * 8.0 * 3 * 12 * 2048.0, 3.0 * (8.0 * 6 * 6 + 4.0 * 12 * 6 + 4.0 * 24 * 5) * 6 * 16384.0, 0.0);
  else
    st.set_issued(3.0 * (8.0 * 3 * 12 + 8.0 * 6 * 48 + 4.0 * 12 * 48 + 4.0 * 24 * 40) * 2048.0, 0.0, 0.0);
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    Dec03Args a{};
    const Tensor &tx = n.tensors[x_in], &ty = n.tensors[y_out];
    a.x = tx.p;
    a.ls_x = tx.ls;
    a.ws_x = (long)tx.win_stride();
    a.y = ty.p;
    a.ls_y = ty.ls;
    a.ws_y = (long)ty.win_stride();
    for (int i = 0; i < 4; ++i) {
      a.af[i] = q[i]->d;
      a.bs[i] = c[i]->bias.d;
      a.af_stride[i] = (long)(c[i]->afrag.h.size() / 3);
    }
    a.edge_w = ew->d;
    a.edge_b = eb->d;
    a.B = B;
    a.n_rows = 3 * B;
    a.even_split = (n.cfg.plan_flags[7] >> 11) & 1;
    a.clk = (n.debug_clock && n.debug_clock->d) ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) + (size_t)n.max_batch * 32 : nullptr;
    const int grid = a.n_rows < 256 ? a.n_rows : 256;
    if (b3) {
      for (int i = 0; i < 3; ++i) {
        a.af3[i] = reinterpret_cast<const uint4*>(p3[i]->d);
        a.af3_stride[i] = (long)(p3[i]->h.size() / 3 / 4);
      }
      hipLaunchKernelGGL(eqt_dec03_kernel<true>, dim3(grid), dim3(D03_NTH), B3_LDS_BYTES, s, a);
    } else {
      hipLaunchKernelGGL(eqt_dec03_kernel<false>, dim3(grid), dim3(D03_NTH), D03_LDS_FLOATS * sizeof(float), s, a);
    }
    return 0;
  };
  if (b3)
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_dec03_kernel<true>), (size_t)B3_LDS_BYTES});
  else
    net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_dec03_kernel<false>), D03_LDS_FLOATS * sizeof(float)});
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 5);
  net.steps.insert(net.steps.begin() + first, std::move(st));
  return VP_OK;
}

}  // namespace vp
