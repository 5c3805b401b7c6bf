// EQTransformer decoder tail (stages 4, 5, 6 + the sigmoid heads) on the bf16 matrix cores with exact three-piece
// operands (conv_b3.h): same time tiling as eqt_tail.hip -- a tile's rows never leave the CU, the halo every stage needs
// is recomputed -- but every product is six v_mfma_f32_16x16x32_bf16 over (hi, mid, lo) pieces instead of eight fp32
// MFMAs: 6 / 16 of the matrix time at fp32 accuracy (what is dropped is below the rounding of one fp32 product).
//
//   tile of outputs [t0, t0 + 1200) of the 6000-sample row, t0 = 1200 j (five tiles per row: 15 tiles per CU at 256 windows):
//     stage 6 columns n6 = t0/2 - 3 + [0, 606 / 640)  reads stage-5 samples [n6 - 3, n6 + 3]  (K = 11 folded to 7 taps)
//     stage 5 columns n5 = t0/4 - 3 + [0, 306 / 320)  reads stage-4 samples [n5 - 2, n5 + 2]  (K = 9 folded to 5 taps)
//     stage 4 columns n4 = t0/8 - 3 + [0, 156 / 160)  reads stage-3 samples [n4 - 2, n4 + 2]  (K = 7 folded to 5 taps)
//   (needed / computed: every wave takes one block of n-tiles of one m-tile per stage: 3 or 2 / 5 / 5)
//
// LDS (157 KB).  Three-piece images take 6 bytes per value against the 4 of fp32 rows, so a 2000-sample tile no longer
// fits beside its staging: 1200-sample tiles, chunk-plane images [piece][8-channel chunk][column][8 channels] (conv_b3.h), and
// two regions used twice per tile --
//     R0 (63 KB): the stage-3 image (32 ch x 208 columns), then the stage-5 output = stage-6 input (16 ch x 656 columns)
//     R1 (62 KB): the stage-4 output = stage-5 input (16 ch x 384 columns), then the stage-6 output staged for the heads
//     + the operands of stages 5 and 6 (18 + 12 KB) and the heads' tap table (2 KB)
// Nothing is zero-filled per tile.  Samples outside a row's signal are WRITTEN as zeros by the producing stage; a column
// a stage reads beyond what its producer wrote this tile (left-overs of the image that lay there before) only ever feeds
// outputs nobody keeps: everything a KEPT output touches -- the zero-weight padded taps of stages 5 / 6 and of the heads
// included, 0 x NaN being NaN -- is computed from this tile's own stage-3 samples (static_assert below; a non-finite
// window must not reach its neighbour: tests/test_gpu_nonfinite.py).
//
// Weights.  16-channel layers pack two taps into the K = 32 of one instruction (conv_b3.h: B3Steps).  The stage-4 operand
// (15 16-byte registers per lane) stays in registers over all tiles of a workgroup, those of stages 5 and 6 rest in
// LDS and are read into registers at the start of their stage (all three in registers spilled: the reloads, counted
// by vmcnt behind the next tile's stage-3 loads, exposed the latency of those); everything is reloaded when the decoder
// changes (twice per workgroup at most).  The heads (8 -> 1, k = 11) are the Toeplitz product of eqt_tail.hip with K
// ordered (tap, channel): a B fragment = the 8 channels of one staged sample (two 8-byte reads: staging [channel quad]
// [sample % 16][sample / 16], odd row pitch: the stores of stage 6 are conflict-free, the reads 2-way), an A fragment =
// w[.][tap - m] or zero: a 43-entry table per piece in LDS, read with a per-lane offset.
//
// Where the time goes (tools/tail_clock.py; 15 tiles of 19.6 k cycles per workgroup at 256 windows): stage 4 / 5 / 6
// 4.3 / 5.2 / 6.6 k cycles for 2.9 / 2.9 / 3.8 k of MFMA issue per SIMD, heads 2.4 k, conversion of the next tile's stage-3
// rows 1.1 k.  The older wave of a SIMD runs its chain of MFMAs first; what the younger one adds behind it, and every
// epilogue (bias, ReLU, split into pieces, stores: 40 vector instructions per n-tile, an MFMA leaves 8 of its 16 cycles to
// them), is what a stage costs beyond its MFMAs.  Bank conflicts are not: three image layouts measured the same (conv_b3.h).
//
// Plan flag plan_flags[7] & 64 keeps the fp32-MFMA kernel of eqt_tail.hip (bit-identical to the layer launches); this one
// agrees with it to fp32 rounding (tests/test_gpu_eqt.py).
#include "conv_b3.h"
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

constexpr int T_OUT = 6000;
constexpr int T3_NTH = 512, T3_WAVES = 8;
constexpr int HSB = 81, OUT_QS = 16 * HSB, OUT_PS = 2 * OUT_QS;  // heads' staging: 8-byte units per row / quad plane / piece
constexpr int HT_N = 43;                                         // head table entries: k = -15 .. 27
// Two tilings of a row.  TW = 1200: five tiles cover the 6000 samples (the form of round 2).  TW = 1256: FOUR tiles cover the
// 5004 samples [496, 5500) that are left when the caller blinds 500 samples at either end of every window (README.md:58,
// BASELINE configs[2]) -- the blinded samples are never stacked, so their tiles are not computed: the launch walks the
// output range [t_lo, t_hi) it is given, in whichever tiling needs fewer tiles (plan_eqt_fuse_tail_b3).  The wider tile
// fits the same 157 KB: stage 6 still computes 640 columns (needs 634), stage 5 still 320 (needs exactly 320: rounds 3-5 used
// TW = 1264, which needs 322 -- one n-tile more for every wave of stage 5, 17 % of that stage, for eight samples nobody keeps),
// stage 4 one n-tile more for the younger waves (192 columns for 163 needed).  TW is a multiple of 8, not of 16: the heads' last
// 16-sample block of a tile is half used.  The zero-weight padded taps of stage 6 reach two columns past what stage 5 writes:
// those columns are zeroed per tile (0 x stale NaN would be NaN).
template <int TW_>
struct T3 {
  static constexpr int TW = TW_;
  // n-tiles per wave and stage: the first four waves of a workgroup (one per SIMD) are the OLDER wave of their SIMD, whose MFMAs
  // issue first; with equal shares the younger wave was still a quarter of a stage behind when the older one was through
  // (tools/tail_clock.py).  TW = 1200: stage 4 needs 10 n-tiles per phase: 3 to the older, 2 to the younger wave; stages 5 and 6
  // measured no faster with 6 + 4 than with 5 + 5
  static constexpr int NB4O = 3, NB4Y = TW_ > 1200 ? 3 : 2, NB5 = 5, NB6 = 5;
  static constexpr int C4 = 2 * (NB4O + NB4Y) * 16, C5 = 4 * NB5 * 16, C6 = T3_WAVES * NB6 * 16;  // columns computed per stage
  // stage-3 samples a tile parks: what its kept outputs need + the one more that reaches them through the zero-weight padded
  // taps of stages 5 / 6 and the heads (0 x stale non-finite = NaN)
  static constexpr int PARK_COLS = TW_ > 1200 ? 176 : 168;
  static constexpr int NC4 = 208, NC5 = (2 * C4 > C5 + 16 ? 2 * C4 : C5 + 16), NC6 = C6 + 16;  // image columns (a place for every column a stage reads or writes)
  static constexpr int BLOCKS = (TW_ + 15) / 16;  // the heads' 16-sample blocks of a tile (the last one half used where TW % 16 = 8)
  using Q4 = B3Chunk<32, NC4>;                                    // chunk-plane images (conv_b3.h)
  using Q5 = B3Chunk<16, NC5>;
  using Q6 = B3Chunk<16, NC6>;
  static constexpr bool GUARD5 = 2 * C5 > NC6;  // stage 5 computes columns beyond the stage-6 image: those stores are skipped
  // columns of the stage-6 input image behind what stage 5 writes that a kept output reaches through zero-weight taps: zeroed per tile
  static constexpr bool ZERO6 = (TW_ + 12) / 2 + 7 >= 2 * C5;
  static constexpr int R0_BYTES = 3 * Q6::PS * 2, R1_BYTES = 3 * OUT_PS * 8;
  static constexpr int A5_N = 2 * B3Steps<16, 5>::STEPS * 3 * 64, A6_N = B3Steps<16, 7>::STEPS * 3 * 64;  // uint4: operands of stages 5 and 6
  static constexpr int OFF_R1 = R0_BYTES, OFF_HT = OFF_R1 + R1_BYTES, OFF_A5 = OFF_HT + 3 * HT_N * 16 + 48, OFF_A6 = OFF_A5 + A5_N * 16;
  static constexpr int LDS_BYTES = OFF_A6 + A6_N * 16;
  static_assert(OFF_A5 % 16 == 0, "16-byte fragments");
  static_assert(3 * Q4::PS * 2 <= R0_BYTES && 3 * Q5::PS * 2 <= R1_BYTES && LDS_BYTES <= 160 * 1024 && OFF_R1 % 16 == 0 && OFF_HT % 16 == 0,
                "LDS budget");
  static_assert(NC4 >= PARK_COLS && NC5 >= C5 + 6 && NC6 >= C6 + 8 && NC5 >= 2 * C4 && (GUARD5 || NC6 >= 2 * C5) && HSB % 2 == 1,
                "every column a stage reads or writes has a place");
  static_assert(TW % 8 == 0, "tile grid: 16-byte stores of the heads, whole stage-3 samples");
  // what a tile needs (file comment) is computed, and what is computed has a place
  static_assert(TW + 11 <= 2 * C6 && C6 / 8 <= HSB - 1 && (BLOCKS + 15) / 16 <= T3_WAVES, "stage 6 / heads");
  static_assert((TW + 12) / 2 + 7 < NC6 && (!ZERO6 || 2 * C5 % 2 == 0), "the padded taps of stage 6 stay inside its input image");
  static_assert(TW / 2 + 6 + 6 <= 2 * C5 && TW / 2 + 6 + 6 <= NC6 && TW / 4 + 6 + 5 <= 2 * C4 && TW / 8 + 6 + 4 <= PARK_COLS, "halo chain");
  // ... and everything a kept output touches, zero-weight taps included, is this tile's data: heads staged t <= TW + 12 -> stage-6
  // column <= (TW + 12) / 2 -> stage-5 sample + 7 -> stage-5 column -> stage-4 sample + 1 + 5 -> stage-4 column -> image column + 4
  static_assert(((((TW + 12) / 2 + 7) / 2 + 1 + 5) / 2 + 4) < PARK_COLS, "closure of the kept outputs under the padded taps");
  static_assert(PARK_COLS * 8 <= 3 * T3_NTH && PARK_COLS * 8 > 2 * T3_NTH, "three (four channels x one sample) items per thread");
  // six-MFMA groups a tile issues
  static constexpr double GROUPS = 4.0 * (NB4O + NB4Y) * 5 + 8.0 * NB5 * 3 + 8.0 * NB6 * 4 + 5.0 * 7;
};

struct Tail3Args {
  const float* x3;  // stage-3 rows [3 B][32][ls]
  int ls3;
  long ws3;
  float* y;         // dense (B, 3, 6000)
  const uint4 *af4, *af5, *af6;  // three-piece operands [set][MT][STEPS][piece][64] (net.hip: b3_operand, rows (phase, channel))
  long af4_stride, af5_stride, af6_stride;  // uint4 per set
  const float *bs4, *bs5, *bs6;  // bias [set][COUT]
  const uint4* head_t;           // [3][piece][HT_N]: the 8 channels' w[.][k] as bf16 pieces, k = entry - 15 (zero outside 0 .. 10)
  const float* head_b;           // [3]
  const float* flags;            // optional [B]: != 0 where the window held a non-finite sample: its predictions are written as NaN
  int B, n_tiles;
  int t_lo, tiles_per_row;       // tile j of a row = outputs [t_lo + j TW, t_lo + (j + 1) TW): t_lo a multiple of 16, the tiles cover what the caller keeps
  unsigned long long* clk;  // debug (plan flag plan_flags[1] & 2): the stamps of eqt_tail.hip's TailArgs::clk, same slots
};

struct Tile3 {
  int d, win, t0;
};
template <int TW>
__device__ __forceinline__ Tile3 tile3_id(const int tile, const Tail3Args& a) {
  const int row = tile / a.tiles_per_row, j = tile - row * a.tiles_per_row;  // row = d * B + b
  return Tile3{row / a.B, row, a.t_lo + j * TW};
}

// relu(acc + bias) of the lane's four rows; zero outside the row's signal
__device__ __forceinline__ void t3_finish(const f32x4 acc, const float (&bias)[4], const bool in, float (&v)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(acc[r] + bias[r], 0.f) : 0.f;
}

template <int TW_>
__global__ __launch_bounds__(T3_NTH) void eqt_tail3_kernel(const Tail3Args a) {
  using K = T3<TW_>;
  using Q5 = typename K::Q5;
  using Q6 = typename K::Q6;
  constexpr int TW = K::TW, NB4O = K::NB4O, NB4Y = K::NB4Y, NB5 = K::NB5, NB6 = K::NB6, PARK_COLS = K::PARK_COLS, NC4 = K::NC4,
                NC5 = K::NC5, NC6 = K::NC6, C5 = K::C5, BLOCKS = K::BLOCKS, OFF_R1 = K::OFF_R1, OFF_HT = K::OFF_HT, OFF_A5 = K::OFF_A5, OFF_A6 = K::OFF_A6,
                A5_N = K::A5_N, A6_N = K::A6_N, T3_LDS_BYTES = K::LDS_BYTES;
  (void)sizeof(Q5), (void)sizeof(Q6);
  extern __shared__ uint4 t3_lds[];
  char* base = reinterpret_cast<char*>(t3_lds);
  bf16_t* IN4 = reinterpret_cast<bf16_t*>(base);            // R0
  bf16_t* IN6 = IN4;
  bf16_t* IN5 = reinterpret_cast<bf16_t*>(base + OFF_R1);   // R1
  uint2* OUT6 = reinterpret_cast<uint2*>(base + OFF_R1);
  uint4* HT = reinterpret_cast<uint4*>(base + OFF_HT);
  uint4* A5 = reinterpret_cast<uint4*>(base + OFF_A5);
  uint4* A6 = reinterpret_cast<uint4*>(base + OFF_A6);
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile = blockIdx.x;
  if (tile >= a.n_tiles) return;
  for (int i = tid; i < T3_LDS_BYTES / 16; i += T3_NTH) t3_lds[i] = make_uint4(0u, 0u, 0u, 0u);
  Tile3 id = tile3_id<TW>(tile, a);
  unsigned long long* clk = (a.clk && tid == 0 && (int)blockIdx.x < a.B) ? a.clk + (long)blockIdx.x * 32 : nullptr;
  int n_done = 0;
#define T3_STAMP(k) \
  if (clk && n_done < 4) clk[n_done * 6 + (k)] = __builtin_readcyclecounter();
  if (clk) clk[30] = __builtin_amdgcn_s_memrealtime();
  if (clk) clk[24] = __builtin_readcyclecounter();  // [24], [25]: shader clock at the kernel's two ends ([30], [31]: the 100 MHz clock)

  // stage-3 samples of a tile: image column x <-> sample t0/8 - 5 + x of the row; an item = four channels of one sample
  float pre[3][4];
  unsigned pre_off[3];  // per-lane part of an item's address; the row of channel r is a uniform step on the base
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int item = tid + k * T3_NTH, cq = item / PARK_COLS, x = item - cq * PARK_COLS;
    pre_off[k] = (unsigned)(4 * cq * a.ls3 + x);
  }
  auto request = [&](const Tile3& t) {
    const float* src = a.x3 + (long)t.win * a.ws3 + (HALO - 5 + t.t0 / 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = src + (long)r * a.ls3;
#pragma unroll
      for (int k = 0; k < 3; ++k) pre[k][r] = (k < 2 || tid + 2 * T3_NTH < 8 * PARK_COLS) ? row[pre_off[k]] : 0.f;
    }
  };
  // the same request in twelve parts, one between the K-steps of stage 6 each: as one burst in front of the stage its twelve loads
  // per wave held the stage up by ~0.9 k cycles (a wave issues in order; the same 0.9 k wherever the burst stands: tools/
  // tail_clock.py, profiles/r06_g_eqt_tail_probes.txt), spread they cost 0.6 k
  auto request_part = [&](const Tile3& t, const int q) {
    const int r = q / 3, k = q - 3 * r;
    const float* row = a.x3 + (long)t.win * a.ws3 + (HALO - 5 + t.t0 / 8) + (long)r * a.ls3;
    pre[k][r] = (k < 2 || tid + 2 * T3_NTH < 8 * PARK_COLS) ? row[pre_off[k]] : 0.f;
  };
  auto park = [&]() {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int item = tid + k * T3_NTH, cq = item / PARK_COLS, x = item - cq * PARK_COLS;
      if (item < 8 * PARK_COLS) b3c_store4<32, NC4>(IN4, x, cq, pre[k]);
    }
  };
  request(id);

  uint4 a4[B3Steps<32, 5>::STEPS * 3];
  float bias4[4], bias5[4], bias6[4], bh;
  auto load_weights = [&](const int d) {
    b3_load_a<32, 5>(a.af4 + d * a.af4_stride, w & 1, lane, a4);
    for (int i = tid; i < A5_N; i += T3_NTH) A5[i] = a.af5[d * a.af5_stride + i];
    for (int i = tid; i < A6_N; i += T3_NTH) A6[i] = a.af6[d * a.af6_stride + i];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias4[r] = a.bs4[d * 16 + 4 * g + r];
      bias5[r] = a.bs5[d * 16 + 4 * g + r];
      bias6[r] = a.bs6[d * 8 + 4 * (g & 1) + r];
    }
    bh = a.head_b[d];
    if (tid < 3 * HT_N) HT[tid] = a.head_t[d * 3 * HT_N + tid];
  };
  __syncthreads();  // the zero fill is through before the table lands
  load_weights(id.d);

  while (true) {
    const int next = tile + gridDim.x;
    const bool more = next < a.n_tiles;
    const Tile3 nid = more ? tile3_id<TW>(next, a) : id;
    const int t0 = id.t0;
    // the window's non-finite flag, read HERE: behind the heads' MFMAs its trip to L2 stood between the last MFMA and the stores
    const float win_flag = a.flags ? a.flags[id.win - id.d * a.B] : 0.f;
    T3_STAMP(0)
    park();
    __syncthreads();
    T3_STAMP(1)
    {  // stage 4: column c, phase p -> stage-4 sample t0/4 - 6 + t, t = 2 c + p: column t of the stage-5 input image
      const bool older = w < 4;  // uniform
      const int ph = w & 1, colb = older ? (w >> 1) * (NB4O * 16) : 2 * NB4O * 16 + ((w - 4) >> 1) * (NB4Y * 16);
      const int lo = 6 - t0 / 4;
      auto finish = [&](const int j, const f32x4 acc) {
        const int t = 2 * (colb + j * 16 + n) + ph;
        float v[4];
        t3_finish(acc, bias4, (unsigned)(t - lo) < 1500u, v);
        b3c_store4<16, NC5>(IN5, t, g, v);
      };
      if (older) b3c_mac_tiles<32, NC4, 5, NB4O>(b3c_lane_ptr<32, NC4, 5>(IN4, colb, lane), a4, finish);
      else b3c_mac_tiles<32, NC4, 5, NB4Y>(b3c_lane_ptr<32, NC4, 5>(IN4, colb, lane), a4, finish);
    }
    __syncthreads();
    T3_STAMP(2)
    {  // stage 5: column c reads the image columns c + 1 + tap; output t = 2 c + p -> column t of the stage-6 input image
      const int ph = w & 1, colb = (w >> 1) * (NB5 * 16);
      uint4 a5[B3Steps<16, 5>::STEPS * 3];  // this wave's operand: LDS -> registers for the stage
#pragma unroll
      for (int i = 0; i < B3Steps<16, 5>::STEPS * 3; ++i) a5[i] = A5[ph * (A5_N / 2) + i * 64 + lane];
      const int lo = 6 - t0 / 2;
      auto finish = [&](const int j, const f32x4 acc) {
        const int t = 2 * (colb + j * 16 + n) + ph;
        float v[4];
        t3_finish(acc, bias5, (unsigned)(t - lo) < 3000u, v);
        if (!K::GUARD5 || t < NC6) b3c_store4<16, NC6>(IN6, t, g, v);  // (tilings whose stage 5 computes columns beyond the stage-6 image: nobody's input)
      };
      if (K::ZERO6) {  // columns [2 C5, NC6) of the stage-6 input image: 16-byte units (column, 8 channels) of every chunk plane
        constexpr int ZC = NC6 - 2 * C5, UNITS = 3 * 2 * ZC;
        for (int i = tid; i < UNITS; i += T3_NTH) {
          const int plane = i / ZC, col = 2 * C5 + (i - plane * ZC);  // plane = piece * 2 + chunk
          *reinterpret_cast<uint4*>(IN6 + (plane >> 1) * Q6::PS + (plane & 1) * Q6::CHS + col * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      b3c_mac_tiles<16, NC5, 5, NB5>(b3c_lane_ptr<16, NC5, 5>(IN5, colb + 1, lane), a5, finish);
    }
    __syncthreads();
    T3_STAMP(3)
    {  // stage 6: column c reads the image columns c + tap; one m-tile: lane group g holds phase g / 2, channels 4 (g % 2) ..;
       // output t = 2 c + p = sample t0 - 6 + t of the row -> 8-byte unit (t % 16) * HSB + t / 16 of its quad's staging plane
      const int colb = w * (NB6 * 16), ph = g >> 1;
      uint4 a6[B3Steps<16, 7>::STEPS * 3];
#pragma unroll
      for (int i = 0; i < B3Steps<16, 7>::STEPS * 3; ++i) a6[i] = A6[i * 64 + lane];
      const int lo = 6 - t0;
      // t = 2 colb + 32 j + (2 n + p): t % 16 is the lane's, t / 16 = colb / 8 + 2 j + n / 8
      uint2* q = OUT6 + (g & 1) * OUT_QS + ((2 * n + ph) & 15) * HSB + (n >> 3) + colb / 8;
      auto finish = [&](const int j, const f32x4 acc) {
        const int t = 2 * (colb + j * 16 + n) + ph;
        float v[4];
        t3_finish(acc, bias6, (unsigned)(t - lo) < (unsigned)T_OUT, v);
        if (B3_EXP & 16) {
          asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          return;
        }
        const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
        const float r0 = v[0] - bf16_lo(h0), r1 = v[1] - bf16_hi(h0), r2 = v[2] - bf16_lo(h1), r3 = v[3] - bf16_hi(h1);
        const unsigned m0 = pack_bf16x2(r0, r1), m1 = pack_bf16x2(r2, r3);
        const unsigned l0 = pack_bf16x2(r0 - bf16_lo(m0), r1 - bf16_hi(m0)), l1 = pack_bf16x2(r2 - bf16_lo(m1), r3 - bf16_hi(m1));
        if (B3_EXP & 4) {
          asm volatile("" ::"v"(h0), "v"(h1), "v"(m0), "v"(m1), "v"(l0), "v"(l1), "v"(q));
          return;
        }
        q[2 * j] = make_uint2(h0, h1);
        q[2 * j + OUT_PS] = make_uint2(m0, m1);
        q[2 * j + 2 * OUT_PS] = make_uint2(l0, l1);
      };
      // (the next tile's stage-3 rows travel under stage 6 and the heads; the last tile asks for its own again)
      b3c_mac_tiles<16, NC6, 7, NB6>(b3c_lane_ptr<16, NC6, 7>(IN6, colb, lane), a6, finish, [&](const int i) {
        if (i < 12) request_part(nid, i);
      });
    }
    __syncthreads();
    T3_STAMP(4)
    if (w < (BLOCKS + 15) / 16) {
      // heads: wave w owns the 16-sample blocks 16 w .. 16 w + 15 of the tile: y[t0 + 16 blk + m] = b + sum_tap sum_ci
      // w[ci][tap - m] x_ci[staged 16 blk + tap + 1]; K-step s = taps 4 s .. 4 s + 3, lane group g the tap 4 s + g
      f32x4 acc = {bh, bh, bh, bh};
      const uint2* bp = OUT6 + (16 * w + n);
      const uint4* ap = HT + (g - n + 15);
      constexpr int HD = 1, HB = HD + 1;  // K-steps of operands in flight ahead of the MFMAs (2 / 3 measured the same: 2.09 / 2.14 k cycles)
      uint4 av[HB][3], bv[HB][3];
      auto load_ab = [&](const int s, uint4 (&aa)[3], uint4 (&bb)[3]) {
        const int e = 4 * s + g + 1, chunk = (e & 15) * HSB + (e >> 4);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          aa[pc] = ap[pc * HT_N + 4 * s];
          const uint2 c0 = bp[pc * OUT_PS + chunk], c1 = bp[pc * OUT_PS + OUT_QS + chunk];  // channels 0-3, 4-7
          bb[pc] = make_uint4(c0.x, c0.y, c1.x, c1.y);
        }
      };
#pragma unroll
      for (int s = 0; s < HD && s < 7; ++s) load_ab(s, av[s % HB], bv[s % HB]);
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        if (s + HD < 7) load_ab(s + HD, av[(s + HD) % HB], bv[(s + HD) % HB]);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_b3, av[s % HB][WP[t]]),
                                                       __builtin_bit_cast(bf16x8_b3, bv[s % HB][XP[t]]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int blk = 16 * w + n;
      if (16 * blk + 4 * g < TW && t0 + 16 * blk + 4 * g < T_OUT) {  // (a tile's last block may be half used; the last tile of a row may reach past it)
        const int b = id.win - id.d * a.B;
        float4 r;
        // sigmoid as v_exp + v_rcp (3e-7 absolute; the library expf and the IEEE division are ~18 instructions per value on a
        // path that is bound by issued instructions)
        r.x = __builtin_amdgcn_rcpf(1.f + __expf(-acc[0]));
        r.y = __builtin_amdgcn_rcpf(1.f + __expf(-acc[1]));
        r.z = __builtin_amdgcn_rcpf(1.f + __expf(-acc[2]));
        r.w = __builtin_amdgcn_rcpf(1.f + __expf(-acc[3]));
        if (win_flag != 0.f) r.x = r.y = r.z = r.w = __builtin_nanf("");  // what launch_poison writes (prepost.h)
        *reinterpret_cast<float4*>(a.y + ((long)b * 3 + id.d) * T_OUT + t0 + 16 * blk + 4 * g) = r;
      }
    }
    T3_STAMP(5)
    ++n_done;
    if (!more) break;
    if (nid.d != id.d) {  // uniform over the workgroup
      __syncthreads();    // every wave is through with the head table
      load_weights(nid.d);
    }
    tile = next;
    id = nid;
    // no barrier here: the next tile's stage-3 image lands in R0, which nobody has read since the barrier behind stage 6
  }
  if (clk) clk[25] = __builtin_readcyclecounter();
  if (clk) clk[31] = __builtin_amdgcn_s_memrealtime();
#undef T3_STAMP
}

// The tiling of one launch, a pure function of the output range the caller keeps (annotate / classify: [blind_l, T - blind_r);
// model(x): the whole row, out_hi <= 0) -- the form that needs fewer tiles; plan_flags[7] bit 10 computes the whole row
// whatever the blinding (A/B, tests).  issued_bf16: matrix work per WINDOW, groups of six bf16 MFMAs (one 16 x 16 x 32
// fp32-accurate product each) over the three decoders' tiles.
struct Tail3Tiling {
  int t_lo, tiles_per_row;
  bool wide;
  double issued_bf16;
};
Tail3Tiling tail3_tiling(const vp_config& cfg, int out_lo, int out_hi) {
  const bool whole = (cfg.plan_flags[7] & 1024) || out_hi <= 0;
  const int t_lo = whole ? 0 : (out_lo / 16) * 16, t_hi = whole ? T_OUT : out_hi;
  const int tiles_a = (t_hi - t_lo + 1199) / 1200, tiles_b = (t_hi - t_lo + 1255) / 1256;
  Tail3Tiling t;
  t.wide = tiles_b < tiles_a;
  t.t_lo = t_lo;
  t.tiles_per_row = t.wide ? tiles_b : tiles_a;
  t.issued_bf16 = 3.0 * t.tiles_per_row * (t.wide ? T3<1256>::GROUPS : T3<1200>::GROUPS) * 6 * 16384.0;
  return t;
}

}  // namespace

// Replaces the steps "decoder.4", "decoder.5", "decoder.6+heads" of the plan by one fused step (bf16-piece form).
int plan_eqt_fuse_tail_b3(Net& net) {
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
  if (!c4 || !c5 || !c6 || c4->n_sets != 3 || c4->g.cinp() != 32 || c5->g.cinp() != 16 || c6->g.cinp() != 16 || c4->g.taps != 5 ||
      c5->g.taps != 5 || c6->g.taps != 7 || c6->g.M() != 16) {
    set_error("fused decoder tail: conv layers missing");
    return VP_ERR_INVALID;
  }
  const int x3 = c4->src1;
  net.need(x3, HALO - 5 + T_OUT / 8 + T3<1256>::PARK_COLS);  // the last tile of a row may read past it: zero margin
  net.tensor_sets[c4->dst] = 0;  // stages 4 and 5 are never materialised by this plan
  net.tensor_sets[c5->dst] = 0;
  HostBlob* p4 = net.add_blob(b3_operand(*c4, true));
  HostBlob* p5 = net.add_blob(b3_operand(*c5, true));
  HostBlob* p6 = net.add_blob(b3_operand(*c6, true));
  // head table: [decoder][piece][entry e][8 channels] bf16, entry e <-> k = e - 15: w[ci][k] for 0 <= k <= 10, else zero
  std::vector<uint16_t> ht((size_t)3 * 3 * HT_N * 8, 0);
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
  for (int d = 0; d < 3; ++d)
    for (int k = 0; k <= 10; ++k)
      for (int ci = 0; ci < 8; ++ci) {
        const float wv = c6->e0.h[(size_t)d * 88 + ci * 11 + k];
        const uint16_t h = rne(wv);
        const float r1 = wv - widen(h);
        const uint16_t md = rne(r1);
        const uint16_t lo = rne(r1 - widen(md));
        const size_t e = ((size_t)d * 3 * HT_N + (k + 15)) * 8 + ci;
        ht[e] = h;
        ht[e + (size_t)HT_N * 8] = md;
        ht[e + (size_t)2 * HT_N * 8] = lo;
      }
  std::vector<float> htf(ht.size() / 2);
  memcpy(htf.data(), ht.data(), ht.size() * 2);
  HostBlob* head_t = net.add_blob(std::move(htf));
  Step st;
  st.name = "fused.tail (decoder.4-6 + heads, time-tiled)";
  st.flops_per_window = 0;
  for (int i = 0; i < 3; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    Tail3Args a{};
    const Tensor& t3 = n.tensors[x3];
    a.x3 = t3.p;
    a.ls3 = t3.ls;
    a.ws3 = (long)t3.win_stride();
    a.y = n.y;
    a.af4 = reinterpret_cast<const uint4*>(p4->d);
    a.af5 = reinterpret_cast<const uint4*>(p5->d);
    a.af6 = reinterpret_cast<const uint4*>(p6->d);
    a.af4_stride = (long)(p4->h.size() / 3 / 4);
    a.af5_stride = (long)(p5->h.size() / 3 / 4);
    a.af6_stride = (long)(p6->h.size() / 3 / 4);
    a.bs4 = c4->bias.d, a.bs5 = c5->bias.d, a.bs6 = c6->bias.d;
    a.head_t = reinterpret_cast<const uint4*>(head_t->d);
    a.head_b = c6->e1.d;
    a.flags = n.win_flags ? n.win_flags->d : nullptr;  // set by gather_normalize_kernel (or by eqt_front_kernel when it cuts the windows itself)
    a.B = B;
    // the outputs the caller keeps (annotate / classify: [blind_l, T - blind_r); model(x): the whole row), in the tiling that
    // needs fewer tiles; plan_flags[7] bit 10 computes the whole row whatever the blinding (A/B, tests)
    const Tail3Tiling tl = tail3_tiling(n.cfg, n.out_lo, n.out_hi);
    const bool wide = tl.wide;
    a.t_lo = tl.t_lo;
    a.tiles_per_row = tl.tiles_per_row;
    a.n_tiles = 3 * B * a.tiles_per_row;
    a.clk = (n.debug_clock && n.debug_clock->d)
                ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) + (size_t)n.max_batch * 32 + 64 * 8
                : nullptr;
    const int grid = a.n_tiles < 256 ? a.n_tiles : 256;
    if (wide)
      hipLaunchKernelGGL(eqt_tail3_kernel<1256>, dim3(grid), dim3(T3_NTH), T3<1256>::LDS_BYTES, s, a);
    else
      hipLaunchKernelGGL(eqt_tail3_kernel<1200>, dim3(grid), dim3(T3_NTH), T3<1200>::LDS_BYTES, s, a);
    return 0;
  };
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_tail3_kernel<1200>), (size_t)T3<1200>::LDS_BYTES});
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_tail3_kernel<1256>), (size_t)T3<1256>::LDS_BYTES});
  st.issued_for_range = [](const Net& n, int lo, int hi, double* w) {
    w[0] = 0.0, w[1] = tail3_tiling(n.cfg, lo, hi).issued_bf16, w[2] = 0.0;
  };
  st.set_issued(0.0, tail3_tiling(net.cfg, 0, 0).issued_bf16, 0.0);  // the whole row
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 3);
  net.steps.push_back(std::move(st));
  net.poison_in_plan = true;  // no poison_kernel launch behind this plan
  return VP_OK;
}

}  // namespace vp
