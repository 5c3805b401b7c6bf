// EQTransformer encoder stages 3-6 (Conv1d + ReLU + MaxPool1d(2) each: 16 x 750 -> 32 x 375 -> 32 x 188 -> 64 x 94 -> 64 x 47)
// on the bf16 matrix cores with exact three-piece operands (conv_b3.h): the kernel of eqt_enc36.hip -- one workgroup per
// window, the four stages back to back with every intermediate row in LDS -- with six v_mfma_f32_16x16x32_bf16 over
// (hi, mid, lo) pieces in the place of eight fp32 MFMAs per product (6 / 16 of the matrix time, fp32 accuracy).
//
// LDS (149 KB): chunk-plane piece images [piece][8-channel chunk][column][8 channels], 6 bytes per value.  The four images
// would take 235 KB side by side, so the input of stage 5 lies where the input of stage 3 was and the input of stage 6
// where the input of stage 4 was; the padding columns of an image are therefore cleared for every window, by spare threads
// of the stage that writes the image (never by the stage that still reads what lies beneath).
// Weights: a stage's operand (12 / 15 / 15 / 18 16-byte registers per lane: every wave keeps one m-tile) is requested
// while the stage before runs.  MaxPool as in the fp32 kernel: neighbouring columns = neighbouring lanes (one DPP move),
// the even lane stores the pooled sample -- four channels of it, split into the three pieces; conv outputs beyond the
// row count as zero (ReLU outputs are >= 0: MaxPool's -1e10 pad for the odd tail of stage 4 and the next stage's zero
// padding at once).  Stage 6 writes the bottleneck row and relu(bn1(row)) for the first residual block to memory.
//
// Plan flag plan_flags[7] & 128 keeps the fp32-MFMA kernel (bit-identical to the layer launches); this one agrees with it
// to fp32 rounding (tests/test_gpu_eqt.py).
#include "conv_b3.h"
#include "eqt_kernels.h"
#include "net.h"
#include "prepost.h"

namespace vp {

namespace {

constexpr int E3_NTH = 512;
constexpr int N3 = 750, N4 = 375, N5 = 188, N6 = 94, N7 = 47;  // conv row lengths of the stages; the bottleneck length
// n-tiles per wave: stage 3: 2 m-tiles x 4 blocks of 12; stage 4: 2 x 4 x 6; stage 5: 4 x 2 x 6; stage 6: 4 x 2 x 3
constexpr int NB3 = 12, NB4 = 6, NB5 = 6, NB6 = 3;
constexpr int C3 = 4 * NB3 * 16, C4 = 4 * NB4 * 16, C5 = 2 * NB5 * 16, C6 = 2 * NB6 * 16;  // columns computed: 768, 384, 192, 96
constexpr int K3 = 3, K4 = 2, K5 = 2, K6 = 1;                     // sample t of an image at column t + K (= the conv's left reach)
constexpr int NC3 = 784, NC4 = 400, NC5 = 208, NC6 = 112;         // image columns, multiples of 16
static_assert(NC3 >= C3 + 2 * K3 && NC4 >= C4 + 2 * K4 && NC5 >= C5 + 2 * K5 && NC6 >= C6 + 2 * K6, "every column a stage reads has a place");
static_assert(C3 >= N3 && C4 >= N4 && C5 >= N5 && C6 >= N6 && N4 == N3 / 2 && N5 == (N4 + 1) / 2 && N6 == N5 / 2 && N7 == N6 / 2, "row lengths");
using Q3 = B3Chunk<16, NC3>;
using Q4 = B3Chunk<32, NC4>;
using Q5 = B3Chunk<32, NC5>;
using Q6 = B3Chunk<64, NC6>;
constexpr int RA_BYTES = 3 * Q3::PS * 2, RB_BYTES = 3 * Q4::PS * 2;  // region A: stage-3 / stage-5 input; region B: stage-4 / stage-6 input
static_assert(3 * Q5::PS * 2 <= RA_BYTES && 3 * Q6::PS * 2 <= RB_BYTES && RA_BYTES % 16 == 0, "aliased images fit");
constexpr int E3_LDS_BYTES = RA_BYTES + RB_BYTES;
static_assert(E3_LDS_BYTES <= 160 * 1024, "LDS budget");

struct Enc36B3Args {
  const float* x;  // encoder.2 rows [B][16][ls]
  int ls_x;
  long ws_x;
  float* y;        // encoder.6 rows [B][64][ls]
  int ls_y;
  long ws_y;
  float* act;      // relu(bn1_0(encoder.6)) rows [B][64][ls]
  int ls_a;
  long ws_a;
  const uint4* af3[4];  // three-piece operands [MT][steps][piece][64] (net.hip: b3_operand)
  const float* bs[4];
  const float* bn_s;    // norm1 of the first residual block, folded: act = relu(s * y + b)
  const float* bn_b;
  int B;
};

// b3_load_a with a uniform base and an opaque 32-bit lane offset: inside the window loop the per-lane 64-bit addresses of
// the 12-18 loads are loop invariants, and hipcc hoists them out of the loop and spills them (89 registers; every reload
// sits in front of the load that needs it).  (Making the POINTER opaque loses its address space: flat loads, which
// count in vmcnt and lgkmcnt and return out of order, so that every stage waited for the operand of the next.)
template <int C, int TAPS>
__device__ __forceinline__ void e3_load_a(const uint4* __restrict__ af3, const int mt, const int lane,
                                          uint4 (&a)[B3Steps<C, TAPS>::STEPS * 3]) {
  constexpr int N = B3Steps<C, TAPS>::STEPS * 3;
  const uint4* p = af3 + (long)mt * (N * 64);  // uniform
  unsigned off = (unsigned)lane;
  asm volatile("" : "+v"(off));
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = p[i * 64 + off];
}

// the lane's four consecutive values of a per-channel vector (uniform base, opaque 32-bit lane offset: see e3_load_a)
__device__ __forceinline__ void e3_load4(const float* __restrict__ base, const int lane, float (&v)[4]) {
  unsigned go = (unsigned)(lane >> 4);
  asm volatile("" : "+v"(go));
  const float4 q = reinterpret_cast<const float4*>(base)[go];
  v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
}

__device__ __forceinline__ float lane_xor1(float v) { return dpp_move<0xB1, 0xF>(v); }  // quad_perm [1,0,3,2]

// relu(acc + bias) of the lane's four rows at conv column c (zero beyond the row), pooled with the neighbouring column
__device__ __forceinline__ void pool4(const f32x4 acc, const float (&bias)[4], const bool in_row, float (&m)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = in_row ? fmaxf(acc[r] + bias[r], 0.f) : 0.f;
    m[r] = fmaxf(v, lane_xor1(v));
  }
}

// the padding columns [0, K) and [K + len, NC) of an image with NCH chunk planes: one 16-byte store per (piece, chunk, column)
template <class Q, int NCH, int K, int NC>
__device__ __forceinline__ void zero_pads(bf16_t* img, const int len, const int tid) {
  const int npad = K + (NC - K - len);
  for (int i = tid; i < npad * 3 * NCH; i += E3_NTH) {
    const int k = i % npad, cp = i / npad;  // cp = piece * NCH + chunk
    const int col = k < K ? k : len + k;
    *reinterpret_cast<uint4*>(img + (cp / NCH) * Q::PS + (cp % NCH) * Q::CHS + col * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
}

__global__ __launch_bounds__(E3_NTH) void eqt_enc36_b3_kernel(const Enc36B3Args a) {
  extern __shared__ uint4 e3_lds[];
  char* base = reinterpret_cast<char*>(e3_lds);
  bf16_t* X3 = reinterpret_cast<bf16_t*>(base);             // region A
  bf16_t* X5 = X3;
  bf16_t* X4 = reinterpret_cast<bf16_t*>(base + RA_BYTES);  // region B
  bf16_t* X6 = X4;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int win = blockIdx.x;
  if (win >= a.B) return;
  const int mt2 = w & 1, blk2 = w >> 1;  // stages with two m-tiles
  const int mt4 = w & 3, blk4 = w >> 2;  // stages with four
  while (true) {
    uint4 a3[B3Steps<16, 7>::STEPS * 3];
    float bias3[4];
    e3_load_a<16, 7>(a.af3[0], mt2, lane, a3);
    e3_load4(a.bs[0] + mt2 * 16, lane, bias3);
    {  // the window's 16 x 750 input -> the whole stage-3 image [0, 784) <-> samples -3 .. 780 (the tensor's margins are zero):
       // an item = four channels of one column; all of a thread's items requested before the first is split
      const float* src = a.x + (long)win * a.ws_x + (HALO - K3);
      constexpr int ITEMS = (4 * NC3 + E3_NTH - 1) / E3_NTH;
      float v[ITEMS][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* row = src + (long)r * a.ls_x;  // uniform base per channel-in-quad; 32-bit per-lane offsets
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
          const int i = tid + k * E3_NTH, cq = i / NC3, col = i - cq * NC3;
          v[k][r] = i < 4 * NC3 ? row[(unsigned)(4 * cq * a.ls_x + col)] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const int i = tid + k * E3_NTH, cq = i / NC3, col = i - cq * NC3;
        if (i < 4 * NC3) b3c_store4<16, NC3>(X3, col, cq, v[k]);
      }
    }
    __syncthreads();  // also: every wave is through with stage 6 of the window before (region B)
    uint4 a4[B3Steps<32, 5>::STEPS * 3];
    float bias4[4];
    e3_load_a<32, 5>(a.af3[1], mt2, lane, a4);
    e3_load4(a.bs[1] + mt2 * 16, lane, bias4);
    {  // stage 3: 16 x 750 -> 32 x 375
      zero_pads<Q4, 4, K4, NC4>(X4, N4, tid);
      const int colb = blk2 * (NB3 * 16);
      b3c_mac_tiles<16, NC3, 7, NB3>(b3c_lane_ptr<16, NC3, 7>(X3, colb, lane), a3, [&](const int j, const f32x4 acc) {
        const int c = colb + j * 16 + n;
        float m[4];
        pool4(acc, bias3, c < N3, m);
        if (!(n & 1) && (c >> 1) < N4) b3c_store4<32, NC4>(X4, (c >> 1) + K4, 4 * mt2 + g, m);
      });
    }
    __syncthreads();
    uint4 a5[B3Steps<32, 5>::STEPS * 3];
    float bias5[4];
    e3_load_a<32, 5>(a.af3[2], mt4, lane, a5);
    e3_load4(a.bs[2] + mt4 * 16, lane, bias5);
    {  // stage 4: 32 x 375 -> 32 x 188 (odd tail pooled alone)
      zero_pads<Q5, 4, K5, NC5>(X5, N5, tid);
      const int colb = blk2 * (NB4 * 16);
      b3c_mac_tiles<32, NC4, 5, NB4>(b3c_lane_ptr<32, NC4, 5>(X4, colb, lane), a4, [&](const int j, const f32x4 acc) {
        const int c = colb + j * 16 + n;
        float m[4];
        pool4(acc, bias4, c < N4, m);
        if (!(n & 1) && (c >> 1) < N5) b3c_store4<32, NC5>(X5, (c >> 1) + K5, 4 * mt2 + g, m);
      });
    }
    __syncthreads();
    uint4 a6[B3Steps<64, 3>::STEPS * 3];
    float bias6[4];
    e3_load_a<64, 3>(a.af3[3], mt4, lane, a6);
    e3_load4(a.bs[3] + mt4 * 16, lane, bias6);
    {  // stage 5: 32 x 188 -> 64 x 94
      zero_pads<Q6, 8, K6, NC6>(X6, N6, tid);
      const int colb = blk4 * (NB5 * 16);
      b3c_mac_tiles<32, NC5, 5, NB5>(b3c_lane_ptr<32, NC5, 5>(X5, colb, lane), a5, [&](const int j, const f32x4 acc) {
        const int c = colb + j * 16 + n;
        float m[4];
        pool4(acc, bias5, c < N5, m);
        if (!(n & 1) && (c >> 1) < N6) b3c_store4<64, NC6>(X6, (c >> 1) + K6, 4 * mt4 + g, m);
      });
    }
    __syncthreads();
    {  // stage 6: 64 x 94 -> 64 x 47, to memory with the first residual block's BN-ReLU beside it
      float* y = a.y + (long)win * a.ws_y + HALO;
      float* act = a.act + (long)win * a.ws_a + HALO;
      float sc[4], sh[4];
      e3_load4(a.bn_s + mt4 * 16, lane, sc);
      e3_load4(a.bn_b + mt4 * 16, lane, sh);
      const int colb = blk4 * (NB6 * 16);
      b3c_mac_tiles<64, NC6, 3, NB6>(b3c_lane_ptr<64, NC6, 3>(X6, colb, lane), a6, [&](const int j, const f32x4 acc) {
        const int c = colb + j * 16 + n;
        float m[4];
        pool4(acc, bias6, c < N6, m);
        if (!(n & 1) && c < N6) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = mt4 * 16 + 4 * g + r;
            y[(long)co * a.ls_y + (c >> 1)] = m[r];
            act[(long)co * a.ls_a + (c >> 1)] = fmaxf(fmaf(sc[r], m[r], sh[r]), 0.f);
          }
        }
      });
    }
    win += gridDim.x;
    if (win >= a.B) break;
    // no barrier here: the next window's input image lands in region A, whose last readers (stage 5) are a barrier back;
    // region B (read by stage 6) is not written before the barrier behind that
  }
}

}  // namespace

// Replaces the steps "encoder.3" .. "encoder.6" of the plan by one fused step (bf16-piece form).
int plan_eqt_fuse_enc36_b3(Net& net) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "encoder.3") first = (int)i;
  if (first < 0 || first + 4 > (int)net.steps.size() || net.steps[first + 3].name != "encoder.6") {
    set_error("fused encoder stages 3-6: layer plan not found");
    return VP_ERR_INVALID;
  }
  ConvLayer* c[4] = {nullptr, nullptr, nullptr, nullptr};
  for (auto& l : net.convs)
    for (int i = 0; i < 4; ++i)
      if (l->name == "encoder." + std::to_string(i + 3)) c[i] = l.get();
  const int cin[4] = {16, 32, 32, 64}, taps[4] = {7, 5, 5, 3}, mrows[4] = {32, 32, 64, 64};
  if (!c[0] || !c[1] || !c[2] || !c[3] || c[3]->dst2 < 0) {
    set_error("fused encoder stages 3-6: conv layers missing");
    return VP_ERR_INVALID;
  }
  for (int i = 0; i < 4; ++i)
    if (c[i]->g.cinp() != cin[i] || c[i]->g.taps != taps[i] || c[i]->g.M() != mrows[i] || c[i]->g.P != 1) {
      set_error("fused encoder stages 3-6: unexpected layer shape");
      return VP_ERR_INVALID;
    }
  HostBlob* p3[4];
  for (int i = 0; i < 4; ++i) p3[i] = net.add_blob(b3_operand(*c[i], false));
  const int x_in = c[0]->src1, y_out = c[3]->dst, act_out = c[3]->dst2;
  net.need(x_in, HALO - K3 + NC3);  // the image row is fetched whole: zero margin up to there
  for (int i = 0; i < 3; ++i) net.tensor_sets[c[i]->dst] = 0;  // encoder.3 - .5 live in LDS under this plan
  Step st;
  st.name = "fused.enc36 (encoder.3-6, one window per workgroup)";
  st.flops_per_window = 0;
  for (int i = 0; i < 4; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  // matrix work issued, as fp32-equivalent FLOP: one group of six bf16 MFMAs = one 16 x 16 x 32 fp32-accurate product
  st.set_issued(0.0, (2.0 * 48 * 4 + 2.0 * 24 * 5 + 4.0 * 12 * 5 + 4.0 * 6 * 6) * 6 * 16384.0, 0.0);  // 1008 groups = 6048 MFMAs
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    Enc36B3Args a{};
    const Tensor &tx = n.tensors[x_in], &ty = n.tensors[y_out], &ta = n.tensors[act_out];
    a.x = tx.p;
    a.ls_x = tx.ls;
    a.ws_x = (long)tx.win_stride();
    a.y = ty.p;
    a.ls_y = ty.ls;
    a.ws_y = (long)ty.win_stride();
    a.act = ta.p;
    a.ls_a = ta.ls;
    a.ws_a = (long)ta.win_stride();
    for (int i = 0; i < 4; ++i) {
      a.af3[i] = reinterpret_cast<const uint4*>(p3[i]->d);
      a.bs[i] = c[i]->bias.d;
    }
    a.bn_s = c[3]->e1.d;
    a.bn_b = c[3]->e2.d;
    a.B = B;
    const int grid = B < 256 ? B : 256;
    hipLaunchKernelGGL(eqt_enc36_b3_kernel, dim3(grid), dim3(E3_NTH), E3_LDS_BYTES, s, a);
    return 0;
  };
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_enc36_b3_kernel), (size_t)E3_LDS_BYTES});
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 4);
  net.steps.insert(net.steps.begin() + first, std::move(st));
  return VP_OK;
}

}  // namespace vp
