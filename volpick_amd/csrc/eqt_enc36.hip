// EQTransformer encoder stages 3-6 (Conv1d + ReLU + MaxPool1d(2) each: 16 x 750 -> 32 x 375 -> 32 x 188 -> 64 x 94 -> 64 x 47)
// as ONE launch: one workgroup per window runs the four stages back to back with every intermediate row in LDS.
//
// As four conv_mfma_kernel launches these stages took 54 us per 256 windows for 26 us of MFMA issue: the rows are short,
// every launch is a fraction of one residency of the chip plus a kernel boundary.  Here one 512-thread workgroup per CU
// owns a window: 157 KB of LDS images (no aliasing: the margins, zeroed once, stay the convolution padding), the A
// operand of a stage (28 / 40 / 40 / 48 fragments per lane: every wave keeps one m-tile) in registers, fetched as
// 16-byte loads while the stage before runs, B fragments out of LDS one K-step ahead (conv_lds_areg).  MaxPool pools
// neighbouring columns = neighbouring lanes (one DPP move), conv outputs beyond the row counted as zero (ReLU outputs are
// >= 0: that is MaxPool's -1e10 pad for the odd tail of stage 4 and the next stage's zero padding at once).  Stage 6
// writes the bottleneck row and, as the launch it replaces, relu(bn1(row)) for the first residual block.
// Same packed fragments, same K order, same arithmetic: bit-identical (plan flag plan_flags[7] & 8 keeps the launches).
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"
#include "prepost.h"

namespace vp {

namespace {

constexpr int E36_NTH = 512;
//                      CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF NB RELU
using E_3 = LdsLayer<16, 0, 32, 1, 7, 1, -3, 0, 6, 1>;  // 2 m-tiles x 8 blocks of 6 n-tiles: two per wave
using E_4 = LdsLayer<32, 0, 32, 1, 5, 1, -2, 0, 6, 1>;  // 2 m-tiles x 4 blocks of 6
using E_5 = LdsLayer<32, 0, 64, 1, 5, 1, -2, 0, 6, 1>;  // 4 m-tiles x 2 blocks of 6
using E_6 = LdsLayer<64, 0, 64, 1, 3, 1, -1, 0, 3, 1>;  // 4 m-tiles x 2 blocks of 3
constexpr int N3 = 750, N4 = 375, N5 = 188, N6 = 94, N7 = 47;    // conv row lengths of the stages; the bottleneck length
constexpr int C3 = 752, C4 = 384, C5 = 192, C6 = 96;              // MFMA columns per stage (47 / 24 / 12 / 6 n-tiles)
constexpr int S3 = 784, S4 = 400, S5 = 208, S6 = 112, BI = 4;     // images: strides == 16 mod 32, sample 0 at column 4
static_assert(S3 % 32 == 16 && S4 % 32 == 16 && S5 % 32 == 16 && S6 % 32 == 16, "bank-conflict-free strides");
static_assert(S3 >= BI + 768 + 3 && S4 >= BI + C4 + 2 && S4 >= BI + 768 / 2 && S5 >= BI + C5 + 2 && S6 >= BI + C6 + 1, "image widths");
constexpr int OFF3 = 0, OFF4 = OFF3 + 16 * S3, OFF5 = OFF4 + 32 * S4, OFF6 = OFF5 + 32 * S5, E36_LDS_FLOATS = OFF6 + 64 * S6;
static_assert(E36_LDS_FLOATS * 4 <= 160 * 1024 && OFF4 % 4 == 0 && OFF5 % 4 == 0 && OFF6 % 4 == 0, "LDS budget");

struct Enc36Args {
  const float* x;  // encoder.2 rows [B][16][ls]
  int ls_x;
  long ws_x;
  float* y;        // encoder.6 rows [B][64][ls]
  int ls_y;
  long ws_y;
  float* act;      // relu(bn1_0(encoder.6)) rows [B][64][ls]
  int ls_a;
  long ws_a;
  const float* af[4];  // A fragments regrouped for 16-byte loads [MT][CB * TAPS / 4][64][4]
  const float* bs[4];
  const float* bn_s;   // norm1 of the first residual block, folded: act = relu(s * y + b)
  const float* bn_b;
  int B;
};

__device__ __forceinline__ float lane_xor1(float v) { return dpp_move<0xB1, 0xF>(v); }  // quad_perm [1,0,3,2]

// MaxPool1d(2) of a stage's ReLU output: neighbouring columns (lanes n, n ^ 1) pool, the even lane stores pooled sample
// c / 2; conv outputs at columns >= len count as zero.
template <int S>
struct PoolStore {
  static constexpr bool custom_block_epilogue = true;
  float* img;    // image + BI
  unsigned len;  // conv row length
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 1 && L::RELU == 1, "encoder stage");
    const bool fast = (unsigned)(colb + L::NB * 16 - 1) < len;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* row = img + (mt * 16 + 4 * g + r) * S + ((colb + n) >> 1);
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        float v = fmaxf(acc[j][r] + biasv[r], 0.f);
        if (!fast) v = ((unsigned)(colb + j * 16 + n) < len) ? v : 0.f;
        const float m = fmaxf(v, lane_xor1(v));
        if (!(n & 1)) row[j * 8] = m;
      }
    }
  }
};

// Stage 6 -> memory: the pooled row and relu(s * row + b) (conv_mfma.h EPI_POOL2_DUAL).
struct PoolDualOut {
  static constexpr bool custom_block_epilogue = true;
  float* y;    // row base of channel 0 (+ HALO)
  float* act;
  int ls_y, ls_a;
  const float* s;
  const float* b;
  template <class L>
  __device__ __forceinline__ void block_epilogue(const f32x4 (&acc)[L::NB], const float (&biasv)[4], const int mt, const int colb,
                                                 const int g, const int n) const {
    static_assert(L::P == 1 && L::RELU == 1, "encoder stage");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = mt * 16 + 4 * g + r;
      const float sc = s[co], sh = b[co];
#pragma unroll
      for (int j = 0; j < L::NB; ++j) {
        const int c = colb + j * 16 + n;
        float v = fmaxf(acc[j][r] + biasv[r], 0.f);
        v = (c < N6) ? v : 0.f;
        const float m = fmaxf(v, lane_xor1(v));
        if (!(n & 1) && c < N6) {
          y[(long)co * ls_y + (c >> 1)] = m;
          act[(long)co * ls_a + (c >> 1)] = fmaxf(fmaf(sc, m, sh), 0.f);
        }
      }
    }
  }
};

__global__ __launch_bounds__(E36_NTH) void eqt_enc36_kernel(const Enc36Args a) {
  extern __shared__ float4 e36_lds_raw[];
  float* lds = reinterpret_cast<float*>(e36_lds_raw);
  int off4 = OFF4 / 4, off5 = OFF5 / 4, off6 = OFF6 / 4;  // opaque image offsets (eqt_tail.hip)
  asm volatile("" : "+v"(off4), "+v"(off5), "+v"(off6));
  float* X3 = lds + OFF3;
  float* X4 = lds + 4 * off4;
  float* X5 = lds + 4 * off5;
  float* X6 = lds + 4 * off6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int win = blockIdx.x;
  if (win >= a.B) return;
  // every column no stage ever writes is convolution padding: zero once, for all windows of this workgroup
  for (int i = tid; i < E36_LDS_FLOATS / 4; i += E36_NTH) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const int mt2 = wave_u & 1, blk2 = wave_u >> 1;  // stages with two m-tiles
  const int mt4 = wave_u & 3, blk4 = wave_u >> 2;  // stages with four
  while (true) {
    {  // the window's 16 x 750 input: the whole image row [0, 784) <-> samples -4 .. 779 (the tensor's margins are zero)
      const float* src = a.x + (long)win * a.ws_x + (HALO - BI);
      constexpr int Q = S3 / 4;
      for (int i = tid; i < 16 * Q; i += E36_NTH) {
        const int c = i / Q, q = i - c * Q;
        *reinterpret_cast<float4*>(X3 + c * S3 + 4 * q) = *reinterpret_cast<const float4*>(src + (long)c * a.ls_x + 4 * q);
      }
    }
    float areg3[E_3::CB * E_3::TAPS], bias3[4];
    load_areg4<E_3>(a.af[0], mt2, lane, areg3);
    load_biasreg<E_3>(a.bs[0], mt2, lane, bias3);
    __syncthreads();
    float areg4[E_4::CB * E_4::TAPS], bias4[4];
    load_areg4<E_4>(a.af[1], mt2, lane, areg4);
    load_biasreg<E_4>(a.bs[1], mt2, lane, bias4);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 3: 16 x 750 -> 32 x 375
      PoolStore<S4> st{X4 + BI, (unsigned)N3};
      conv_lds_areg<E_3, S3, BI, S3, BI>(X3, X3, areg3, bias3, mt2, C3, st, blk2, 4, lane);
    }
    __syncthreads();
    float areg5[E_5::CB * E_5::TAPS], bias5[4];
    load_areg4<E_5>(a.af[2], mt4, lane, areg5);
    load_biasreg<E_5>(a.bs[2], mt4, lane, bias5);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 4: 32 x 375 -> 32 x 188 (odd tail pooled alone)
      PoolStore<S5> st{X5 + BI, (unsigned)N4};
      conv_lds_areg<E_4, S4, BI, S4, BI>(X4, X4, areg4, bias4, mt2, C4, st, blk2, 4, lane);
    }
    __syncthreads();
    float areg6[E_6::CB * E_6::TAPS], bias6[4];
    load_areg4<E_6>(a.af[3], mt4, lane, areg6);
    load_biasreg<E_6>(a.bs[3], mt4, lane, bias6);
    __builtin_amdgcn_sched_barrier(0);
    {  // stage 5: 32 x 188 -> 64 x 94
      PoolStore<S6> st{X6 + BI, (unsigned)N5};
      conv_lds_areg<E_5, S5, BI, S5, BI>(X5, X5, areg5, bias5, mt4, C5, st, blk4, 2, lane);
    }
    __syncthreads();
    {  // stage 6: 64 x 94 -> 64 x 47, to memory with the first residual block's BN-ReLU beside it
      PoolDualOut st{a.y + (long)win * a.ws_y + HALO, a.act + (long)win * a.ws_a + HALO, a.ls_y, a.ls_a, a.bn_s, a.bn_b};
      conv_lds_areg<E_6, S6, BI, S6, BI>(X6, X6, areg6, bias6, mt4, C6, st, blk4, 2, lane);
    }
    win += gridDim.x;
    if (win >= a.B) break;
    // no barrier: the next window's load writes the stage-3 input image, whose readers are three barriers back
  }
  static_assert(N7 * 2 == N6, "bottleneck length");
}

}  // namespace

// Replaces the steps "encoder.3" .. "encoder.6" of the plan by one fused step.
int plan_eqt_fuse_enc36(Net& net) {
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
  if (!c[0] || !c[1] || !c[2] || !c[3] || c[3]->dst2 < 0) {
    set_error("fused encoder stages 3-6: conv layers missing");
    return VP_ERR_INVALID;
  }
  HostBlob* q[4];
  for (int i = 0; i < 4; ++i) q[i] = c[i]->afrag_q4 ? c[i]->afrag_q4 : net.add_blob(regroup_afrag4(*c[i]));
  const int x_in = c[0]->src1, y_out = c[3]->dst, act_out = c[3]->dst2;
  net.need(x_in, HALO - BI + S3);  // the image row is fetched whole: zero margin up to there
  for (int i = 0; i < 3; ++i) net.tensor_sets[c[i]->dst] = 0;  // encoder.3 - .5 live in LDS under this plan
  Step st;
  st.name = "fused.enc36 (encoder.3-6, one window per workgroup)";
  st.flops_per_window = 0;
  for (int i = 0; i < 4; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  // issued MFMA work: 2 m-tiles x 48 n-tiles x 28 K-steps, 2 x 24 x 40, 4 x 12 x 40, 4 x 6 x 48 (2048 FLOP each)
  st.set_issued((2.0 * 48 * 28 + 2.0 * 24 * 40 + 4.0 * 12 * 40 + 4.0 * 6 * 48) * 2048.0, 0.0, 0.0);
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    Enc36Args a{};
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
      a.af[i] = q[i]->d;
      a.bs[i] = c[i]->bias.d;
    }
    a.bn_s = c[3]->e1.d;
    a.bn_b = c[3]->e2.d;
    a.B = B;
    const int grid = B < 256 ? B : 256;
    hipLaunchKernelGGL(eqt_enc36_kernel, dim3(grid), dim3(E36_NTH), E36_LDS_FLOATS * sizeof(float), s, a);
    return 0;
  };
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_enc36_kernel), E36_LDS_FLOATS * sizeof(float)});
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 4);
  net.steps.insert(net.steps.begin() + first, std::move(st));
  return VP_OK;
}

}  // namespace vp
