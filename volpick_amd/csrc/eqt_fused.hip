// EQTransformer ResCNN stack as ONE launch: seven residual blocks (14 convs 64->64, k 3/2, length 47)
// run back to back inside one workgroup per window with the residual stream, the BN-ReLU'd conv
// input and the mid activation in LDS (60 KB); only the packed weights stream in from L2.
// Same arithmetic and the same packed A-fragments as the 14 conv_mfma_kernel launches it replaces
// (plan flag reserved[0] = 1 keeps those for A/B timing and the layer-by-layer parity test).
#include "conv_lds.h"
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

constexpr int RT = EQT_T;   // 47
constexpr int RS = 80;      // image stride (== 16 mod 32): logical 0 at column 4, 29 zero columns to the right
constexpr int RB = 4;
using R_k3 = LdsLayer<64, 0, 64, 1, 3, 1, -1, 0, 3, 1>;   // conv1: (BN folded) + ReLU
using R_k2 = LdsLayer<64, 0, 64, 1, 2, 1, 0, 0, 3, 1>;
using R_k3n = LdsLayer<64, 0, 64, 1, 3, 1, -1, 0, 3, 0>;  // conv2: no activation, residual epilogue
using R_k2n = LdsLayer<64, 0, 64, 1, 2, 1, 0, 0, 3, 0>;

struct ResArgs {
  const float* x0;    // encoder output [B][64][ls]
  const float* act0;  // relu(bn1_0(x0)) [B][64][ls]
  int ls_x, ls_a;
  long ws_x, ws_a;
  float* out;         // [B][64][ls]
  int ls_out;
  long ws_out;
  const float* af1[7];
  const float* bs1[7];
  const float* af2[7];
  const float* bs2[7];
  const float* s_next[7];  // BN scale/shift of the NEXT block's norm1 (block 6: unused)
  const float* b_next[7];
};

struct MidStore {  // conv1 epilogue -> mid image
  float* img;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)RT) img[co * RS + RB + t] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < RT; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * RS + RB + t] = v; }
};

struct ResStore {  // conv2 epilogue: x += v (in place); act = relu(s*x + b) for the next block's conv1
  float* x;
  float* act;
  const float* s;
  const float* b;
  bool last;
  __device__ __forceinline__ void unchecked(int co, int t, float v) const {
    const float o = v + x[co * RS + RB + t];
    x[co * RS + RB + t] = o;
    if (!last) act[co * RS + RB + t] = fmaxf(fmaf(s[co], o, b[co]), 0.f);
  }
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)RT) unchecked(co, t, v);
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < RT; }
};

__global__ __launch_bounds__(256) void eqt_res_kernel(const ResArgs a) {
  __shared__ __attribute__((aligned(16))) float X[64 * RS];
  __shared__ __attribute__((aligned(16))) float ACT[64 * RS];
  __shared__ __attribute__((aligned(16))) float MID[64 * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, win = blockIdx.x;
  {
    const float* x0 = a.x0 + (long)win * a.ws_x + HALO;
    const float* a0 = a.act0 + (long)win * a.ws_a + HALO;
    for (int i = tid; i < 64 * RS; i += 256) {
      const int c = i / RS, col = i - c * RS, t = col - RB;
      const bool in = (unsigned)t < (unsigned)RT;
      X[i] = in ? x0[(long)c * a.ls_x + t] : 0.f;
      ACT[i] = in ? a0[(long)c * a.ls_a + t] : 0.f;  // halo columns stay zero for the whole kernel
      MID[i] = 0.f;
    }
  }
  __syncthreads();
  constexpr int kers[7] = {3, 3, 3, 3, 2, 3, 2};
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    MidStore ms{MID};
    ResStore rs{X, ACT, a.s_next[i], a.b_next[i], i == 6};
    if (kers[i] == 3) {
      conv_lds<R_k3, RS, RB, RS, RB, false>(ACT, ACT, a.af1[i], a.bs1[i], RT, ms, wave, 4, lane);
      __syncthreads();
      conv_lds<R_k3n, RS, RB, RS, RB, false>(MID, MID, a.af2[i], a.bs2[i], RT, rs, wave, 4, lane);
    } else {
      conv_lds<R_k2, RS, RB, RS, RB, false>(ACT, ACT, a.af1[i], a.bs1[i], RT, ms, wave, 4, lane);
      __syncthreads();
      conv_lds<R_k2n, RS, RB, RS, RB, false>(MID, MID, a.af2[i], a.bs2[i], RT, rs, wave, 4, lane);
    }
    __syncthreads();
  }
  float* out = a.out + (long)win * a.ws_out + HALO;
  for (int i = tid; i < 64 * RT; i += 256) {
    const int c = i / RT, t = i - c * RT;
    out[(long)c * a.ls_out + t] = X[c * RS + RB + t];
  }
}


// ---------------------------------------------------------------------------------------------
// Decoder tail as ONE launch per 512 output samples: decoder.5 (16->16, k9) -> decoder.6 (16->8, k11)
// -> Conv1d(8,1,11) + sigmoid head, for the three decoders (weight set = window / B).  The 16x3000
// rows between stages 5 and 6 (147 MB per 256-window batch, written and read back by the layer plan)
// stay in LDS.  Local coordinates: y4 col c <-> global 128*tile - 8 + c; y5 col <-> global 256*tile - 16 + c;
// staged stage-6 col s <-> output sample 512*tile - 16 + s.
// ---------------------------------------------------------------------------------------------
constexpr int DT_TT = 512;
constexpr int DT_S4 = 176, DT_S5 = 304, DT_SO = 548;  // image strides (== 16 mod 32 for the conv inputs)
using DT_d5 = LdsLayer<16, 0, 16, 2, 5, 1, -2, 0, 5, 1>;
using DT_d6 = LdsLayer<16, 0, 8, 2, 7, 1, 5, 0, 5, 1>;   // column j reads y5 local j + 8 + d, d in [-3, 3]
constexpr int DT_Y4 = 0, DT_Y5 = 16 * DT_S4, DT_ST = DT_Y5 + 16 * DT_S5, DT_LDS_FLOATS = DT_ST + 8 * DT_SO;

struct DecTailArgs {
  const float* y4;  // decoder.4 rows [3B][16][ls]
  int ls4;
  long ws4;
  float* y;         // dense (B, 3, T)
  const float* af5; // [3 sets]
  const float* bs5;
  const float* af6;
  const float* bs6;
  long af5_stride, af6_stride;
  const float* wh;  // [3][8][11]
  const float* bh;  // [3]
  int B, T;
};

__global__ __launch_bounds__(256) void eqt_dec_tail_kernel(const DecTailArgs a) {
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, win = blockIdx.y;
  const int set = win / a.B, b = win - set * a.B;
  const int o5 = 256 * tile - 16, t_st = DT_TT * tile - 16;  // global index of y5 col 0 / staged col 0
  {  // y4 rows: cols [0, 152); physical = HALO + 128*tile - 8 + 4q
    const float* src = a.y4 + (long)win * a.ws4 + 128 * tile;
    for (int i = tid; i < 16 * 38; i += 256) {
      const int c = i / 38, q = i - c * 38;
      *reinterpret_cast<float4*>(lds + DT_Y4 + c * DT_S4 + RB + 4 * q) =
          *reinterpret_cast<const float4*>(src + (long)c * a.ls4 + 4 * q);
    }
  }
  __syncthreads();
  {
    ImageStore<DT_S5, RB> st{lds + DT_Y5, 0, 288, -o5, 3000 - o5};
    conv_lds<DT_d5, DT_S4, RB, DT_S4, RB, false>(lds + DT_Y4, lds + DT_Y4, a.af5 + set * a.af5_stride, a.bs5 + set * 16,
                                                   144, st, wave, 4, lane);
  }
  __syncthreads();
  {
    ImageStore<DT_SO, 0> st{lds + DT_ST, 0, 544, -t_st, a.T - t_st};
    conv_lds<DT_d6, DT_S5, RB, DT_S5, RB, false>(lds + DT_Y5, lds + DT_Y5, a.af6 + set * a.af6_stride, a.bs6 + set * 8,
                                                   272, st, wave, 4, lane);
  }
  __syncthreads();
  // head: thread i reads staged cols 4i..4i+15 of each channel, owns cols 4i+5..4i+8 inside [16, 528)
  if (tid < 132) {
    const float* w = a.wh + set * 88;
    const float bias_h = a.bh[set];
    float acc[4] = {bias_h, bias_h, bias_h, bias_h};
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
      float v[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 x4 = *reinterpret_cast<const float4*>(lds + DT_ST + ci * DT_SO + 4 * tid + 4 * q4);
        v[4 * q4] = x4.x;
        v[4 * q4 + 1] = x4.y;
        v[4 * q4 + 2] = x4.z;
        v[4 * q4 + 3] = x4.w;
      }
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wk = w[ci * 11 + k];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk, v[o + k], acc[o]);
      }
    }
    float* y = a.y + ((long)b * 3 + set) * a.T;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int col = 4 * tid + 5 + o, t = t_st + col;
      if (col >= 16 && col < 16 + DT_TT && t < a.T) y[t] = 1.f / (1.f + expf(-acc[o]));
    }
  }
}

}  // namespace

// Replaces the steps "res0.conv1" .. "res6.conv2" of the layer plan by one fused step.
int plan_eqt_fuse_res(Net& net) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "res0.conv1") first = (int)i;
  if (first < 0 || first + 14 > (int)net.steps.size() || net.steps[first + 13].name != "res6.conv2") {
    set_error("fused ResCNN: layer plan not found");
    return VP_ERR_INVALID;
  }
  std::vector<ConvLayer*> c1(7), c2(7);
  for (auto& c : net.convs)
    for (int i = 0; i < 7; ++i) {
      if (c->name == "res" + std::to_string(i) + ".conv1") c1[i] = c.get();
      if (c->name == "res" + std::to_string(i) + ".conv2") c2[i] = c.get();
    }
  for (int i = 0; i < 7; ++i)
    if (!c1[i] || !c2[i]) {
      set_error("fused ResCNN: conv layer %d missing", i);
      return VP_ERR_INVALID;
    }
  const int x0 = c2[0]->res, act0 = c1[0]->src1, out = c2[6]->dst;
  Step st;
  st.name = "fused.rescnn (7 residual blocks)";
  st.flops_per_window = 0;
  for (int i = 0; i < 14; ++i) st.flops_per_window += net.steps[first + i].flops_per_window;
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    ResArgs a{};
    const Tensor &tx = n.tensors[x0], &ta = n.tensors[act0], &to = n.tensors[out];
    a.x0 = tx.p;
    a.act0 = ta.p;
    a.ls_x = tx.ls;
    a.ws_x = (long)tx.win_stride();
    a.ls_a = ta.ls;
    a.ws_a = (long)ta.win_stride();
    a.out = to.p;
    a.ls_out = to.ls;
    a.ws_out = (long)to.win_stride();
    for (int i = 0; i < 7; ++i) {
      a.af1[i] = c1[i]->afrag.d;
      a.bs1[i] = c1[i]->bias.d;
      a.af2[i] = c2[i]->afrag.d;
      a.bs2[i] = c2[i]->bias.d;
      a.s_next[i] = c2[i]->e1.d;
      a.b_next[i] = c2[i]->e2.d;
    }
    hipLaunchKernelGGL(eqt_res_kernel, dim3(B), dim3(256), 0, s, a);
    return 0;
  };
  net.steps.erase(net.steps.begin() + first, net.steps.begin() + first + 14);
  net.steps.insert(net.steps.begin() + first, std::move(st));
  return VP_OK;
}

}  // namespace vp

namespace vp {
// Replaces "decoder.5" and "decoder.6+heads" by the fused decoder tail.
int plan_eqt_fuse_dec_tail(Net& net) {
  int first = -1;
  for (size_t i = 0; i < net.steps.size(); ++i)
    if (net.steps[i].name == "decoder.5") first = (int)i;
  if (first < 0 || first + 2 != (int)net.steps.size() || net.steps[first + 1].name != "decoder.6+heads") {
    set_error("fused decoder tail: layer plan not found");
    return VP_ERR_INVALID;
  }
  ConvLayer *c5 = nullptr, *c6 = nullptr;
  for (auto& c : net.convs) {
    if (c->name == "decoder.5") c5 = c.get();
    if (c->name == "decoder.6") c6 = c.get();
  }
  if (!c5 || !c6) {
    set_error("fused decoder tail: conv layers missing");
    return VP_ERR_INVALID;
  }
  const int y4 = c5->src1;
  const int n_tiles = (net.in_samples + DT_TT - 1) / DT_TT;
  net.need(y4, HALO + 128 * (n_tiles - 1) - 8 + 152);
  Step st;
  st.name = "fused.decoder_tail (decoder.5+6+heads)";
  st.flops_per_window = net.steps[first].flops_per_window + net.steps[first + 1].flops_per_window;
  st.run = [=](Net& n, int B, hipStream_t s) -> int {
    DecTailArgs a{};
    const Tensor& t4 = n.tensors[y4];
    a.y4 = t4.p;
    a.ls4 = t4.ls;
    a.ws4 = (long)t4.win_stride();
    a.y = n.y;
    a.af5 = c5->afrag.d;
    a.bs5 = c5->bias.d;
    a.af6 = c6->afrag.d;
    a.bs6 = c6->bias.d;
    a.af5_stride = (long)(c5->afrag.h.size() / 3);
    a.af6_stride = (long)(c6->afrag.h.size() / 3);
    a.wh = c6->e0.d;
    a.bh = c6->e1.d;
    a.B = B;
    a.T = n.in_samples;
    hipLaunchKernelGGL(eqt_dec_tail_kernel, dim3(n_tiles, 3 * B), dim3(256), DT_LDS_FLOATS * sizeof(float), s, a);
    return 0;
  };
  net.steps.erase(net.steps.begin() + first, net.steps.end());
  net.steps.push_back(std::move(st));
  net.extra_kernels.push_back({reinterpret_cast<const void*>(&eqt_dec_tail_kernel), DT_LDS_FLOATS * sizeof(float)});
  return VP_OK;
}
}  // namespace vp
