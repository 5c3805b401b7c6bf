// PhaseNet training step (BASELINE config 5): forward in training mode (batch
// statistics), vector cross entropy, backward, Adam -- the arithmetic of one
// PhaseNetLit.training_step + optimizer.step of the reference
// (/root/reference volpick/model/models.py:34-51 loss, :160-164 shared_step/training_step,
// :177-185 Adam; the SeisBench PhaseNet module it wraps is restated in oracle/models.py).
//
// Layout: every layer keeps z (raw conv output), a = relu(bn(z)) and the gradients gz, ga as haloed
// [B][C][ls] rows.  Forward convs and input-gradient convs run on conv_mfma_kernel with fragments
// re-packed from the live weights every step (gather_pack_kernel, index maps built once with the
// inference packers); weight gradients on wgrad_kernel; everything else in train_kernels.h.
// ConvTranspose outputs are kept at full length (4 L + 3): BatchNorm sees the samples the U-Net
// crops away, and they receive gradient through the batch statistics.
#include <hip/hip_ext.h>

#include <memory>

#include "conv_mfma.h"
#include "conv_train_b16.h"
#include "train_kernels.h"

namespace vp {
namespace {

constexpr int T0 = 3001, T1 = 751, T2 = 188, T3 = 47, T4 = 12;
constexpr int NLAYER = 18;
static int wg_rows_cap(int out_n);  // rows of partial weight-gradient results a launch may write (defined below)

//                    CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF WM WN NW RELU EPI
// (BF = 1: rows stored as bf16, conv_mfma.h)
// forward (no activation: BatchNorm needs the raw output)
template <int BF> using F_inc = ConvCfg<3, 0, 8, 2, 8, 2, -3, 0, 1, 4, 8, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d0s = ConvCfg<8, 0, 8, 2, 8, 2, -3, 0, 1, 4, 8, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d0d = ConvCfg<8, 0, 8, 2, 11, 8, -3, 0, 1, 4, 2, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d1s = ConvCfg<8, 0, 16, 1, 7, 1, -3, 0, 1, 4, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d1d = ConvCfg<16, 0, 16, 1, 7, 4, -2, 0, 1, 4, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d2s = ConvCfg<16, 0, 32, 1, 7, 1, -3, 0, 2, 2, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d2d = ConvCfg<32, 0, 32, 1, 7, 4, -1, 0, 2, 2, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d3s = ConvCfg<32, 0, 64, 1, 7, 1, -3, 0, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d3d = ConvCfg<64, 0, 64, 1, 7, 4, -2, 0, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_d4s = ConvCfg<64, 0, 128, 1, 7, 1, -3, 0, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u0T = ConvCfg<128, 0, 64, 4, 2, 1, -1, 0, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u0s = ConvCfg<64, 64, 64, 1, 7, 1, -3, 0, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u1T = ConvCfg<64, 0, 32, 4, 2, 1, -1, 0, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u1s = ConvCfg<32, 32, 32, 1, 7, 1, -3, 0, 2, 2, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u2T = ConvCfg<32, 0, 16, 4, 2, 1, -1, 0, 4, 1, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u2s = ConvCfg<16, 16, 16, 1, 7, 1, -3, 0, 1, 4, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u3T = ConvCfg<16, 0, 8, 4, 2, 1, -1, 0, 2, 2, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using F_u3s = ConvCfg<8, 8, 8, 2, 8, 2, -3, 0, 1, 4, 8, 0, EPI_STORE, 0, 0, BF>;
// input gradients: conv(k7, same) -> conv with the flipped, transposed kernel
template <int BF> using G_d0s = ConvCfg<8, 0, 8, 2, 8, 2, -3, 0, 1, 4, 8, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d1s = ConvCfg<16, 0, 8, 2, 8, 2, -3, 0, 1, 4, 2, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d2s = ConvCfg<32, 0, 16, 1, 7, 1, -3, 0, 1, 4, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d3s = ConvCfg<64, 0, 32, 1, 7, 1, -3, 0, 2, 2, 2, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d4s = ConvCfg<128, 0, 64, 1, 7, 1, -3, 0, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u0s = ConvCfg<64, 0, 128, 1, 7, 1, -3, 0, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u1s = ConvCfg<32, 0, 64, 1, 7, 1, -3, 0, 2, 2, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u2s = ConvCfg<16, 0, 32, 1, 7, 1, -3, 0, 2, 2, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u3s = ConvCfg<8, 0, 16, 1, 7, 1, -3, 0, 1, 4, 8, 0, EPI_STORE, 0, 0, BF>;
// conv(k7, s4, left pad p) -> ConvTranspose(k7, s4) of the same kernel, output shifted by -p
template <int BF> using G_d0d = ConvCfg<8, 0, 8, 4, 2, 1, -1, -3, 2, 2, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d1d = ConvCfg<16, 0, 16, 4, 2, 1, -1, -2, 4, 1, 4, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d2d = ConvCfg<32, 0, 32, 4, 2, 1, -1, -1, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_d3d = ConvCfg<64, 0, 64, 4, 2, 1, -1, -2, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
// ConvTranspose(k7, s4) -> conv(k7, s4, no pad) of the same kernel over the full-length gradient
template <int BF> using G_u0T = ConvCfg<64, 0, 128, 1, 7, 4, 0, 0, 4, 1, 1, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u1T = ConvCfg<32, 0, 64, 1, 7, 4, 0, 0, 4, 1, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u2T = ConvCfg<16, 0, 32, 1, 7, 4, 0, 0, 2, 2, 3, 0, EPI_STORE, 0, 0, BF>;
template <int BF> using G_u3T = ConvCfg<8, 0, 16, 1, 7, 4, 0, 0, 1, 4, 4, 0, EPI_STORE, 0, 0, BF>;
// (level-0 stride-1 layers, bf16 rows: 512-sample chunks -- half the barriers per byte; same-box 1.394 -> 1.384 ms per step)
constexpr int W0TT = 512;
//                     LO HI1 HI2 K S NWAVE TT [WB windows per item]
template <int BF> using W_inc = WgradCfg<8, 3, 0, 7, 1, 8, BF ? W0TT : 256>;
template <int BF> using W_d0s = WgradCfg<8, 8, 0, 7, 1, 8, BF ? W0TT : 256>;
using W_d0d = WgradCfg<8, 8, 0, 7, 4, 8, 256>;
using W_d1s = WgradCfg<16, 8, 0, 7, 1, 8, 256>;
using W_d1d = WgradCfg<16, 16, 0, 7, 4, 8, 96>;
using W_d2s = WgradCfg<32, 16, 0, 7, 1, 8, 192>;
using W_d2d = WgradCfg<32, 32, 0, 7, 4, 4, 48>;
using W_d3s = WgradCfg<64, 32, 0, 7, 1, 8, 48>;
using W_d3d = WgradCfg<64, 64, 0, 7, 4, 16, 12, 2>;
using W_d4s = WgradCfg<128, 64, 0, 7, 1, 16, 12, 4>;
using W_u0T = WgradCfg<128, 64, 0, 7, 4, 16, 12, 2>;
using W_u0s = WgradCfg<64, 64, 64, 7, 1, 16, 48>;
using W_u1T = WgradCfg<64, 32, 0, 7, 4, 8, 48>;
using W_u1s = WgradCfg<32, 32, 32, 7, 1, 8, 96>;
using W_u2T = WgradCfg<32, 16, 0, 7, 4, 8, 96>;
using W_u2s = WgradCfg<16, 16, 16, 7, 1, 8, 256>;
using W_u3T = WgradCfg<16, 8, 0, 7, 4, 8, 256>;
template <int BF> using W_u3s = WgradCfg<8, 8, 8, 7, 1, 8, BF ? W0TT : 256>;

// (No layer keeps the fp32 form for its matrix time any more: with two K-steps of fragments in flight up0.same and its input
// gradient -- 57 k weights per 48-column tile, six bytes each as pieces instead of four -- were 1-7 us slower on the bf16
// pipe; with six in flight they are even.  The trait stays for the next such case.)
template <class Cfg> constexpr bool b16_loses = false;

struct ConvOp {
  bool used = false;
  ConvGeom g{};
  int (*launch)(const ConvArgs&, int, hipStream_t) = nullptr;
  // bf16 rows, filters of >= 7 taps: the bf16-MFMA form (conv_train_b16.h); its A operand is cut from the fp32 fragments
  int (*launch_b16)(const ConvArgs&, const uint4*, float*, int, hipStream_t) = nullptr;
  int tn = 0, cout = 0;  // columns per tile / output channels (the statistics a forward launch leaves: [cout][tiles x B][2])
  size_t a3_n = 0, a3_off = 0;  // uint4 words of the operand / its place in Trainer::frag3
  const void* kernel = nullptr;
  size_t lds_bytes = 0;
  int src1 = -1, src2 = -1, dst = -1;
  int cols = 0, l_out = 0;
  size_t frag_off = 0;
  long bias_off = -1;  // offset of a real bias in the weight blob, -1 = zeros
};

struct WgradOp {
  int (*launch)(const WgradArgs&, int, hipStream_t) = nullptr;
  int lo = -1, hi1 = -1, hi2 = -1;
  int Ln = 0, off = 0, TT = 0, WB = 1, out_n = 0;
  long grad_off = 0;
  size_t partial_off = 0;  // slice of the shared partial buffer (all layers are folded by one launch at the end)
};

struct BnOp {
  int z = -1, a = -1, gz = -1;
  int ga1 = -1, ga1_ch = 0, ga2 = -1, ga2_ch = 0;
  int C = 0, Lz = 0, La = 0, crop = 0;
  long gamma_off = 0, beta_off = 0, rm_off = 0, rv_off = 0;
  size_t stats_off = 0;
};

struct Layer {
  std::string name;
  ConvOp fwd, dgrad;
  BnOp bn;
  WgradOp wg;
};

template <class Cfg>
void set_conv(ConvOp* op, int src1, int src2, int dst, int cols, int l_out) {
  op->used = true;
  op->g = Cfg::geom();
  op->launch = &launch_conv<Cfg>;
  op->kernel = reinterpret_cast<const void*>(&conv_mfma_kernel<Cfg>);
  op->lds_bytes = Cfg::LDS_FLOATS * sizeof(float);
  op->src1 = src1;
  op->src2 = src2;
  op->dst = dst;
  op->cols = cols;
  op->l_out = l_out;
  if constexpr (Cfg::BF16 && (Cfg::TAPS >= 7 || (Cfg::TAPS == 2 && Cfg::CB % 4 == 0)) && !b16_loses<Cfg>) {
    op->launch_b16 = &launch_conv_b16<Cfg>;
    op->a3_n = ConvB16<Cfg>::A_UINT4;
    op->tn = Cfg::TN;
    op->cout = Cfg::COUT;
  }
}

template <class Cfg, class T>
void set_wgrad(WgradOp* op, int lo, int hi1, int hi2, int Ln, int off) {
  // bf16 rows: exact products on the bf16 matrix cores (wgrad_bf16_kernel).  (Round 4 kept the four deepest layers -- >= 4096
  // channel pairs, <= 48 samples per row -- on the fp32-MFMA kernel: with 256 partial rows per tensor their cost was the
  // 115-230 KB every workgroup writes.  With the rows capped by bytes (wg_rows_cap: 64 workgroups for 57 k weights) the fp32
  // form is bound by its matrix time on those few CUs: up0.same 99-118 us against 51 here; the step itself does not move.)
  if constexpr (sizeof(T) == 2) {
    op->launch = &launch_wgrad_bf16<Cfg>;
  } else {
    op->launch = &launch_wgrad<Cfg, T>;
  }
  op->lo = lo;
  op->hi1 = hi1;
  op->hi2 = hi2;
  op->Ln = Ln;
  op->off = off;
  op->TT = Cfg::TT;
  op->WB = Cfg::WB;
  op->out_n = Cfg::OUT;
}

// dense (B, C, T) -> haloed rows
__global__ __launch_bounds__(256) void load_rows_kernel(const float* __restrict__ x, Rows r, int C, int T) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) r.p[(long)b * r.ws + (long)c * r.ls + HALO + t] = x[((long)b * C + c) * T + t];
}
// the same into bf16 rows, an even-aligned pair per thread: grid (ceil(T / 512), C, B)
__global__ __launch_bounds__(256) void load_rows_bf16_kernel(const float* __restrict__ x, Rows r, int C, int T) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int t = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (t < T) {
    const float* xp = x + ((long)b * C + c) * T + t;
    Elem<bf16_t>::store2(r.row<bf16_t>(b, c) + t, xp[0], t + 1 < T ? xp[1] : 0.f);
  }
}

}  // namespace

thread_local int g_launches = 0;  // kernel launches of the step being enqueued (vp_train_launch_count)
#define TRL(...)                      \
  do {                                \
    ++g_launches;                     \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

struct Trainer {
  int device = 0, max_batch = 0;
  int launches_last_step = 0;
  bool bf16 = false;  // activation / gradient rows stored as bfloat16 (weights, gradients, statistics, Adam: fp32)
  int esize = 4;      // bytes per row element
  hipStream_t stream = nullptr;
  // The weight gradients run on a stream of their own: wgrad of layer L needs gz of L and the activations below it, nothing
  // downstream needs its result before the fold at the end of the step, and the backward chain (BatchNorm backward -> input
  // gradient -> next layer's BatchNorm backward) is a string of short launches that leave most of the chip idle in the deep
  // layers.  Events order the two: ev_gz[L] (gz of L complete) -> wgrad of L; ev_wg (all weight gradients) -> fold + Adam.
  hipStream_t stream_wg = nullptr;
  hipEvent_t ev_gz[32] = {}, ev_wg = nullptr, ev_wg1 = nullptr;  // ev_wg1: every weight gradient but the first layer's
  std::vector<Tensor> tensors;
  std::vector<Layer> layers;
  std::map<std::string, long> poff;  // parameter name -> offset in the flat blob
  std::map<std::string, size_t> psize;
  size_t n_params = 0;
  // device memory
  float* arena = nullptr;   // activation / gradient tensors
  float* w = nullptr;       // weights (flat blob, canonical order)
  float* grad = nullptr;
  float* adam_m = nullptr;
  float* adam_v = nullptr;
  float* mask = nullptr;
  float* ema = nullptr;     // EMA of the weights (allocated by vp_train_set_ema)
  // behind the last reader of a step's x / y (the head kernel), a ring over the steps in flight:
  // vp_train_wait_inputs_consumed (the latest), vp_train_inputs_consumed_upto (which steps are past it)
  static constexpr int EV_RING = 8;
  hipEvent_t ev_inputs[EV_RING] = {};
  long long ev_inputs_seq[EV_RING] = {-1, -1, -1, -1, -1, -1, -1, -1};
  long long seq = 0;  // steps enqueued so far
  float ema_decay = 0.f;
  int* frag_idx = nullptr;
  float* frag = nullptr;
  size_t frag_n = 0;
  float* zeros = nullptr;   // 256 zero floats (bias of bias-free convs)
  float* stats = nullptr;
  double* bn_partial = nullptr;
  unsigned* bn_counter = nullptr;
  float* conv_stat = nullptr;  // BatchNorm sums a forward bf16-MFMA conv leaves per workgroup (one layer at a time)
  uint4* frag3 = nullptr;  // three-piece A operands of the bf16-MFMA convs, cut from `frag` every step
  ConvB16PackJobs b16_jobs{};
  int b16_blocks = 0;
  float* wg_partial = nullptr;
  size_t wg_partial_floats = 0;
  double* head_partial = nullptr;
  double* head_sums = nullptr;  // [28] + loss at [28]
  double* head_stage = nullptr; // [64][28]
  float* x_dev = nullptr;       // staging for host inputs
  float* y_dev = nullptr;
  float* p_dev = nullptr;       // predictions of the last step
  long step = 0;
  float beta1 = 0.9f, beta2 = 0.999f, adam_eps = 1e-8f, bn_eps = 1e-3f, bn_momentum = 0.1f, loss_eps = 1e-5f;
  int t_x = -1, t_ga_last = -1;
  std::vector<int> frag_idx_host;

  int add_tensor(const std::string& name, int C, int L) {
    Tensor t;
    t.name = name;
    t.C = C;
    t.L = L;
    // rows are walked in 8-sample vectors, a cropped operand one vector further (train_kernels.h load8_at)
    t.need = HALO + round_up(L, 8) + 16;
    tensors.push_back(t);
    return (int)tensors.size() - 1;
  }
  void need(int id, int phys) {
    if (id >= 0 && tensors[id].need < phys) tensors[id].need = phys;
  }
  Rows rows(int id, int ch = 0) const {
    const Tensor& t = tensors[id];  // (t.p is a byte address in disguise when the rows are bf16)
    return Rows{reinterpret_cast<float*>(reinterpret_cast<char*>(t.p) + (long)ch * t.ls * esize), t.ls, (long)t.win_stride()};
  }
  ~Trainer() {
    for (void* p : {(void*)arena, (void*)w, (void*)grad, (void*)adam_m, (void*)adam_v, (void*)mask, (void*)ema, (void*)frag_idx,
                    (void*)frag, (void*)zeros, (void*)stats, (void*)bn_partial, (void*)bn_counter, (void*)frag3, (void*)conv_stat, (void*)wg_partial, (void*)head_partial,
                    (void*)head_sums, (void*)head_stage, (void*)x_dev, (void*)y_dev, (void*)p_dev})
      if (p) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
    if (stream_wg) (void)hipStreamDestroy(stream_wg);
    for (hipEvent_t e : ev_gz)
      if (e) (void)hipEventDestroy(e);
    if (ev_wg) (void)hipEventDestroy(ev_wg);
    if (ev_wg1) (void)hipEventDestroy(ev_wg1);
    for (hipEvent_t e : ev_inputs)
      if (e) (void)hipEventDestroy(e);
  }
};

namespace {

// index map of one conv's packed fragments: pack "weights" whose values are their own flat index + 1
std::vector<int> to_index(const std::vector<float>& packed) {
  std::vector<int> out(packed.size());
  for (size_t i = 0; i < packed.size(); ++i) out[i] = (int)(packed[i] + 0.5f);
  return out;
}
std::vector<float> index_weights(long off, size_t n) {
  std::vector<float> w(n);
  for (size_t i = 0; i < n; ++i) w[i] = (float)(off + (long)i + 1);  // exact below 2^24
  return w;
}
// W[co][ci][K] -> W'[ci][co][K] with the taps reversed (input gradient of a stride-1 "same" conv)
std::vector<float> flip_transpose(const std::vector<float>& w, int cout, int cin, int K) {
  std::vector<float> o(w.size());
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int k = 0; k < K; ++k) o[((size_t)ci * cout + co) * K + (K - 1 - k)] = w[((size_t)co * cin + ci) * K + k];
  return o;
}

void append_frag(Trainer& tr, ConvOp* op, const std::vector<float>& amat) {
  const std::vector<int> idx = to_index(pack_afrag(amat, op->g.M(), op->g.cinp(), op->g.taps));
  op->frag_off = tr.frag_idx_host.size();
  tr.frag_idx_host.insert(tr.frag_idx_host.end(), idx.begin(), idx.end());
}

template <int BF>
int build_plan(Trainer& tr) {
  using ET = typename std::conditional<BF != 0, bf16_t, float>::type;
  const ParamDesc* table;
  const int np = param_table(VP_MODEL_PHASENET, &table);
  long off = 0;
  for (int i = 0; i < np; ++i) {
    tr.poff[table[i].name] = off;
    tr.psize[table[i].name] = table[i].size;
    off += (long)table[i].size;
  }
  tr.n_params = (size_t)off;
  const int len[5] = {T0, T1, T2, T3, T4};
  const int ch[5] = {8, 16, 32, 64, 128};
  const int padl[4] = {3, 2, 1, 2};
  tr.layers.resize(NLAYER);
  // layer order: 0 inc, 1+2i down{i}.same, 2+2i down{i}.down (i<4), 9 down4.same, 10+2j up{j}.convT, 11+2j up{j}.same
  auto P = [&](const std::string& n) { return tr.poff.at(n); };
  auto mk_bn = [&](Layer& L, const std::string& bn, int C, int Lz, int La, int crop) {
    L.bn.C = C;
    L.bn.Lz = Lz;
    L.bn.La = La;
    L.bn.crop = crop;
    L.bn.gamma_off = P(bn + ".weight");
    L.bn.beta_off = P(bn + ".bias");
    L.bn.rm_off = P(bn + ".running_mean");
    L.bn.rv_off = P(bn + ".running_var");
    L.bn.z = tr.add_tensor(L.name + ".z", C, Lz);
    L.bn.a = tr.add_tensor(L.name + ".a", C, La);
    L.bn.gz = tr.add_tensor(L.name + ".gz", C, Lz);
  };
  tr.t_x = tr.add_tensor("x", 3, T0);
  Layer* Ls = tr.layers.data();
  Ls[0].name = "inc";
  mk_bn(Ls[0], "in_bn", 8, T0, T0, 0);
  for (int i = 0; i < 5; ++i) {
    Layer& S = Ls[i < 4 ? 1 + 2 * i : 9];
    S.name = "down" + std::to_string(i) + ".same";
    mk_bn(S, "down_branch." + std::to_string(i) + ".1", ch[i], len[i], len[i], 0);
    if (i < 4) {
      Layer& D = Ls[2 + 2 * i];
      D.name = "down" + std::to_string(i) + ".down";
      mk_bn(D, "down_branch." + std::to_string(i) + ".3", ch[i], len[i + 1], len[i + 1], 0);
    }
  }
  for (int j = 0; j < 4; ++j) {
    const int lv = 3 - j;  // output level
    Layer& U = Ls[10 + 2 * j];
    U.name = "up" + std::to_string(j) + ".convT";
    const int lfull = 4 * len[lv + 1] + 3;
    const int crop = 1 + ((lfull - 3) - len[lv]) / 2;  // x[:, :, 1:-2] then the centre crop to the skip length
    mk_bn(U, "up_branch." + std::to_string(j) + ".1", ch[lv], lfull, len[lv], crop);
    Layer& S = Ls[11 + 2 * j];
    S.name = "up" + std::to_string(j) + ".same";
    mk_bn(S, "up_branch." + std::to_string(j) + ".3", ch[lv], len[lv], len[lv], 0);
  }
  // gradient-of-activation tensors
  int ga[NLAYER];       // plain ga tensors (single producer)
  int gcat[4], gskip[4];
  for (int i = 0; i < NLAYER; ++i) ga[i] = -1;
  for (int j = 0; j < 4; ++j) {
    const int lv = 3 - j;
    gcat[j] = tr.add_tensor("up" + std::to_string(j) + ".gcat", 2 * ch[lv], len[lv]);   // d(concat(skip, up)) of up{j}.same
    gskip[lv] = tr.add_tensor("down" + std::to_string(lv) + ".gskip", ch[lv], len[lv]);  // from down{lv}.down
  }
  auto mk_ga = [&](int li) {
    ga[li] = tr.add_tensor(Ls[li].name + ".ga", Ls[li].bn.C, Ls[li].bn.La);
    Ls[li].bn.ga1 = ga[li];
  };
  mk_ga(0);                                    // inc        <- dgrad of down0.same
  for (int i = 0; i < 4; ++i) mk_ga(2 + 2 * i);  // down{i}.down <- dgrad of down{i+1}.same
  mk_ga(9);                                    // down4.same <- dgrad of up0.convT
  for (int j = 0; j < 4; ++j) mk_ga(11 + 2 * j);  // up{j}.same <- dgrad of up{j+1}.convT, or the head for j = 3
  tr.t_ga_last = ga[17];
  for (int i = 0; i < 4; ++i) {  // skips: from the up path (first half of gcat) + from the strided conv below
    BnOp& b = Ls[1 + 2 * i].bn;
    b.ga1 = gcat[3 - i];
    b.ga1_ch = 0;
    b.ga2 = gskip[i];
  }
  for (int j = 0; j < 4; ++j) {  // convT outputs: second half of gcat
    BnOp& b = Ls[10 + 2 * j].bn;
    b.ga1 = gcat[j];
    b.ga1_ch = ch[3 - j];
  }

  // ---- convolutions: forward, input gradient, weight gradient ------------------------------
  auto W = [&](const std::string& n) { return index_weights(P(n), tr.psize.at(n)); };
#define FWD_CONV(LI, CFG, WNAME, COUT, CIN, STRIDE, SRC1, SRC2, COLS)                                        \
  set_conv<CFG<BF>>(&Ls[LI].fwd, SRC1, SRC2, Ls[LI].bn.z, COLS, Ls[LI].bn.Lz);                                    \
  append_frag(tr, &Ls[LI].fwd, amat_conv(W(WNAME).data(), COUT, CIN, 7, STRIDE, CFG<BF>::P, CFG<BF>::CINP, nullptr));
#define FWD_CONVT(LI, CFG, WNAME, CIN, COUT, SRC, LIN)                                                       \
  set_conv<CFG<BF>>(&Ls[LI].fwd, SRC, -1, Ls[LI].bn.z, (LIN) + 1, Ls[LI].bn.Lz);                                  \
  append_frag(tr, &Ls[LI].fwd, amat_convT_k7s4(W(WNAME).data(), CIN, COUT, CFG<BF>::CINP, nullptr));
  // dgrad of a stride-1 conv W[COUT][CIN][7]: conv of gz (COUT channels) with the flipped transpose -> CIN channels
#define BWD_SAME(LI, CFG, WNAME, COUT, CIN, DST, LDST)                                                       \
  set_conv<CFG<BF>>(&Ls[LI].dgrad, Ls[LI].bn.gz, -1, DST, ((LDST) + CFG<BF>::P - 1) / CFG<BF>::P, LDST);                  \
  append_frag(tr, &Ls[LI].dgrad,                                                                             \
              amat_conv(flip_transpose(W(WNAME), COUT, CIN, 7).data(), CIN, COUT, 7, 1, CFG<BF>::P, CFG<BF>::CINP, nullptr));
  // dgrad of a stride-4 conv W[C][C][7] with left pad PADL: transposed conv of gz, output index shifted by -PADL
#define BWD_DOWN(LI, CFG, WNAME, C, DST, LDST, PADL)                                                         \
  set_conv<CFG<BF>>(&Ls[LI].dgrad, Ls[LI].bn.gz, -1, DST, ((LDST)-1 + (PADL)) / 4 + 1, LDST);                     \
  append_frag(tr, &Ls[LI].dgrad, amat_convT_k7s4(W(WNAME).data(), C, C, CFG<BF>::CINP, nullptr));
  // dgrad of ConvTranspose Wt[CIN][COUT][7]: stride-4 conv of the full-length gz (COUT channels) -> CIN channels
#define BWD_UPT(LI, CFG, WNAME, CIN, COUT, DST, LIN)                                                         \
  set_conv<CFG<BF>>(&Ls[LI].dgrad, Ls[LI].bn.gz, -1, DST, LIN, LIN);                                              \
  append_frag(tr, &Ls[LI].dgrad, amat_conv(W(WNAME).data(), CIN, COUT, 7, 4, 1, CFG<BF>::CINP, nullptr));

  const int x = tr.t_x;
  auto A = [&](int li) { return Ls[li].bn.a; };
  FWD_CONV(0, F_inc, "inc.weight", 8, 3, 1, x, -1, (T0 + 1) / 2)
  Ls[0].fwd.bias_off = P("inc.bias");
  FWD_CONV(1, F_d0s, "down_branch.0.0.weight", 8, 8, 1, A(0), -1, (T0 + 1) / 2)
  FWD_CONV(2, F_d0d, "down_branch.0.2.weight", 8, 8, 4, A(1), -1, (T1 + 1) / 2)
  FWD_CONV(3, F_d1s, "down_branch.1.0.weight", 16, 8, 1, A(2), -1, T1)
  FWD_CONV(4, F_d1d, "down_branch.1.2.weight", 16, 16, 4, A(3), -1, T2)
  FWD_CONV(5, F_d2s, "down_branch.2.0.weight", 32, 16, 1, A(4), -1, T2)
  FWD_CONV(6, F_d2d, "down_branch.2.2.weight", 32, 32, 4, A(5), -1, T3)
  FWD_CONV(7, F_d3s, "down_branch.3.0.weight", 64, 32, 1, A(6), -1, T3)
  FWD_CONV(8, F_d3d, "down_branch.3.2.weight", 64, 64, 4, A(7), -1, T4)
  FWD_CONV(9, F_d4s, "down_branch.4.0.weight", 128, 64, 1, A(8), -1, T4)
  FWD_CONVT(10, F_u0T, "up_branch.0.0.weight", 128, 64, A(9), T4)
  FWD_CONV(11, F_u0s, "up_branch.0.2.weight", 64, 128, 1, A(7), A(10), T3)
  FWD_CONVT(12, F_u1T, "up_branch.1.0.weight", 64, 32, A(11), T3)
  FWD_CONV(13, F_u1s, "up_branch.1.2.weight", 32, 64, 1, A(5), A(12), T2)
  FWD_CONVT(14, F_u2T, "up_branch.2.0.weight", 32, 16, A(13), T2)
  FWD_CONV(15, F_u2s, "up_branch.2.2.weight", 16, 32, 1, A(3), A(14), T1)
  FWD_CONVT(16, F_u3T, "up_branch.3.0.weight", 16, 8, A(15), T1)
  FWD_CONV(17, F_u3s, "up_branch.3.2.weight", 8, 16, 1, A(1), A(16), (T0 + 1) / 2)

  BWD_SAME(1, G_d0s, "down_branch.0.0.weight", 8, 8, ga[0], T0)
  BWD_SAME(3, G_d1s, "down_branch.1.0.weight", 16, 8, ga[2], T1)
  BWD_SAME(5, G_d2s, "down_branch.2.0.weight", 32, 16, ga[4], T2)
  BWD_SAME(7, G_d3s, "down_branch.3.0.weight", 64, 32, ga[6], T3)
  BWD_SAME(9, G_d4s, "down_branch.4.0.weight", 128, 64, ga[8], T4)
  BWD_SAME(11, G_u0s, "up_branch.0.2.weight", 64, 128, gcat[0], T3)
  BWD_SAME(13, G_u1s, "up_branch.1.2.weight", 32, 64, gcat[1], T2)
  BWD_SAME(15, G_u2s, "up_branch.2.2.weight", 16, 32, gcat[2], T1)
  BWD_SAME(17, G_u3s, "up_branch.3.2.weight", 8, 16, gcat[3], T0)
  BWD_DOWN(2, G_d0d, "down_branch.0.2.weight", 8, gskip[0], T0, padl[0])
  BWD_DOWN(4, G_d1d, "down_branch.1.2.weight", 16, gskip[1], T1, padl[1])
  BWD_DOWN(6, G_d2d, "down_branch.2.2.weight", 32, gskip[2], T2, padl[2])
  BWD_DOWN(8, G_d3d, "down_branch.3.2.weight", 64, gskip[3], T3, padl[3])
  BWD_UPT(10, G_u0T, "up_branch.0.0.weight", 128, 64, ga[9], T4)
  BWD_UPT(12, G_u1T, "up_branch.1.0.weight", 64, 32, ga[11], T3)
  BWD_UPT(14, G_u2T, "up_branch.2.0.weight", 32, 16, ga[13], T2)
  BWD_UPT(16, G_u3T, "up_branch.3.0.weight", 16, 8, ga[15], T1)
#undef FWD_CONV
#undef FWD_CONVT
#undef BWD_SAME
#undef BWD_DOWN
#undef BWD_UPT

  auto GZ = [&](int li) { return Ls[li].bn.gz; };
#define WG(LI, CFG, WNAME, LO, HI1, HI2, LN, OFF)            \
  set_wgrad<CFG, ET>(&Ls[LI].wg, LO, HI1, HI2, LN, OFF);          \
  Ls[LI].wg.grad_off = P(WNAME);
  WG(0, W_inc<BF>, "inc.weight", GZ(0), x, -1, T0, -3)
  WG(1, W_d0s<BF>, "down_branch.0.0.weight", GZ(1), A(0), -1, T0, -3)
  WG(2, W_d0d, "down_branch.0.2.weight", GZ(2), A(1), -1, T1, -padl[0])
  WG(3, W_d1s, "down_branch.1.0.weight", GZ(3), A(2), -1, T1, -3)
  WG(4, W_d1d, "down_branch.1.2.weight", GZ(4), A(3), -1, T2, -padl[1])
  WG(5, W_d2s, "down_branch.2.0.weight", GZ(5), A(4), -1, T2, -3)
  WG(6, W_d2d, "down_branch.2.2.weight", GZ(6), A(5), -1, T3, -padl[2])
  WG(7, W_d3s, "down_branch.3.0.weight", GZ(7), A(6), -1, T3, -3)
  WG(8, W_d3d, "down_branch.3.2.weight", GZ(8), A(7), -1, T4, -padl[3])
  WG(9, W_d4s, "down_branch.4.0.weight", GZ(9), A(8), -1, T4, -3)
  WG(10, W_u0T, "up_branch.0.0.weight", A(9), GZ(10), -1, T4, 0)   // ConvTranspose: lo = layer input, hi = gz
  WG(11, W_u0s, "up_branch.0.2.weight", GZ(11), A(7), A(10), T3, -3)
  WG(12, W_u1T, "up_branch.1.0.weight", A(11), GZ(12), -1, T3, 0)
  WG(13, W_u1s, "up_branch.1.2.weight", GZ(13), A(5), A(12), T2, -3)
  WG(14, W_u2T, "up_branch.2.0.weight", A(13), GZ(14), -1, T2, 0)
  WG(15, W_u2s, "up_branch.2.2.weight", GZ(15), A(3), A(14), T1, -3)
  WG(16, W_u3T, "up_branch.3.0.weight", A(15), GZ(16), -1, T1, 0)
  WG(17, W_u3s<BF>, "up_branch.3.2.weight", GZ(17), A(1), A(16), T0, -3)
#undef WG

  for (Layer& L : tr.layers)
    for (ConvOp* op : {&L.fwd, &L.dgrad})
      if (op->used) {
        tr.need(op->src1, op->g.src_need(op->cols));
        tr.need(op->src2, op->g.src_need(op->cols));
      }
  size_t so = 0;
  for (Layer& L : tr.layers) {
    L.bn.stats_off = so;
    so += 4 * (size_t)L.bn.C;
  }
  return VP_OK;
}

#define TR_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);       \
      return VP_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

int upload(Trainer& tr, const float* weights) {
  const int B = tr.max_batch;
  size_t total = 0;
  std::vector<size_t> toff(tr.tensors.size());
  for (size_t i = 0; i < tr.tensors.size(); ++i) {
    Tensor& t = tr.tensors[i];
    t.ls = round_up(t.need, tr.bf16 ? 8 : 4);  // rows start on 16-byte boundaries either way
    toff[i] = total;
    total += (size_t)t.C * t.ls * B;
    total = (total + 63) / 64 * 64;
  }
  TR_HIP(hipMalloc(&tr.arena, total * tr.esize));
  TR_HIP(hipMemset(tr.arena, 0, total * tr.esize));  // (a zero of either type)
  for (size_t i = 0; i < tr.tensors.size(); ++i)
    tr.tensors[i].p = reinterpret_cast<float*>(reinterpret_cast<char*>(tr.arena) + toff[i] * tr.esize);
  const size_t np = tr.n_params;
  for (float** p : {&tr.w, &tr.grad, &tr.adam_m, &tr.adam_v, &tr.mask}) {
    TR_HIP(hipMalloc(p, np * sizeof(float)));
    TR_HIP(hipMemset(*p, 0, np * sizeof(float)));
  }
  TR_HIP(hipMemcpy(tr.w, weights, np * sizeof(float), hipMemcpyHostToDevice));
  std::vector<float> mask(np, 1.f);
  for (auto& kv : tr.poff) {
    const std::string& n = kv.first;
    const bool running = n.size() > 12 && (n.rfind("running_mean") == n.size() - 12 || n.rfind("running_var") == n.size() - 11);
    if (running)
      for (size_t i = 0; i < tr.psize[n]; ++i) mask[kv.second + i] = 0.f;
  }
  TR_HIP(hipMemcpy(tr.mask, mask.data(), np * sizeof(float), hipMemcpyHostToDevice));
  tr.frag_n = tr.frag_idx_host.size();
  TR_HIP(hipMalloc(&tr.frag_idx, tr.frag_n * sizeof(int)));
  TR_HIP(hipMemcpy(tr.frag_idx, tr.frag_idx_host.data(), tr.frag_n * sizeof(int), hipMemcpyHostToDevice));
  TR_HIP(hipMalloc(&tr.frag, tr.frag_n * sizeof(float)));
  {  // the bf16-MFMA convs' operands
    size_t n3 = 0;
    for (Layer& L : tr.layers)
      for (ConvOp* op : {&L.fwd, &L.dgrad})
        if (op->used && op->launch_b16) {
          op->a3_off = n3;
          n3 += op->a3_n;
        }
    if (n3) {
      size_t ns2 = 0;
      for (Layer& L : tr.layers)
        if (L.fwd.launch_b16) {
          const size_t need = (size_t)L.fwd.cout * ((L.fwd.cols + L.fwd.tn - 1) / L.fwd.tn) * tr.max_batch * 2;
          if (need > ns2) ns2 = need;
        }
      if (ns2) TR_HIP(hipMalloc(&tr.conv_stat, ns2 * sizeof(float)));
      TR_HIP(hipMalloc(&tr.frag3, n3 * sizeof(uint4)));
      for (Layer& L : tr.layers)
        for (ConvOp* op : {&L.fwd, &L.dgrad})
          if (op->used && op->launch_b16) {
            if (tr.b16_jobs.count >= MAX_B16_JOBS) return VP_ERR_UNSUPPORTED;
            ConvB16PackJob& jb = tr.b16_jobs.job[tr.b16_jobs.count++];
            const int cb = op->g.cinp() / 4, tpr = op->g.taps <= 2 ? 2 : 8, tg = tpr == 8 ? (op->g.taps + 7) / 8 : 1, mt = op->g.M() / 16;
            jb.frag = tr.frag + op->frag_off;
            jb.out = tr.frag3 + op->a3_off;
            jb.MT = mt;
            jb.CB = cb;
            jb.TAPS = op->g.taps;
            jb.TG = tg;
            jb.TPR = tpr;
            jb.first_block = tr.b16_blocks;
            tr.b16_blocks += (mt * (tpr == 8 ? cb * tg : cb / 4) + 3) / 4;
          }
    }
  }
  TR_HIP(hipMalloc(&tr.zeros, 256 * sizeof(float)));
  TR_HIP(hipMemset(tr.zeros, 0, 256 * sizeof(float)));
  size_t ns = 0;
  for (Layer& L : tr.layers) ns += 4 * (size_t)L.bn.C;
  TR_HIP(hipMalloc(&tr.stats, ns * sizeof(float)));
  TR_HIP(hipMemset(tr.stats, 0, ns * sizeof(float)));
  TR_HIP(hipMalloc(&tr.bn_partial, (size_t)128 * 256 * 2 * sizeof(double)));
  TR_HIP(hipMalloc(&tr.bn_counter, 128 * sizeof(unsigned)));
  TR_HIP(hipMemset(tr.bn_counter, 0, 128 * sizeof(unsigned)));
  size_t wtot = 0;
  for (Layer& L : tr.layers) {
    L.wg.partial_off = wtot;
    wtot += (size_t)wg_rows_cap(L.wg.out_n) * (size_t)L.wg.out_n;  // what the launches can reach (512 rows each was 550 MB)
  }
  tr.wg_partial_floats = wtot;
  TR_HIP(hipMalloc(&tr.wg_partial, wtot * sizeof(float)));
  const size_t hb = (size_t)((T0 + 255) / 256) * B;
  TR_HIP(hipMalloc(&tr.head_partial, hb * 28 * sizeof(double)));
  TR_HIP(hipMalloc(&tr.head_sums, 32 * sizeof(double)));
  TR_HIP(hipMemset(tr.head_sums, 0, 32 * sizeof(double)));
  TR_HIP(hipMalloc(&tr.head_stage, 64 * 28 * sizeof(double)));
  const size_t dense = (size_t)B * 3 * T0;
  TR_HIP(hipMalloc(&tr.x_dev, dense * sizeof(float)));
  TR_HIP(hipMalloc(&tr.y_dev, dense * sizeof(float)));
  TR_HIP(hipMalloc(&tr.p_dev, dense * sizeof(float)));
  for (Layer& L : tr.layers)
    for (ConvOp* op : {&L.fwd, &L.dgrad})
      if (op->used && op->lds_bytes > 48 * 1024)
        TR_HIP(hipFuncSetAttribute(op->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)op->lds_bytes));
  // Non-blocking: a blocking stream synchronises implicitly with the legacy default stream, so every event a binder
  // records there for the step's inputs (volpick_amd/train.py: torch's current stream) serialised the step against its
  // predecessor on the device -- 2.75 instead of 1.72 ms per 512-window step (tools/train_probe.py).  The uploads above ran
  // on the default stream: they are complete before the first step can be enqueued.
  TR_HIP(hipDeviceSynchronize());
  // The weight gradients' stream gets the LOWEST priority, the main chain the default one: streams of different priority
  // never share a hardware queue (two default-priority streams created after other streams of the process had been
  // destroyed did: the step ran 1.96 instead of 1.56 ms, its two chains serialised -- tools/train_after_forward_probe.py),
  // and where both have workgroups waiting the chain the step's length hangs on goes first.
  {
    int pr_least = 0, pr_greatest = 0;
    TR_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    TR_HIP(hipStreamCreateWithFlags(&tr.stream, hipStreamNonBlocking));
    TR_HIP(hipStreamCreateWithPriority(&tr.stream_wg, hipStreamNonBlocking, pr_least));  // (main chain at the highest: the same)
  }
  // consumed by another stream of this device only: a device-scope release at the event is enough (the default release
  // to the system writes the L2 back: ~5 us in front of the stream's next launch, once per layer)
  const unsigned ev_dev = hipEventDisableTiming | hipEventReleaseToDevice;
  for (int i = 0; i < NLAYER; ++i) TR_HIP(hipEventCreateWithFlags(&tr.ev_gz[i], ev_dev));
  TR_HIP(hipEventCreateWithFlags(&tr.ev_wg, ev_dev));
  TR_HIP(hipEventCreateWithFlags(&tr.ev_wg1, ev_dev));
  for (hipEvent_t& e : tr.ev_inputs) TR_HIP(hipEventCreateWithFlags(&e, ev_dev));
  return VP_OK;
}

// Returns the number of per-workgroup BatchNorm sums the launch left in tr.conv_stat (want_stat, bf16-MFMA form), else 0.
int run_conv(Trainer& tr, const ConvOp& op, int B, bool want_stat = false) {
  ++g_launches;
  ConvArgs a{};
  const Tensor& s1 = tr.tensors[op.src1];
  a.src1 = s1.p;
  a.ls1 = s1.ls;
  a.ws1 = (long)s1.win_stride();
  if (op.src2 >= 0) {
    const Tensor& s2 = tr.tensors[op.src2];
    a.src2 = s2.p;
    a.ls2 = s2.ls;
    a.ws2 = (long)s2.win_stride();
  }
  const Tensor& d = tr.tensors[op.dst];
  a.dst = d.p;
  a.lsd = d.ls;
  a.wsd = (long)d.win_stride();
  a.dst_halo = HALO;
  a.afrag = tr.frag + op.frag_off;
  a.bias = op.bias_off >= 0 ? tr.w + op.bias_off : tr.zeros;
  a.win_per_set = B;
  a.n_windows = B;
  a.l_out = op.l_out;
  a.l_dst = op.l_out;
  static const bool b16 = [] { const char* e = getenv("VP_CONV_B16"); return !e || atoi(e) != 0; }();
  static const bool estat = [] { const char* e = getenv("VP_CONV_STAT"); return !e || atoi(e) != 0; }();
  if (op.launch_b16 && tr.frag3 && b16) {
    const bool st = want_stat && estat && tr.conv_stat;
    op.launch_b16(a, tr.frag3 + op.a3_off, st ? tr.conv_stat : nullptr, op.cols, tr.stream);
    return st ? ((op.cols + op.tn - 1) / op.tn) * B : 0;
  }
  op.launch(a, op.cols, tr.stream);
  return 0;
}

BnArgs bn_args(Trainer& tr, const BnOp& b, int B) {
  BnArgs a{};
  a.z = tr.rows(b.z);
  a.a = tr.rows(b.a);
  a.gz = tr.rows(b.gz);
  a.ga1 = tr.rows(b.ga1, b.ga1_ch);
  a.ga2 = b.ga2 >= 0 ? tr.rows(b.ga2, b.ga2_ch) : Rows{nullptr, 0, 0};
  a.C = b.C;
  a.B = B;
  a.Lz = b.Lz;
  a.La = b.La;
  a.crop = b.crop;
  a.gamma = tr.w + b.gamma_off;
  a.beta = tr.w + b.beta_off;
  a.running_mean = tr.w + b.rm_off;
  a.running_var = tr.w + b.rv_off;
  a.stats = tr.stats + b.stats_off;
  a.partial = tr.bn_partial;
  {  // 64 >> sl2 rows per wave and trip, ~2 k workgroups per launch at most
    const int nv = (b.Lz + 7) / 8;
    a.sl2 = nv <= 2 ? 1 : nv <= 4 ? 2 : nv <= 8 ? 3 : nv <= 16 ? 4 : nv <= 32 ? 5 : 6;
    const int rows_per_block = 4 * (64 >> a.sl2);
    int gb = 2048 / b.C;
    if (gb > (B + rows_per_block - 1) / rows_per_block) gb = (B + rows_per_block - 1) / rows_per_block;
    a.GB = gb < 1 ? 1 : (gb > 256 ? 256 : gb);
  }
  a.g_gamma = tr.grad + b.gamma_off;
  a.g_beta = tr.grad + b.beta_off;
  a.eps = tr.bn_eps;
  a.momentum = tr.bn_momentum;
  return a;
}

// BatchNorm passes: the crop of the ConvTranspose layers (0, 1 or 2 samples) selects the instantiation
#define BN_CROP_LAUNCH(KERNEL, GRID, NTH)                                              \
  do {                                                                                 \
    if (a.crop == 0) {                                                                 \
      TRL((KERNEL<T, 0>), GRID, dim3(NTH), 0, s, a);                     \
    } else if (a.crop == 1) {                                                          \
      TRL((KERNEL<T, 1>), GRID, dim3(NTH), 0, s, a);                     \
    } else {                                                                           \
      TRL((KERNEL<T, 2>), GRID, dim3(NTH), 0, s, a);                     \
    }                                                                                  \
  } while (0)
// the same with an event bound to the launch's own completion (hipExtLaunchKernel's stop event): what another stream
// waits for is this kernel's end, without the marker packet a hipEventRecord puts into the queue (6-8 us in front of the
// next launch of this stream, eighteen times per step)
#define BN_CROP_LAUNCH_EV(KERNEL, GRID, NTH, EV)                                                         \
  do {                                                                                                   \
    ++g_launches;                                                                                        \
    if (a.crop == 0) {                                                                                   \
      hipExtLaunchKernelGGL((KERNEL<T, 0>), GRID, dim3(NTH), 0, s, nullptr, EV, 0, a);                   \
    } else if (a.crop == 1) {                                                                            \
      hipExtLaunchKernelGGL((KERNEL<T, 1>), GRID, dim3(NTH), 0, s, nullptr, EV, 0, a);                   \
    } else {                                                                                             \
      hipExtLaunchKernelGGL((KERNEL<T, 2>), GRID, dim3(NTH), 0, s, nullptr, EV, 0, a);                   \
    }                                                                                                    \
  } while (0)
template <class T>
void bn_forward_v(const BnArgs& a, hipStream_t s) {
  const dim3 grid(a.C, a.GB);
  if (!a.fpart) TRL(bnv_stats_partial_kernel<T>, grid, dim3(256), 0, s, a);  // (else the conv launch left the sums)
  BN_CROP_LAUNCH(bnv_apply_kernel, grid, 256);
}
template <class T>
void bn_backward_v(const BnArgs& a, hipStream_t s, hipEvent_t gz_done) {
  const dim3 grid(a.C, a.GB);
  BN_CROP_LAUNCH(bnv_bwd_partial_kernel, grid, 256);
  BN_CROP_LAUNCH_EV(bnv_bwd_apply_kernel, grid, 256, gz_done);
}
#undef BN_CROP_LAUNCH
#undef BN_CROP_LAUNCH_EV

// Partial results of a weight gradient: one row of out_n floats per workgroup, folded by sum_rows_multi_kernel.  With 256
// rows for every tensor of > 10 k weights the rows were 267 MB per step -- written by the weight-gradient launches, read
// again by the fold (68 us, on the step's critical path) -- for 1.1 MB of gradients: the rows are capped by their bytes.
static int wg_rows_cap(int out_n) {
  constexpr long budget = 1L << 22;  // same-box sweep: 1.64 ms per step unbounded, 1.60 at 8 M floats, 1.57 at 4-6 M, 1.60 at 2-3 M
                                     // (below that the deep layers' launches have too few workgroups and end behind the main chain)
  int cap = out_n > 10000 ? 256 : 512;
  while (cap > 32 && (long)cap * out_n > budget) cap >>= 1;
  return cap;
}

int forward_backward(Trainer& tr, const float* x_dev, const float* y_dev, int B, bool update, float lr) {
  hipStream_t s = tr.stream;
  g_launches = 0;
  TRL(gather_pack_kernel, dim3((unsigned)((tr.frag_n + 255) / 256)), dim3(256), 0, s, tr.frag_idx, tr.w,
                     tr.frag, (long)tr.frag_n);
  if (tr.b16_blocks) TRL(conv_b16_pack_kernel, dim3(tr.b16_blocks), dim3(256), 0, s, tr.b16_jobs);
  if (tr.bf16) {
    TRL(load_rows_bf16_kernel, dim3((T0 + 511) / 512, 3, B), dim3(256), 0, s, x_dev, tr.rows(tr.t_x), 3, T0);
  } else {
    TRL(load_rows_kernel, dim3((T0 + 255) / 256, 3, B), dim3(256), 0, s, x_dev, tr.rows(tr.t_x), 3, T0);
  }
  for (Layer& L : tr.layers) {
    const int n_stat = run_conv(tr, L.fwd, B, true);
    BnArgs a = bn_args(tr, L.bn, B);
    a.fpart = n_stat ? tr.conv_stat : nullptr;
    a.n_fpart = n_stat;
    if (tr.bf16) {
      bn_forward_v<bf16_t>(a, s);
    } else {
      bn_forward_v<float>(a, s);
    }
  }
  {
    HeadArgs2 h{};
    h.a = tr.rows(tr.layers[17].bn.a);
    h.ga = tr.rows(tr.t_ga_last);
    h.y = y_dev;
    h.p = tr.p_dev;
    h.w = tr.w + tr.poff.at("out.weight");
    h.b = tr.w + tr.poff.at("out.bias");
    h.partial = tr.head_partial;
    h.B = B;
    h.T = T0;
    h.eps = tr.loss_eps;
    int gx = (T0 + 255) / 256;
    const int slot = (int)(tr.seq % Trainer::EV_RING);
    hipEvent_t ev_in = tr.ev_inputs[slot];
    tr.ev_inputs_seq[slot] = tr.seq++;
    if (tr.bf16) {
      gx = (T0 + 1023) / 1024;
      ++g_launches;
      hipExtLaunchKernelGGL((head_fwd_bwd_pair_kernel<bf16_t, 4>), dim3(gx, B), dim3(256), 0, s, nullptr, ev_in, 0, h);
    } else {
      ++g_launches;
      hipExtLaunchKernelGGL(head_fwd_bwd_kernel, dim3(gx, B), dim3(256), 0, s, nullptr, ev_in, 0, h);
    }
    // load_rows above read x, this launch read y: nothing behind its end (ev_inputs) touches the caller's buffers
    // two stages: 64 row groups, then the 64 group sums
    TRL((sum_rows_kernel<double, double>), dim3(1, 64), dim3(256), 0, s, tr.head_partial, gx * B, 28,
                       tr.head_stage);
    TRL((sum_rows_kernel<double, double>), dim3(1, 1), dim3(256), 0, s, tr.head_stage, 64, 28,
                       tr.head_sums);
    TRL(head_final_kernel, dim3(1), dim3(32), 0, s, tr.head_sums, tr.head_sums + 28,
                       tr.grad + tr.poff.at("out.bias"), tr.grad + tr.poff.at("out.weight"));
  }
  // The partial rows of the weight gradients are folded by sum_rows_multi_kernel: those of the layers >= FOLD_EARLY each
  // right behind its weight gradient on that stream (where the stream still waits for the main chain between its
  // launches), the rest in one launch at the end of the step, in front of Adam, alone on the chip.
  // (same-box sweep of FOLD_EARLY: none 1.414 ms per step, 13: 1.389, 11: 1.405, 9: 1.411, 7: 1.418, 5: 1.436 -- the folds of
  // the deep layers' 15 MB rows take more from the main chain beside them than they save at the end)
  constexpr int FOLD_EARLY = 13;
  SumJobs jobs{}, jobs_last{};
  int sum_blocks_main = 0, sum_blocks_last = 0;
  // Layers whose gz gets an event for the weight-gradient stream; the layers between hand their launch to the next event.
  // An event costs the main chain ~5 us (the launch behind a kernel with a completion signal starts that much later):
  // every other layer above level 0, every layer of the last four (their weight gradients are the step's tail).
  constexpr unsigned ev_mask = 0x2aaaf;
  // ev_wg1 is recorded at li == 1 on the premise that layer 1's launch has just been flushed, li == 0 flushes the rest, and the
  // first layer handled (NLAYER - 1) opens the chain: a retuned mask must keep those three bits
  static_assert((ev_mask & 3u) == 3u && ((ev_mask >> (NLAYER - 1)) & 1u), "ev_mask: layers 0, 1 and NLAYER - 1 carry an event");
  struct { const WgradOp* w; WgradArgs g; int grid, li; } held[NLAYER];
  int n_held = 0;
  static_assert(NLAYER <= MAX_SUM_JOBS, "one sum job per layer");
  for (int li = NLAYER - 1; li >= 0; --li) {
    Layer& L = tr.layers[li];
    const BnArgs a = bn_args(tr, L.bn, B);
    if (tr.bf16) {
      bn_backward_v<bf16_t>(a, s, ((ev_mask >> li) & 1) ? tr.ev_gz[li] : nullptr);  // the event: gz of this layer is complete
    } else {
      bn_backward_v<float>(a, s, ((ev_mask >> li) & 1) ? tr.ev_gz[li] : nullptr);
    }
    {
      const WgradOp& w = L.wg;
      WgradArgs g{};
      g.lo = tr.rows(w.lo);
      g.hi1 = tr.rows(w.hi1);
      g.lim_hi1 = tr.tensors[w.hi1].ls - HALO;
      if (w.hi2 >= 0) {
        g.hi2 = tr.rows(w.hi2);
        g.lim_hi2 = tr.tensors[w.hi2].ls - HALO;
      }
      g.Ln = w.Ln;
      g.off = w.off;
      g.B = B;
      g.chunks = (w.Ln + w.TT - 1) / w.TT;
      g.partial = tr.wg_partial + w.partial_off;
      const int items = ((B + w.WB - 1) / w.WB) * g.chunks;
      const int cap = wg_rows_cap(w.out_n);  // one partial result per workgroup: fewer, longer-lived workgroups for the big weight tensors
      const int grid = items < cap ? items : cap;
      ++g_launches;
      held[n_held].w = &w;
      held[n_held].g = g;
      held[n_held].li = li;
      held[n_held++].grid = grid;
      if ((ev_mask >> li) & 1) {
        (void)hipStreamWaitEvent(tr.stream_wg, tr.ev_gz[li], 0);
        for (int k = 0; k < n_held; ++k) {
          const WgradOp& hw = *held[k].w;
          hw.launch(held[k].g, held[k].grid, tr.stream_wg);
          if (held[k].li >= FOLD_EARLY) {
            SumJobs one{};
            one.count = 1;
            SumJob& ej = one.job[0];
            ej.partial = held[k].g.partial;
            ej.out = tr.grad + hw.grad_off;
            ej.rows = held[k].grid;
            ej.n = hw.out_n;
            ej.first_block = 0;
            ej.cq = sum_job_cq(held[k].grid);
            ++g_launches;
            hipLaunchKernelGGL(sum_rows_multi_kernel, dim3((hw.out_n + 4 * ej.cq - 1) / (4 * ej.cq)), dim3(256), 0, tr.stream_wg, one);
          }
        }
        n_held = 0;
      }
      if (li == 1) (void)hipEventRecord(tr.ev_wg1, tr.stream_wg);  // (layer 1 carries an event: its launch is out)
      if (li < FOLD_EARLY) {
        SumJobs& jset = li == 0 ? jobs_last : jobs;
        int& sum_blocks = li == 0 ? sum_blocks_last : sum_blocks_main;
        SumJob& jb = jset.job[jset.count++];
        jb.partial = g.partial;
        jb.out = tr.grad + w.grad_off;
        jb.rows = grid;
        jb.n = w.out_n;
        jb.first_block = sum_blocks;
        jb.cq = sum_job_cq(grid);
        sum_blocks += (w.out_n + 4 * jb.cq - 1) / (4 * jb.cq);
      }
    }
    if (li == 0) {  // conv bias of `inc`: sum of gz per channel (zero up to rounding: BatchNorm removes the mean)
      const int GB = B < 64 ? B : 64;
      if (tr.bf16) {
        TRL(channel_sum_partial_v_kernel<bf16_t>, dim3(8, GB), dim3(256), 0, s, tr.rows(L.bn.gz), B, T0, GB,
                           tr.bn_partial);
      } else {
        TRL(channel_sum_partial_v_kernel<float>, dim3(8, GB), dim3(256), 0, s, tr.rows(L.bn.gz), B, T0, GB,
                           tr.bn_partial);
      }
      TRL((sum_rows_kernel<double, float>), dim3(1, 1), dim3(256), 0, s, tr.bn_partial, GB, 8,
                         tr.grad + tr.poff.at("inc.bias"));
    }
    if (L.dgrad.used) run_conv(tr, L.dgrad, B);
  }
  // the fold of everything but the first layer's rows runs beside that layer's weight gradient, the step's last launch on
  // the other stream; its own few rows behind it
  (void)hipStreamWaitEvent(s, tr.ev_wg1, 0);
  TRL(sum_rows_multi_kernel, dim3(sum_blocks_main), dim3(256), 0, s, jobs);
  (void)hipEventRecord(tr.ev_wg, tr.stream_wg);
  (void)hipStreamWaitEvent(s, tr.ev_wg, 0);  // every weight gradient's partial rows are written
  TRL(sum_rows_multi_kernel, dim3(sum_blocks_last), dim3(256), 0, s, jobs_last);
  if (update) {
    tr.step += 1;
    const float bc1 = 1.f - powf(tr.beta1, (float)tr.step);
    const float bc2 = 1.f - powf(tr.beta2, (float)tr.step);
    TRL(adam_kernel, dim3((unsigned)((tr.n_params + 255) / 256)), dim3(256), 0, s, tr.w, tr.grad,
                       tr.adam_m, tr.adam_v, tr.mask, tr.ema, (int)tr.n_params, lr, tr.beta1, tr.beta2, tr.adam_eps, bc1,
                       sqrtf(bc2), tr.ema_decay);
  }
  tr.launches_last_step = g_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("training step: kernel launch failed: %s", hipGetErrorString(e));
    return VP_ERR_HIP;
  }
  return VP_OK;
}

}  // namespace
}  // namespace vp

using namespace vp;

extern "C" {

int vp_train_create(int device_id, int model_kind, const float* weights, size_t n_floats, int max_batch,
                    vp_trainer** out) {
  return vp_train_create_dtype(device_id, model_kind, weights, n_floats, max_batch, VP_TRAIN_FP32, out);
}

int vp_train_dtype(const vp_trainer* h) {
  return h ? (reinterpret_cast<const Trainer*>(h)->bf16 ? VP_TRAIN_BF16 : VP_TRAIN_FP32) : VP_ERR_INVALID;
}

int vp_train_create_dtype(int device_id, int model_kind, const float* weights, size_t n_floats, int max_batch, int dtype,
                          vp_trainer** out) {
  VP_REQUIRE(out && weights && max_batch > 0, "vp_train_create: bad argument");
  VP_REQUIRE(dtype == VP_TRAIN_FP32 || dtype == VP_TRAIN_BF16, "vp_train_create: dtype %d is neither VP_TRAIN_FP32 nor VP_TRAIN_BF16", dtype);
  if (model_kind != VP_MODEL_PHASENET) {
    set_error("vp_train_create: only PhaseNet has a training step");
    return VP_ERR_UNSUPPORTED;
  }
  auto tr = std::make_unique<Trainer>();
  tr->device = device_id;
  tr->max_batch = max_batch;
  tr->bf16 = dtype == VP_TRAIN_BF16;
  tr->esize = tr->bf16 ? 2 : 4;
  int rc = tr->bf16 ? build_plan<1>(*tr) : build_plan<0>(*tr);
  if (rc != VP_OK) return rc;
  VP_REQUIRE(n_floats == tr->n_params, "vp_train_create: expected %zu weights, got %zu", tr->n_params, n_floats);
  VP_HIP(hipSetDevice(device_id));
  rc = upload(*tr, weights);
  if (rc != VP_OK) return rc;
  *out = reinterpret_cast<vp_trainer*>(tr.release());
  return VP_OK;
}

int vp_train_destroy(vp_trainer* h) {
  delete reinterpret_cast<Trainer*>(h);
  return VP_OK;
}

int vp_train_set_hyper(vp_trainer* h, float beta1, float beta2, float adam_eps, float bn_momentum, float loss_eps) {
  VP_REQUIRE(h, "vp_train_set_hyper: null handle");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  tr.beta1 = beta1;
  tr.beta2 = beta2;
  tr.adam_eps = adam_eps;
  tr.bn_momentum = bn_momentum;
  tr.loss_eps = loss_eps;
  return VP_OK;
}

int vp_train_set_ema(vp_trainer* h, float decay) {
  VP_REQUIRE(h && decay >= 0.f && decay < 1.f, "vp_train_set_ema: bad argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_HIP(hipSetDevice(tr.device));
  VP_HIP(hipStreamSynchronize(tr.stream));
  if (!tr.ema) VP_HIP(hipMalloc(&tr.ema, tr.n_params * sizeof(float)));
  // starts at the current weights; on the trainer's (non-blocking) stream, so that the next step's Adam / EMA update is
  // ordered behind the copy (a null-stream device-to-device copy may return before it has run)
  VP_HIP(hipMemcpyAsync(tr.ema, tr.w, tr.n_params * sizeof(float), hipMemcpyDeviceToDevice, tr.stream));
  VP_HIP(hipStreamSynchronize(tr.stream));
  tr.ema_decay = decay;
  return VP_OK;
}

int vp_train_step(vp_trainer* h, const float* x, const float* y, int mem, int B, float lr, int update, double* loss) {
  VP_REQUIRE(h && x && y, "vp_train_step: null argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_REQUIRE(B >= 2 && B <= tr.max_batch, "vp_train_step: batch %d outside [2, %d]", B, tr.max_batch);
  VP_HIP(hipSetDevice(tr.device));
  const float *xd = x, *yd = y;
  if (mem == VP_MEM_HOST) {
    const size_t n = (size_t)B * 3 * T0 * sizeof(float);
    VP_HIP(hipMemcpyAsync(tr.x_dev, x, n, hipMemcpyHostToDevice, tr.stream));
    VP_HIP(hipMemcpyAsync(tr.y_dev, y, n, hipMemcpyHostToDevice, tr.stream));
    xd = tr.x_dev;
    yd = tr.y_dev;
  }
  const int rc = forward_backward(tr, xd, yd, B, update != 0, lr);
  if (rc != VP_OK) return rc;
  if (loss) {
    VP_HIP(hipMemcpyAsync(loss, tr.head_sums + 28, sizeof(double), hipMemcpyDeviceToHost, tr.stream));
    VP_HIP(hipStreamSynchronize(tr.stream));
  }
  return VP_OK;
}

// Makes `stream` (a hipStream_t of the same device; NULL = the legacy default stream) wait until the latest vp_train_step
// has read its x / y for the last time: work enqueued on `stream` afterwards may overwrite or free them.  No host wait.
int vp_train_wait_inputs_consumed(vp_trainer* h, void* stream) {
  VP_REQUIRE(h, "vp_train_wait_inputs_consumed: null handle");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_HIP(hipSetDevice(tr.device));
  if (tr.seq > 0)
    VP_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), tr.ev_inputs[(tr.seq - 1) % Trainer::EV_RING], 0));
  return VP_OK;
}

// The highest step number (0 = the first vp_train_step of this trainer) whose reads of x / y have completed, -1 if none is
// known to have: a caller that keeps device batches alive for queued steps drops the ones up to it.  No host wait.
long long vp_train_inputs_consumed_upto(vp_trainer* h) {
  if (!h) return -1;
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  if (hipSetDevice(tr.device) != hipSuccess) return -1;
  long long best = -1;
  for (int i = 0; i < Trainer::EV_RING; ++i)
    if (tr.ev_inputs_seq[i] > best && hipEventQuery(tr.ev_inputs[i]) == hipSuccess) best = tr.ev_inputs_seq[i];
  (void)hipGetLastError();  // hipErrorNotReady of the pending ones is not an error
  return best;
}

// Steps enqueued on this trainer so far: the step a vp_train_step call just queued has the number (this - 1).
long long vp_train_steps_enqueued(const vp_trainer* h) { return h ? reinterpret_cast<const Trainer*>(h)->seq : 0; }

int vp_train_synchronize(vp_trainer* h) {
  VP_REQUIRE(h, "vp_train_synchronize: null handle");
  VP_HIP(hipStreamSynchronize(reinterpret_cast<Trainer*>(h)->stream));
  return VP_OK;
}

// which: 0 weights (incl. BN running statistics), 1 gradients of the last step, 2 Adam m, 3 Adam v, 4 EMA weights
int vp_train_read(vp_trainer* h, int which, float* out, size_t n_floats) {
  VP_REQUIRE(h && out, "vp_train_read: null argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_REQUIRE(n_floats == tr.n_params && which >= 0 && which <= 4, "vp_train_read: bad size or selector");
  VP_REQUIRE(which != 4 || tr.ema, "vp_train_read: EMA is off (vp_train_set_ema)");
  const float* src = which == 0 ? tr.w : which == 1 ? tr.grad : which == 2 ? tr.adam_m : which == 3 ? tr.adam_v : tr.ema;
  VP_HIP(hipSetDevice(tr.device));
  VP_HIP(hipStreamSynchronize(tr.stream));
  VP_HIP(hipMemcpy(out, src, n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return VP_OK;
}

int vp_train_write_weights(vp_trainer* h, const float* weights, size_t n_floats) {
  VP_REQUIRE(h && weights, "vp_train_write_weights: null argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_REQUIRE(n_floats == tr.n_params, "vp_train_write_weights: bad size");
  VP_HIP(hipSetDevice(tr.device));
  VP_HIP(hipStreamSynchronize(tr.stream));
  VP_HIP(hipMemcpy(tr.w, weights, n_floats * sizeof(float), hipMemcpyHostToDevice));
  return VP_OK;
}

// predictions (B, 3, 3001) of the last step's forward pass
int vp_train_predictions(vp_trainer* h, float* out, int B) {
  VP_REQUIRE(h && out, "vp_train_predictions: null argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_REQUIRE(B > 0 && B <= tr.max_batch, "vp_train_predictions: bad batch");
  VP_HIP(hipSetDevice(tr.device));
  VP_HIP(hipStreamSynchronize(tr.stream));
  VP_HIP(hipMemcpy(out, tr.p_dev, (size_t)B * 3 * T0 * sizeof(float), hipMemcpyDeviceToHost));
  return VP_OK;
}

int vp_train_tensor_count(const vp_trainer* h) {
  return h ? (int)reinterpret_cast<const Trainer*>(h)->tensors.size() : VP_ERR_INVALID;
}
int vp_train_tensor_info(const vp_trainer* h, int index, const char** name, int* channels, int* length) {
  VP_REQUIRE(h, "vp_train_tensor_info: null handle");
  const Trainer& tr = *reinterpret_cast<const Trainer*>(h);
  VP_REQUIRE(index >= 0 && index < (int)tr.tensors.size(), "vp_train_tensor_info: bad index");
  if (name) *name = tr.tensors[index].name.c_str();
  if (channels) *channels = tr.tensors[index].C;
  if (length) *length = tr.tensors[index].L;
  return VP_OK;
}
// dense (B, C, L) copy of an activation / gradient tensor of the last step (parity tests)
int vp_train_tensor_read(vp_trainer* h, int index, int B, float* out) {
  VP_REQUIRE(h && out, "vp_train_tensor_read: null argument");
  Trainer& tr = *reinterpret_cast<Trainer*>(h);
  VP_REQUIRE(index >= 0 && index < (int)tr.tensors.size() && B > 0 && B <= tr.max_batch, "vp_train_tensor_read: bad argument");
  const Tensor& t = tr.tensors[index];
  VP_HIP(hipSetDevice(tr.device));
  VP_HIP(hipStreamSynchronize(tr.stream));
  if (tr.bf16) {
    const size_t n = (size_t)B * t.C * t.L;
    std::vector<uint16_t> raw(n);
    VP_HIP(hipMemcpy2D(raw.data(), (size_t)t.L * 2, reinterpret_cast<const uint16_t*>(t.p) + HALO, (size_t)t.ls * 2, (size_t)t.L * 2,
                       (size_t)B * t.C, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      const uint32_t u = (uint32_t)raw[i] << 16;
      memcpy(out + i, &u, 4);
    }
    return VP_OK;
  }
  VP_HIP(hipMemcpy2D(out, (size_t)t.L * sizeof(float), t.p + HALO, (size_t)t.ls * sizeof(float), (size_t)t.L * sizeof(float),
                     (size_t)B * t.C, hipMemcpyDeviceToHost));
  return VP_OK;
}
int vp_train_launch_count(const vp_trainer* h) { return h ? reinterpret_cast<const Trainer*>(h)->launches_last_step : 0; }

void* vp_train_stream(const vp_trainer* h) { return h ? reinterpret_cast<const Trainer*>(h)->stream : nullptr; }

}  // extern "C"
