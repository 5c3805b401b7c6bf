// Device plan of one picker model: activation tensors, packed weights and the ordered list
// of kernel launches of one forward pass.  Built host-only (plan_*), then uploaded.
#pragma once
#include <functional>
#include <memory>

#include "conv_mfma.h"
#include "prepost.h"

namespace vp {

struct Net;

struct HostBlob {  // host staging of a device constant array
  std::vector<float> h;
  float* d = nullptr;
};

struct ConvLayer {
  std::string name;
  ConvGeom g;
  int src1 = -1, src2 = -1, dst = -1, dst2 = -1, res = -1;  // tensor ids; dst == kDenseOut -> net.y
  int cols = 0, l_out = 0, l_dst = 0;
  int n_sets = 1;  // weight sets (EQT decoders x3); windows = n_sets * B
  HostBlob afrag, bias, e0, e1, e2;
  HostBlob* afrag_q4 = nullptr;  // ConvCfg AQ4: the A operand regrouped for 16-byte loads (what the kernel reads)
  int (*launch)(const ConvArgs&, int, hipStream_t) = nullptr;
  const void* kernel = nullptr;
  size_t lds_bytes = 0;
  double flops_per_window = 0;  // algorithmic 2*MAC of the layer (not the padded MFMA work)
};

struct Step {
  std::string name;
  std::function<int(Net&, int /*B*/, hipStream_t)> run;
  double flops_per_window = 0;
  // Work the launch ISSUES per window (whole tiles, padded channels, recomputed halos, folded taps), by the pipe it runs on:
  double issued_f32 = 0;   // FLOP as v_mfma_f32_16x16x4_f32 (2,048 each)
  double issued_bf16 = 0;  // FLOP as v_mfma_f32_16x16x32_bf16 (16,384 each; an exact three-piece product group is SIX of them)
  double issued_valu = 0;  // FLOP on the vector ALUs (direct convolutions, recurrences, attention scores)
  double issued_flops_per_window = 0;  // fp32-equivalent of the three: issued_f32 + issued_bf16 / 6 + issued_valu (set_issued)
  void set_issued(double f32, double bf16, double valu) {
    issued_f32 = f32, issued_bf16 = bf16, issued_valu = valu;
    issued_flops_per_window = f32 + bf16 / 6.0 + valu;
  }
  // A launch whose tiling depends on the output range the caller keeps (Net::out_lo / out_hi) says what it issues for a
  // given range through this pure function; the fields above then hold the whole-row figure and run() never touches them.
  std::function<void(const Net&, int /*out_lo*/, int /*out_hi*/, double* /*f32, bf16, valu*/)> issued_for_range;
};

constexpr int kDenseOut = -2;

// The A operand of a conv layer regrouped for 16-byte loads (conv_lds_q4): [mt][step][64] -> [mt][step / 4][64][4],
// step = channel block * taps + tap; needs (channel blocks * taps) % 4 == 0.
std::vector<float> regroup_afrag4(const ConvLayer& L);
// three-piece bf16 operands of every weight set of a layer (conv_b3.h) / the fp32 operand with phase-major rows (net.hip)
std::vector<float> b3_operand(const ConvLayer& L, bool mperm);
std::vector<float> regroup_afrag4_phase_major(const ConvLayer& L);

struct Net {
  int model_kind = 0;
  vp_config cfg{};
  int in_samples = 0, n_out = 3;
  int max_batch = 0;
  std::vector<Tensor> tensors;
  std::vector<int> tensor_sets;  // capacity multiplier per tensor (3 for batched EQT decoders)
  std::vector<std::unique_ptr<ConvLayer>> convs;
  std::vector<HostBlob*> blobs;  // every constant array to upload (owned by layers / extra)
  std::vector<std::unique_ptr<HostBlob>> extra;
  std::vector<Step> steps;
  std::vector<std::pair<const void*, size_t>> extra_kernels;  // (kernel, dynamic LDS bytes) needing the LDS attribute
  std::map<std::string, HostBlob*> named;  // constant arrays a later fusion pass looks up by name (e.g. "decoder.2.edge.w")
  int input = -1;          // tensor id of the normalised haloed input windows
  float* y = nullptr;      // dense output [max_batch][n_out][in_samples]
  float* arena = nullptr;  // one device allocation for tensors + constants
  size_t arena_floats = 0;
  double flops_per_window = 0;
  int warm_launches = 1;  // launches of the fused PhaseNet kernels that still pre-touch the weights (first launch of a plan:
                          // cold L2); in steady state the weights are L2-resident and the touch only delays 8 workgroups
  HostBlob* debug_clock = nullptr;  // fused PhaseNet core: per-layer shader-clock stamps (debug plan flag)
  HostBlob* win_flags = nullptr;    // [max_batch]: 1 where annotate_batch_pre met a non-finite window (its predictions become NaN, as the reference's)
  bool fused_pre = false;           // the plan's first launch can gather + normalise its windows itself (pn_window_kernel, eqt_front_kernel)
  int out_lo = 0, out_hi = 0;       // set by the caller around run(): the output samples [out_lo, out_hi) of every window that it keeps
                                    // (annotate / classify blind the rest); 0, 0 = all.  A last launch that tiles the time axis skips the
                                    // tiles outside (eqt_tail3_kernel)
  bool fused_pre_poisons = false;   // ... and that launch writes the NaN predictions of a non-finite window itself (pn_window_kernel)
  bool poison_in_plan = false;      // the plan's LAST launch reads win_flags and writes NaN for flagged windows, whoever set the
                                    // flags (eqt_tail3_kernel): poison_kernel is never launched
  const PreArgs* pre = nullptr;     // set by the caller around run() when fused_pre: where the windows of this batch come from

  int add_tensor(const std::string& name, int C, int L, int sets = 1);
  HostBlob* add_blob(std::vector<float> v);
  void need(int tensor, int phys_end) {
    if (tensor >= 0 && tensors[tensor].need < phys_end) tensors[tensor].need = phys_end;
  }

  template <class Cfg>
  ConvLayer* add_conv(const std::string& name, int src1, int src2, int dst, int cols, int l_out,
                      std::vector<float> afrag, std::vector<float> bias, int n_sets = 1) {
    auto L = std::make_unique<ConvLayer>();
    L->name = name;
    L->g = Cfg::geom();
    L->src1 = src1;
    L->src2 = src2;
    L->dst = dst;
    L->cols = cols;
    L->l_out = l_out;
    L->l_dst = l_out;
    L->n_sets = n_sets;
    L->afrag.h = std::move(afrag);
    L->bias.h = std::move(bias);
    L->launch = &launch_conv<Cfg>;
    L->kernel = reinterpret_cast<const void*>(&conv_mfma_kernel<Cfg>);
    L->lds_bytes = Cfg::LDS_FLOATS * sizeof(float);
    need(src1, L->g.src_need(cols));
    need(src2, L->g.src_need(cols));
    ConvLayer* raw = L.get();
    convs.push_back(std::move(L));
    if constexpr (Cfg::AQ4) raw->afrag_q4 = add_blob(regroup_afrag4(*raw));  // the kernel reads this copy (16-byte groups)
    add_conv_step(raw);
    return raw;
  }
  void add_conv_step(ConvLayer* L);
  int finalize_layout();  // host-only: row strides, arena offsets
  int upload();           // hipMalloc + copy constants + zero activations
  int run(int B, hipStream_t stream);
  void release();
};

int plan_phasenet(Net& net, const ParamView& pv);
int plan_phasenet_fused(Net& net, const ParamView& pv, int debug_flags);  // swaps the 18 layer steps for 3 fused launches; bit0: dump LDS intermediates, bit1: clock stamps
int plan_eqt(Net& net, const ParamView& pv);
int plan_eqt_fuse_res(Net& net);  // swaps the 14 ResCNN conv steps for one fused launch
int plan_eqt_fuse_tail_b3(Net& net);  // the same on the bf16 matrix cores, exact three-piece operands (eqt_tail_b3.hip)
int plan_eqt_fuse_tail(Net& net);  // swaps decoder.4 / .5 / .6+heads for one time-tiled fused launch (eqt_tail.hip)
int plan_eqt_fuse_front(Net& net, bool b3);  // swaps encoder.0 / .1 / .2 for one time-tiled fused launch (eqt_front.hip)
int plan_eqt_fuse_enc36_b3(Net& net);  // the same on the bf16 matrix cores, exact three-piece operands (eqt_enc36_b3.hip)
int plan_eqt_fuse_enc36(Net& net);  // swaps encoder.3 .. .6 for one launch per window (eqt_enc36.hip)
int plan_eqt_fuse_dec03(Net& net, bool b3);  // swaps decoder.0 / .1 / .2 / .2.edge / .3 for one launch per (decoder, window) row (eqt_dec03.hip)


// BatchNorm (eval) folded into the preceding conv: scale = gamma / sqrt(var + eps),
// shift = beta - mean * scale (+ conv bias * scale).
void bn_fold(const ParamView& pv, const std::string& bn, int C, float eps, const float* conv_bias,
             std::vector<float>* scale, std::vector<float>* shift);

}  // namespace vp
