// PhaseNet (SeisBench topology, volpick weights) as 18 launches of conv_mfma_kernel.
// Reference contract: model(x) on (B,3,3001) -> (B,3,3001) softmax over channels in
// `phases` order (volpick/model/eval_taks0.py:85-89).
// BatchNorm (eval) and its eps are folded into the conv weights at plan time; the 1x1
// output conv + softmax run in the epilogue of the last conv.
#include "net.h"

namespace vp {

namespace {
constexpr int T0 = 3001, T1 = 751, T2 = 188, T3 = 47, T4 = 12;

//                      CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF WM WN NW RELU EPI
using PN_inc = ConvCfg<3, 0, 8, 2, 8, 2, -3, 0, 1, 4, 8, 1, EPI_STORE>;
using PN_d0same = ConvCfg<8, 0, 8, 2, 8, 2, -3, 0, 1, 4, 8, 1, EPI_STORE>;
using PN_d0down = ConvCfg<8, 0, 8, 2, 11, 8, -3, 0, 1, 4, 2, 1, EPI_STORE>;
using PN_d1same = ConvCfg<8, 0, 16, 1, 7, 1, -3, 0, 1, 4, 4, 1, EPI_STORE>;
using PN_d1down = ConvCfg<16, 0, 16, 1, 7, 4, -2, 0, 1, 4, 1, 1, EPI_STORE>;
using PN_d2same = ConvCfg<16, 0, 32, 1, 7, 1, -3, 0, 2, 2, 3, 1, EPI_STORE>;
using PN_d2down = ConvCfg<32, 0, 32, 1, 7, 4, -1, 0, 2, 2, 1, 1, EPI_STORE>;
using PN_d3same = ConvCfg<32, 0, 64, 1, 7, 1, -3, 0, 4, 1, 3, 1, EPI_STORE>;
using PN_d3down = ConvCfg<64, 0, 64, 1, 7, 4, -2, 0, 4, 1, 1, 1, EPI_STORE>;
using PN_d4same = ConvCfg<64, 0, 128, 1, 7, 1, -3, 0, 4, 1, 1, 1, EPI_STORE>;
using PN_u0T = ConvCfg<128, 0, 64, 4, 2, 1, -1, -1, 4, 1, 1, 1, EPI_STORE>;
using PN_u0same = ConvCfg<64, 64, 64, 1, 7, 1, -3, 0, 4, 1, 3, 1, EPI_STORE>;
using PN_u1T = ConvCfg<64, 0, 32, 4, 2, 1, -1, -1, 4, 1, 3, 1, EPI_STORE>;
using PN_u1same = ConvCfg<32, 32, 32, 1, 7, 1, -3, 0, 2, 2, 3, 1, EPI_STORE>;
using PN_u2T = ConvCfg<32, 0, 16, 4, 2, 1, -1, -1, 4, 1, 4, 1, EPI_STORE>;
using PN_u2same = ConvCfg<16, 16, 16, 1, 7, 1, -3, 0, 1, 4, 4, 1, EPI_STORE>;
using PN_u3T = ConvCfg<16, 0, 8, 4, 2, 1, -1, -2, 2, 2, 4, 1, EPI_STORE>;
using PN_u3same = ConvCfg<8, 8, 8, 2, 8, 2, -3, 0, 1, 4, 8, 1, EPI_SOFTMAX3>;

struct Folded {
  std::vector<float> afrag, bias;
};

// Conv1d(bias optional) -> BatchNorm -> (ReLU in the kernel)
Folded fold_conv(const ParamView& pv, const std::string& conv, const std::string& bn, int cout, int cin, int K,
                 int stride, const ConvGeom& g, float eps, bool has_bias) {
  const float* W = pv.get(conv + ".weight");
  const float* b = has_bias ? pv.get(conv + ".bias") : nullptr;
  std::vector<float> scale, shift;
  bn_fold(pv, bn, cout, eps, b, &scale, &shift);
  Folded f;
  f.afrag = pack_afrag(amat_conv(W, cout, cin, K, stride, g.P, g.cinp(), scale.data()), g.M(), g.cinp(), g.taps);
  f.bias = shift;
  return f;
}

Folded fold_convT(const ParamView& pv, const std::string& conv, const std::string& bn, int cin, int cout,
                  const ConvGeom& g, float eps) {
  const float* Wt = pv.get(conv + ".weight");
  std::vector<float> scale, shift;
  bn_fold(pv, bn, cout, eps, nullptr, &scale, &shift);
  Folded f;
  f.afrag = pack_afrag(amat_convT_k7s4(Wt, cin, cout, g.cinp(), scale.data()), g.M(), g.cinp(), g.taps);
  f.bias = shift;
  return f;
}
}  // namespace

int plan_phasenet(Net& net, const ParamView& pv) {
  const float eps = net.cfg.bn_eps;
  net.in_samples = T0;
  net.n_out = 3;
  net.win_flags = net.add_blob(std::vector<float>((size_t)std::max(net.max_batch, 1), 0.f));
  const int len[5] = {T0, T1, T2, T3, T4};
  const int ch[5] = {8, 16, 32, 64, 128};

  const int x = net.add_tensor("input", 3, T0);
  net.input = x;
  const int h0 = net.add_tensor("inc", 8, T0);
  int skip[4], down[4];
  for (int i = 0; i < 4; ++i) {
    skip[i] = net.add_tensor("down" + std::to_string(i) + ".same", ch[i], len[i]);
    down[i] = net.add_tensor("down" + std::to_string(i) + ".down", ch[i], len[i + 1]);
  }
  const int bottom = net.add_tensor("down4.same", 128, T4);
  int upT[4], upS[3];
  for (int j = 0; j < 4; ++j) {
    upT[j] = net.add_tensor("up" + std::to_string(j) + ".convT", ch[3 - j], len[3 - j]);
    if (j < 3) upS[j] = net.add_tensor("up" + std::to_string(j) + ".same", ch[3 - j], len[3 - j]);
  }

  auto flops_conv = [](int cout, int cin, int K, int lout) { return 2.0 * cout * cin * K * lout; };
  ConvLayer* L;
  Folded f;

  f = fold_conv(pv, "inc", "in_bn", 8, 3, 7, 1, PN_inc::geom(), eps, true);
  L = net.add_conv<PN_inc>("inc", x, -1, h0, (T0 + 1) / 2, T0, f.afrag, f.bias);
  L->flops_per_window = flops_conv(8, 3, 7, T0);
  net.steps.back().flops_per_window = L->flops_per_window;

#define PN_CONV(CFG, NAME, CONV, BN, COUT, CIN, STRIDE, SRC1, SRC2, DST, COLS, LOUT)                 \
  f = fold_conv(pv, CONV, BN, COUT, CIN, 7, STRIDE, CFG::geom(), eps, false);                        \
  L = net.add_conv<CFG>(NAME, SRC1, SRC2, DST, COLS, LOUT, f.afrag, f.bias);                         \
  L->flops_per_window = flops_conv(COUT, CIN, 7, LOUT);                                              \
  net.steps.back().flops_per_window = L->flops_per_window;
#define PN_CONVT(CFG, NAME, CONV, BN, CIN, COUT, SRC, DST, LIN, LOUT)                                \
  f = fold_convT(pv, CONV, BN, CIN, COUT, CFG::geom(), eps);                                         \
  L = net.add_conv<CFG>(NAME, SRC, -1, DST, (LIN) + 1, LOUT, f.afrag, f.bias);                       \
  L->flops_per_window = 2.0 * (CIN) * (COUT) * 7 * (LIN);                                            \
  net.steps.back().flops_per_window = L->flops_per_window;

  PN_CONV(PN_d0same, "down0.same", "down_branch.0.0", "down_branch.0.1", 8, 8, 1, h0, -1, skip[0], (T0 + 1) / 2, T0)
  PN_CONV(PN_d0down, "down0.down", "down_branch.0.2", "down_branch.0.3", 8, 8, 4, skip[0], -1, down[0], (T1 + 1) / 2, T1)
  PN_CONV(PN_d1same, "down1.same", "down_branch.1.0", "down_branch.1.1", 16, 8, 1, down[0], -1, skip[1], T1, T1)
  PN_CONV(PN_d1down, "down1.down", "down_branch.1.2", "down_branch.1.3", 16, 16, 4, skip[1], -1, down[1], T2, T2)
  PN_CONV(PN_d2same, "down2.same", "down_branch.2.0", "down_branch.2.1", 32, 16, 1, down[1], -1, skip[2], T2, T2)
  PN_CONV(PN_d2down, "down2.down", "down_branch.2.2", "down_branch.2.3", 32, 32, 4, skip[2], -1, down[2], T3, T3)
  PN_CONV(PN_d3same, "down3.same", "down_branch.3.0", "down_branch.3.1", 64, 32, 1, down[2], -1, skip[3], T3, T3)
  PN_CONV(PN_d3down, "down3.down", "down_branch.3.2", "down_branch.3.3", 64, 64, 4, skip[3], -1, down[3], T4, T4)
  PN_CONV(PN_d4same, "down4.same", "down_branch.4.0", "down_branch.4.1", 128, 64, 1, down[3], -1, bottom, T4, T4)

  PN_CONVT(PN_u0T, "up0.convT", "up_branch.0.0", "up_branch.0.1", 128, 64, bottom, upT[0], T4, T3)
  PN_CONV(PN_u0same, "up0.same", "up_branch.0.2", "up_branch.0.3", 64, 128, 1, skip[3], upT[0], upS[0], T3, T3)
  PN_CONVT(PN_u1T, "up1.convT", "up_branch.1.0", "up_branch.1.1", 64, 32, upS[0], upT[1], T3, T2)
  PN_CONV(PN_u1same, "up1.same", "up_branch.1.2", "up_branch.1.3", 32, 64, 1, skip[2], upT[1], upS[1], T2, T2)
  PN_CONVT(PN_u2T, "up2.convT", "up_branch.2.0", "up_branch.2.1", 32, 16, upS[1], upT[2], T2, T1)
  PN_CONV(PN_u2same, "up2.same", "up_branch.2.2", "up_branch.2.3", 16, 32, 1, skip[1], upT[2], upS[2], T1, T1)
  PN_CONVT(PN_u3T, "up3.convT", "up_branch.3.0", "up_branch.3.1", 16, 8, upS[2], upT[3], T1, T0)
  PN_CONV(PN_u3same, "up3.same+out", "up_branch.3.2", "up_branch.3.3", 8, 16, 1, skip[0], upT[3], kDenseOut, (T0 + 1) / 2, T0)
#undef PN_CONV
#undef PN_CONVT

  // 1x1 output conv (8 -> 3) + softmax live in the last layer's epilogue.
  const float* ow = pv.get("out.weight");
  const float* ob = pv.get("out.bias");
  L->e0.h.assign(ow, ow + 24);
  L->e1.h.assign(ob, ob + 3);
  L->flops_per_window += 2.0 * 3 * 8 * T0;
  net.steps.back().flops_per_window = L->flops_per_window;

  net.flops_per_window = 0;
  for (auto& s : net.steps) net.flops_per_window += s.flops_per_window;
  // plan_flags[0] = 1 keeps the layer-by-layer plan (debug / A-B timing); default is the fused plan.
  if (net.cfg.plan_flags[0] != 1) return plan_phasenet_fused(net, pv, net.cfg.plan_flags[1]);
  return VP_OK;
}

}  // namespace vp
