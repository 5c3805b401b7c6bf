// EQTransformer (SeisBench topology, volpick weights): launch plan.
// Reference contract: model(x) on (B,3,6000) -> (detection, P, S), each (B,6000)
// (volpick/model/eval_taks0.py:68-72, volpick/model/models.py:543; SURVEY.md §8a row A5,
// Appendix A.4/B.2).
//
//   encoder   7 x [Conv1d + ReLU + MaxPool1d(2)]            conv_mfma_kernel, pooled epilogue
//   ResCNN    7 x [BN-ReLU-Conv, BN-ReLU-Conv, + skip]      conv_mfma_kernel (BN folded / dual epilogue)
//   BiLSTM    3 x [LSTM(bi) + Conv1d(32,16,1) + BN]         bilstm_kernel
//   transf.   2 x [additive attention + LN + FF + LN]       transformer_kernel
//   picks     2 x [LSTM + banded attention]                  pick_branch_kernel
//   decoders  3 x 7 x [Upsample(2) + Conv1d + ReLU]          conv_mfma_kernel, three weight sets per launch,
//                                                            the producer writes its rows x2-upsampled
//   heads     3 x [Conv1d(8,1,11) + sigmoid]                 epilogue of decoder.6 (EPI_HEAD)
#include "eqt_kernels.h"
#include "net.h"

namespace vp {

namespace {

// APRE (conv_mfma.h) where it measured faster: encoder.0 / .1 and decoder.6; it cost 2-4 us on the decoder stages 2-5.
// AQ4 (16-byte weight loads) where it measured faster: encoder.4-6, decoder.1 / .2 / .4 / .5 (same-box A/B per launch:
// -0.1 .. -3.0 us); not decoder.3.
//                      CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF WM WN NW RELU EPI APRE AQ4
using EQ_e0 = ConvCfg<3, 0, 8, 2, 12, 2, -5, 0, 1, 4, 8, 1, EPI_POOL2, 1>;
using EQ_e1 = ConvCfg<8, 0, 16, 1, 9, 1, -4, 0, 1, 4, 8, 1, EPI_POOL2, 1>;
using EQ_e2 = ConvCfg<16, 0, 16, 1, 7, 1, -3, 0, 1, 4, 6, 1, EPI_POOL2>;
using EQ_e3 = ConvCfg<16, 0, 32, 1, 7, 1, -3, 0, 2, 2, 6, 1, EPI_POOL2>;
using EQ_e4 = ConvCfg<32, 0, 32, 1, 5, 1, -2, 0, 2, 2, 6, 1, EPI_POOL2, 0, 1>;
using EQ_e5 = ConvCfg<32, 0, 64, 1, 5, 1, -2, 0, 4, 1, 6, 1, EPI_POOL2, 0, 1>;
using EQ_e6 = ConvCfg<64, 0, 64, 1, 3, 1, -1, 0, 4, 1, 3, 1, EPI_POOL2_DUAL, 0, 1>;  // two 48-column tiles per window (one 96-column tile: 11.8 vs 10.5 us)
using EQ_r1k3 = ConvCfg<64, 0, 64, 1, 3, 1, -1, 0, 4, 1, 3, 1, EPI_STORE>;
using EQ_r1k2 = ConvCfg<64, 0, 64, 1, 2, 1, 0, 0, 4, 1, 3, 1, EPI_STORE>;
using EQ_r2k3 = ConvCfg<64, 0, 64, 1, 3, 1, -1, 0, 4, 1, 3, 0, EPI_RES>;
using EQ_r2k2 = ConvCfg<64, 0, 64, 1, 2, 1, 0, 0, 4, 1, 3, 0, EPI_RES>;
// Decoder stage = Upsample(2, nearest) + Conv1d(K, pad K/2) + ReLU.  The upsample is folded into the
// conv as a 2-phase polyphase filter on the NOT-upsampled input (conv_pack.cpp amat_upconv):
//   y[co][2n+p] = sum_ci sum_d Wp[co][ci][d] x[ci][n+d],  d in [floor(-pad/2), floor((K-pad)/2)]
// K = 3/5/7/9/11 -> 3/3/5/5/7 taps for two outputs instead of 2K: 0.6-0.7x the MACs, half the reads,
// and the x2 intermediate is never written.  Stage 2 (188 -> 375, odd): the reference crops the upsampled row by one
// sample, which a polyphase filter cannot express at the right edge — its last two outputs would see the cropped
// copy of the last input sample.  The stage runs folded like the others and decoder2_edge_kernel then recomputes
// exactly those two samples per channel from the definition (2 x 32 x 320 MACs per decoder and window).
//                      CIN1 CIN2 COUT P TAPS SN IN_OFF OUT_OFF WM WN NW RELU EPI
using EQ_d0 = ConvCfg<16, 0, 64, 2, 3, 1, -1, 0, 4, 1, 3, 1, EPI_STORE>;
using EQ_d1 = ConvCfg<64, 0, 64, 2, 3, 1, -1, 0, 4, 1, 6, 1, EPI_STORE, 0, 1>;
using EQ_d2 = ConvCfg<64, 0, 32, 2, 3, 1, -1, 0, 2, 2, 6, 1, EPI_STORE, 0, 1>;
using EQ_d3 = ConvCfg<32, 0, 32, 2, 5, 1, -2, 0, 2, 2, 6, 1, EPI_STORE>;  // AQ4 measured 62.5 vs 58.4 us here (MW = 2 with 6 n-tiles: registers)
using EQ_d4 = ConvCfg<32, 0, 16, 2, 5, 1, -2, 0, 2, 2, 6, 1, EPI_STORE, 0, 1>;
using EQ_d5 = ConvCfg<16, 0, 16, 2, 5, 1, -2, 0, 2, 2, 6, 1, EPI_STORE, 0, 1>;
// NW = 6 (376-column steps): 8 tiles x 768 rows = 6144 workgroups at 6 per CU = exactly four residencies of the chip
// (NW = 8: 4608 workgroups at 4 per CU = 4.5, the last one half empty); 96.6 -> 95 us.
using EQ_d6 = ConvCfg<16, 0, 8, 2, 7, 1, -3, 0, 1, 4, 6, 1, EPI_HEAD, 1>;  // + Conv1d(8,1,11) + sigmoid head
// A/B tile variants (plan flag plan_flags[3] = 1): half-width tiles, twice the workgroups per CU
using EQ_d3b = ConvCfg<32, 0, 32, 2, 5, 1, -2, 0, 2, 2, 3, 1, EPI_STORE>;
using EQ_d4b = ConvCfg<32, 0, 16, 2, 5, 1, -2, 0, 2, 2, 3, 1, EPI_STORE>;
using EQ_d5b = ConvCfg<16, 0, 16, 2, 5, 1, -2, 0, 2, 2, 3, 1, EPI_STORE>;
using EQ_d6b = ConvCfg<16, 0, 8, 2, 7, 1, -3, 0, 1, 4, 4, 1, EPI_HEAD>;

// Right edge of decoder stage 2 (module comment): y[t] = relu(b + sum_ci sum_k W[co][ci][k] u[t + k - 2]) for the last
// two samples t = l_out - 2, l_out - 1 (373, 374), with u[i] = x[ci][i >> 1] inside the cropped upsampled row [0, l_out)
// and 0 outside.  Both samples only see x[ci][185 .. 187]; the taps that hit the same input sample are pre-summed on the
// host: e[set][ci][thread][3] with thread = 2 co + which.  One 64-thread workgroup per (decoder, window).
struct EdgeArgs {
  const float* x;  // stage-1 rows [3 B][64][ls]
  int ls_x;
  long ws_x;
  float* y;        // stage-2 rows [3 B][32][ls]
  int ls_y;
  long ws_y;
  const float* e;  // [3][64][64][3]
  const float* b;  // [3][32]
  int win_per_set, l_out;
};
__global__ __launch_bounds__(64) void decoder2_edge_kernel(const EdgeArgs a) {
  __shared__ float xs[64][3];
  const int win = blockIdx.x, set = win / a.win_per_set, tid = threadIdx.x;
  const int n0 = (a.l_out - 2 - 2) >> 1;  // first input sample the two outputs see (185)
  {
    const float* x = a.x + (long)win * a.ws_x + HALO + (long)tid * a.ls_x + n0;
    xs[tid][0] = x[0], xs[tid][1] = x[1], xs[tid][2] = x[2];
  }
  __syncthreads();
  const float* e = a.e + ((long)set * 64 * 64 + tid) * 3;
  float acc = a.b[set * 32 + (tid >> 1)];
#pragma unroll 8
  for (int ci = 0; ci < 64; ++ci) {
    const float* ec = e + (long)ci * 64 * 3;
    acc = fmaf(ec[0], xs[ci][0], acc);
    acc = fmaf(ec[1], xs[ci][1], acc);
    acc = fmaf(ec[2], xs[ci][2], acc);
  }
  a.y[(long)win * a.ws_y + HALO + (long)(tid >> 1) * a.ls_y + a.l_out - 2 + (tid & 1)] = fmaxf(acc, 0.f);
}

std::vector<float> vec(const float* p, size_t n) { return std::vector<float>(p, p + n); }

std::vector<float> pack_plain(const ParamView& pv, const std::string& conv, int cout, int cin, int K, const ConvGeom& g,
                              const float* row_scale = nullptr) {
  return pack_afrag(amat_conv(pv.get(conv + ".weight"), cout, cin, K, 1, g.P, g.cinp(), row_scale), g.M(), g.cinp(),
                    g.taps);
}

template <class Cfg>
ConvLayer* add_plain(Net& net, const ParamView& pv, const std::string& name, const std::string& conv, int cout,
                     int cin, int K, int src, int dst, int cols, int l_out) {
  ConvLayer* L = net.add_conv<Cfg>(name, src, -1, dst, cols, l_out, pack_plain(pv, conv, cout, cin, K, Cfg::geom()),
                                   vec(pv.get(conv + ".bias"), cout));
  L->flops_per_window = 2.0 * cout * cin * K * l_out;
  net.steps.back().flops_per_window = L->flops_per_window;
  return L;
}

LstmWeights lstm_weights(Net& net, const ParamView& pv, const std::string& prefix, const std::string& suffix, int cin) {
  const float* bi = pv.get(prefix + "bias_ih_l0" + suffix);
  const float* bh = pv.get(prefix + "bias_hh_l0" + suffix);
  std::vector<float> b(64);
  for (int i = 0; i < 64; ++i) b[i] = bi[i] + bh[i];
  LstmWeights w{};
  // device pointers are patched after upload(); stash the blobs
  HostBlob* wih = net.add_blob(vec(pv.get(prefix + "weight_ih_l0" + suffix), (size_t)64 * cin));
  HostBlob* whh = net.add_blob(vec(pv.get(prefix + "weight_hh_l0" + suffix), (size_t)64 * 16));
  HostBlob* bb = net.add_blob(b);
  w.w_ih = reinterpret_cast<const float*>(wih);
  w.w_hh = reinterpret_cast<const float*>(whh);
  w.b = reinterpret_cast<const float*>(bb);
  return w;
}
// The structs above temporarily hold HostBlob* in their float* fields; resolve at run time.
inline const float* dptr(const float* stash) { return reinterpret_cast<const HostBlob*>(stash)->d; }
LstmWeights resolve(const LstmWeights& w) { return LstmWeights{dptr(w.w_ih), dptr(w.w_hh), dptr(w.b)}; }

AttnWeights attn_weights(Net& net, const ParamView& pv, const std::string& prefix) {
  AttnWeights w{};
  w.Wt = reinterpret_cast<const float*>(net.add_blob(vec(pv.get(prefix + "Wt"), 16 * 32)));
  w.Wx = reinterpret_cast<const float*>(net.add_blob(vec(pv.get(prefix + "Wx"), 16 * 32)));
  w.bh = reinterpret_cast<const float*>(net.add_blob(vec(pv.get(prefix + "bh"), 32)));
  w.Wa = reinterpret_cast<const float*>(net.add_blob(vec(pv.get(prefix + "Wa"), 32)));
  return w;
}
AttnWeights resolve(const AttnWeights& w) { return AttnWeights{dptr(w.Wt), dptr(w.Wx), dptr(w.bh), dptr(w.Wa)}; }

}  // namespace

int plan_eqt(Net& net, const ParamView& pv) {
  const float eps = net.cfg.bn_eps;
  const int T = 6000;
  net.in_samples = T;
  net.n_out = 3;
  net.win_flags = net.add_blob(std::vector<float>((size_t)std::max(net.max_batch, 1), 0.f));
  const int filt[7] = {8, 16, 16, 32, 32, 64, 64};
  const int rker[7] = {3, 3, 3, 3, 2, 3, 2};
  int len[8];
  len[0] = T;
  for (int i = 0; i < 7; ++i) len[i + 1] = (len[i] + 1) / 2;  // 3000 1500 750 375 188 94 47

  // ---- encoder -----------------------------------------------------------------------
  const int x = net.add_tensor("input", 3, T);
  net.input = x;
  int enc[7];
  for (int i = 0; i < 7; ++i) enc[i] = net.add_tensor("encoder." + std::to_string(i), filt[i], len[i + 1]);
  const int act0 = net.add_tensor("res.act", 64, EQT_T);  // relu(bn1(x)) feeding each block's conv1
  ConvLayer* L;
  add_plain<EQ_e0>(net, pv, "encoder.0", "encoder.convs.0", 8, 3, 11, x, enc[0], 3000, 6000);
  add_plain<EQ_e1>(net, pv, "encoder.1", "encoder.convs.1", 16, 8, 9, enc[0], enc[1], 3000, 3000);
  add_plain<EQ_e2>(net, pv, "encoder.2", "encoder.convs.2", 16, 16, 7, enc[1], enc[2], 1500, 1500);
  add_plain<EQ_e3>(net, pv, "encoder.3", "encoder.convs.3", 32, 16, 7, enc[2], enc[3], 750, 750);
  add_plain<EQ_e4>(net, pv, "encoder.4", "encoder.convs.4", 32, 32, 5, enc[3], enc[4], 375, 375);
  add_plain<EQ_e5>(net, pv, "encoder.5", "encoder.convs.5", 64, 32, 5, enc[4], enc[5], 188, 188);
  L = add_plain<EQ_e6>(net, pv, "encoder.6", "encoder.convs.6", 64, 64, 3, enc[5], enc[6], 94, 94);
  {
    std::vector<float> s, b;
    bn_fold(pv, "res_cnn_stack.members.0.norm1", 64, eps, nullptr, &s, &b);
    L->dst2 = act0;
    L->e1.h = s;
    L->e2.h = b;
  }

  // ---- ResCNN stack ---------------------------------------------------------------------
  const int mid = net.add_tensor("res.mid", 64, EQT_T);
  const int ping[2] = {net.add_tensor("res.xa", 64, EQT_T), net.add_tensor("res.xb", 64, EQT_T)};
  int x_cur = enc[6];  // residual stream; block i reads x_cur and writes ping[i & 1]
  for (int i = 0; i < 7; ++i) {
    const std::string p = "res_cnn_stack.members." + std::to_string(i);
    const int K = rker[i];
    // conv1: relu(bn1 x) -> conv -> bn2 -> relu       [bn2 folded into conv1's weights]
    std::vector<float> s2, b2;
    bn_fold(pv, p + ".norm2", 64, eps, pv.get(p + ".conv1.bias"), &s2, &b2);
    if (K == 3) {
      L = net.add_conv<EQ_r1k3>("res" + std::to_string(i) + ".conv1", act0, -1, mid, EQT_T, EQT_T,
                                pack_plain(pv, p + ".conv1", 64, 64, 3, EQ_r1k3::geom(), s2.data()), b2);
    } else {
      L = net.add_conv<EQ_r1k2>("res" + std::to_string(i) + ".conv1", act0, -1, mid, EQT_T, EQT_T,
                                pack_plain(pv, p + ".conv1", 64, 64, 2, EQ_r1k2::geom(), s2.data()), b2);
    }
    L->flops_per_window = 2.0 * 64 * 64 * K * EQT_T;
    net.steps.back().flops_per_window = L->flops_per_window;
    // conv2 + skip; second output = relu(bn1 of the NEXT block), the input of its conv1
    const int x_next = ping[i & 1];
    if (K == 3) {
      L = net.add_conv<EQ_r2k3>("res" + std::to_string(i) + ".conv2", mid, -1, x_next, EQT_T, EQT_T,
                                pack_plain(pv, p + ".conv2", 64, 64, 3, EQ_r2k3::geom()),
                                vec(pv.get(p + ".conv2.bias"), 64));
    } else {
      L = net.add_conv<EQ_r2k2>("res" + std::to_string(i) + ".conv2", mid, -1, x_next, EQT_T, EQT_T,
                                pack_plain(pv, p + ".conv2", 64, 64, 2, EQ_r2k2::geom()),
                                vec(pv.get(p + ".conv2.bias"), 64));
    }
    L->res = x_cur;
    std::vector<float> s1(64, 0.f), b1(64, 0.f);
    if (i < 6) {
      bn_fold(pv, "res_cnn_stack.members." + std::to_string(i + 1) + ".norm1", 64, eps, nullptr, &s1, &b1);
      L->dst2 = act0;
    }
    L->e1.h = s1;
    L->e2.h = b1;
    L->flops_per_window = 2.0 * 64 * 64 * K * EQT_T;
    net.steps.back().flops_per_window = L->flops_per_window;
    x_cur = x_next;
  }
  const int res_out = x_cur;

  // ---- BiLSTM stack ---------------------------------------------------------------------
  std::vector<std::function<BiLstmArgs(Net&)>> mk_lstm;
  std::vector<std::function<TransformerArgs(Net&)>> mk_tr;
  int mid_first = -1;  // first of the six steps that eqt_mid_kernel replaces
  double mid_flops = 0;
  int lstm_in = res_out;
  for (int i = 0; i < 3; ++i) {
    const std::string p = "bi_lstm_stack.members." + std::to_string(i);
    const int cin = (i == 0) ? 64 : 16;
    const int out = net.add_tensor("bilstm." + std::to_string(i), 16, EQT_T);
    LstmWeights fw = lstm_weights(net, pv, p + ".lstm.", "", cin);
    LstmWeights bw = lstm_weights(net, pv, p + ".lstm.", "_reverse", cin);
    std::vector<float> s, b;
    bn_fold(pv, p + ".norm", 16, eps, pv.get(p + ".conv.bias"), &s, &b);
    std::vector<float> wc = vec(pv.get(p + ".conv.weight"), 16 * 32);
    for (int co = 0; co < 16; ++co)
      for (int c = 0; c < 32; ++c) wc[co * 32 + c] *= s[co];
    HostBlob* wcb = net.add_blob(wc);
    HostBlob* bcb = net.add_blob(b);
    Step st;
    st.name = "bilstm." + std::to_string(i);
    st.flops_per_window = 2.0 * 2 * EQT_T * (64.0 * cin + 64.0 * 16) + 2.0 * 16 * 32 * EQT_T;
    const int src_t = lstm_in;
    auto mk = [=](Net& n) -> BiLstmArgs {
      BiLstmArgs a{};
      const Tensor& s0 = n.tensors[src_t];
      const Tensor& d0 = n.tensors[out];
      a.src = s0.p;
      a.ls_src = s0.ls;
      a.ws_src = (long)s0.win_stride();
      a.dst = d0.p;
      a.ls_dst = d0.ls;
      a.ws_dst = (long)d0.win_stride();
      a.fwd = resolve(fw);
      a.bwd = resolve(bw);
      a.wc = wcb->d;
      a.bc = bcb->d;
      return a;
    };
    mk_lstm.push_back(mk);
    st.run = [=](Net& n, int B, hipStream_t s_) -> int { return launch_bilstm(mk(n), cin, B, s_); };
    mid_first = (i == 0) ? (int)net.steps.size() : mid_first;
    mid_flops += st.flops_per_window;
    net.steps.push_back(std::move(st));
    lstm_in = out;
  }

  // ---- transformers -----------------------------------------------------------------------
  const int dec_in = net.add_tensor("decoder.in", 16, EQT_T, 3);  // inputs of the 3 decoders (set-major)
  int tr_in = lstm_in;
  const char* tr_names[2] = {"transformer_d0", "transformer_d"};
  for (int i = 0; i < 2; ++i) {
    const std::string p = tr_names[i];
    const int out = net.add_tensor(p, 16, EQT_T);
    AttnWeights aw = attn_weights(net, pv, p + ".attention.");
    std::vector<float> ln4;  // gamma1 | beta1 | gamma2 | beta2 in ONE blob: eqt_mid_kernel fetches it as ln4[lane]
    for (const char* nm : {".norm1.gamma", ".norm1.beta", ".norm2.gamma", ".norm2.beta"}) {
      const std::vector<float> part = vec(pv.get(p + nm), 16);
      ln4.insert(ln4.end(), part.begin(), part.end());
    }
    HostBlob* ln = net.add_blob(ln4);
    HostBlob* w1 = net.add_blob(vec(pv.get(p + ".ff.lin1.weight"), 128 * 16));
    HostBlob* bb1 = net.add_blob(vec(pv.get(p + ".ff.lin1.bias"), 128));
    HostBlob* w2 = net.add_blob(vec(pv.get(p + ".ff.lin2.weight"), 16 * 128));
    HostBlob* bb2 = net.add_blob(vec(pv.get(p + ".ff.lin2.bias"), 16));
    Step st;
    st.name = p;
    st.flops_per_window = 2.0 * (2 * EQT_T * 16 * 32 + EQT_T * EQT_T * 32 * 2 + EQT_T * EQT_T * 16 + 2 * EQT_T * 16 * 128);
    const int src_t = tr_in;
    const bool last = (i == 1);
    const float attn_eps = net.cfg.attention_eps, ln_eps = net.cfg.layernorm_eps;
    auto mk = [=](Net& n) -> TransformerArgs {
      TransformerArgs a{};
      const Tensor& s0 = n.tensors[src_t];
      const Tensor& d0 = n.tensors[out];
      a.src = s0.p;
      a.ls_src = s0.ls;
      a.ws_src = (long)s0.win_stride();
      a.dst = d0.p;
      a.ls_dst = d0.ls;
      a.ws_dst = (long)d0.win_stride();
      if (last) {
        const Tensor& u = n.tensors[dec_in];
        a.up = u.p;
        a.ls_up = u.ls;
        a.ws_up = (long)u.win_stride();
      }
      a.att = resolve(aw);
      a.g1 = ln->d;
      a.b1 = ln->d + 16;
      a.g2 = ln->d + 32;
      a.b2 = ln->d + 48;
      a.w1 = w1->d;
      a.bb1 = bb1->d;
      a.w2 = w2->d;
      a.bb2 = bb2->d;
      a.attn_eps = attn_eps;
      a.ln_eps = ln_eps;
      return a;
    };
    mk_tr.push_back(mk);
    st.run = [=](Net& n, int B, hipStream_t s_) -> int { return launch_transformer(mk(n), B, s_); };
    mid_flops += st.flops_per_window;
    net.steps.push_back(std::move(st));
    tr_in = out;
  }

  // ---- P / S branches: LSTM + banded attention ------------------------------------------------
  {
    LstmWeights lw[2];
    AttnWeights aw[2];
    for (int br = 0; br < 2; ++br) {
      lw[br] = lstm_weights(net, pv, "pick_lstms." + std::to_string(br) + ".", "", 16);
      aw[br] = attn_weights(net, pv, "pick_attentions." + std::to_string(br) + ".");
    }
    Step st;
    st.name = "pick_branches";
    st.flops_per_window = 2.0 * (2.0 * EQT_T * (64 * 16 + 64 * 16) + 2.0 * (2 * EQT_T * 16 * 32 + EQT_T * EQT_T * 32 * 2 + EQT_T * EQT_T * 16));
    const int src_t = tr_in;
    const float attn_eps = net.cfg.attention_eps;
    auto mk_pick = [=](Net& n, int B) -> PickBranchArgs {
      PickBranchArgs a{};
      const Tensor& s0 = n.tensors[src_t];
      const Tensor& u = n.tensors[dec_in];
      a.src = s0.p;
      a.ls_src = s0.ls;
      a.ws_src = (long)s0.win_stride();
      a.up = u.p;
      a.ls_up = u.ls;
      a.ws_up = (long)u.win_stride();
      a.B = B;
      for (int br = 0; br < 2; ++br) {
        a.lstm[br] = resolve(lw[br]);
        a.att[br] = resolve(aw[br]);
      }
      a.attn_eps = attn_eps;
      a.width = 3;
      return a;
    };
    st.run = [=](Net& n, int B, hipStream_t s_) -> int { return launch_pick_branch(mk_pick(n, B), s_); };
    mid_flops += st.flops_per_window;
    net.steps.push_back(std::move(st));
    // plan_flags[2] = 1 keeps the six separate launches (A/B timing); default: one launch for the whole latency-bound chain,
    // four windows per workgroup in teams of four waves (eqt_mid4.hip); plan_flags[2] = 3: two windows per workgroup in teams
    // of eight waves (the default of rounds 3-5); plan_flags[2] = 2: one window per workgroup (the form of rounds 1-2)
    if (net.cfg.plan_flags[2] != 1 && mid_first >= 0 && (int)net.steps.size() == mid_first + 6) {
      Step fused;
      fused.name = "fused.mid (3 BiLSTM + 2 transformer blocks + pick branches)";
      fused.flops_per_window = mid_flops;
      // issued per window (eqt_kernels.hip): 16 x 16 x 4 fp32 tiles of the dense products -- BiLSTM input projections 8 waves x 3
      // n-tiles x CIN / 4 K-steps (CIN 64, 16, 16) + Conv1d(32,16,1) 3 x 8; per transformer q / k 12 x 4, a.x 3 x 12, the two
      // feed-forward layers 24 x 4 and 6 x 16; pick branches 24 x 4 + 2 x (12 x 4 + 3 x 12) -- and, on the vector ALUs, the
      // eight 47-step recurrences (64 gate rows x 16 units) and the four 47 x 47 x 32 score loops (two packed FMAs per pair)
      fused.set_issued((24.0 * 16 + 24 + 2 * (24.0 * 4 + 24) + 2 * (48.0 + 36 + 96 + 96) + 96 + 2 * (48.0 + 36)) * 2048.0, 0.0,
                       8 * 47 * 2.0 * 64 * 16 + 4 * 47.0 * 47 * 32 * 4);
      fused.run = [=](Net& n, int B, hipStream_t s_) -> int {
        MidArgs m{};
        for (int i = 0; i < 3; ++i) m.lstm[i] = mk_lstm[i](n);
        for (int i = 0; i < 2; ++i) m.tr[i] = mk_tr[i](n);
        m.pick = mk_pick(n, B);
        m.clk = n.debug_clock ? reinterpret_cast<unsigned long long*>(n.debug_clock->d) : nullptr;  // slots [B][8] (tools/mid_clock.py)
        m.B = B;
        if (n.cfg.plan_flags[2] == 2 || n.cfg.plan_flags[2] == 3) return launch_eqt_mid(m, B, s_, n.cfg.plan_flags[2] == 2);
        return launch_eqt_mid4(m, B, s_);
      };
      net.steps.erase(net.steps.begin() + mid_first, net.steps.end());
      net.steps.push_back(std::move(fused));
    }
  }

  // ---- three decoders, one launch per stage ----------------------------------------------------
  const int din[7] = {47, 94, 188, 375, 750, 1500, 3000};    // length of the stage's input rows
  const int dout[7] = {94, 188, 375, 750, 1500, 3000, 6000};  // conv output length
  const int dco[7] = {64, 64, 32, 32, 16, 16, 8};
  const int dci[7] = {16, 64, 64, 32, 32, 16, 16};
  const int dk[7] = {3, 5, 5, 7, 7, 9, 11};
  const char* dec_prefix[3] = {"decoder_d", "pick_decoders.0", "pick_decoders.1"};
  int dsrc = dec_in;
  const bool alt = net.cfg.plan_flags[3] == 1;
  for (int i = 0; i < 7; ++i) {
    const bool polyphase = true;
    const int dst_len = dout[i];
    const int dst = (i == 6) ? kDenseOut : net.add_tensor("decoder." + std::to_string(i), dco[i], dst_len, 3);
    auto pack3 = [&](const ConvGeom& g, std::vector<float>* af, std::vector<float>* bs) {
      for (int d = 0; d < 3; ++d) {
        const std::string c = std::string(dec_prefix[d]) + ".convs." + std::to_string(i);
        std::vector<float> a =
            polyphase ? pack_afrag(amat_upconv(pv.get(c + ".weight"), dco[i], dci[i], dk[i], g.cinp()), g.M(), g.cinp(),
                                   g.taps)
                      : pack_plain(pv, c, dco[i], dci[i], dk[i], g);
        af->insert(af->end(), a.begin(), a.end());
        const float* b = pv.get(c + ".bias");
        bs->insert(bs->end(), b, b + dco[i]);
      }
    };
    std::vector<float> af, bs;
    const std::string nm = "decoder." + std::to_string(i);
    const int cols = din[i];  // polyphase: one column per input sample
#define EQ_DEC(CFG)                                                  \
  pack3(CFG::geom(), &af, &bs);                                      \
  L = net.add_conv<CFG>(nm, dsrc, -1, dst, cols, dout[i], af, bs, 3);
    switch (i) {
      case 0: EQ_DEC(EQ_d0) break;
      case 1: EQ_DEC(EQ_d1) break;
      case 2: EQ_DEC(EQ_d2) break;
      case 3: if (alt) { EQ_DEC(EQ_d3b) } else { EQ_DEC(EQ_d3) } break;
      case 4: if (alt) { EQ_DEC(EQ_d4b) } else { EQ_DEC(EQ_d4) } break;
      case 5: if (alt) { EQ_DEC(EQ_d5b) } else { EQ_DEC(EQ_d5) } break;
      default: if (alt) { EQ_DEC(EQ_d6b) } else { EQ_DEC(EQ_d6) } break;
    }
#undef EQ_DEC
    L->l_dst = dst_len;
    L->flops_per_window = 3 * 2.0 * dco[i] * dci[i] * dk[i] * dout[i];  // algorithmic (reference) MACs
    net.steps.back().flops_per_window = L->flops_per_window;
    if (i == 2) {  // the two samples at the cropped right edge, recomputed from the definition
      std::vector<float> ew((size_t)3 * 64 * 64 * 3), eb;
      for (int d = 0; d < 3; ++d) {
        const std::string c = std::string(dec_prefix[d]) + ".convs.2";
        const float* w = pv.get(c + ".weight");  // [32][64][5]
        const float* b = pv.get(c + ".bias");
        for (int ci = 0; ci < 64; ++ci)
          for (int co = 0; co < 32; ++co) {
            const float* k = w + ((size_t)co * 64 + ci) * 5;
            float* e0 = &ew[(((size_t)d * 64 + ci) * 64 + 2 * co) * 3];  // t = 373: u[371..375] = x185 x186 x186 x187 (cropped)
            e0[0] = k[0], e0[1] = k[1] + k[2], e0[2] = k[3];
            float* e1 = e0 + 3;                                           // t = 374: u[372..376] = x186 x186 x187 (cropped) (0)
            e1[0] = 0.f, e1[1] = k[0] + k[1], e1[2] = k[2];
          }
        eb.insert(eb.end(), b, b + 32);
      }
      HostBlob* hw = net.add_blob(std::move(ew));
      HostBlob* hb = net.add_blob(std::move(eb));
      net.named["decoder.2.edge.w"] = hw;
      net.named["decoder.2.edge.b"] = hb;
      Step st;
      st.name = "decoder.2.edge";
      st.flops_per_window = 0;
      const int src_t = dsrc, dst_t = dst;
      st.run = [=](Net& n, int B, hipStream_t s_) -> int {
        EdgeArgs a{};
        const Tensor &tx = n.tensors[src_t], &ty = n.tensors[dst_t];
        a.x = tx.p;
        a.ls_x = tx.ls;
        a.ws_x = (long)tx.win_stride();
        a.y = ty.p;
        a.ls_y = ty.ls;
        a.ws_y = (long)ty.win_stride();
        a.e = hw->d;
        a.b = hb->d;
        a.win_per_set = B;
        a.l_out = 375;
        hipLaunchKernelGGL(decoder2_edge_kernel, dim3(3 * B), dim3(64), 0, s_, a);
        return 0;
      };
      net.steps.push_back(std::move(st));
    }
    dsrc = dst;
  }

  // ---- heads: Conv1d(8,1,11,pad 5) + sigmoid of the three decoders, fused into decoder.6's epilogue ----
  {
    const char* head[3] = {"conv_d", "pick_convs.0", "pick_convs.1"};
    for (int d = 0; d < 3; ++d) {
      const float* hw = pv.get(std::string(head[d]) + ".weight");
      L->e0.h.insert(L->e0.h.end(), hw, hw + 88);
      L->e1.h.push_back(pv.get(std::string(head[d]) + ".bias")[0]);
    }
    L->flops_per_window += 3 * 2.0 * 8 * 11 * T;
    net.steps.back().flops_per_window = L->flops_per_window;
    net.steps.back().name = "decoder.6+heads";
    L->name = "decoder.6";
  }

  net.flops_per_window = 0;
  for (auto& s : net.steps) net.flops_per_window += s.flops_per_window;
  if (net.cfg.plan_flags[1] & 2)  // debug clock stamps of every conv launch (tools/conv_clock.py)
    // [max_batch][32] eqt_mid_kernel | [64][8] conv launches | [max_batch][32] eqt_tail_kernel  (64-bit words)
    net.debug_clock = net.add_blob(std::vector<float>(((size_t)net.max_batch * 64 + 64 * 8) * 2, 0.f));
  // plan_flags[0] = 1 keeps the 14 ResCNN conv launches (layer-by-layer debug / A-B plan)
  if (net.cfg.plan_flags[0] != 1) {
    int rc = plan_eqt_fuse_res(net);
    if (rc != VP_OK) return rc;
  }
  // plan_flags[7] bit 0 keeps decoder.4 / .5 / .6+heads as three launches (layer tests, A/B timing)
  if (!(net.cfg.plan_flags[7] & 1) && !alt) {
    int rc = (net.cfg.plan_flags[7] & 64) ? plan_eqt_fuse_tail(net) : plan_eqt_fuse_tail_b3(net);  // bit 6: the fp32-MFMA kernel
    if (rc != VP_OK) return rc;
  }
  // bit 2 keeps encoder.0 .. .2 as three launches
  if (!(net.cfg.plan_flags[7] & 4)) {
    int rc = plan_eqt_fuse_front(net, !(net.cfg.plan_flags[7] & 256));  // bit 8: stages 1 and 2 on the fp32 MFMA too
    if (rc != VP_OK) return rc;
  }
  // bit 3 keeps encoder.3 .. .6 as four launches
  if (!(net.cfg.plan_flags[7] & 8)) {
    int rc = (net.cfg.plan_flags[7] & 128) ? plan_eqt_fuse_enc36(net) : plan_eqt_fuse_enc36_b3(net);  // bit 7: the fp32-MFMA kernel
    if (rc != VP_OK) return rc;
  }
  // bit 1 keeps decoder.0 .. .3 (+ the stage-2 edge fix) as five launches
  if (!(net.cfg.plan_flags[7] & 2)) {
    int rc = plan_eqt_fuse_dec03(net, !(net.cfg.plan_flags[7] & 32));  // bit 5: every stage on the fp32 MFMA
    if (rc != VP_OK) return rc;
  }
  return VP_OK;
}

}  // namespace vp
