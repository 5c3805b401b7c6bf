// Plan bookkeeping shared by both models: tensor layout, constant upload, launch loop.
#include "net.h"

#include <cmath>

namespace vp {

#include "param_tables.inc"

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

int param_table(int model_kind, const ParamDesc** table) {
  if (model_kind == VP_MODEL_PHASENET) {
    *table = kPhaseNetParams;
    return kPhaseNetParamsCount;
  }
  if (model_kind == VP_MODEL_EQTRANSFORMER) {
    *table = kEqtParams;
    return kEqtParamsCount;
  }
  *table = nullptr;
  return 0;
}

bool build_param_view(int model_kind, const float* blob, size_t n_floats, ParamView* out) {
  const ParamDesc* t;
  const int n = param_table(model_kind, &t);
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (off + t[i].size > n_floats) return false;
    out->by_name[t[i].name] = {blob + off, &t[i]};
    off += t[i].size;
  }
  return off == n_floats;
}

const float* ParamView::get(const std::string& name, const ParamDesc** d) const {
  auto it = by_name.find(name);
  if (it == by_name.end()) {
    set_error("missing parameter %s", name.c_str());
    return nullptr;
  }
  if (d) *d = it->second.second;
  return it->second.first;
}

std::vector<float> regroup_afrag4(const ConvLayer& L) {
  const int steps = L.g.cinp() / 4 * L.g.taps, mt_n = L.g.M() / 16 * L.n_sets;
  std::vector<float> v(L.afrag.h.size());
  for (int mt = 0; mt < mt_n; ++mt)
    for (int st = 0; st < steps; ++st)
      for (int l = 0; l < 64; ++l)
        v[(((size_t)mt * (steps / 4) + st / 4) * 64 + l) * 4 + (st & 3)] = L.afrag.h[((size_t)mt * steps + st) * 64 + l];
  return v;
}

// fp32 MFMA-order fragments [set][mt][cb][tap][64] (pack_afrag) -> three-piece bf16 operands
// [set][mt'][tap * KS + step][piece][64][8] (conv_b3.h): an fp32 weight is exactly hi + mid + lo in bfloat16.
// mperm: GEMM rows regrouped (channel, phase) -> (phase, channel).  Two bf16 per float slot of the blob.
std::vector<float> b3_operand(const ConvLayer& L, bool mperm) {
  const int cinp = L.g.cinp(), taps = L.g.taps, M = L.g.M(), P = L.g.P, cout = L.g.cout, CB = cinp / 4, MT = M / 16;
  // K of one instruction = 32: cinp >= 32: 32 channels of one tap, steps (tap, channel step); cinp = 16 / 8: all channels of
  // 2 / 4 consecutive taps (taps beyond the filter carry zero weights), a lane's eight values = eight channels of one tap
  // cinp = 4 (PhaseNet's inc, three input channels padded to four): all channels of EIGHT consecutive taps per step, a lane's
  // eight values = four channels of two taps (lane group g: taps 2 g, 2 g + 1) in the order its two 8-byte reads deliver them:
  // (tap a: ch 0, 1), (tap b: ch 0, 1), (tap a: ch 2, 3), (tap b: ch 2, 3)
  const int KS = cinp >= 32 ? cinp / 32 : 1, TPK = cinp >= 32 ? 1 : 32 / cinp, LPT = cinp >= 32 ? 4 : (cinp >= 8 ? cinp / 8 : 1);  // lanes groups per tap
  const int steps = cinp >= 32 ? taps * KS : (taps + TPK - 1) / TPK;
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
  const size_t set_in = (size_t)M * cinp * taps, set_out = (size_t)MT * steps * 3 * 64 * 8;
  std::vector<uint16_t> o(set_out * L.n_sets);
  for (int set = 0; set < L.n_sets; ++set) {
    const float* af = L.afrag.h.data() + set * set_in;
    uint16_t* os = o.data() + set * set_out;
    for (int mt = 0; mt < MT; ++mt)
      for (int st = 0; st < steps; ++st)
        for (int l = 0; l < 64; ++l)
          for (int i = 0; i < 8; ++i) {
            const int mp = mt * 16 + (l & 15), g = l >> 4;
            const int m = mperm ? (mp % cout) * P + mp / cout : mp;  // row of the packed fp32 operand
            const int tap = cinp >= 32 ? st / KS : cinp == 4 ? st * TPK + 2 * g + ((i >> 1) & 1) : st * TPK + g / LPT;
            const int ci = cinp >= 32 ? (st % KS) * 32 + 8 * g + i : cinp == 4 ? (i & 1) + 2 * (i >> 2) : 8 * (g % LPT) + i;
            const float w = tap < taps ? af[(((size_t)(m / 16) * CB + ci / 4) * taps + tap) * 64 + (ci % 4) * 16 + (m % 16)] : 0.f;
            const uint16_t h = rne(w);
            const float r1 = w - widen(h);
            const uint16_t md = rne(r1);
            const uint16_t lo = rne(r1 - widen(md));
            const size_t base = ((((size_t)mt * steps + st) * 3) * 64 + l) * 8 + i;
            os[base] = h;
            os[base + 64 * 8] = md;
            os[base + 2 * 64 * 8] = lo;
          }
  }
  std::vector<float> f(o.size() / 2);
  memcpy(f.data(), o.data(), o.size() * 2);
  return f;
}

// the fp32 operand with its GEMM rows regrouped (channel, phase) -> (phase, channel), in the 16-byte-load order of
// regroup_afrag4: for a P-phase layer whose epilogue wants a lane's four rows to be four consecutive channels
std::vector<float> regroup_afrag4_phase_major(const ConvLayer& L) {
  const int cinp = L.g.cinp(), taps = L.g.taps, M = L.g.M(), P = L.g.P, cout = L.g.cout, steps = cinp / 4 * taps, MT = M / 16;
  const size_t set = (size_t)M * cinp * taps;
  std::vector<float> v(L.afrag.h.size());
  for (int s = 0; s < L.n_sets; ++s)
    for (int mt = 0; mt < MT; ++mt)
      for (int st = 0; st < steps; ++st)
        for (int l = 0; l < 64; ++l) {
          const int mp = mt * 16 + (l & 15), m = (mp % cout) * P + mp / cout;
          v[s * set + (((size_t)mt * (steps / 4) + st / 4) * 64 + l) * 4 + (st & 3)] =
              L.afrag.h[s * set + ((size_t)(m / 16) * steps + st) * 64 + (l & 48) + (m % 16)];
        }
  return v;
}

void bn_fold(const ParamView& pv, const std::string& bn, int C, float eps, const float* conv_bias,
             std::vector<float>* scale, std::vector<float>* shift) {
  const float* w = pv.get(bn + ".weight");
  const float* b = pv.get(bn + ".bias");
  const float* m = pv.get(bn + ".running_mean");
  const float* v = pv.get(bn + ".running_var");
  scale->resize(C);
  shift->resize(C);
  for (int c = 0; c < C; ++c) {
    const float s = w[c] / std::sqrt(v[c] + eps);
    (*scale)[c] = s;
    (*shift)[c] = b[c] + ((conv_bias ? conv_bias[c] : 0.f) - m[c]) * s;
  }
}

int Net::add_tensor(const std::string& name, int C, int L, int sets) {
  Tensor t;
  t.name = name;
  t.C = C;
  t.L = L;
  t.need = HALO + round_up(L, 4);
  tensors.push_back(t);
  tensor_sets.push_back(sets);
  return (int)tensors.size() - 1;
}

HostBlob* Net::add_blob(std::vector<float> v) {
  extra.push_back(std::make_unique<HostBlob>());
  extra.back()->h = std::move(v);
  return extra.back().get();
}

void Net::add_conv_step(ConvLayer* L) {
  Step s;
  s.name = L->name;
  s.run = [L](Net& net, int B, hipStream_t stream) -> int {
    ConvArgs a{};
    const Tensor& s1 = net.tensors[L->src1];
    a.src1 = s1.p;
    a.ls1 = s1.ls;
    a.ws1 = (long)s1.win_stride();
    if (L->src2 >= 0) {
      const Tensor& s2 = net.tensors[L->src2];
      a.src2 = s2.p;
      a.ls2 = s2.ls;
      a.ws2 = (long)s2.win_stride();
    }
    if (L->dst == kDenseOut) {
      a.dst = net.y;
      a.lsd = net.in_samples;
      a.wsd = (long)net.n_out * net.in_samples;
      a.dst_halo = 0;
    } else {
      const Tensor& d = net.tensors[L->dst];
      a.dst = d.p;
      a.lsd = d.ls;
      a.wsd = (long)d.win_stride();
      a.dst_halo = HALO;
    }
    if (L->dst2 >= 0) {
      const Tensor& d2 = net.tensors[L->dst2];
      a.dst2 = d2.p;
      a.lsd2 = d2.ls;
      a.wsd2 = (long)d2.win_stride();
    }
    a.afrag = L->afrag_q4 ? L->afrag_q4->d : L->afrag.d;
    a.bias = L->bias.d;
    a.afrag_set_stride = (long)(L->afrag.h.size() / L->n_sets);
    a.bias_set_stride = (long)(L->bias.h.size() / L->n_sets);
    a.win_per_set = B;
    a.n_windows = B * L->n_sets;
    a.l_out = L->l_out;
    a.l_dst = L->l_dst;
    a.e0 = (L->res >= 0) ? net.tensors[L->res].p : L->e0.d;
    if (L->res >= 0) {
      a.ls_res = net.tensors[L->res].ls;
      a.ws_res = (long)net.tensors[L->res].win_stride();
    }
    a.e1 = L->e1.d;
    a.e2 = L->e2.d;
    a.e_set_stride = L->e1.h.empty() ? 0 : (long)(L->e1.h.size() / L->n_sets);
    if (net.debug_clock && net.debug_clock->d) {
      int li = 0;
      for (auto& c : net.convs) {
        if (c.get() == L) break;
        ++li;
      }
      a.clk = reinterpret_cast<unsigned long long*>(net.debug_clock->d) + (size_t)net.max_batch * 32 + (size_t)li * 8;
    }
    return L->launch(a, L->cols, stream);
  };
  s.flops_per_window = L->flops_per_window;
  {  // what the matrix cores are asked to do: whole 16-column tiles, channels padded to 4, folded / polyphase taps
    const int step_cols = L->g.tn() - (L->g.epi == EPI_HEAD ? 8 : 0);
    const double tiles = (L->cols + step_cols - 1) / step_cols;
    s.set_issued(2.0 * tiles * L->g.M() * L->g.tn() * L->g.cinp() * L->g.taps * L->n_sets, 0.0, 0.0);
  }
  steps.push_back(std::move(s));
}

int Net::finalize_layout() {
  size_t off = 0;
  for (size_t i = 0; i < tensors.size(); ++i) {
    Tensor& t = tensors[i];
    t.ls = round_up(t.need, 4);
    // offsets are turned into pointers by upload(); stash the offset in `need`'s place
    off += (size_t)t.C * t.ls * max_batch * tensor_sets[i];
    off = (off + 63) / 64 * 64;
  }
  (void)off;
  return VP_OK;
}

int Net::upload() {
  // collect every constant blob
  blobs.clear();
  for (auto& c : convs) {
    for (HostBlob* b : {&c->afrag, &c->bias, &c->e0, &c->e1, &c->e2})
      if (!b->h.empty()) blobs.push_back(b);
  }
  for (auto& e : extra)
    if (!e->h.empty()) blobs.push_back(e.get());

  size_t total = 0;
  std::vector<size_t> toff(tensors.size()), boff(blobs.size());
  for (size_t i = 0; i < tensors.size(); ++i) {
    toff[i] = total;
    total += (size_t)tensors[i].C * tensors[i].ls * max_batch * tensor_sets[i];
    total = (total + 63) / 64 * 64;
  }
  const size_t y_off = total;
  total += (size_t)max_batch * n_out * in_samples;
  total = (total + 63) / 64 * 64;
  for (size_t i = 0; i < blobs.size(); ++i) {
    boff[i] = total;
    total += blobs[i]->h.size();
    total = (total + 63) / 64 * 64;
  }
  arena_floats = total;
  VP_HIP(hipMalloc(&arena, total * sizeof(float)));
  VP_HIP(hipMemset(arena, 0, total * sizeof(float)));  // zero halos / margins, once
  for (size_t i = 0; i < tensors.size(); ++i) tensors[i].p = arena + toff[i];
  y = arena + y_off;
  for (size_t i = 0; i < blobs.size(); ++i) {
    blobs[i]->d = arena + boff[i];
    VP_HIP(hipMemcpy(blobs[i]->d, blobs[i]->h.data(), blobs[i]->h.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  for (auto& c : convs) {
    if (c->lds_bytes > 48 * 1024) {
      VP_HIP(hipFuncSetAttribute(c->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes));
    }
  }
  for (auto& k : extra_kernels) {
    if (k.second > 48 * 1024) {
      VP_HIP(hipFuncSetAttribute(k.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.second));
    }
  }
  return VP_OK;
}

int Net::run(int B, hipStream_t stream) {
  if (B <= 0 || B > max_batch) {
    set_error("batch %d outside (0, %d]", B, max_batch);
    return VP_ERR_INVALID;
  }
  for (auto& s : steps) {
    int rc = s.run(*this, B, stream);
    if (rc != 0) return rc;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch failed: %s", hipGetErrorString(e));
    return VP_ERR_HIP;
  }
  return VP_OK;
}

void Net::release() {
  if (arena) (void)hipFree(arena);
  arena = nullptr;
}

}  // namespace vp
