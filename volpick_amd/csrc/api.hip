// C ABI of libvolpick_hip.so (include/volpick_hip.h).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <numeric>

#include "net.h"
#include "prepost.h"

namespace vp {
const char* last_error();
}

struct vp_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  vp::Net net;
  hipEvent_t ev[5] = {};
  float total_ms = 0.f;
  bool timing = false;  // stage events cost ~5 us of stream bubble each: off unless vp_set_timing(h, 1)
  float stage_ms[4] = {0.f, 0.f, 0.f, 0.f};
  // the window description of the latest preprocessing batch: the vp_profile_* calls replay it through plans whose first
  // launch cuts its windows itself -- after checking that what it points to is still allocated (pre_is_replayable)
  vp::PreArgs last_pre{};
  int last_pre_windows = 0;
  int last_out_lo = 0, last_out_hi = 0;  // ... and the kept output range of that batch (Net::out_lo / out_hi)
  // growable device scratch
  float* d_in = nullptr;    // staged host input (stream or windows)
  size_t d_in_cap = 0;
  float* d_pred = nullptr;  // [n_windows][n_out][T] predictions of one annotate call
  size_t d_pred_cap = 0;
  float* d_out = nullptr;   // stacked output when the caller's buffer is on the host
  size_t d_out_cap = 0;
  // vp_classify_multi: tables (window table, block table, scan rows) and the result block
  char* d_tab = nullptr;
  size_t d_tab_cap = 0;
  char* m_pick_d = nullptr;
  char* m_pick_h = nullptr;
  size_t m_pick_bytes = 0;
  struct Slot {             // one in-flight vp_classify_submit
    char* d_pick = nullptr;  // trigger results: counters + per-spec on/off/peak/value arrays
    char* h_pick = nullptr;  // pinned host mirror, filled by ONE async copy per submit
    size_t pick_bytes = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
    int n_specs = 0, cap = 0;
    int64_t fv = -1, lv = -1, nwin = 0;
  } slot[VP_MAX_INFLIGHT];
};

namespace {

int grow(float** p, size_t* cap, size_t need) {
  if (need <= *cap) return VP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  VP_HIP(hipMalloc(p, need * sizeof(float)));
  *cap = need;
  return VP_OK;
}

int64_t count_windows(int64_t N, int T, int overlap, int64_t* n_regular, int* has_tail) {
  if (N < T) {
    *n_regular = 0;
    *has_tail = 0;
    return 0;
  }
  const int64_t step = T - overlap;
  *n_regular = (N - T) / step + 1;
  *has_tail = ((*n_regular - 1) * step + T < N) ? 1 : 0;
  return *n_regular + *has_tail;
}

vp::PreArgs pre_args(const vp_handle* h, const float* src, int dense, long N, long step, long first, int preprocess) {
  const vp::Net& net = h->net;
  const vp::Tensor& in = net.tensors[net.input];
  vp::PreArgs a{};
  a.src = src;
  a.N = N;
  a.dense = dense;
  a.T = net.in_samples;
  a.step = step;
  a.first_window = first;
  a.preprocess = preprocess;
  a.norm = net.cfg.norm;
  a.per_comp = (net.model_kind == VP_MODEL_PHASENET) ? 1 : net.cfg.norm_amp_per_comp;
  // SeisBench EQTransformer.annotate_batch_pre: norm_amp_per_comp divides by the per-channel PEAK whatever `norm`
  // says; `norm` only selects peak / std for the amplitude taken over all three channels.
  if (net.model_kind == VP_MODEL_EQTRANSFORMER && net.cfg.norm_amp_per_comp) a.norm = VP_NORM_PEAK;
  a.taper = net.cfg.taper_samples;
  a.norm_eps = net.cfg.norm_eps;
  a.dst = in.p;
  a.lsd = in.ls;
  a.wsd = (long)in.win_stride();
  a.flags = net.win_flags ? net.win_flags->d : nullptr;
  return a;
}

// ---- first come, first served between the device contexts of one GPU (PhaseNet) ----------------------------------------------
// A PhaseNet forward pass is ONE launch that fills the chip: 256 workgroups of 1024 threads, each owning its CU's LDS and
// register file.  Three device contexts keep three such launches pending, and the hardware shares the CUs that a finishing
// launch frees between ALL pending launches, workgroup by workgroup: three launches then progress together, end together,
// their post-processing and the host's turn-around come in one burst, and the chip idles ~30 us in every four steps
// (tools/trace_timeline.py, round 4).  The gate below lets forward launch n + 2 wait for the END of launch n (an event, no
// host synchronisation): at any time one launch runs and ONE waits for its CUs, so launches run in submission order, each
// step's stacking / trigger scan / publish run under the NEXT step's forward pass (they fit beside its workgroups: prepost.hip),
// and the host collects step n while step n + 1 computes.  Distance 2, not 1: the successor's workgroups still move onto
// CUs as the predecessor's leave them.
constexpr int FORWARD_GATE_MIN_WINDOWS = 192;  // 3/4 of the 256 CUs: below it a launch leaves room for another one
struct ForwardGate {
  static constexpr int RING = 8, DIST = 2;
  std::mutex mu;
  hipEvent_t done[RING] = {};
  hipStream_t by[RING] = {};  // the stream launch (seq % RING) went to
  bool made = false;
  unsigned long long seq = 0;
};
ForwardGate& forward_gate(int device) {
  static ForwardGate gates[64];
  return gates[(unsigned)device % 64];
}
struct ForwardTurn {  // RAII around one forward launch of a gated plan
  ForwardGate* g = nullptr;
  hipStream_t s;
  ForwardTurn(vp_handle* h, bool gated) : s(h->stream) {
    if (!gated) return;
    g = &forward_gate(h->device);
    g->mu.lock();
    if (!g->made) {
      for (auto& e : g->done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
      g->made = true;
    }
    if (g->seq >= ForwardGate::DIST) {
      const int k = (int)((g->seq - ForwardGate::DIST) % ForwardGate::RING);
      if (g->done[k] && g->by[k] != s) (void)hipStreamWaitEvent(s, g->done[k], 0);  // (its own stream is in order anyway)
    }
  }
  ~ForwardTurn() {
    if (!g) return;
    const int k = (int)(g->seq % ForwardGate::RING);
    if (g->done[k]) (void)hipEventRecord(g->done[k], s);
    g->by[k] = s;
    ++g->seq;
    g->mu.unlock();
  }
};

// One batch through annotate_batch_pre + the forward pass.  Plans whose first launch gathers and normalises the
// windows itself take the window description with them; otherwise gather_normalize fills the input tensor first.
int run_batch(vp_handle* h, const vp::PreArgs& pa, int nb) {
  vp::Net& net = h->net;
  if (pa.preprocess) {
    h->last_pre = pa;
    h->last_pre_windows = nb;
    h->last_out_lo = net.out_lo, h->last_out_hi = net.out_hi;
  }
  // (one-launch plans only: a plan of several launches gains from the contexts' launches interleaving; and only launches
  // that fill the chip -- one workgroup per window, 256 CUs: the few-window launches of a classify() over many short
  // blocks overlap freely across contexts, as before the gate)
  const ForwardTurn turn(h, net.model_kind == VP_MODEL_PHASENET && net.steps.size() == 1 && net.cfg.plan_flags[3] != 64 &&
                                nb >= FORWARD_GATE_MIN_WINDOWS);
  if (net.fused_pre && pa.preprocess) {
    net.pre = &pa;
    const int rc = net.run(nb, h->stream);
    net.pre = nullptr;
    if (rc == VP_OK && pa.flags && !net.fused_pre_poisons && !net.poison_in_plan)
      vp::launch_poison(net.y, pa.flags, nb, (long)net.n_out * net.in_samples, h->stream);
    return rc;
  }
  vp::launch_gather_normalize(pa, nb, h->stream);
  const int rc = net.run(nb, h->stream);
  if (rc == VP_OK && pa.flags && !net.poison_in_plan)
    vp::launch_poison(net.y, pa.flags, nb, (long)net.n_out * net.in_samples, h->stream);
  return rc;
}

}  // namespace

extern "C" {

const char* vp_last_error(void) { return vp::last_error(); }
const char* vp_version(void) { return "volpick_hip 0.4 (gfx950)"; }
int vp_abi_version(void) { return VP_ABI_VERSION; }
size_t vp_config_size(void) { return sizeof(vp_config); }

int vp_default_config(int model_kind, vp_config* cfg) {
  VP_REQUIRE(cfg != nullptr, "cfg is null");
  VP_REQUIRE(model_kind == VP_MODEL_PHASENET || model_kind == VP_MODEL_EQTRANSFORMER, "unknown model kind %d",
             model_kind);
  memset(cfg, 0, sizeof(*cfg));
  cfg->norm = VP_NORM_PEAK;
  cfg->norm_amp_per_comp = 0;
  cfg->max_batch = 256;
  cfg->bn_eps = 1e-3f;
  cfg->attention_eps = 1e-5f;
  cfg->layernorm_eps = 1e-14f;
  cfg->norm_eps = 1e-10f;
  cfg->taper_samples = (model_kind == VP_MODEL_EQTRANSFORMER) ? 6 : 0;
  return VP_OK;
}

size_t vp_weight_count(int model_kind) {
  const vp::ParamDesc* t;
  const int n = vp::param_table(model_kind, &t);
  size_t s = 0;
  for (int i = 0; i < n; ++i) s += t[i].size;
  return s;
}
int vp_param_count(int model_kind) {
  const vp::ParamDesc* t;
  return vp::param_table(model_kind, &t);
}
const char* vp_param_name(int model_kind, int index) {
  const vp::ParamDesc* t;
  const int n = vp::param_table(model_kind, &t);
  return (index >= 0 && index < n) ? t[index].name : nullptr;
}
size_t vp_param_size(int model_kind, int index) {
  const vp::ParamDesc* t;
  const int n = vp::param_table(model_kind, &t);
  return (index >= 0 && index < n) ? t[index].size : 0;
}

int vp_create(int device_id, int model_kind, const float* weights, size_t n_floats, int weights_mem,
              const vp_config* cfg, vp_handle** out) {
  VP_REQUIRE(out != nullptr && weights != nullptr, "null argument");
  VP_REQUIRE(model_kind == VP_MODEL_PHASENET || model_kind == VP_MODEL_EQTRANSFORMER, "unknown model kind %d",
             model_kind);
  VP_REQUIRE(n_floats == vp_weight_count(model_kind), "expected %zu weight floats, got %zu",
             vp_weight_count(model_kind), n_floats);
  vp_config c;
  if (cfg) {
    c = *cfg;
  } else {
    vp_default_config(model_kind, &c);
  }
  VP_REQUIRE(c.max_batch > 0 && c.max_batch <= 65535, "max_batch %d out of range", c.max_batch);
  for (int v : c.reserved) VP_REQUIRE(v == 0, "vp_config.reserved must be zero (plan selectors live in plan_flags)");
  VP_HIP(hipSetDevice(device_id));

  std::vector<float> host;
  if (weights_mem == VP_MEM_DEVICE) {  // e.g. the buffer an RCCL broadcast delivered
    host.resize(n_floats);
    VP_HIP(hipMemcpy(host.data(), weights, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    weights = host.data();
  }
  vp::ParamView pv;
  VP_REQUIRE(vp::build_param_view(model_kind, weights, n_floats, &pv), "weight blob does not match the table");

  auto* h = new vp_handle();
  h->device = device_id;
  h->net.model_kind = model_kind;
  h->net.cfg = c;
  h->net.max_batch = c.max_batch;
  int rc = (model_kind == VP_MODEL_PHASENET) ? vp::plan_phasenet(h->net, pv) : vp::plan_eqt(h->net, pv);
  if (rc == VP_OK) rc = h->net.finalize_layout();
  if (rc == VP_OK) rc = h->net.upload();
  if (rc != VP_OK) {
    h->net.release();
    delete h;
    return rc;
  }
  VP_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (auto& e : h->ev) VP_HIP(hipEventCreate(&e));
  for (auto& sl : h->slot) VP_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  *out = h;
  return VP_OK;
}

int vp_destroy(vp_handle* h) {
  if (!h) return VP_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->d_in) (void)hipFree(h->d_in);
  if (h->d_pred) (void)hipFree(h->d_pred);
  if (h->d_out) (void)hipFree(h->d_out);
  if (h->d_tab) (void)hipFree(h->d_tab);
  if (h->m_pick_d) (void)hipFree(h->m_pick_d);
  if (h->m_pick_h) (void)hipHostFree(h->m_pick_h);
  for (auto& sl : h->slot) {
    if (sl.d_pick) (void)hipFree(sl.d_pick);
    if (sl.h_pick) (void)hipHostFree(sl.h_pick);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  h->net.release();
  delete h;
  return VP_OK;
}

int vp_in_samples(const vp_handle* h) { return h ? h->net.in_samples : VP_ERR_INVALID; }
int vp_n_outputs(const vp_handle* h) { return h ? h->net.n_out : VP_ERR_INVALID; }
void* vp_stream(const vp_handle* h) { return h ? (void*)h->stream : nullptr; }

int vp_synchronize(vp_handle* h) {
  VP_REQUIRE(h != nullptr, "null handle");
  VP_HIP(hipStreamSynchronize(h->stream));
  return VP_OK;
}

int vp_forward(vp_handle* h, const float* x, int x_mem, int B, int preprocess, float* y, int y_mem) {
  VP_REQUIRE(h && x && y, "null argument");
  VP_REQUIRE(B > 0, "B must be positive");
  VP_HIP(hipSetDevice(h->device));
  vp::Net& net = h->net;
  const int T = net.in_samples;
  const size_t in_w = (size_t)3 * T, out_w = (size_t)net.n_out * T;
  float* y_saved = net.y;
  VP_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int b0 = 0; b0 < B; b0 += net.max_batch) {
    const int nb = std::min(net.max_batch, B - b0);
    const float* src = x + (size_t)b0 * in_w;
    if (x_mem == VP_MEM_HOST) {
      int rc = grow(&h->d_in, &h->d_in_cap, (size_t)net.max_batch * in_w);
      if (rc != VP_OK) return rc;
      VP_HIP(hipMemcpyAsync(h->d_in, src, nb * in_w * sizeof(float), hipMemcpyHostToDevice, h->stream));
      src = h->d_in;
    }
    if (y_mem == VP_MEM_DEVICE) net.y = y + (size_t)b0 * out_w;  // last layer writes straight into y
    int rc = run_batch(h, pre_args(h, src, 1, 0, 0, 0, preprocess), nb);
    net.y = y_saved;
    if (rc != VP_OK) return rc;
    if (y_mem == VP_MEM_HOST) {
      VP_HIP(hipMemcpyAsync(y + (size_t)b0 * out_w, net.y, nb * out_w * sizeof(float), hipMemcpyDeviceToHost,
                            h->stream));
      if (b0 + nb < B) VP_HIP(hipStreamSynchronize(h->stream));  // net.y is reused by the next chunk
    }
  }
  VP_HIP(hipEventRecord(h->ev[1], h->stream));
  VP_HIP(hipStreamSynchronize(h->stream));
  VP_HIP(hipEventElapsedTime(&h->total_ms, h->ev[0], h->ev[1]));
  h->stage_ms[0] = h->stage_ms[2] = h->stage_ms[3] = 0.f;
  h->stage_ms[1] = h->total_ms;
  return VP_OK;
}

int64_t vp_window_starts(int64_t N, int in_samples, int overlap, int64_t* starts, int64_t cap) {
  if (in_samples <= 0 || overlap < 0 || overlap >= in_samples) {
    vp::set_error("overlap %d must be in [0, in_samples=%d)", overlap, in_samples);
    return VP_ERR_INVALID;
  }
  int64_t n_reg;
  int tail;
  const int64_t n = count_windows(N, in_samples, overlap, &n_reg, &tail);
  const int64_t step = in_samples - overlap;
  for (int64_t i = 0; i < n && i < cap; ++i) starts[i] = (i < n_reg) ? i * step : N - in_samples;
  return n;
}

// Shared body of vp_annotate / vp_classify: everything up to the stacked (n_out, N) rows in
// device memory.  No host synchronisation.
static int annotate_device(vp_handle* h, const float* stream, int stream_mem, int64_t N, int overlap, int blind_l,
                           int blind_r, int stacking, int batch, float* d_out, int64_t* first_valid,
                           int64_t* last_valid, int64_t* n_windows) {
  vp::Net& net = h->net;
  const int T = net.in_samples;
  VP_REQUIRE(N > 0, "N must be positive");
  VP_REQUIRE(overlap >= 0 && overlap < T, "overlap %d must be in [0, %d)", overlap, T);
  VP_REQUIRE(blind_l >= 0 && blind_r >= 0 && blind_l + blind_r < T, "blinding (%d, %d) leaves no samples", blind_l,
             blind_r);
  VP_REQUIRE(stacking == VP_STACK_AVG || stacking == VP_STACK_MAX, "unknown stacking %d", stacking);
  if (batch <= 0 || batch > net.max_batch) batch = net.max_batch;

  int64_t n_reg;
  int tail;
  const int64_t nwin = count_windows(N, T, overlap, &n_reg, &tail);
  if (n_windows) *n_windows = nwin;
  const long step = T - overlap;
  // valid (un-blinded) output range: union of [s_i + blind_l, s_i + T - blind_r)
  int64_t fv = -1, lv = -1;
  if (nwin > 0) {
    fv = blind_l;
    lv = (tail ? N - T : (n_reg - 1) * step) + T - blind_r - 1;
  }
  if (first_valid) *first_valid = fv;
  if (last_valid) *last_valid = lv;

  if (h->timing) VP_HIP(hipEventRecord(h->ev[0], h->stream));
  const float* d_stream = stream;
  if (stream_mem == VP_MEM_HOST) {
    int rc = grow(&h->d_in, &h->d_in_cap, std::max((size_t)3 * N, (size_t)net.max_batch * 3 * T));
    if (rc != VP_OK) return rc;
    VP_HIP(hipMemcpyAsync(h->d_in, stream, (size_t)3 * N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    d_stream = h->d_in;
  }
  if (nwin > 0) {
    const size_t out_w = (size_t)net.n_out * T;
    int rc = grow(&h->d_pred, &h->d_pred_cap, (size_t)nwin * out_w);
    if (rc != VP_OK) return rc;
    float* y_saved = net.y;
    for (int64_t w0 = 0; w0 < nwin; w0 += batch) {
      const int nb = (int)std::min<int64_t>(batch, nwin - w0);
      net.y = h->d_pred + (size_t)w0 * out_w;
      net.out_lo = blind_l, net.out_hi = T - blind_r;  // what stacking will read of every window
      rc = run_batch(h, pre_args(h, d_stream, 0, N, step, w0, 1), nb);
      net.out_lo = net.out_hi = 0;
      net.y = y_saved;
      if (rc != VP_OK) return rc;
    }
  }
  if (h->timing) VP_HIP(hipEventRecord(h->ev[1], h->stream));
  vp::StackArgs sa{};
  sa.pred = h->d_pred;
  sa.out = d_out;
  sa.N = N;
  sa.T = T;
  sa.n_out = net.n_out;
  sa.step = step;
  sa.n_regular = n_reg;
  sa.has_tail = tail;
  sa.blind_l = blind_l;
  sa.blind_r = blind_r;
  sa.mode = stacking;
  vp::launch_stack(sa, h->stream);  // with zero windows this writes all-NaN rows
  if (h->timing) VP_HIP(hipEventRecord(h->ev[2], h->stream));
  return VP_OK;
}

static void read_stage_timing(vp_handle* h, bool with_scan) {
  if (!h->timing) return;
  float fwd = 0.f, stk = 0.f, scan = 0.f;
  (void)hipEventElapsedTime(&fwd, h->ev[0], h->ev[1]);
  (void)hipEventElapsedTime(&stk, h->ev[1], h->ev[2]);
  if (with_scan) (void)hipEventElapsedTime(&scan, h->ev[3], h->ev[4]);
  h->stage_ms[0] = 0.f;
  h->stage_ms[1] = fwd;  // gather/normalise + forward of every batch
  h->stage_ms[2] = stk;
  h->stage_ms[3] = scan;
  h->total_ms = fwd + stk + scan;
}

int vp_annotate(vp_handle* h, const float* stream, int stream_mem, int64_t N, int overlap, int blind_l,
                int blind_r, int stacking, int batch, float* out, int out_mem, int64_t* first_valid,
                int64_t* last_valid, int64_t* n_windows) {
  VP_REQUIRE(h && stream && out, "null argument");
  VP_HIP(hipSetDevice(h->device));
  float* d_out = out;
  if (out_mem == VP_MEM_HOST) {
    int rc = grow(&h->d_out, &h->d_out_cap, (size_t)h->net.n_out * std::max<int64_t>(N, 1));
    if (rc != VP_OK) return rc;
    d_out = h->d_out;
  }
  int rc = annotate_device(h, stream, stream_mem, N, overlap, blind_l, blind_r, stacking, batch, d_out, first_valid,
                           last_valid, n_windows);
  if (rc != VP_OK) return rc;
  if (out_mem == VP_MEM_HOST) {
    VP_HIP(hipMemcpyAsync(out, d_out, (size_t)h->net.n_out * N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  }
  VP_HIP(hipStreamSynchronize(h->stream));
  read_stage_timing(h, false);
  return VP_OK;
}

// ---- trigger scan of one or more device rows; ONE device->host copy, one synchronisation ----
struct ScanLayout {
  size_t header, per_spec, total;
  int cap;
};
static ScanLayout scan_layout(int n_specs, int cap) {
  ScanLayout L;
  L.cap = std::max(cap, 1);
  L.header = ((size_t)n_specs * 2 * sizeof(int) + 255) / 256 * 256;
  L.per_spec = (size_t)L.cap * (3 * sizeof(int64_t) + sizeof(float));
  L.per_spec = (L.per_spec + 255) / 256 * 256;
  L.total = L.header + L.per_spec * n_specs;
  return L;
}

// Enqueue the trigger scan of `n_specs` device rows into slot `sl` (no host synchronisation).
static int scan_submit(vp_handle* h, vp_handle::Slot& sl, const float* const* rows, const int64_t* lens,
                       const float* thr_on, const float* thr_off, int n_specs, int cap) {
  const ScanLayout L = scan_layout(n_specs, cap);
  if (L.total > sl.pick_bytes) {
    if (sl.d_pick) (void)hipFree(sl.d_pick);
    if (sl.h_pick) (void)hipHostFree(sl.h_pick);
    sl.d_pick = sl.h_pick = nullptr;
    sl.pick_bytes = 0;
    VP_HIP(hipMalloc((void**)&sl.d_pick, L.total));
    VP_HIP(hipMemset(sl.d_pick, 0, L.total));  // counters start at zero; publish_kernel re-arms them
    VP_HIP(hipHostMalloc((void**)&sl.h_pick, L.total, hipHostMallocMapped));
    sl.pick_bytes = L.total;
  }
  if (h->timing) VP_HIP(hipEventRecord(h->ev[3], h->stream));
  for (int i0 = 0; i0 < n_specs; i0 += vp::kMaxPickRows) {
    vp::PickBatch batch{};
    for (int i = i0; i < n_specs && i < i0 + vp::kMaxPickRows; ++i) {
      char* base = sl.d_pick + L.header + L.per_spec * i;
      vp::PickArgs& a = batch.a[batch.n++];
      a.trace = rows[i];
      a.n = lens[i] > 0 ? lens[i] : 0;
      a.thr_on = thr_on[i];
      a.thr_off = thr_off[i];
      a.count = (int*)sl.d_pick + 2 * i;
      a.on = (int64_t*)base;
      a.off = a.on + L.cap;
      a.peak = a.off + L.cap;
      a.value = (float*)(a.peak + L.cap);
      a.cap = cap;
    }
    vp::launch_pick(batch, h->stream);
  }
  vp::launch_publish(sl.d_pick, sl.h_pick, n_specs, cap, (long)L.header, (long)L.per_spec, h->stream);
  if (h->timing) VP_HIP(hipEventRecord(h->ev[4], h->stream));
  sl.n_specs = n_specs;
  sl.cap = cap;
  return VP_OK;
}

// After the slot's `done` event: sort each spec's triggers by onset and hand them out.
static int scan_collect(const vp_handle::Slot& sl, int64_t* on, int64_t* off, int64_t* peak, float* value,
                        int32_t* spec_of, int cap, int* n_found) {
  const ScanLayout L = scan_layout(sl.n_specs, sl.cap);
  int total = 0, written = 0;
  for (int i = 0; i < sl.n_specs; ++i) {
    const int found = ((const int*)sl.h_pick)[2 * i];
    total += found;
    const int m = std::min(found, sl.cap);
    const char* base = sl.h_pick + L.header + L.per_spec * i;
    const int64_t* t_on = (const int64_t*)base;
    const int64_t* t_off = t_on + L.cap;
    const int64_t* t_pk = t_off + L.cap;
    const float* t_v = (const float*)(t_pk + L.cap);
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return t_on[x] < t_on[y]; });
    for (int k = 0; k < m && written < cap; ++k, ++written) {
      on[written] = t_on[order[k]];
      off[written] = t_off[order[k]];
      peak[written] = t_pk[order[k]];
      value[written] = t_v[order[k]];
      if (spec_of) spec_of[written] = i;
    }
  }
  *n_found = total;
  return VP_OK;
}

int vp_classify_submit(vp_handle* h, int slot, const float* stream, int stream_mem, int64_t N, int overlap,
                       int blind_l, int blind_r, int stacking, int batch, const vp_trigger_spec* specs, int n_specs,
                       float* out, int out_mem, int cap) {
  VP_REQUIRE(h && stream, "null argument");
  VP_REQUIRE(slot >= 0 && slot < VP_MAX_INFLIGHT, "slot %d outside [0, %d)", slot, VP_MAX_INFLIGHT);
  VP_REQUIRE(!h->slot[slot].busy, "slot %d has an uncollected submit", slot);
  VP_REQUIRE(n_specs >= 0 && n_specs <= 16 && (n_specs == 0 || specs), "bad trigger specs");
  VP_REQUIRE(cap >= 0, "negative cap");
  VP_HIP(hipSetDevice(h->device));
  const int n_out = h->net.n_out;
  for (int i = 0; i < n_specs; ++i) {
    VP_REQUIRE(specs[i].row >= 0 && specs[i].row < n_out, "spec %d: row %d out of range", i, specs[i].row);
    VP_REQUIRE(specs[i].thr_off <= specs[i].thr_on, "spec %d: thr_off must not exceed thr_on", i);
  }
  vp_handle::Slot& sl = h->slot[slot];
  float* d_out = (out && out_mem == VP_MEM_DEVICE) ? out : nullptr;
  if (!d_out) {
    const size_t need = (size_t)n_out * std::max<int64_t>(N, 1);
    if (need > h->d_out_cap) VP_HIP(hipStreamSynchronize(h->stream));  // in-flight work may still read the old buffer
    int rc = grow(&h->d_out, &h->d_out_cap, need);
    if (rc != VP_OK) return rc;
    d_out = h->d_out;
  }
  int rc = annotate_device(h, stream, stream_mem, N, overlap, blind_l, blind_r, stacking, batch, d_out, &sl.fv, &sl.lv,
                           &sl.nwin);
  if (rc != VP_OK) return rc;
  if (out && out_mem == VP_MEM_HOST) {
    VP_HIP(hipMemcpyAsync(out, d_out, (size_t)n_out * N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  }
  const float* rows[16];
  int64_t lens[16];
  float t_on[16], t_off[16];
  for (int i = 0; i < n_specs; ++i) {
    rows[i] = d_out + (size_t)specs[i].row * N;  // NaN outside [fv, lv] never triggers
    lens[i] = N;
    t_on[i] = specs[i].thr_on;
    t_off[i] = specs[i].thr_off;
  }
  sl.n_specs = 0;
  if (n_specs > 0) {
    rc = scan_submit(h, sl, rows, lens, t_on, t_off, n_specs, cap);
    if (rc != VP_OK) return rc;
  }
  VP_HIP(hipEventRecord(sl.done, h->stream));
  sl.busy = true;
  return VP_OK;
}

int vp_classify_collect(vp_handle* h, int slot, int64_t* first_valid, int64_t* last_valid, int64_t* n_windows,
                        int64_t* on, int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap,
                        int* n_found) {
  VP_REQUIRE(h && n_found, "null argument");
  VP_REQUIRE(slot >= 0 && slot < VP_MAX_INFLIGHT, "slot %d outside [0, %d)", slot, VP_MAX_INFLIGHT);
  VP_REQUIRE(h->slot[slot].busy, "slot %d has nothing to collect", slot);
  VP_REQUIRE(cap >= 0 && (cap == 0 || (on && off && peak && value)), "null output arrays");
  vp_handle::Slot& sl = h->slot[slot];
  VP_HIP(hipSetDevice(h->device));
  VP_HIP(hipEventSynchronize(sl.done));
  sl.busy = false;
  read_stage_timing(h, sl.n_specs > 0);  // no-op unless vp_set_timing(h, 1); events of the latest submit
  if (first_valid) *first_valid = sl.fv;
  if (last_valid) *last_valid = sl.lv;
  if (n_windows) *n_windows = sl.nwin;
  *n_found = 0;
  if (sl.n_specs > 0) return scan_collect(sl, on, off, peak, value, spec_of, cap, n_found);
  return VP_OK;
}

int vp_classify(vp_handle* h, const float* stream, int stream_mem, int64_t N, int overlap, int blind_l,
                int blind_r, int stacking, int batch, const vp_trigger_spec* specs, int n_specs, float* out,
                int out_mem, int64_t* first_valid, int64_t* last_valid, int64_t* n_windows, int64_t* on,
                int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap, int* n_found) {
  VP_REQUIRE(h && stream && n_found, "null argument");
  VP_REQUIRE(cap >= 0 && (cap == 0 || (on && off && peak && value)), "null output arrays");
  int slot = -1;
  for (int i = 0; i < VP_MAX_INFLIGHT; ++i)
    if (!h->slot[i].busy) {
      slot = i;
      break;
    }
  VP_REQUIRE(slot >= 0, "all %d in-flight slots are busy", VP_MAX_INFLIGHT);
  int rc = vp_classify_submit(h, slot, stream, stream_mem, N, overlap, blind_l, blind_r, stacking, batch, specs,
                              n_specs, out, out_mem, cap);
  if (rc != VP_OK) return rc;
  rc = vp_classify_collect(h, slot, first_valid, last_valid, n_windows, on, off, peak, value, spec_of, cap, n_found);
  if (rc != VP_OK) return rc;
  VP_HIP(hipStreamSynchronize(h->stream));  // also covers a host-side `out` copy
  read_stage_timing(h, n_specs > 0);
  return VP_OK;
}

// The trigger scan alone, over several rows of one device array in ONE launch and ONE synchronisation (what classify() runs
// on the stacked rows of a day-long block that annotate has left on the device).
int vp_pick_rows(vp_handle* h, const float* rows_dev, int64_t N, const vp_trigger_spec* specs, int n_specs, int64_t* on,
                 int64_t* off, int64_t* peak, float* value, int32_t* spec_of, int cap, int* n_found) {
  VP_REQUIRE(h && rows_dev && n_found && N > 0, "null / empty argument");
  VP_REQUIRE(n_specs > 0 && n_specs <= 16 && specs, "bad trigger specs");
  VP_REQUIRE(cap >= 0 && (cap == 0 || (on && off && peak && value)), "null output arrays");
  VP_HIP(hipSetDevice(h->device));
  int slot = -1;
  for (int i = 0; i < VP_MAX_INFLIGHT; ++i)
    if (!h->slot[i].busy) {
      slot = i;
      break;
    }
  VP_REQUIRE(slot >= 0, "all %d in-flight slots are busy", VP_MAX_INFLIGHT);
  const float* rows[16];
  int64_t lens[16];
  float t_on[16], t_off[16];
  for (int i = 0; i < n_specs; ++i) {
    VP_REQUIRE(specs[i].row >= 0 && specs[i].row < h->net.n_out, "spec %d: row %d out of range", i, specs[i].row);
    VP_REQUIRE(specs[i].thr_off <= specs[i].thr_on, "spec %d: thr_off must not exceed thr_on", i);
    rows[i] = rows_dev + (size_t)specs[i].row * N;
    lens[i] = N;
    t_on[i] = specs[i].thr_on;
    t_off[i] = specs[i].thr_off;
  }
  vp_handle::Slot& sl = h->slot[slot];
  int rc = scan_submit(h, sl, rows, lens, t_on, t_off, n_specs, cap);
  if (rc != VP_OK) return rc;
  VP_HIP(hipStreamSynchronize(h->stream));
  return scan_collect(sl, on, off, peak, value, spec_of, cap, n_found);
}

// K stream blocks in one call: the windows of ALL blocks fill the forward batches together, as the
// reference's batch_size spans every fragment of the stream it is given (README.md:54-66); stacking and
// the trigger scan of all blocks are one launch each.
int vp_classify_multi(vp_handle* h, const float* streams, int stream_mem, const int64_t* offsets,
                      const int64_t* lengths, int K, int overlap, int blind_l, int blind_r, int stacking, int batch,
                      const vp_trigger_spec* specs, int n_specs, float* out, int out_mem, int64_t* first_valid,
                      int64_t* last_valid, int64_t* n_windows, int64_t* on, int64_t* off, int64_t* peak, float* value,
                      int32_t* spec_of, int32_t* block_of, int cap_per_row, int cap, int* n_found) {
  VP_REQUIRE(h && streams && offsets && lengths && n_found, "null argument");
  VP_REQUIRE(K > 0 && K <= 65535, "block count %d out of range", K);
  VP_REQUIRE(n_specs >= 0 && n_specs <= 16 && (n_specs == 0 || specs), "bad trigger specs");
  VP_REQUIRE(cap >= 0 && cap_per_row > 0 && (cap == 0 || (on && off && peak && value)), "bad result capacity");
  vp::Net& net = h->net;
  const int T = net.in_samples, n_out = net.n_out;
  VP_REQUIRE(n_out == 3, "vp_classify_multi expects three output rows per block");
  VP_REQUIRE(overlap >= 0 && overlap < T, "overlap %d must be in [0, %d)", overlap, T);
  VP_REQUIRE(blind_l >= 0 && blind_r >= 0 && blind_l + blind_r < T, "blinding (%d, %d) leaves no samples", blind_l,
             blind_r);
  VP_REQUIRE(stacking == VP_STACK_AVG || stacking == VP_STACK_MAX, "unknown stacking %d", stacking);
  for (int i = 0; i < n_specs; ++i) {
    VP_REQUIRE(specs[i].row >= 0 && specs[i].row < n_out, "spec %d: row %d out of range", i, specs[i].row);
    VP_REQUIRE(specs[i].thr_off <= specs[i].thr_on, "spec %d: thr_off must not exceed thr_on", i);
  }
  for (int i = 0; i < VP_MAX_INFLIGHT; ++i) VP_REQUIRE(!h->slot[i].busy, "uncollected vp_classify_submit on this handle");
  if (batch <= 0 || batch > net.max_batch) batch = net.max_batch;
  VP_HIP(hipSetDevice(h->device));
  const long step = T - overlap;

  // ---- host tables ------------------------------------------------------------------------
  std::vector<long> wtab;
  std::vector<vp::StackBlock> blocks(K);
  int64_t total = 0, span = 0, n_max = 0;
  for (int k = 0; k < K; ++k) {
    const int64_t N = lengths[k];
    VP_REQUIRE(N > 0 && offsets[k] >= 0, "block %d: bad offset / length", k);
    int64_t n_reg;
    int tail;
    const int64_t nw = count_windows(N, T, overlap, &n_reg, &tail);
    vp::StackBlock& b = blocks[k];
    b.off = offsets[k];
    b.N = N;
    b.n_regular = n_reg;
    b.has_tail = tail;
    b.w0 = (long)(wtab.size() / 3);
    b.cum = total;
    b.pad = 0;
    for (int64_t i = 0; i < nw; ++i) {
      wtab.push_back(offsets[k]);
      wtab.push_back(N);
      wtab.push_back(i < n_reg ? i * step : N - T);
    }
    if (n_windows) n_windows[k] = nw;
    if (first_valid) first_valid[k] = nw > 0 ? blind_l : -1;
    if (last_valid) last_valid[k] = nw > 0 ? (tail ? N - T : (n_reg - 1) * step) + T - blind_r - 1 : -1;
    total += N;
    span = std::max<int64_t>(span, offsets[k] + 3 * N);
    n_max = std::max(n_max, N);
  }
  const int64_t W = (int64_t)(wtab.size() / 3);
  const int R = K * n_specs;

  // ---- device buffers ------------------------------------------------------------------------
  const float* d_streams = streams;
  if (stream_mem == VP_MEM_HOST) {
    int rc = grow(&h->d_in, &h->d_in_cap, std::max((size_t)span, (size_t)net.max_batch * 3 * T));
    if (rc != VP_OK) return rc;
    VP_HIP(hipMemcpyAsync(h->d_in, streams, (size_t)span * sizeof(float), hipMemcpyHostToDevice, h->stream));
    d_streams = h->d_in;
  }
  float* d_out = (out && out_mem == VP_MEM_DEVICE) ? out : nullptr;
  if (!d_out) {
    int rc = grow(&h->d_out, &h->d_out_cap, (size_t)span);
    if (rc != VP_OK) return rc;
    d_out = h->d_out;
  }
  const size_t wt_bytes = (wtab.size() * sizeof(long) + 255) / 256 * 256;
  const size_t bt_bytes = ((size_t)K * sizeof(vp::StackBlock) + 255) / 256 * 256;
  const size_t rt_bytes = ((size_t)std::max(R, 1) * sizeof(vp::PickArgs) + 255) / 256 * 256;
  if (wt_bytes + bt_bytes + rt_bytes > h->d_tab_cap) {
    if (h->d_tab) (void)hipFree(h->d_tab);
    h->d_tab = nullptr;
    h->d_tab_cap = 0;
    VP_HIP(hipMalloc((void**)&h->d_tab, wt_bytes + bt_bytes + rt_bytes));
    h->d_tab_cap = wt_bytes + bt_bytes + rt_bytes;
  }
  long* d_wtab = reinterpret_cast<long*>(h->d_tab);
  vp::StackBlock* d_blocks = reinterpret_cast<vp::StackBlock*>(h->d_tab + wt_bytes);
  vp::PickArgs* d_rows = reinterpret_cast<vp::PickArgs*>(h->d_tab + wt_bytes + bt_bytes);
  const ScanLayout L = scan_layout(std::max(R, 1), cap_per_row);
  if (L.total > h->m_pick_bytes) {
    if (h->m_pick_d) (void)hipFree(h->m_pick_d);
    if (h->m_pick_h) (void)hipHostFree(h->m_pick_h);
    h->m_pick_d = h->m_pick_h = nullptr;
    h->m_pick_bytes = 0;
    VP_HIP(hipMalloc((void**)&h->m_pick_d, L.total));
    VP_HIP(hipHostMalloc((void**)&h->m_pick_h, L.total, hipHostMallocMapped));
    h->m_pick_bytes = L.total;
  }
  VP_HIP(hipMemsetAsync(h->m_pick_d, 0, L.header, h->stream));  // counters
  std::vector<vp::PickArgs> rows(R);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < n_specs; ++i) {
      const int r = k * n_specs + i;
      char* base = h->m_pick_d + L.header + L.per_spec * r;
      vp::PickArgs& a = rows[r];
      a.trace = d_out + offsets[k] + (size_t)specs[i].row * lengths[k];
      a.n = lengths[k];
      a.thr_on = specs[i].thr_on;
      a.thr_off = specs[i].thr_off;
      a.count = (int*)h->m_pick_d + 2 * r;
      a.on = (int64_t*)base;
      a.off = a.on + L.cap;
      a.peak = a.off + L.cap;
      a.value = (float*)(a.peak + L.cap);
      a.cap = cap_per_row;
    }
  if (W > 0) VP_HIP(hipMemcpyAsync(d_wtab, wtab.data(), wtab.size() * sizeof(long), hipMemcpyHostToDevice, h->stream));
  VP_HIP(hipMemcpyAsync(d_blocks, blocks.data(), (size_t)K * sizeof(vp::StackBlock), hipMemcpyHostToDevice, h->stream));
  if (R > 0) VP_HIP(hipMemcpyAsync(d_rows, rows.data(), (size_t)R * sizeof(vp::PickArgs), hipMemcpyHostToDevice, h->stream));
  // the three tables above are pageable host memory: the copies are complete when the calls return

  // ---- forward over all windows, stack, scan ------------------------------------------------
  if (W > 0) {
    const size_t out_w = (size_t)n_out * T;
    int rc = grow(&h->d_pred, &h->d_pred_cap, (size_t)W * out_w);
    if (rc != VP_OK) return rc;
    float* y_saved = net.y;
    for (int64_t w0 = 0; w0 < W; w0 += batch) {
      const int nb = (int)std::min<int64_t>(batch, W - w0);
      vp::PreArgs pa = pre_args(h, d_streams, 0, 0, step, w0, 1);
      pa.table = d_wtab;
      net.y = h->d_pred + (size_t)w0 * out_w;
      rc = run_batch(h, pa, nb);
      net.y = y_saved;
      if (rc != VP_OK) return rc;
    }
  }
  vp::StackMultiArgs sa{};
  sa.pred = h->d_pred;
  sa.out = d_out;
  sa.blocks = d_blocks;
  sa.n_blocks = K;
  sa.total = total;
  sa.T = T;
  sa.n_out = n_out;
  sa.step = step;
  sa.blind_l = blind_l;
  sa.blind_r = blind_r;
  sa.mode = stacking;
  vp::launch_stack_multi(sa, h->stream);
  if (out && out_mem == VP_MEM_HOST)
    VP_HIP(hipMemcpyAsync(out, d_out, (size_t)span * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (R > 0) {
    vp::launch_pick_table(d_rows, R, n_max, h->stream);
    vp::launch_publish_table(h->m_pick_d, h->m_pick_h, R, cap_per_row, (long)L.header, (long)L.per_spec, h->stream);
  }
  VP_HIP(hipStreamSynchronize(h->stream));

  // ---- results: per (block, spec) sorted by onset; a row that overflowed its cap_per_row is reported whole
  // in *n_found (found > written tells the caller to retry with more room) ---------------------------------
  int total_found = 0, written = 0;
  for (int r = 0; r < R; ++r) {
    const int found = ((const int*)h->m_pick_h)[2 * r];
    total_found += found;
    const int m = std::min(found, cap_per_row);
    const char* base = h->m_pick_h + L.header + L.per_spec * r;
    const int64_t* t_on = (const int64_t*)base;
    const int64_t* t_off = t_on + L.cap;
    const int64_t* t_pk = t_off + L.cap;
    const float* t_v = (const float*)(t_pk + L.cap);
    std::vector<int> order(m);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return t_on[x] < t_on[y]; });
    for (int k = 0; k < m && written < cap; ++k, ++written) {
      on[written] = t_on[order[k]];
      off[written] = t_off[order[k]];
      peak[written] = t_pk[order[k]];
      value[written] = t_v[order[k]];
      if (spec_of) spec_of[written] = r % n_specs;
      if (block_of) block_of[written] = r / n_specs;
    }
    if (found > cap_per_row) total_found = std::max(total_found, cap + 1);  // forces the retry path
  }
  *n_found = total_found;
  return VP_OK;
}

int vp_pick_host(const float* trace, int64_t n, float thr_on, float thr_off, int64_t* on, int64_t* off,
                 int64_t* peak, float* value, int cap, int* n_found) {
  VP_REQUIRE(trace && n_found, "null argument");
  VP_REQUIRE(cap == 0 || (on && off && peak && value), "null output arrays");
  return vp::pick_host(trace, n, thr_on, thr_off, on, off, peak, value, cap, n_found);
}

int vp_pick(vp_handle* h, const float* trace, int trace_mem, int64_t n, float thr_on, float thr_off, int64_t* on,
            int64_t* off, int64_t* peak, float* value, int cap, int* n_found) {
  VP_REQUIRE(h && trace && n_found, "null argument");
  VP_REQUIRE(cap >= 0 && (cap == 0 || (on && off && peak && value)), "null output arrays");
  if (trace_mem == VP_MEM_HOST || thr_off > thr_on) {
    // host-resident traces are scanned where they are; thr_off > thr_on uses the general rule
    std::vector<float> tmp;
    if (trace_mem == VP_MEM_DEVICE) {
      tmp.resize(n);
      VP_HIP(hipSetDevice(h->device));
      VP_HIP(hipMemcpy(tmp.data(), trace, n * sizeof(float), hipMemcpyDeviceToHost));
      trace = tmp.data();
    }
    return vp::pick_host(trace, n, thr_on, thr_off, on, off, peak, value, cap, n_found);
  }
  VP_HIP(hipSetDevice(h->device));
  const float* rows[1] = {trace};
  const int64_t lens[1] = {n};
  int slot = -1;
  for (int i = 0; i < VP_MAX_INFLIGHT; ++i)
    if (!h->slot[i].busy) {
      slot = i;
      break;
    }
  VP_REQUIRE(slot >= 0, "all %d in-flight slots are busy", VP_MAX_INFLIGHT);
  vp_handle::Slot& sl = h->slot[slot];
  int rc = scan_submit(h, sl, rows, lens, &thr_on, &thr_off, 1, cap);
  if (rc != VP_OK) return rc;
  VP_HIP(hipStreamSynchronize(h->stream));
  if (h->timing) (void)hipEventElapsedTime(&h->stage_ms[3], h->ev[3], h->ev[4]);
  rc = scan_collect(sl, on, off, peak, value, nullptr, cap, n_found);
  if (rc != VP_OK || *n_found <= cap || cap == 0) return rc;
  // More triggers than `cap`: the device kept whichever `cap` of them were appended first.  The contract (and the
  // host mirror) is the EARLIEST cap by onset, so scan again with room for all and hand out the head of the list.
  const int all = *n_found;
  std::vector<int64_t> t_on(all), t_off(all), t_pk(all);
  std::vector<float> t_v(all);
  rc = scan_submit(h, sl, rows, lens, &thr_on, &thr_off, 1, all);
  if (rc != VP_OK) return rc;
  VP_HIP(hipStreamSynchronize(h->stream));
  int again = 0;
  rc = scan_collect(sl, t_on.data(), t_off.data(), t_pk.data(), t_v.data(), nullptr, all, &again);
  if (rc != VP_OK) return rc;
  for (int i = 0; i < cap && i < again; ++i) {
    on[i] = t_on[i];
    off[i] = t_off[i];
    peak[i] = t_pk[i];
    value[i] = t_v[i];
  }
  return VP_OK;
}

// Replaces the per-sample Python loop of the reference's evaluate() (eval_taks0.py:96-142).
int vp_pick_windows(vp_handle* h, const float* prob, int prob_mem, int B, int n_rows, int row, const int32_t* lo,
                    const int32_t* hi, float thr_on, float thr_off, int K, int32_t* count, int32_t* peak,
                    float* value) {
  VP_REQUIRE(h && prob && count && peak && value, "null argument");
  VP_REQUIRE(B > 0 && K > 0 && n_rows > 0 && row >= 0 && row < n_rows, "bad shape arguments");
  VP_REQUIRE(thr_off <= thr_on, "thr_off must not exceed thr_on");
  VP_HIP(hipSetDevice(h->device));
  const int T = h->net.in_samples;
  const size_t n_prob = (size_t)B * n_rows * T;
  // scratch layout (floats): [prob copy if host][lo B][hi B][count B][peak B*K][value B*K]
  const size_t need = (prob_mem == VP_MEM_HOST ? n_prob : 0) + (size_t)B * 3 + (size_t)B * K * 2 + 64;
  int rc = grow(&h->d_in, &h->d_in_cap, std::max(need, (size_t)h->net.max_batch * 3 * T));
  if (rc != VP_OK) return rc;
  float* p = h->d_in;
  const float* d_prob = prob;
  if (prob_mem == VP_MEM_HOST) {
    VP_HIP(hipMemcpyAsync(p, prob, n_prob * sizeof(float), hipMemcpyHostToDevice, h->stream));
    d_prob = p;
    p += n_prob;
  }
  int* d_lo = reinterpret_cast<int*>(p);
  int* d_hi = d_lo + B;
  int* d_count = d_hi + B;
  int* d_peak = d_count + B;
  float* d_value = reinterpret_cast<float*>(d_peak + (size_t)B * K);
  if (lo) VP_HIP(hipMemcpyAsync(d_lo, lo, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (hi) VP_HIP(hipMemcpyAsync(d_hi, hi, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  vp::WindowPickArgs a{};
  a.prob = d_prob;
  a.B = B;
  a.n_rows = n_rows;
  a.T = T;
  a.row = row;
  a.lo = lo ? d_lo : nullptr;
  a.hi = hi ? d_hi : nullptr;
  a.thr_on = thr_on;
  a.thr_off = thr_off;
  a.K = K;
  a.count = d_count;
  a.peak = d_peak;
  a.value = d_value;
  VP_HIP(hipMemsetAsync(d_count, 0, B * sizeof(int), h->stream));
  vp::launch_window_pick(a, h->stream);
  VP_HIP(hipMemcpyAsync(count, d_count, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  VP_HIP(hipMemcpyAsync(peak, d_peak, (size_t)B * K * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  VP_HIP(hipMemcpyAsync(value, d_value, (size_t)B * K * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  VP_HIP(hipStreamSynchronize(h->stream));
  return VP_OK;
}

int vp_set_timing(vp_handle* h, int enable) {
  VP_REQUIRE(h != nullptr, "null handle");
  h->timing = enable != 0;
  return VP_OK;
}

int vp_last_timing(const vp_handle* h, float* total_ms, float stage_ms[4]) {
  VP_REQUIRE(h != nullptr, "null handle");
  if (total_ms) *total_ms = h->total_ms;
  if (stage_ms)
    for (int i = 0; i < 4; ++i) stage_ms[i] = h->stage_ms[i];
  return VP_OK;
}

// ---- introspection used by bench.py (per-kernel HIP-event timing) and the parity tests ----
int vp_step_count(const vp_handle* h) { return h ? (int)h->net.steps.size() : VP_ERR_INVALID; }

int vp_step_info(const vp_handle* h, int index, const char** name, double* flops_per_window) {
  VP_REQUIRE(h && index >= 0 && index < (int)h->net.steps.size(), "bad step index");
  if (name) *name = h->net.steps[index].name.c_str();
  if (flops_per_window) *flops_per_window = h->net.steps[index].flops_per_window;
  return VP_OK;
}

int vp_step_issued_flops(const vp_handle* h, int index, double* issued_flops_per_window) {
  VP_REQUIRE(h && issued_flops_per_window && index >= 0 && index < (int)h->net.steps.size(), "bad step index");
  vp_issued_work w;
  const int rc = vp_step_issued_work(h, index, &w);
  if (rc != VP_OK) return rc;
  *issued_flops_per_window = w.mfma_f32_flop + w.mfma_bf16_flop / 6.0 + w.valu_flop;
  return VP_OK;
}

int vp_step_issued_work(const vp_handle* h, int index, vp_issued_work* out) {
  VP_REQUIRE(h && out && index >= 0 && index < (int)h->net.steps.size(), "bad step index");
  const vp::Step& s = h->net.steps[index];
  out->mfma_f32_flop = s.issued_f32;
  out->mfma_bf16_flop = s.issued_bf16;
  out->valu_flop = s.issued_valu;
  if (s.issued_for_range) {  // the tiling the vp_profile_* calls time: the output range the latest preprocessing batch kept
    double w[3];
    s.issued_for_range(h->net, h->last_out_lo, h->last_out_hi, w);
    out->mfma_f32_flop = w[0], out->mfma_bf16_flop = w[1], out->valu_flop = w[2];
  }
  return VP_OK;
}

int vp_step_issued_work_for_range(const vp_handle* h, int index, int out_lo, int out_hi, vp_issued_work* out) {
  VP_REQUIRE(h && out && index >= 0 && index < (int)h->net.steps.size(), "bad step index");
  const vp::Step& s = h->net.steps[index];
  double w[3] = {s.issued_f32, s.issued_bf16, s.issued_valu};
  if (s.issued_for_range) s.issued_for_range(h->net, out_lo, out_hi, w);
  out->mfma_f32_flop = w[0], out->mfma_bf16_flop = w[1], out->valu_flop = w[2];
  return VP_OK;
}
double vp_flops_per_window(const vp_handle* h) { return h ? h->net.flops_per_window : 0.0; }

// Plans whose first launch cuts and normalises its windows itself are profiled the way they run: on the windows of the
// handle's latest preprocessing batch (when it had at least B of them); otherwise on the input tensor.
// The description is only a description: its src / table pointers may be the caller's device stream, the handle's staging
// buffer or the window table of a vp_classify_multi call, any of which may have been freed or regrown since.  It is
// replayed only if every byte the first B windows would read still lies inside a live device allocation
// (hipMemGetAddressRange fails for a pointer that has been freed); otherwise the launches read the input tensor.
static bool range_is_live(const void* p, size_t bytes) {
  if (!p) return false;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  const char* lo = static_cast<const char*>(base);
  const char* q = static_cast<const char*>(p);
  return q >= lo && bytes <= size && (size_t)(q - lo) <= size - bytes;
}

static bool pre_is_replayable(const vp::PreArgs& pa, int B) {
  if (!pa.src || B <= 0) return false;
  if (pa.dense) return range_is_live(pa.src, (size_t)B * 3 * pa.T * sizeof(float));
  if (!pa.table) return pa.N >= pa.T && range_is_live(pa.src, (size_t)3 * pa.N * sizeof(float));
  const size_t n = (size_t)(pa.first_window + B) * 3;
  if (!range_is_live(pa.table, n * sizeof(long))) return false;
  std::vector<long> tab(n);
  if (hipMemcpy(tab.data(), pa.table, n * sizeof(long), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  for (long w = pa.first_window; w < pa.first_window + B; ++w) {  // (block offset in src, block length, window start)
    const long off = tab[3 * w], len = tab[3 * w + 1], start = tab[3 * w + 2];
    if (off < 0 || len < pa.T || start < 0 || start + pa.T > len) return false;
    if (!range_is_live(pa.src + off, (size_t)3 * len * sizeof(float))) return false;
  }
  return true;
}

struct ProfilePre {
  vp::Net& net;
  ProfilePre(vp_handle* h, int B) : net(h->net) {
    if (net.fused_pre && h->last_pre_windows >= B && pre_is_replayable(h->last_pre, B)) net.pre = &h->last_pre;
    net.out_lo = h->last_out_lo, net.out_hi = h->last_out_hi;  // the launches are profiled over the range the last call kept
  }
  ~ProfilePre() {
    net.pre = nullptr;
    net.out_lo = net.out_hi = 0;
  }
};

// Runs every launch of the forward pass `iters` times on B windows (whatever the input
// tensor currently holds) and reports the mean duration of each launch in milliseconds,
// measured with HIP events on the handle's stream.
int vp_profile_steps(vp_handle* h, int B, int iters, float* step_ms, int cap) {
  VP_REQUIRE(h && step_ms && iters > 0, "bad argument");
  vp::Net& net = h->net;
  VP_REQUIRE(B > 0 && B <= net.max_batch, "B outside (0, max_batch]");
  VP_HIP(hipSetDevice(h->device));
  ProfilePre pre_scope(h, B);
  const int n = (int)net.steps.size();
  for (int s = 0; s < n && s < cap; ++s) {
    int rc = net.steps[s].run(net, B, h->stream);  // warm
    if (rc != 0) return rc;
    VP_HIP(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < iters; ++i) net.steps[s].run(net, B, h->stream);
    VP_HIP(hipEventRecord(h->ev[1], h->stream));
    VP_HIP(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    VP_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    step_ms[s] = ms / iters;
  }
  return VP_OK;
}

// One launch of the plan, `iters` times back to back on the handle's stream (inputs = whatever the buffers hold).
// Two handles driven from two host threads show whether two kernels share the chip (tools/overlap_probe.py).
int vp_profile_one_step(vp_handle* h, int B, int iters, int index, float* ms) {
  VP_REQUIRE(h && ms && iters > 0, "bad argument");
  vp::Net& net = h->net;
  VP_REQUIRE(B > 0 && B <= net.max_batch, "B outside (0, max_batch]");
  VP_REQUIRE(index >= 0 && index < (int)net.steps.size(), "step index %d outside [0, %d)", index, (int)net.steps.size());
  VP_HIP(hipSetDevice(h->device));
  ProfilePre pre_scope(h, B);
  int rc = net.steps[index].run(net, B, h->stream);  // warm
  if (rc != 0) return rc;
  VP_HIP(hipEventRecord(h->ev[0], h->stream));
  for (int i = 0; i < iters; ++i) net.steps[index].run(net, B, h->stream);
  VP_HIP(hipEventRecord(h->ev[1], h->stream));
  VP_HIP(hipStreamSynchronize(h->stream));
  VP_HIP(hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
  *ms /= iters;
  return VP_OK;
}

// Mean duration of ONE step measured where it runs in practice: the whole step list executes in order
// `iters` times and only step `index` is bracketed by events, so its inputs arrive from the preceding
// kernel (not from a warm re-run of itself) -- the number rocprofv3's per-kernel AverageNs is compared with.
int vp_profile_step_in_pipeline(vp_handle* h, int B, int iters, int index, float* ms) {
  VP_REQUIRE(h && ms && iters > 0 && iters <= 4096, "bad argument");
  vp::Net& net = h->net;
  VP_REQUIRE(B > 0 && B <= net.max_batch, "B outside (0, max_batch]");
  const int n = (int)net.steps.size();
  VP_REQUIRE(index >= 0 && index < n, "step index %d outside [0, %d)", index, n);
  VP_HIP(hipSetDevice(h->device));
  ProfilePre pre_scope(h, B);
  int rc = net.run(B, h->stream);  // warm
  if (rc != VP_OK) return rc;
  // every pass is enqueued before the first one is waited for: the launch is timed the way it runs in a busy
  // pipeline (a synchronisation per pass lets the GPU idle between passes, and its clock with it)
  std::vector<hipEvent_t> ev(2 * (size_t)iters, nullptr);
  for (auto& e : ev) VP_HIP(hipEventCreate(&e));
  for (int i = 0; i < iters && rc == VP_OK; ++i) {
    for (int s = 0; s < n && rc == VP_OK; ++s) {
      if (s == index) (void)hipEventRecord(ev[2 * i], h->stream);
      rc = net.steps[s].run(net, B, h->stream);
      if (s == index) (void)hipEventRecord(ev[2 * i + 1], h->stream);
    }
  }
  hipError_t err = hipStreamSynchronize(h->stream);
  double total = 0.0;
  for (int i = 0; i < iters && rc == VP_OK && err == hipSuccess; ++i) {
    float t = 0.f;
    err = hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]);
    total += t;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc != VP_OK) return rc;
  VP_HIP(err);
  *ms = (float)(total / iters);
  return VP_OK;
}

int vp_debug_tensor_count(const vp_handle* h) { return h ? (int)h->net.tensors.size() : VP_ERR_INVALID; }

int vp_debug_tensor_info(const vp_handle* h, int index, const char** name, int* channels, int* length) {
  VP_REQUIRE(h && index >= 0 && index < (int)h->net.tensors.size(), "bad tensor index");
  const vp::Tensor& t = h->net.tensors[index];
  if (name) *name = t.name.c_str();
  if (channels) *channels = t.C;
  if (length) *length = t.L;
  return VP_OK;
}

int vp_debug_tensor_read(vp_handle* h, int index, int B, float* host_out) {
  VP_REQUIRE(h && host_out && index >= 0 && index < (int)h->net.tensors.size(), "bad argument");
  const vp::Tensor& t = h->net.tensors[index];
  if (h->net.tensor_sets[index] == 0) {  // lives in LDS only under this plan (e.g. decoder.4 / .5 inside eqt_tail_kernel)
    vp::set_error("tensor %s is not materialised by this plan", t.name.c_str());
    return VP_ERR_UNSUPPORTED;
  }
  VP_REQUIRE(B > 0 && B <= h->net.max_batch * h->net.tensor_sets[index], "bad B");
  VP_HIP(hipSetDevice(h->device));
  VP_HIP(hipStreamSynchronize(h->stream));
  // strided copy: (B*C rows) x L floats out of rows of stride ls, skipping the halo
  VP_HIP(hipMemcpy2D(host_out, (size_t)t.L * sizeof(float), t.p + vp::HALO, (size_t)t.ls * sizeof(float),
                     (size_t)t.L * sizeof(float), (size_t)B * t.C, hipMemcpyDeviceToHost));
  return VP_OK;
}

// Debug guard of the halo-is-padding invariant (vp_common.h): the margins of every activation row are zeroed once
// at vp_create and no kernel may ever write them.  Scans all rows of all tensors of the handle's arena.
int vp_debug_check_halos(vp_handle* h, int self_test, int64_t* n_bad, const char** first_bad_tensor) {
  VP_REQUIRE(h && n_bad, "null argument");
  VP_HIP(hipSetDevice(h->device));
  vp::Net& net = h->net;
  const size_t nt = net.tensors.size();
  VP_REQUIRE(self_test >= 0 && self_test <= (int)nt, "self_test names tensor %d of %zu", self_test - 1, nt);
  float* planted = nullptr;
  if (self_test > 0) {  // plant ONE stray word in the right margin of the tensor's last row, restored below
    const vp::Tensor& t = net.tensors[self_test - 1];
    const size_t rows = (size_t)t.C * net.max_batch * net.tensor_sets[self_test - 1];
    VP_REQUIRE(rows > 0, "tensor %s is not materialised by this plan", t.name.c_str());
    planted = t.p + (rows - 1) * t.ls + vp::HALO + t.L;
    const float one = 1.f;
    VP_HIP(hipMemcpyAsync(planted, &one, sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  int* d_bad = nullptr;
  VP_HIP(hipMalloc((void**)&d_bad, (nt + 1) * sizeof(int)));
  VP_HIP(hipMemsetAsync(d_bad, 0, (nt + 1) * sizeof(int), h->stream));
  for (size_t i = 0; i < nt; ++i) {
    const vp::Tensor& t = net.tensors[i];
    vp::launch_halo_check(t.p, (long)t.C * net.max_batch * net.tensor_sets[i], t.ls, t.L, d_bad + i, h->stream);
  }
  std::vector<int> bad(nt + 1, 0);
  hipError_t e = hipMemcpyAsync(bad.data(), d_bad, (nt + 1) * sizeof(int), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && planted) e = hipMemsetAsync(planted, 0, sizeof(float), h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d_bad);
  VP_HIP(e);
  *n_bad = 0;
  if (first_bad_tensor) *first_bad_tensor = nullptr;
  for (size_t i = 0; i < nt; ++i) {
    if (bad[i] && *n_bad == 0 && first_bad_tensor) *first_bad_tensor = net.tensors[i].name.c_str();
    *n_bad += bad[i];
  }
  return VP_OK;
}

// Debug: per-window shader-clock stamps of the fused PhaseNet core kernel (plan flag plan_flags[1]).
// Debug: 8 stamps per conv launch (workgroup tile 1 / window 7): start, loaded, mfma done, staged, stored.
int vp_debug_conv_clock(vp_handle* h, unsigned long long* out, int max_layers) {
  VP_REQUIRE(h && out && h->net.debug_clock && h->net.debug_clock->d, "no clock stamps (create with plan_flags[1] & 2)");
  VP_HIP(hipSetDevice(h->device));
  VP_HIP(hipStreamSynchronize(h->stream));
  const int n = std::min<int>(max_layers, 64);
  const unsigned long long* src =
      reinterpret_cast<const unsigned long long*>(h->net.debug_clock->d) + (size_t)h->net.max_batch * 32;
  VP_HIP(hipMemcpy(out, src, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return (int)h->net.convs.size();
}

int vp_debug_core_clock(vp_handle* h, int B, unsigned long long* out32) {
  VP_REQUIRE(h && out32 && h->net.debug_clock && h->net.debug_clock->d, "no clock stamps (create with plan_flags[1]=1)");
  VP_HIP(hipSetDevice(h->device));
  VP_HIP(hipStreamSynchronize(h->stream));
  VP_HIP(hipMemcpy(out32, h->net.debug_clock->d, (size_t)B * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return VP_OK;
}

// Debug: the stamps of eqt_tail_kernel (third region of the EQTransformer debug clock buffer, eqt.hip).
int vp_debug_tail_clock(vp_handle* h, int B, unsigned long long* out32) {
  VP_REQUIRE(h && out32 && h->net.debug_clock && h->net.debug_clock->d, "no clock stamps (create with plan_flags[1] & 2)");
  VP_REQUIRE(h->net.model_kind == VP_MODEL_EQTRANSFORMER && B > 0 && B <= h->net.max_batch, "EQTransformer handles only");
  VP_HIP(hipSetDevice(h->device));
  VP_HIP(hipStreamSynchronize(h->stream));
  const unsigned long long* src =
      reinterpret_cast<const unsigned long long*>(h->net.debug_clock->d) + (size_t)h->net.max_batch * 32 + 64 * 8;
  VP_HIP(hipMemcpy(out32, src, (size_t)B * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return VP_OK;
}

// Host-only: packed constants of one conv layer of a freshly planned model (no GPU
// needed); lets the CPU test-suite check BN folding and MFMA fragment packing.
int vp_debug_plan_conv(int model_kind, const float* weights, size_t n_floats, const vp_config* cfg,
                       int conv_index, int* geom13, float* afrag, size_t afrag_cap, float* bias, size_t bias_cap,
                       const char** name, int* cols, int* l_out) {
  static thread_local std::string name_store;
  vp::ParamView pv;
  VP_REQUIRE(vp::build_param_view(model_kind, weights, n_floats, &pv), "weight blob does not match the table");
  vp::Net net;
  net.model_kind = model_kind;
  if (cfg) {
    net.cfg = *cfg;
  } else {
    vp_default_config(model_kind, &net.cfg);
  }
  net.max_batch = 1;
  int rc = (model_kind == VP_MODEL_PHASENET) ? vp::plan_phasenet(net, pv) : vp::plan_eqt(net, pv);
  if (rc != VP_OK) return rc;
  if (conv_index < 0 || conv_index >= (int)net.convs.size()) return -1000 - (int)net.convs.size();
  const vp::ConvLayer& L = *net.convs[conv_index];
  const vp::ConvGeom& g = L.g;
  const int gv[13] = {g.cin1, g.cin2, g.cout, g.P, g.taps, g.sn, g.in_off, g.out_off, g.waves_m, g.waves_n, g.nw,
                      g.relu, g.epi};
  if (geom13) memcpy(geom13, gv, sizeof(gv));
  if (afrag) memcpy(afrag, L.afrag.h.data(), std::min(afrag_cap, L.afrag.h.size()) * sizeof(float));
  if (bias) memcpy(bias, L.bias.h.data(), std::min(bias_cap, L.bias.h.size()) * sizeof(float));
  name_store = L.name;
  if (name) *name = name_store.c_str();
  if (cols) *cols = L.cols;
  if (l_out) *l_out = L.l_out;
  return (int)L.afrag.h.size();
}

}  // extern "C"
