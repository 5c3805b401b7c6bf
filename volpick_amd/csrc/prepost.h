// Argument blocks of the HBM-bound pre/post kernels (prepost.hip).
#pragma once
#include "vp_common.h"

namespace vp {

struct PreArgs {
  const float* src;  // stream rows (3, N) or dense windows (B, 3, T)
  long N;            // stream length (row stride) when !dense
  int dense;         // 1: src is (B,3,T)
  int T;             // in_samples
  long step;         // in_samples - overlap
  long first_window; // index of window 0 of this batch within the stream
  int preprocess;    // 0: copy only (model(x) semantics)
  int norm;          // VP_NORM_*
  int per_comp;
  int taper;
  float norm_eps;
  float* dst;        // haloed input tensor
  int lsd;
  long wsd;
  float* flags;      // optional [n_windows]: 1 where the window holds a non-finite sample (NaN / Inf statistics), else 0
  const long* table; // optional, 3 longs per window of the whole call: block offset in src, block length (row stride),
                     // window start; window w of this launch is entry first_window + w (multi-block classify)
};
int launch_gather_normalize(const PreArgs& a, int n_windows, hipStream_t stream);
// The reference (torch) carries a NaN / Inf of the input into every output sample of its window: demeaning spreads it
// over the channel, the first conv over all channels.  The kernels here lose it at the first ReLU (v_max is maxNum), so the
// predictions of flagged windows are overwritten with NaN: y[w][0 .. floats_per_window) for every w with flags[w] != 0.
int launch_poison(float* y, const float* flags, int n_windows, long floats_per_window, hipStream_t stream);

// Wavefront reductions on the DPP network (six VALU steps: xor 1, xor 2, half-row mirror, row mirror, row broadcast 15,
// row broadcast 31, then lane 63 holds the result) instead of six ds_bpermute round trips through the LDS hardware
// (~100 cycles each).  Every lane receives the result.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_move(float v) {
  const int x = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false));
}
__device__ inline float wave_sum(float v) {
  v += dpp_move<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_move<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row's sum
  const float r1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
  v += r1;
  const float r2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3
  v += r2;
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xF>(v));
  v = fmaxf(v, dpp_move<0x4E, 0xF>(v));
  v = fmaxf(v, dpp_move<0x141, 0xF>(v));
  v = fmaxf(v, dpp_move<0x140, 0xF>(v));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v));  // lanes outside the row mask keep their own value
  v = fmaxf(v, dpp_move<0x143, 0xC>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Three independent wave reductions, interleaved by hand: every step is ONE DPP instruction per row and the two
// instructions of the other rows in between cover the two wait states a DPP read needs after a VALU write.  (From
// wave_max / wave_sum above hipcc makes mov_dpp + canonicalising max + max + s_nop per step and did not interleave the rows:
// ~80 serial instructions per row, 3.6 k cycles for the six rows of a wave.)
#define VP_DPP3(OP, CTRL)                                                                            \
  "v_" OP "_dpp %0, %0, %0 " CTRL "\n\tv_" OP "_dpp %1, %1, %1 " CTRL "\n\tv_" OP "_dpp %2, %2, %2 " CTRL "\n\t"
#define VP_REDUCE3(OP)                                                           \
  asm volatile("s_nop 1\n\t" VP_DPP3(OP, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")      \
               VP_DPP3(OP, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")                     \
               VP_DPP3(OP, "row_half_mirror row_mask:0xf bank_mask:0xf")                         \
               VP_DPP3(OP, "row_mirror row_mask:0xf bank_mask:0xf")                              \
               VP_DPP3(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                            \
               VP_DPP3(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"                  \
               : "+v"(a), "+v"(b), "+v"(c));                                                    \
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));                            \
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));                            \
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
__device__ __forceinline__ void wave_max3(float& a, float& b, float& c) { VP_REDUCE3("max_f32") }
__device__ __forceinline__ void wave_sum3(float& a, float& b, float& c) { VP_REDUCE3("add_f32") }
#undef VP_REDUCE3
#undef VP_DPP3


// n / den for many n and ONE den (the amplitude of a window channel): the compiler's own division sequence -- v_rcp_f32, one
// Newton step on the reciprocal, quotient, two residual corrections -- with the part that depends on den alone done once.
// Bit for bit the quotient `n / den` yields wherever that sequence does not rescale its operands (v_div_scale: exponents
// near the ends of the range; a window whose amplitude is 0, Inf or NaN is flagged and poisoned whatever comes out here).
// 5 vector instructions per sample instead of 12 (annotate_batch_pre inside pn_window_kernel: 9 divisions per lane).
struct NormDiv {
  float den, r;
};
__device__ __forceinline__ NormDiv norm_div_prepare(const float den) {
  float r = __builtin_amdgcn_rcpf(den);
  r = fmaf(fmaf(-den, r, 1.f), r, r);
  return NormDiv{den, r};
}
__device__ __forceinline__ float norm_div(const float n, const NormDiv d) {
  float q = n * d.r;
  q = fmaf(fmaf(-d.den, q, n), d.r, q);
  return fmaf(fmaf(-d.den, q, n), d.r, q);
}

struct StackArgs {
  const float* pred;  // [n_windows][n_out][T]
  float* out;         // [n_out][N]
  long N;
  int T, n_out;
  long step;
  long n_regular;     // windows at i*step
  int has_tail;       // one more window at N - T
  int blind_l, blind_r;
  int mode;           // VP_STACK_*
};
int launch_stack(const StackArgs& a, hipStream_t stream);

// Several stream blocks stacked by one launch (vp_classify_multi).  Block k: rows of N samples at float offset
// `off` of both the input and the output buffer, windows [w0, w0 + n_regular + has_tail) of pred.
struct StackBlock {
  long off, N, n_regular, w0, cum;  // cum: samples of all earlier blocks (grid index space)
  int has_tail, pad;
};
struct StackMultiArgs {
  const float* pred;
  float* out;
  const StackBlock* blocks;
  int n_blocks;
  long total;  // sum of N
  int T, n_out;
  long step;
  int blind_l, blind_r, mode;
};
int launch_stack_multi(const StackMultiArgs& a, hipStream_t stream);

struct PickArgs {
  const float* trace;
  long n;
  float thr_on, thr_off;
  int64_t *on, *off, *peak;
  float* value;
  int cap;
  int* count;
};
constexpr int kMaxPickRows = 4;
struct PickBatch {
  PickArgs a[kMaxPickRows];
  int n;
};
int launch_pick(const PickBatch& b, hipStream_t stream);
// the same scan over a table of rows in device memory (any number of rows; n_max = longest row)
int launch_pick_table(const PickArgs* rows, int n_rows, long n_max, hipStream_t stream);
int launch_publish_table(char* dev, char* host, int n_rows, int cap, long header, long per_row, hipStream_t stream);
struct WindowPickArgs {
  const float* prob;  // [B][n_rows][T]
  int B, n_rows, T, row;
  const int* lo;      // per-window borders (may be null: whole window)
  const int* hi;
  float thr_on, thr_off;
  int K;              // slots per window
  int* count;         // [B] triggers found (may exceed K)
  int* peak;          // [B][K] peak index relative to lo
  float* value;       // [B][K]
};
int launch_window_pick(const WindowPickArgs& a, hipStream_t stream);
int launch_publish(char* dev, char* host, int n_specs, int cap, long header, long per_spec, hipStream_t stream);
// Debug guard of the halo-is-padding invariant (vp_common.h): counts the non-zero words in the margins
// [0, HALO) and [HALO + L, ls) of `rows` rows of stride ls; adds the count to *bad (device int).
int launch_halo_check(const float* rows_base, long rows, int ls, int L, int* bad, hipStream_t stream);
int pick_host(const float* x, int64_t n, float thr_on, float thr_off, int64_t* on, int64_t* off, int64_t* peak,
              float* value, int cap, int* n_found);

}  // namespace vp
