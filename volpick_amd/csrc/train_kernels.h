// Kernels of the PhaseNet training step (BASELINE config 5) other than the
// convolutions themselves (forward and input-gradient convs run on conv_mfma_kernel):
// BatchNorm in training mode (batch statistics, forward and backward), the 1x1 head + softmax +
// vector cross entropy (volpick/model/models.py:34-51) with its backward, the weight-gradient
// GEMM on the fp32 matrix cores, weight packing, deterministic partial-sum reduction and Adam
// (models.py:177-185 configure_optimizers).  All arithmetic is fp32; reductions finish in fp64.
#pragma once
#include <type_traits>

#include "bf16.h"
#include "vp_common.h"

namespace vp {

// ------------------------------------------------------------------------------------------
// rows of an activation / gradient tensor: element (b, c, t) at p[b * ws + c * ls + HALO + t]
// (bf16 storage mode: p holds the address of bf16_t elements, strides count elements; the *_v kernels below take the
// element type as a template parameter and cast)
struct Rows {
  float* p;
  int ls;
  long ws;
  template <class T>
  __device__ __forceinline__ T* row(int b, int c) const {
    return reinterpret_cast<T*>(p) + (long)b * ws + (long)c * ls + HALO;
  }
};

struct BnArgs {
  Rows z;          // raw conv output, C channels, Lz samples
  Rows a;          // relu(bn(z[crop + t])), t in [0, La)
  Rows ga1, ga2;   // gradient wrt a (ga2.p may be null; the two are summed)
  Rows gz;         // gradient wrt z, Lz samples
  int C, B, Lz, La, crop;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* stats;     // [4][C]: batch mean, rstd, S1 = sum(da'), S2 = sum(da' * xhat)
  double* partial;  // [C][GB][2]
  int GB;
  const float* fpart;  // forward statistics from the conv's epilogue instead (conv_train_b16.h): [C][n_fpart][2] sums, or null
  int n_fpart;
  int sl2;          // vector kernels: log2 of the lanes that share a row (VecWalk)
  float* g_gamma;   // gradient blob slots
  float* g_beta;
  float eps, momentum;
};

// one wavefront per channel: totals of the GB partial pairs (fixed order), valid in lane 0
__device__ inline void channel_totals(const BnArgs& a, int c, double* s0, double* s1) {
  double x = 0.0, y = 0.0;
  for (int j = threadIdx.x; j < a.GB; j += 64) {
    x += a.partial[((long)c * a.GB + j) * 2];
    y += a.partial[((long)c * a.GB + j) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    x += __shfl_xor(x, o, 64);
    y += __shfl_xor(y, o, 64);
  }
  *s0 = x;
  *s1 = y;
}

// ---- BatchNorm passes, T = float or bf16_t rows ---------------------------------------------------------------------
// Rows are walked in 8-sample vectors (16 bytes of bf16).  A wave covers 64 >> sl2 rows at a time with 1 << sl2 lanes
// each (BnArgs::sl2: 6 = a whole wave per row for the 3001-sample rows ... 1 = 32 rows of 12 samples per wave), so the
// short rows of the deep layers keep all lanes busy.  grid (C, GB), 256 threads: block j, wave w take rows
// (j + GB * (w + 4 k)) * RP + lane / slots.  A few thousand long-lived workgroups instead of one short block per 1024
// samples: such per-1024-sample grids (round 1) were bound by workgroup dispatch, not by memory, once the rows shrank
// to half the bytes (fp32 rows through these kernels: 2.45 vs 2.68 ms per 512-window step).  Margins: a row's samples behind its logical length are zero and stay zero (the
// consumers' padding): vectors are read whole, written with the tail masked.
// The per-channel finalisation (mean / rstd / running statistics; S1, S2 and the gamma / beta gradients) is done by every
// apply block for itself from the GB partial sums (fixed order: deterministic, and every block gets the same values);
// block 0 of a channel writes them out.  That removes two launches per layer and direction.
//
// CROP (ConvTranspose layers: a = relu(bn(z[CROP + t]))) shifts z against a / ga by one or two samples: the shifted
// operand is read as two aligned vectors and re-indexed in registers.
template <class T, int CROP>
__device__ __forceinline__ void load8_at(const T* row, int t0, float (&v)[8]) {  // v[i] = row[t0 + CROP + i], t0 % 8 == 0
  if constexpr (CROP == 0) {
    Elem<T>::load8(row + t0, v);
  } else if constexpr (CROP > 0) {
    float f[16];
    Elem<T>::load8(row + t0, f);
    Elem<T>::load8(row + t0 + 8, f + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f[CROP + i];
  } else {  // CROP < 0: the vector starts -CROP samples before t0 (t0 = 0 reads the row's zero halo)
    float f[16];
    Elem<T>::load8(row + t0 - 8, f);
    Elem<T>::load8(row + t0, f + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f[8 + CROP + i];
  }
}

// sums of two per-thread values over a block of NW waves (fixed order), valid in every thread
template <int NW>
__device__ inline void block_sum2_f64(double& x, double& y, double* sh /*[2 * NW]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    x += __shfl_xor(x, o, 64);
    y += __shfl_xor(y, o, 64);
  }
  __syncthreads();
  if (lane == 0) {
    sh[2 * wave] = x;
    sh[2 * wave + 1] = y;
  }
  __syncthreads();
  x = y = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    x += sh[2 * w];
    y += sh[2 * w + 1];
  }
}

// row walker of the vector kernels: rows b (lane group rs of wave-pass b0), vectors v = vs, vs + slots, ...
struct VecWalk {
  int vs, rs, slots, RP;
  __device__ explicit VecWalk(int sl2) {
    const int lane = threadIdx.x & 63;
    slots = 1 << sl2;
    RP = 64 >> sl2;
    vs = lane & (slots - 1);
    rs = lane >> sl2;
  }
};

// the four per-element passes: rows b0 + rs for b0 = first, first + stride, ...
// (one 1024-thread workgroup per channel doing reduce + finalise + apply in a single launch was tried for the deep
// layers, as round 1 had it: 1.90 vs 1.86 ms per step -- the chip-wide grids win even at 12 samples per row)
template <class T>
__device__ __forceinline__ void bnv_pass_stats(const BnArgs& a, int c, const VecWalk& w, int first, int stride, float& s, float& q) {
  const int nv = (a.Lz + 7) >> 3;
  for (int b0 = first; b0 < a.B; b0 += stride) {
    const int b = b0 + w.rs;
    if (b >= a.B) continue;
    const T* z = a.z.row<T>(b, c);
    for (int v = w.vs; v < nv; v += w.slots) {
      float f[8];
      Elem<T>::load8(z + 8 * v, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s += f[i];
        q = fmaf(f[i], f[i], q);
      }
    }
  }
}

template <class T, int CROP>
__device__ __forceinline__ void bnv_pass_apply(const BnArgs& a, int c, const VecWalk& w, int first, int stride, float sc, float sh) {
  const int nv = (a.La + 7) >> 3;
  for (int b0 = first; b0 < a.B; b0 += stride) {
    const int b = b0 + w.rs;
    if (b >= a.B) continue;
    const T* z = a.z.row<T>(b, c);
    T* o = a.a.row<T>(b, c);
    for (int v = w.vs; v < nv; v += w.slots) {
      float f[8], r[8];
      load8_at<T, CROP>(z, 8 * v, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = (8 * v + i < a.La) ? fmaxf(fmaf(f[i], sc, sh), 0.f) : 0.f;
      Elem<T>::store8(o + 8 * v, r);
    }
  }
}

template <class T, int CROP>
__device__ __forceinline__ void bnv_pass_bwd_sums(const BnArgs& a, int c, const VecWalk& w, int first, int stride, float mean,
                                                  float rstd, float& s1, float& s2) {
  // The ReLU gate comes from z, not from a: a = relu(fma(z, sc, sh)) (rounded to the row type) is positive exactly where the
  // fma is, with the sc / sh of bnv_apply_kernel -- one tensor less to read in each backward pass (a quarter / a fifth of
  // their traffic).  Outside [0, La) the gradient rows hold zeros (halo, margin), so no gate is needed there.
  const float sc = a.gamma[c] * rstd, shv = a.beta[c] - mean * sc;
  const int nv = (a.La + 7) >> 3;
  for (int b0 = first; b0 < a.B; b0 += stride) {
    const int b = b0 + w.rs;
    if (b >= a.B) continue;
    const T* z = a.z.row<T>(b, c);
    const T* g1 = a.ga1.row<T>(b, c);
    const T* g2 = a.ga2.p ? a.ga2.row<T>(b, c) : nullptr;
    for (int v = w.vs; v < nv; v += w.slots) {
      float fz[8], fg[8];
      load8_at<T, CROP>(z, 8 * v, fz);
      Elem<T>::load8(g1 + 8 * v, fg);
      if (g2) {
        float f2[8];
        Elem<T>::load8(g2 + 8 * v, f2);
#pragma unroll
        for (int i = 0; i < 8; ++i) fg[i] += f2[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float g = (fmaf(fz[i], sc, shv) > 0.f) ? fg[i] : 0.f;  // ga's margin is zero: nothing behind La counts
        s1 += g;
        s2 = fmaf(g, (fz[i] - mean) * rstd, s2);
      }
    }
  }
}

template <class T, int CROP>
__device__ __forceinline__ void bnv_pass_bwd_apply(const BnArgs& a, int c, const VecWalk& w, int first, int stride, float mean,
                                                   float rstd, float m1, float m2, float k) {
  const float sc = a.gamma[c] * rstd, shv = a.beta[c] - mean * sc;  // the ReLU gate from z (bnv_pass_bwd_sums)
  const int nv = (a.Lz + 7) >> 3;
  for (int b0 = first; b0 < a.B; b0 += stride) {
    const int b = b0 + w.rs;
    if (b >= a.B) continue;
    const T* z = a.z.row<T>(b, c);
    const T* g1 = a.ga1.row<T>(b, c);
    const T* g2 = a.ga2.p ? a.ga2.row<T>(b, c) : nullptr;
    T* gz = a.gz.row<T>(b, c);
    for (int v = w.vs; v < nv; v += w.slots) {
      float fz[8], fg[8], r[8];
      Elem<T>::load8(z + 8 * v, fz);
      load8_at<T, -CROP>(g1, 8 * v, fg);  // ga at t = j - CROP
      if (g2) {
        float f2[8];
        load8_at<T, -CROP>(g2, 8 * v, f2);
#pragma unroll
        for (int i = 0; i < 8; ++i) fg[i] += f2[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float g = (fmaf(fz[i], sc, shv) > 0.f) ? fg[i] : 0.f;  // zero halo / zero margin of ga: nothing outside [0, La) counts
        const float xh = (fz[i] - mean) * rstd;
        r[i] = (8 * v + i < a.Lz) ? k * (g - m1 - xh * m2) : 0.f;
      }
      Elem<T>::store8(gz + 8 * v, r);
    }
  }
}

// finalisations from the channel's sums: batch mean / biased variance -> mean, rstd; running statistics as torch
// (momentum 0.1, unbiased variance); S1, S2 = the beta / gamma gradients
__device__ __forceinline__ void bn_finish_stats(const BnArgs& a, int c, double S, double Q, bool write, float* mean_out, float* rstd_out) {
  const double N = (double)a.B * a.Lz;
  const double mean = S / N;
  double var = Q / N - mean * mean;
  if (var < 0.0) var = 0.0;
  *mean_out = (float)mean;
  *rstd_out = (float)(1.0 / sqrt(var + (double)a.eps));
  if (write) {
    a.stats[c] = *mean_out;
    a.stats[a.C + c] = *rstd_out;
    a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)(var * N / (N - 1.0));
  }
}
__device__ __forceinline__ void bn_finish_bwd(const BnArgs& a, int c, double S1, double S2, bool write) {
  if (write) {
    a.stats[2 * a.C + c] = (float)S1;
    a.stats[3 * a.C + c] = (float)S2;
    a.g_beta[c] = (float)S1;
    a.g_gamma[c] = (float)S2;
  }
}
// totals of the GB partial pairs of channel c, the same values in every thread of the block (fixed order)
__device__ inline void channel_totals_block(const BnArgs& a, int c, double* sh /*[2]*/, double* s0, double* s1) {
  if (threadIdx.x < 64) {
    double x, y;
    channel_totals(a, c, &x, &y);
    if (threadIdx.x == 0) {
      sh[0] = x;
      sh[1] = y;
    }
  }
  __syncthreads();
  *s0 = sh[0];
  *s1 = sh[1];
}

template <class T>
__global__ __launch_bounds__(256) void bnv_stats_partial_kernel(const BnArgs a) {
  __shared__ double sh[8];
  const int c = blockIdx.x, j = blockIdx.y, wave = threadIdx.x >> 6;
  const VecWalk w(a.sl2);
  float s = 0.f, q = 0.f;
  bnv_pass_stats<T>(a, c, w, (j + wave * a.GB) * w.RP, 4 * a.GB * w.RP, s, q);
  double S = (double)s, Q = (double)q;
  block_sum2_f64<4>(S, Q, sh);
  if (threadIdx.x == 0) {
    a.partial[((long)c * a.GB + j) * 2] = S;
    a.partial[((long)c * a.GB + j) * 2 + 1] = Q;
  }
}

// the same from the per-workgroup sums a conv launch left (BnArgs::fpart): all four waves, fixed order
__device__ inline void channel_totals_fpart(const BnArgs& a, int c, double* sh /*[8]*/, double* s0, double* s1) {
  const float2* p = reinterpret_cast<const float2*>(a.fpart) + (long)c * a.n_fpart;
  double x = 0.0, y = 0.0;
  for (int i0 = threadIdx.x; i0 < a.n_fpart; i0 += 256 * 8) {  // eight loads in flight (rolled: one L2 round trip per 256 entries)
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (i0 + 256 * k < a.n_fpart) ? p[i0 + 256 * k] : make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x += (double)v[k].x;
      y += (double)v[k].y;
    }
  }
  block_sum2_f64<4>(x, y, sh);
  *s0 = x;
  *s1 = y;
}

template <class T, int CROP>
__global__ __launch_bounds__(256) void bnv_apply_kernel(const BnArgs a) {
  __shared__ double sh[8];
  const int c = blockIdx.x, j = blockIdx.y, wave = threadIdx.x >> 6;
  double S, Q;
  if (a.fpart) {
    channel_totals_fpart(a, c, sh, &S, &Q);
  } else {
    channel_totals_block(a, c, sh, &S, &Q);
  }
  float mean, rstd;
  bn_finish_stats(a, c, S, Q, j == 0 && threadIdx.x == 0, &mean, &rstd);
  const float sc = a.gamma[c] * rstd, shv = a.beta[c] - mean * sc;
  const VecWalk w(a.sl2);
  bnv_pass_apply<T, CROP>(a, c, w, (j + wave * a.GB) * w.RP, 4 * a.GB * w.RP, sc, shv);
}

template <class T, int CROP>
__global__ __launch_bounds__(256) void bnv_bwd_partial_kernel(const BnArgs a) {
  __shared__ double sh[8];
  const int c = blockIdx.x, j = blockIdx.y, wave = threadIdx.x >> 6;
  const float mean = a.stats[c], rstd = a.stats[a.C + c];
  const VecWalk w(a.sl2);
  float s1 = 0.f, s2 = 0.f;
  bnv_pass_bwd_sums<T, CROP>(a, c, w, (j + wave * a.GB) * w.RP, 4 * a.GB * w.RP, mean, rstd, s1, s2);
  double S1 = (double)s1, S2 = (double)s2;
  block_sum2_f64<4>(S1, S2, sh);
  if (threadIdx.x == 0) {
    a.partial[((long)c * a.GB + j) * 2] = S1;
    a.partial[((long)c * a.GB + j) * 2 + 1] = S2;
  }
}

template <class T, int CROP>
__global__ __launch_bounds__(256) void bnv_bwd_apply_kernel(const BnArgs a) {
  __shared__ double sh[2];
  const int c = blockIdx.x, j = blockIdx.y, wave = threadIdx.x >> 6;
  double S1, S2;
  channel_totals_block(a, c, sh, &S1, &S2);
  bn_finish_bwd(a, c, S1, S2, j == 0 && threadIdx.x == 0);
  const float mean = a.stats[c], rstd = a.stats[a.C + c];
  const float invN = 1.f / ((float)a.B * (float)a.Lz);
  const float m1 = (float)S1 * invN, m2 = (float)S2 * invN;
  const VecWalk w(a.sl2);
  bnv_pass_bwd_apply<T, CROP>(a, c, w, (j + wave * a.GB) * w.RP, 4 * a.GB * w.RP, mean, rstd, m1, m2, a.gamma[c] * rstd);
}

// ------------------------------------------------------------------------------------------
// Head: logits = W (3x8) a + b, p = softmax, loss = -(1/B) sum_b sum_c (1/T) sum_t y log(p + eps)
// (vector_cross_entropy, models.py:34-51), and its backward down to ga (8 channels) plus
// block partials of [loss, db(3), dW(24)].
struct HeadArgs2 {
  Rows a;       // 8 channels
  Rows ga;      // 8 channels
  const float* y;  // dense labels [B][3][T]
  float* p;        // dense predictions [B][3][T] (may be null)
  const float* w;  // [3][8]
  const float* b;  // [3]
  double* partial; // [blocks][28]
  int B, T;
  float eps;
};

__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(const HeadArgs2 h) {
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  float w[3][8], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bb[c] = h.b[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[c][k] = h.w[c * 8 + k];
  }
  float vals[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) vals[i] = 0.f;
  if (t < h.T) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = h.a.p[(long)b * h.a.ws + (long)k * h.a.ls + HALO + t];
    float z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      z[c] = bb[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) z[c] = fmaf(w[c][k], x[k], z[c]);
    }
    const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
    float e[3] = {expf(z[0] - mx), expf(z[1] - mx), expf(z[2] - mx)};
    const float inv = 1.f / (e[0] + e[1] + e[2]);
    float p[3], g[3], gl[3];
    const float scale = 1.f / ((float)h.B * (float)h.T);
    float dot = 0.f, loss = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      p[c] = e[c] * inv;
      const float yv = h.y[((long)b * 3 + c) * h.T + t];
      loss -= yv * logf(p[c] + h.eps);
      g[c] = -yv / (p[c] + h.eps) * scale;
      dot = fmaf(g[c], p[c], dot);
      if (h.p) h.p[((long)b * 3 + c) * h.T + t] = p[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gl[c] = p[c] * (g[c] - dot);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      h.ga.p[(long)b * h.ga.ws + (long)k * h.ga.ls + HALO + t] = w[0][k] * gl[0] + w[1][k] * gl[1] + w[2][k] * gl[2];
    vals[0] = loss * scale;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vals[1 + c] = gl[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) vals[4 + c * 8 + k] = gl[c] * x[k];
    }
  }
  // 28 block sums: wavefront butterflies, then the four wave results through LDS (one barrier)
  __shared__ float shv[4][28];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    float v = vals[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) shv[wave][i] = v;
  }
  __syncthreads();
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x < 28)
    h.partial[blk * 28 + threadIdx.x] = ((double)shv[0][threadIdx.x] + (double)shv[1][threadIdx.x]) +
                                        ((double)shv[2][threadIdx.x] + (double)shv[3][threadIdx.x]);
}

// The same head with NS consecutive samples per thread (NS = 2: even-aligned pairs, 4-byte accesses of bf16 rows; NS = 4:
// quads, 8-byte accesses -- half the blocks, half the 28-value block reductions per sample: 36 -> 27 us per step).
// grid (ceil(T / (256 NS)), B), 256 threads.
template <class T, int NS = 2>
__global__ __launch_bounds__(256) void head_fwd_bwd_pair_kernel(const HeadArgs2 h) {
  static_assert(NS == 2 || (NS == 4 && sizeof(T) == 2), "pairs, or quads of bf16");
  const int b = blockIdx.y, t0 = NS * (blockIdx.x * 256 + threadIdx.x);
  float w[3][8], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bb[c] = h.b[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[c][k] = h.w[c * 8 + k];
  }
  float vals[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) vals[i] = 0.f;
  if (t0 < h.T) {
    float x[8][NS], ga[8][NS];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (NS == 2) {
        Elem<T>::load2(h.a.row<T>(b, k) + t0, reinterpret_cast<float(&)[2]>(x[k]));
      } else {
        const uint2 r = *reinterpret_cast<const uint2*>(h.a.row<T>(b, k) + t0);
        x[k][0] = bf16_lo(r.x), x[k][1] = bf16_hi(r.x), x[k][2] = bf16_lo(r.y), x[k][3] = bf16_hi(r.y);
      }
    }
    const float scale = 1.f / ((float)h.B * (float)h.T);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int t = t0 + u;
      if (t < h.T) {
        float z[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          z[c] = bb[c];
#pragma unroll
          for (int k = 0; k < 8; ++k) z[c] = fmaf(w[c][k], x[k][u], z[c]);
        }
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        float e[3] = {expf(z[0] - mx), expf(z[1] - mx), expf(z[2] - mx)};
        const float inv = 1.f / (e[0] + e[1] + e[2]);
        float p[3], g[3], gl[3];
        float dot = 0.f, loss = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          p[c] = e[c] * inv;
          const float yv = h.y[((long)b * 3 + c) * h.T + t];
          loss -= yv * logf(p[c] + h.eps);
          g[c] = -yv / (p[c] + h.eps) * scale;
          dot = fmaf(g[c], p[c], dot);
          if (h.p) h.p[((long)b * 3 + c) * h.T + t] = p[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gl[c] = p[c] * (g[c] - dot);
#pragma unroll
        for (int k = 0; k < 8; ++k) ga[k][u] = w[0][k] * gl[0] + w[1][k] * gl[1] + w[2][k] * gl[2];
        vals[0] += loss * scale;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vals[1 + c] += gl[c];
#pragma unroll
          for (int k = 0; k < 8; ++k) vals[4 + c * 8 + k] = fmaf(gl[c], x[k][u], vals[4 + c * 8 + k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) ga[k][u] = 0.f;  // the row's margin stays zero
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (NS == 2) {
        Elem<T>::store2(h.ga.row<T>(b, k) + t0, ga[k][0], ga[k][1]);
      } else {
        *reinterpret_cast<uint2*>(h.ga.row<T>(b, k) + t0) = make_uint2(pack_bf16x2(ga[k][0], ga[k][1]), pack_bf16x2(ga[k][2], ga[k][3]));
      }
    }
  }
  __shared__ float shv[4][28];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    float v = vals[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) shv[wave][i] = v;
  }
  __syncthreads();
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x < 28)
    h.partial[blk * 28 + threadIdx.x] = ((double)shv[0][threadIdx.x] + (double)shv[1][threadIdx.x]) +
                                        ((double)shv[2][threadIdx.x] + (double)shv[3][threadIdx.x]);
}

// out[r][i] = sum over rows g = r, r + R, ... of partial[g][i]  (fixed order: run-to-run deterministic).
// grid (ceil(n / 32), R): a block owns 32 columns, its 8 row lanes stride over the rows, LDS folds the lanes.
// R = 1 gives the final sum; a larger R is the first stage of a two-stage reduction over many rows.
template <typename T, typename TOut>
__global__ __launch_bounds__(256) void sum_rows_kernel(const T* __restrict__ partial, int G, int n, TOut* __restrict__ out) {
  __shared__ double sh[8][33];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  const int R = gridDim.y, r = blockIdx.y;
  double s = 0.0;
  if (col < n)
    for (int g = r + rl * R; g < G; g += 8 * R) s += (double)partial[(long)g * n + col];
  sh[rl][threadIdx.x & 31] = s;
  __syncthreads();
  if (rl == 0 && col < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][threadIdx.x & 31];
    out[(long)r * n + col] = (TOut)t;
  }
}

// All weight-gradient partials of a step folded in ONE launch: job j sums `rows` rows of `n` floats into `out`.
// A block owns 4 * cq columns and walks the rows with 256 / cq row lanes: cq = 32 (128 columns, 8 row lanes) for the wide
// jobs of the deep layers (57 k columns x 32 rows), cq = 4 (16 columns, 64 row lanes) for the level-0 layers, whose few
// hundred columns x 512 rows left seven blocks walking 64 rows each behind one another (68 us for the launch; the
// memory round trips of those seven blocks, not bytes).
struct SumJob {
  const float* partial;
  float* out;
  int rows, n, first_block, cq;  // blocks [first_block, next job's first_block) own 4 * cq columns each
};
constexpr int MAX_SUM_JOBS = 24;
struct SumJobs {
  SumJob job[MAX_SUM_JOBS];
  int count;
};
__host__ __device__ constexpr int sum_job_cq(int rows) { return rows > 100000 ? 4 : 32; }  // the 16-column form measured slower for every job of the step (bytes, not round trips, bound the launch)
__global__ __launch_bounds__(256) void sum_rows_multi_kernel(const SumJobs jobs) {
  __shared__ double sh[1024 + 256];
  int j = 0;
  while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;
  const SumJob jb = jobs.job[j];
  const int cq = jb.cq, W = 4 * cq, RL = 256 / cq;
  const int ql = (int)threadIdx.x % cq, rl = (int)threadIdx.x / cq;
  const int col = ((int)blockIdx.x - jb.first_block) * W + 4 * ql;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (col < jb.n) {  // n is a multiple of 4: a quad is inside or outside
    const float* p = jb.partial + col;
    int g = rl;
    for (; g + 3 * RL < jb.rows; g += 4 * RL) {  // four 16-byte loads in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(g + RL * u) * jb.n);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[0] += (double)v[u].x;
        s[1] += (double)v[u].y;
        s[2] += (double)v[u].z;
        s[3] += (double)v[u].w;
      }
    }
    for (; g < jb.rows; g += RL) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long)g * jb.n);
      s[0] += (double)v.x;
      s[1] += (double)v.y;
      s[2] += (double)v.z;
      s[3] += (double)v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) sh[rl * W + 4 * ql + i] = s[i];
  __syncthreads();
  // fixed order: 256 / W groups of four row lanes, then the groups
  const int c = (int)threadIdx.x % W, part = (int)threadIdx.x / W, P = 256 / W;  // RL / P = 4 row lanes per part
  {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) t += sh[(part * 4 + k) * W + c];
    sh[1024 + part * W + c] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < W) {
    const int oc = ((int)blockIdx.x - jb.first_block) * W + c;
    if (oc < jb.n) {
      double t = 0.0;
      for (int k = 0; k < P; ++k) t += sh[1024 + k * W + c];
      jb.out[oc] = (float)t;
    }
  }
}

// head partial sums -> loss (double), gradient slots of out.bias / out.weight
__global__ void head_final_kernel(const double* __restrict__ sums, double* loss, float* g_b, float* g_w) {
  const int i = threadIdx.x;
  if (i == 0) *loss = sums[0];
  if (i >= 1 && i < 4) g_b[i - 1] = (float)sums[i];
  if (i >= 4 && i < 28) g_w[i - 4] = (float)sums[i];
}

// per-channel sum over (B, L) of a tensor (the conv bias gradient of `inc`): grid (C, GB) partials, then sum_rows;
// 8-sample vectors, a wave per row (the rows' margins are zero)
template <class T>
__global__ __launch_bounds__(256) void channel_sum_partial_v_kernel(const Rows r, int B, int L, int GB, double* partial) {
  __shared__ double sh[8];
  const int c = blockIdx.x, j = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (L + 7) >> 3;
  float acc = 0.f;
  for (int b = j + wave * GB; b < B; b += 4 * GB) {
    const T* p = r.row<T>(b, c);
    for (int v = lane; v < nv; v += 64) {
      float f[8];
      Elem<T>::load8(p + 8 * v, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += f[i];
    }
  }
  double s = (double)acc, zero = 0.0;
  block_sum2_f64<4>(s, zero, sh);
  if (threadIdx.x == 0) partial[(long)j * gridDim.x + c] = s;  // [GB][C]
}

// packed MFMA A-fragments of every conv from the current weights: idx[i] = 1 + flat weight index, 0 = structural zero
__global__ __launch_bounds__(256) void gather_pack_kernel(const int* __restrict__ idx, const float* __restrict__ w,
                                                          float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = idx[i];
  out[i] = k > 0 ? w[k - 1] : 0.f;
}

// torch.optim.Adam (no weight decay, no amsgrad); `mask` = 1 for trainable entries, 0 for the BN running statistics.
// ema (may be null): exponential moving average of the weights after the step, ema = d * ema + (1 - d) * w
// (the reference's optional EMA callback, volpick/model/train.py:153-176, decay 0.999, every step); the running
// statistics are buffers, not parameters: their EMA slots simply track the current values.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ mask, float* __restrict__ ema, int n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float ema_decay) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float wi = w[i];
  if (mask[i] != 0.f) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    wi -= (lr / bc1) * (mi / denom);
    w[i] = wi;
    if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * wi;
  } else if (ema) {
    ema[i] = wi;
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient on the matrix cores:
//   D[m][h][k] = sum_b sum_n lo[b][m][n] * hi[b][h][S * n + k + off]      m < LO, h < HI1 + HI2, k < K
// conv:            lo = gz (cout rows),  hi = layer input (cin rows)       -> D = dW[cout][cin][K]
// ConvTranspose1d: lo = layer input,     hi = gz over the full output      -> D = dWt[cin][cout][K]
// GEMM view: M = LO rows, N = (k, h) columns (k-major), reduction over time in steps of 4
// (v_mfma_f32_16x16x4_f32: lane l feeds A[l & 15][l >> 4] = lo[m][n0 + (l >> 4)] and
// B[l >> 4][l & 15] = hi[h][S * (n0 + (l >> 4)) + k]).  A workgroup walks (window, time-chunk)
// items grid-stride, staging lo / hi chunks in LDS, and keeps its D tiles in registers; it writes
// one partial D per workgroup, summed afterwards in a fixed order (sum_partials_f32_kernel).
struct WgradArgs {
  Rows lo;
  Rows hi1, hi2;
  int lim_hi1, lim_hi2;  // floats readable after HALO in a hi row
  int Ln;                // valid low-rate length
  int off;               // hi sample offset of tap 0 (minus the left padding)
  int B;
  int chunks;            // time chunks per window (1 when an item holds several windows)
  float* partial;        // [gridDim.x][LO * HI * K]
};

// TT: low-rate samples per chunk; WB: windows per item (short layers batch several windows per LDS stage)
template <int LO_, int HI1_, int HI2_, int K_, int S_, int NWAVE_, int TT_, int WB_ = 1>
struct WgradCfg {
  static constexpr int LO = LO_, HI1 = HI1_, HI2 = HI2_, HI = HI1_ + HI2_, K = K_, S = S_, NWAVE = NWAVE_, TT = TT_;
  static constexpr int WB = WB_;
  static constexpr int LOP = (LO + 15) / 16 * 16, MT = LOP / 16;
  static constexpr int COLS = HI * K, NT = (COLS + 15) / 16, TILES = MT * NT;
  static constexpr int WH = S * TT + K - 1;           // staged hi samples per row and window
  static constexpr int S_LO = WB * TT + 2;            // row strides chosen to spread rows over LDS banks
  static constexpr int S_HI = ((WB * WH) | 1) + 2;
  static constexpr int LDS_FLOATS = LOP * S_LO + HI * S_HI;
  static constexpr int OUT = LO * HI * K;
  // few output tiles: every wave keeps ALL tiles and takes every NWAVE-th time step (split-K), folded through
  // LDS at the end; many tiles: a wave owns one 16-row block (A read once per step) and every NPG-th column tile
  static constexpr bool SPLITK = TILES <= 14;
  static constexpr int NPG = SPLITK ? 1 : NWAVE / MT;       // waves sharing a row block
  static constexpr int NACC = SPLITK ? TILES : NT / (NPG > 0 ? NPG : 1);
  static_assert(SPLITK || (NWAVE % MT == 0 && NT % NPG == 0), "tile assignment: NWAVE = MT * NPG, NT = NPG * NACC");
  static_assert(!SPLITK || TILES * 256 <= LDS_FLOATS, "split-K fold reuses the staging LDS");
  static_assert(TT % 4 == 0, "time chunk must be a multiple of the MFMA k (4)");
  // (the LDS budget of wgrad_kernel is asserted there: wgrad_bf16_kernel has images of its own, WgradB)
};

template <class C, class T = float>
__global__ __launch_bounds__(64 * C::NWAVE) void wgrad_kernel(const WgradArgs a) {
  // bf16 rows are fetched as even-aligned PAIRS (4-byte loads) and widened while they are parked in LDS: a hi window
  // whose first sample is odd (tap offsets -3 / -1) is staged from the sample before it, so a window holds WHS >= WH
  // samples and every B read is shifted by that one sample (par).  fp32 rows: sample by sample, as before.
  constexpr bool PAIRS = sizeof(T) == 2;
  constexpr int WHS = PAIRS ? (C::WH + 2) / 2 * 2 : C::WH;
  constexpr int S_HI = PAIRS ? ((C::WB * WHS) | 1) + 2 : C::S_HI;
  constexpr int LDS_FLOATS = C::LOP * C::S_LO + C::HI * S_HI;
  static_assert(LDS_FLOATS * 4 <= 64 * 1024 && (!C::SPLITK || C::TILES * 256 <= LDS_FLOATS), "static LDS budget");
  __shared__ float lds[LDS_FLOATS];
  float* lo_s = lds;
  float* hi_s = lds + C::LOP * C::S_LO;
  constexpr int NTH = 64 * C::NWAVE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l16 = lane & 15;
  const int par = PAIRS ? (a.off & 1) : 0;
  const int my_mt = C::SPLITK ? 0 : wave % C::MT, my_ng = C::SPLITK ? 0 : wave / C::MT;
  auto tile_of = [&](const int i, int* mt, int* nt) __attribute__((always_inline)) {
    if (C::SPLITK) {
      *mt = i / C::NT;
      *nt = i - *mt * C::NT;
    } else {
      *mt = my_mt;
      *nt = my_ng + i * C::NPG;
    }
  };
  f32x4 acc[C::NACC];
  int b_base[C::NACC];
#pragma unroll
  for (int i = 0; i < C::NACC; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int mt, nt;
    tile_of(i, &mt, &nt);
    int col = nt * 16 + l16;
    if (col >= C::COLS) col = 0;  // padding columns compute garbage that is never stored
    const int k = col / C::HI, h = col - k * C::HI;
    b_base[i] = h * S_HI + C::S * g + k + par;
  }
  for (int i = tid; i < (C::LOP - C::LO) * C::S_LO; i += NTH) lo_s[C::LO * C::S_LO + i] = 0.f;  // padding rows
  // The next item's lo / hi chunks are fetched into registers while the MFMAs of the current one run
  // (all loads of an item are issued back to back: a load -> LDS-store loop exposes the full HBM latency
  // once per element and made this kernel 3-5x slower).
  constexpr int EPL = PAIRS ? 2 : 1;  // elements per load
  constexpr int N_LO = C::LO * C::WB * C::TT / EPL, N_HI = C::HI * C::WB * WHS / EPL;
  constexpr int NLO = (N_LO + NTH - 1) / NTH, NHI = (N_HI + NTH - 1) / NTH;
  static_assert(C::TT % 2 == 0 && WHS % EPL == 0, "pairs never straddle rows");
  typedef typename std::conditional<PAIRS, unsigned, float>::type raw_t;
  raw_t pre_lo[NLO], pre_hi[NHI];
  const int groups = (a.B + C::WB - 1) / C::WB;
  const int items = groups * a.chunks;
  const T* lo_p = reinterpret_cast<const T*>(a.lo.p);
  const T* hi1_p = reinterpret_cast<const T*>(a.hi1.p);
  const T* hi2_p = reinterpret_cast<const T*>(a.hi2.p);
  auto fetch = [&](const int item) __attribute__((always_inline)) {
    const int grp = item / a.chunks, n_start = (item - grp * a.chunks) * C::TT;
    const int b0 = grp * C::WB;
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int i = (tid + k * NTH) * EPL;
      const int m = i / (C::WB * C::TT), r = i - m * (C::WB * C::TT);
      const int w = r / C::TT, n = r - w * C::TT;
      raw_t v = 0;
      if (i < N_LO * EPL && b0 + w < a.B && n_start + n < a.Ln)  // (a pair's second sample behind Ln is the row's zero margin)
        v = *reinterpret_cast<const raw_t*>(lo_p + (long)(b0 + w) * a.lo.ws + (long)m * a.lo.ls + HALO + n_start + n);
      pre_lo[k] = v;
    }
    const int s0 = C::S * n_start + a.off - par;  // hi sample index of staged column 0 (even for pairs)
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      const int i = (tid + k * NTH) * EPL;
      const int h = i / (C::WB * WHS), r = i - h * (C::WB * WHS);
      const int w = r / WHS, j = r - w * WHS;
      const int idx = s0 + j;
      raw_t v = 0;
      if (i < N_HI * EPL && b0 + w < a.B && idx >= -HALO) {
        if (h < C::HI1) {
          if (idx < a.lim_hi1) v = *reinterpret_cast<const raw_t*>(hi1_p + (long)(b0 + w) * a.hi1.ws + (long)h * a.hi1.ls + HALO + idx);
        } else {
          if (idx < a.lim_hi2)
            v = *reinterpret_cast<const raw_t*>(hi2_p + (long)(b0 + w) * a.hi2.ws + (long)(h - C::HI1) * a.hi2.ls + HALO + idx);
        }
      }
      pre_hi[k] = v;
    }
  };
  if ((int)blockIdx.x < items) fetch(blockIdx.x);
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int grp = item / a.chunks, n_start = (item - grp * a.chunks) * C::TT;
    __syncthreads();  // previous item's MFMA reads are done
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int i = (tid + k * NTH) * EPL;
      const int m = i / (C::WB * C::TT), r = i - m * (C::WB * C::TT);
      if (i < N_LO * EPL) {
        if constexpr (PAIRS) {
          lo_s[m * C::S_LO + r] = bf16_lo(pre_lo[k]);
          lo_s[m * C::S_LO + r + 1] = bf16_hi(pre_lo[k]);
        } else {
          lo_s[m * C::S_LO + r] = pre_lo[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      const int i = (tid + k * NTH) * EPL;
      const int h = i / (C::WB * WHS), r = i - h * (C::WB * WHS);
      if (i < N_HI * EPL) {
        if constexpr (PAIRS) {
          hi_s[h * S_HI + r] = bf16_lo(pre_hi[k]);
          hi_s[h * S_HI + r + 1] = bf16_hi(pre_hi[k]);
        } else {
          hi_s[h * S_HI + r] = pre_hi[k];
        }
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < items) fetch(item + gridDim.x);
    int n_len = a.Ln - n_start;
    if (n_len > C::TT) n_len = C::TT;
#pragma unroll
    for (int w = 0; w < C::WB; ++w) {
      const float* lo_w = lo_s + w * C::TT + g;
      const float* hi_w = hi_s + w * WHS;
      if constexpr (C::SPLITK) {
        for (int n0 = 4 * wave; n0 < n_len; n0 += 4 * C::NWAVE) {
          float av[C::MT];
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) av[mt] = lo_w[(mt * 16 + l16) * C::S_LO + n0];
#pragma unroll
          for (int i = 0; i < C::NACC; ++i) {
            const float bv = hi_w[b_base[i] + C::S * n0];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i / C::NT], bv, acc[i], 0, 0, 0);
          }
        }
      } else {
        for (int n0 = 0; n0 < n_len; n0 += 4) {
          const float av = lo_w[(my_mt * 16 + l16) * C::S_LO + n0];
#pragma unroll
          for (int i = 0; i < C::NACC; ++i) {
            const float bv = hi_w[b_base[i] + C::S * n0];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
          }
        }
      }
    }
  }
  if constexpr (C::SPLITK) {  // fold the waves' partial tiles into wave 0, one wave per round
    for (int w = 1; w < C::NWAVE; ++w) {
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < C::NACC; ++i) *reinterpret_cast<f32x4*>(lds + (i * 64 + lane) * 4) = acc[i];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < C::NACC; ++i) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(lds + (i * 64 + lane) * 4);
          acc[i] += o;
        }
      }
    }
    if (wave != 0) return;
  }
  float* out = a.partial + (long)blockIdx.x * C::OUT;
#pragma unroll
  for (int i = 0; i < C::NACC; ++i) {
    int mt, nt;
    tile_of(i, &mt, &nt);
    const int col = nt * 16 + l16;
    const int k = col / C::HI, h = col - k * C::HI;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + 4 * g + r;
      if (m < C::LO && col < C::COLS) out[((long)m * C::HI + h) * C::K + k] = acc[i][r];
    }
  }
}

// ---- the same weight gradient on the bf16 matrix cores (bf16 rows only) ------------------------------------------------
// Stored activations and gradients ARE bfloat16, so lo[n] * hi[n] is exact in fp32 and v_mfma_f32_16x16x32_bf16
// (16 cycles for 32 time samples: 16x the rate of the fp32 form) accumulates the same products in fp32 -- the result
// differs from the fp32-MFMA kernel above by summation order only.
//   K of an instruction = 32 consecutive time samples; lane (l & 15, g = l >> 4) supplies samples 8 g .. 8 g + 7:
//     A: row m of lo                      -> one 16-byte LDS read
//     B: column = input channel h, one TAP per output tile: hi[h][S (n + .) + k]
//        S = 1: the 8 samples of tap k start k (+ parity) samples into a 16-sample window = two aligned 16-byte reads per
//               lane for ALL seven taps; an even shift is a register renaming, an odd one four v_alignbit_b32
//        S = 4: the staging de-interleaves a hi row into four phase planes (plane p holds samples 4 j + p), tap k reads
//               plane (k + parity) & 3 shifted by (k + parity) >> 2 in {0, 1}: a 16-byte and a 4-byte read per plane
//   Output tiles are (m-tile, tap, 16-channel group); a "unit" = the seven taps of one (m-tile, channel group).  Waves
//   own units (all K-steps), or -- fewer units than waves -- share one and split the (window, K-step) pairs, folded
//   through LDS at the end.  LDS images are bf16, zero-initialised once: the padding (rows up to 16, samples up to a
//   multiple of 32) never holds anything else, so the padded products are 0 * 0.
typedef __bf16 bf16x8_native __attribute__((ext_vector_type(8)));

template <class C>
struct WgradB {
  static constexpr int NWAVE = C::NWAVE > 8 ? 8 : C::NWAVE;  // 512 threads at most: 256 registers per lane for up to 28 accumulators
  static constexpr int LOP = C::LOP, MT = C::MT, HIP = (C::HI + 15) / 16 * 16, HG = HIP / 16, NU = MT * HG;
  static constexpr int TTP = (C::TT + 31) / 32 * 32, KS = TTP / 32;
  static constexpr int WHP = (C::S == 1) ? TTP + 16 : TTP + 8;  // staged samples per window (and plane)
  static constexpr int PLANES = (C::S == 1) ? 1 : 4;
  static constexpr int RS_LO = (C::WB * TTP + 127) / 128 * 128 + 8;  // row strides == 8 elements (16 bytes) mod 128:
  static constexpr int RS_HI = (C::WB * WHP + 127) / 128 * 128 + 8;  // the 16 rows of a 16-byte read fall on disjoint banks
  static constexpr int PLANE = HIP * RS_HI + 16;                      // planes 32 bytes apart mod 256: their writes spread too
  static constexpr int LDS_ELEMS = (LOP * RS_LO + PLANES * PLANE + 7) / 8 * 8;
  static constexpr int UPW = (NU >= NWAVE) ? NU / NWAVE : 1;      // units per wave
  static constexpr int KSPLIT = (NU >= NWAVE) ? 1 : NWAVE / NU;   // waves sharing a unit
  // samples fetched per window: even-aligned pairs (S = 1), groups of eight (S = 4: a lane takes samples j, j + 1, j + 4,
  // j + 5 and writes one dword to each of two planes)
  static constexpr int WHS = (C::S == 1) ? (C::WH + 2) / 2 * 2 : (C::WH + 1 + 7) / 8 * 8;
  static_assert(C::S == 1 || C::S == 4, "stride 1 or 4");
  static_assert((NU >= NWAVE) ? (NU % NWAVE == 0) : (NWAVE % NU == 0), "units over waves");
  static_assert(HG % UPW == 0 || UPW % HG == 0, "a wave's units share an m-tile or cover whole channel-group rows");
  static_assert(LDS_ELEMS * 2 <= 160 * 1024 && (KSPLIT == 1 || NU * 7 * 256 * 4 <= LDS_ELEMS * 2), "LDS budget; the fold reuses the images");
  static_assert(C::S == 1 ? WHS <= WHP : WHS / 4 <= WHP, "a staged window fits its image");
};

template <class C, int PAR>
__global__ __launch_bounds__(64 * WgradB<C>::NWAVE) void wgrad_bf16_kernel(const WgradArgs a) {
  using W = WgradB<C>;
  extern __shared__ uint4 wg_lds_raw[];
  bf16_t* lo_s = reinterpret_cast<bf16_t*>(wg_lds_raw);
  bf16_t* hi_s = lo_s + W::LOP * W::RS_LO;
  constexpr int NTH = 64 * W::NWAVE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  for (int i = tid; i < W::LDS_ELEMS / 8; i += NTH) wg_lds_raw[i] = make_uint4(0u, 0u, 0u, 0u);
  // units of this wave: u = u0 .. u0 + UPW - 1, u = mt * HG + hg; ksub = its share of the (window, K-step) pairs
  const int u0 = (W::KSPLIT == 1) ? wave * W::UPW : wave / W::KSPLIT;
  const int ksub = (W::KSPLIT == 1) ? 0 : wave % W::KSPLIT;
  f32x4 acc[W::UPW][7];
#pragma unroll
  for (int u = 0; u < W::UPW; ++u)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fetch units: lo = an even-aligned pair; hi = a pair (S = 1) or two pairs four samples apart (S = 4)
  constexpr int HP = (C::S == 1) ? 1 : 2;                                  // pair loads per hi unit
  constexpr int N_LO = C::LO * C::WB * C::TT / 2, N_HI = C::HI * C::WB * W::WHS / (2 * HP);
  constexpr int NLO = (N_LO + NTH - 1) / NTH, NHI = (N_HI + NTH - 1) / NTH;
  unsigned pre_lo[NLO], pre_hi[NHI][HP];
  const int groups = (a.B + C::WB - 1) / C::WB;
  const int items = groups * a.chunks;
  const bf16_t* lo_p = reinterpret_cast<const bf16_t*>(a.lo.p);
  const bf16_t* hi1_p = reinterpret_cast<const bf16_t*>(a.hi1.p);
  const bf16_t* hi2_p = reinterpret_cast<const bf16_t*>(a.hi2.p);
  // hi unit i -> (row h, window w, first sample j of its first pair)
  auto hi_unit = [&](const int i, int* h, int* w, int* j) __attribute__((always_inline)) {
    constexpr int UPWIN = W::WHS / (2 * HP);  // units per (row, window)
    *h = i / (C::WB * UPWIN);
    const int r = i - *h * (C::WB * UPWIN);
    *w = r / UPWIN;
    const int q = r - *w * UPWIN;
    *j = (C::S == 1) ? 2 * q : 8 * (q >> 1) + 2 * (q & 1);
  };
  auto fetch = [&](const int item) __attribute__((always_inline)) {
    const int grp = item / a.chunks, n_start = (item - grp * a.chunks) * C::TT;
    const int b0 = grp * C::WB;
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int i = (tid + k * NTH) * 2;
      const int m = i / (C::WB * C::TT), r = i - m * (C::WB * C::TT);
      const int w = r / C::TT, n = r - w * C::TT;
      unsigned v = 0u;
      if (i < N_LO * 2 && b0 + w < a.B && n_start + n < a.Ln)
        v = *reinterpret_cast<const unsigned*>(lo_p + (long)(b0 + w) * a.lo.ws + (long)m * a.lo.ls + HALO + n_start + n);
      pre_lo[k] = v;
    }
    const int s0 = C::S * n_start + a.off - PAR;  // hi sample of staged position 0 (even)
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      int h, w, j;
      hi_unit(tid + k * NTH, &h, &w, &j);
      const bool in = tid + k * NTH < N_HI && b0 + w < a.B;
      const bf16_t* row = (h < C::HI1) ? hi1_p + (long)(b0 + w) * a.hi1.ws + (long)h * a.hi1.ls + HALO
                                       : hi2_p + (long)(b0 + w) * a.hi2.ws + (long)(h - C::HI1) * a.hi2.ls + HALO;
      const int lim = (h < C::HI1) ? a.lim_hi1 : a.lim_hi2;
#pragma unroll
      for (int e = 0; e < HP; ++e) {
        const int idx = s0 + j + 4 * e;
        pre_hi[k][e] = (in && idx >= -HALO && idx < lim) ? *reinterpret_cast<const unsigned*>(row + idx) : 0u;
      }
    }
  };
  if ((int)blockIdx.x < items) fetch(blockIdx.x);
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    __syncthreads();  // the zero fill (first trip) / the previous item's reads are done
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int i = (tid + k * NTH) * 2;
      const int m = i / (C::WB * C::TT), r = i - m * (C::WB * C::TT);
      const int w = r / C::TT, n = r - w * C::TT;
      if (i < N_LO * 2) *reinterpret_cast<unsigned*>(lo_s + m * W::RS_LO + w * W::TTP + n) = pre_lo[k];
    }
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      int h, w, j;
      hi_unit(tid + k * NTH, &h, &w, &j);
      if (tid + k * NTH < N_HI) {
        if constexpr (C::S == 1) {
          *reinterpret_cast<unsigned*>(hi_s + h * W::RS_HI + w * W::WHP + j) = pre_hi[k][0];
        } else {  // samples (j, j + 1) and (j + 4, j + 5): plane j & 3 gets (j, j + 4), plane (j & 3) + 1 gets (j + 1, j + 5)
          bf16_t* row = hi_s + (j & 3) * W::PLANE + h * W::RS_HI + w * W::WHP + (j >> 2);
          const unsigned p0 = pre_hi[k][0], p1 = pre_hi[k][HP - 1];
          *reinterpret_cast<unsigned*>(row) = (p0 & 0xffffu) | (p1 << 16);
          *reinterpret_cast<unsigned*>(row + W::PLANE) = (p0 >> 16) | (p1 & 0xffff0000u);
        }
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < items) fetch(item + gridDim.x);
    for (int q = ksub; q < C::WB * W::KS; q += W::KSPLIT) {
      const int w = q / W::KS, ks = q - w * W::KS;
      // A: one fragment per distinct m-tile of this wave's units (consecutive units share the m-tile while hg runs)
      constexpr int NA = (W::UPW + W::HG - 1) / W::HG, NBG = (W::UPW < W::HG) ? W::UPW : W::HG;
      bf16x8_native av[NA];
#pragma unroll
      for (int ia = 0; ia < NA; ++ia) {
        const int mt = (u0 + ia * W::HG) / W::HG;
        av[ia] = __builtin_bit_cast(bf16x8_native, *reinterpret_cast<const uint4*>(lo_s + (mt * 16 + l16) * W::RS_LO + w * W::TTP + ks * 32 + 8 * g));
      }
#pragma unroll
      for (int ib = 0; ib < NBG; ++ib) {
        const int hg = (u0 + ib) % W::HG;
        const bf16_t* hrow = hi_s + (hg * 16 + l16) * W::RS_HI + w * W::WHP + ks * 32 + 8 * g;
        uint4 bop[7];
        if constexpr (C::S == 1) {
          const uint4 q0 = *reinterpret_cast<const uint4*>(hrow), q1 = *reinterpret_cast<const uint4*>(hrow + 8);
          const unsigned d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const int sh = k + PAR;  // samples
            unsigned r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              r[j] = (sh & 1) ? __builtin_amdgcn_alignbit(d[(sh >> 1) + j + 1], d[(sh >> 1) + j], 16) : d[(sh >> 1) + j];
            bop[k] = make_uint4(r[0], r[1], r[2], r[3]);
          }
        } else {
          uint4 pq[4];
          unsigned px[4];
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) {
            const bf16_t* pr = hrow + pl * W::PLANE;
            pq[pl] = *reinterpret_cast<const uint4*>(pr);
            px[pl] = *reinterpret_cast<const unsigned*>(pr + 8);
          }
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const int kk = k + PAR, pl = kk & 3, sh = kk >> 2;  // sh in {0, 1}
            const uint4 q4 = pq[pl];
            bop[k] = sh ? make_uint4(__builtin_amdgcn_alignbit(q4.y, q4.x, 16), __builtin_amdgcn_alignbit(q4.z, q4.y, 16),
                                     __builtin_amdgcn_alignbit(q4.w, q4.z, 16), __builtin_amdgcn_alignbit(px[pl], q4.w, 16))
                        : q4;
          }
        }
        // the units of this wave with channel group hg: local index ib + HG * ia
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) {
          const int u = ib + W::HG * ia;
          if (u < W::UPW) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
              acc[u][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia], __builtin_bit_cast(bf16x8_native, bop[k]), acc[u][k], 0, 0, 0);
          }
        }
      }
    }
  }
  float* fold = reinterpret_cast<float*>(wg_lds_raw);
  if constexpr (W::KSPLIT > 1) {  // fold the waves that share a unit into the first of them, one wave per round
    for (int r = 1; r < W::KSPLIT; ++r) {
      __syncthreads();
      if (ksub == r) {
#pragma unroll
        for (int k = 0; k < 7; ++k) *reinterpret_cast<f32x4*>(fold + ((u0 * 7 + k) * 64 + lane) * 4) = acc[0][k];
      }
      __syncthreads();
      if (ksub == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[0][k] += *reinterpret_cast<const f32x4*>(fold + ((u0 * 7 + k) * 64 + lane) * 4);
      }
    }
    if (ksub != 0) return;
  }
  float* out = a.partial + (long)blockIdx.x * C::OUT;
#pragma unroll
  for (int u = 0; u < W::UPW; ++u) {
    const int mt = (u0 + u) / W::HG, hg = (u0 + u) % W::HG;
    const int h = hg * 16 + l16;
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + 4 * g + r;
        if (m < C::LO && h < C::HI) out[((long)m * C::HI + h) * C::K + k] = acc[u][k][r];
      }
  }
}

template <class C>
int launch_wgrad_bf16(const WgradArgs& a, int grid, hipStream_t s) {
  static bool attr = false;
  constexpr size_t lds = (size_t)WgradB<C>::LDS_ELEMS * 2;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<C, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<C, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (a.off & 1) {
    hipLaunchKernelGGL((wgrad_bf16_kernel<C, 1>), dim3(grid), dim3(64 * WgradB<C>::NWAVE), lds, s, a);
  } else {
    hipLaunchKernelGGL((wgrad_bf16_kernel<C, 0>), dim3(grid), dim3(64 * WgradB<C>::NWAVE), lds, s, a);
  }
  return 0;
}

template <class C, class T = float>
int launch_wgrad(const WgradArgs& a, int grid, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_kernel<C, T>), dim3(grid), dim3(64 * C::NWAVE), 0, s, a);
  return 0;
}

}  // namespace vp
