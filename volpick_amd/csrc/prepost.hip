// HBM-bound stages around the model forward pass:
//   window gather + annotate_batch_pre, NaN blinding + overlap stacking, trigger/peak scan.
#include "prepost.h"

#include <cmath>

namespace vp {

// ---------------------------------------------------------------------------------------
// A2 + A3: one workgroup per window.  Gathers window w of the (3, N) stream (or of a dense
// (B,3,T) batch), subtracts the per-channel mean, divides by the peak / std amplitude
// (per channel or over all three channels) + eps, applies the 6-sample half-cosine taper
// (EQT) and writes the haloed model input row.  The window (<= 72 KB) is read once into registers;
// wavefront shuffle + LDS reductions.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gather_normalize_kernel(const PreArgs a) {
  constexpr int NTH = 1024, NWV = NTH / 64;
  __shared__ float red[3][NWV];
  __shared__ float stat[3][2];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.T;
  long start = a.dense ? 0 : (long)(a.first_window + w) * a.step;
  if (!a.dense && start > a.N - T) start = a.N - T;  // tail window flush with the end
  const float* src = a.src + (a.dense ? (long)w * 3 * T : start);
  long cs = a.dense ? T : a.N;
  if (a.table) {  // multi-block call: the window's block and start come from the table
    const long* e = a.table + 3 * (a.first_window + w);
    src = a.src + e[0] + e[2];
    cs = e[1];
  }
  float* dst = a.dst + (long)w * a.wsd + HALO;

  if (!a.preprocess) {
    bool bad = false;
    for (int c = 0; c < 3; ++c)
      for (int t = tid; t < T; t += NTH) {
        const float x = src[c * cs + t];
        bad |= !isfinite(x);
        dst[(long)c * a.lsd + t] = x;
      }
    if (a.flags) {
      const int any = __syncthreads_or(bad);
      if (tid == 0) a.flags[w] = any ? 1.f : 0.f;
    }
    return;
  }
  // The window lives in registers: one read of the stream, all three steps (mean, amplitude, scale) from there.
  constexpr int MAXE = 6;  // samples per thread and channel: T <= 6144
  float v[3][MAXE];
  float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int t = tid + k * NTH;
      v[c][k] = t < T ? src[c * cs + t] : 0.f;
      s[c] += v[c][k];
    }
  // norm = peak in ONE reduction round: max_k |v_k - mean| = max(vmax - mean, mean - vmin) bit for bit (rounding is monotonic and
  // symmetric), so the maxima and minima travel with the sums.  A NaN / Inf sample makes the mean non-finite: the window is
  // flagged and its predictions become NaN whatever the amplitude says.
  const bool one_pass = a.norm == VP_NORM_PEAK;  // uniform
  __shared__ float red_hi[3][NWV], red_lo[3][NWV];
  if (one_pass) {
    float vhi[3] = {-INFINITY, -INFINITY, -INFINITY}, vlo[3] = {INFINITY, INFINITY, INFINITY};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < MAXE; ++k)
        if (tid + k * NTH < T) vhi[c] = fmaxf(vhi[c], v[c][k]), vlo[c] = fminf(vlo[c], v[c][k]);
    // (three reductions interleaved by hand: the same DPP tree as wave_max / wave_sum, bit for bit, a third of the time)
    float nlo[3] = {-vlo[0], -vlo[1], -vlo[2]};
    wave_max3(vhi[0], vhi[1], vhi[2]);
    wave_max3(nlo[0], nlo[1], nlo[2]);
    if (lane == 0)
      for (int c = 0; c < 3; ++c) red_hi[c][wave] = vhi[c], red_lo[c][wave] = -nlo[c];
  }
  wave_sum3(s[0], s[1], s[2]);
  if (lane == 0)
    for (int c = 0; c < 3; ++c) red[c][wave] = s[c];
  __syncthreads();
  if (tid < 3) {
    float acc = 0.f;
    for (int i = 0; i < NWV; ++i) acc += red[tid][i];
    const float mu = acc / (float)T;
    stat[tid][0] = mu;
    if (one_pass) {
      float h = red_hi[tid][0], l = red_lo[tid][0];
      for (int i = 1; i < NWV; ++i) h = fmaxf(h, red_hi[tid][i]), l = fminf(l, red_lo[tid][i]);
      stat[tid][1] = fmaxf(h - mu, mu - l);
    }
  }
  __syncthreads();
  const float mean[3] = {stat[0][0], stat[1][0], stat[2][0]};
  if (!one_pass) {  // norm = std: the sum of squares needs the mean first
    float m[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int t = tid + k * NTH;
        if (t < T) {
          const float d = v[c][k] - mean[c];
          m[c] += d * d;
        }
      }
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
      const float r = wave_sum(m[c]);
      if (lane == 0) red[c][wave] = r;
    }
    __syncthreads();
    if (tid < 3) {
      const float* r = red[tid];
      float acc = r[0];
      for (int i = 1; i < NWV; ++i) acc = acc + r[i];
      stat[tid][1] = acc;
    }
    __syncthreads();
  }
  if (a.flags && tid == 0) {  // a non-finite sample shows in the statistics of its channel
    bool bad = false;
    for (int c = 0; c < 3; ++c) bad |= !isfinite(stat[c][0]) || !isfinite(stat[c][1]);
    a.flags[w] = bad ? 1.f : 0.f;
  }
  float amp[3];
  if (a.per_comp) {
    for (int c = 0; c < 3; ++c)
      amp[c] = (a.norm == VP_NORM_PEAK) ? stat[c][1] : sqrtf(stat[c][1] / (float)(T - 1));
  } else {
    const float g = (a.norm == VP_NORM_PEAK) ? fmaxf(stat[0][1], fmaxf(stat[1][1], stat[2][1]))
                                             : sqrtf((stat[0][1] + stat[1][1] + stat[2][1]) / (float)(3 * T - 1));
    amp[0] = amp[1] = amp[2] = g;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const NormDiv den = norm_div_prepare(amp[c] + a.norm_eps);
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int t = tid + k * NTH;
      if (t < T) {
        float o = norm_div(v[c][k] - mean[c], den);
        if (a.taper > 0) {
          const int e = (t < a.taper) ? t : ((T - 1 - t < a.taper) ? T - 1 - t : -1);
          if (e >= 0) {  // 0.5 * (1 + cos(linspace(pi, 2 pi, taper)[e]))
            const float ang = 3.14159265358979323846f * (1.f + (float)e / (float)(a.taper - 1));
            o *= 0.5f * (1.f + cosf(ang));
          }
        }
        dst[(long)c * a.lsd + t] = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void poison_kernel(float* y, const float* flags, long n) {
  if (flags[blockIdx.x] == 0.f) return;
  float* p = y + (long)blockIdx.x * n;
  const float q = __builtin_nanf("");
  for (long i = threadIdx.x; i < n; i += 256) p[i] = q;
}

int launch_poison(float* y, const float* flags, int n_windows, long floats_per_window, hipStream_t stream) {
  hipLaunchKernelGGL(poison_kernel, dim3(n_windows), dim3(256), 0, stream, y, flags, floats_per_window);
  return 0;
}

int launch_gather_normalize(const PreArgs& a, int n_windows, hipStream_t stream) {
  if (a.T > 6 * 1024) {  // the kernel keeps a window in registers: 6 samples per thread and channel
    set_error("gather_normalize: in_samples %d exceeds 6144", a.T);
    return VP_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(gather_normalize_kernel, dim3(n_windows), dim3(1024), 0, stream, a);
  return 0;
}

// ---------------------------------------------------------------------------------------
// A6 + A7: blinding + overlap stacking as a gather.  Output sample t of channel c averages
// (or maxes) pred[i][c][t - s_i] over every window i whose un-blinded range
// [s_i + blind_l, s_i + T - blind_r) contains t; none -> NaN.  The reference scatters into
// a (L, n_out, coverage) NaN buffer and takes nanmean / nanmax; the gather form reads each
// prediction once, writes each output once, and needs no coverage-deep buffer.
// ---------------------------------------------------------------------------------------
// I = int when the stream is shorter than 2^31 - T samples (launch_stack): the two divisions per sample are then
// 32-bit (a 64-bit division is ~60 instructions on this target; they were a third of the kernel's time)
template <class I>
__global__ __launch_bounds__(256) void stack_kernel(const StackArgs a) {
  const long tl = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (tl >= a.N) return;
  const I t = (I)tl, T = (I)a.T, step = (I)a.step, blind_l = (I)a.blind_l, blind_r = (I)a.blind_r;
  const I last_start = (I)(a.N - a.T);
  // regular windows i*step, i in [lo, hi]
  I hi = (t - blind_l >= 0) ? (t - blind_l) / step : -1;
  const I lo_num = t - T + blind_r;  // need i*step > lo_num
  const I lo = (lo_num < 0) ? 0 : lo_num / step + 1;
  if (hi > (I)a.n_regular - 1) hi = (I)a.n_regular - 1;
  float acc = (a.mode == VP_STACK_AVG) ? 0.f : -INFINITY;
  int cnt = 0;
  for (I i = lo; i <= hi; ++i) {
    const float v = a.pred[((long)i * a.n_out + c) * a.T + (long)(t - i * step)];
    if (v != v) continue;  // nanmean / nanmax skip NaN predictions
    acc = (a.mode == VP_STACK_AVG) ? acc + v : fmaxf(acc, v);
    ++cnt;
  }
  if (a.has_tail) {
    const I j = t - last_start;
    if (j >= blind_l && j < T - blind_r) {
      const float v = a.pred[((long)a.n_regular * a.n_out + c) * a.T + (long)j];
      if (v == v) {
        acc = (a.mode == VP_STACK_AVG) ? acc + v : fmaxf(acc, v);
        ++cnt;
      }
    }
  }
  float r = NAN;
  if (cnt > 0) r = (a.mode == VP_STACK_AVG) ? acc / (float)cnt : acc;
  a.out[(long)c * a.N + tl] = r;
}

// One thread per (sample of any block, channel): the block is found by bisection over the cumulative lengths.
__global__ __launch_bounds__(256) void stack_multi_kernel(const StackMultiArgs a) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (g >= a.total) return;
  int lo_b = 0, hi_b = a.n_blocks - 1;
  while (lo_b < hi_b) {
    const int mid = (lo_b + hi_b + 1) >> 1;
    if (a.blocks[mid].cum <= g) {
      lo_b = mid;
    } else {
      hi_b = mid - 1;
    }
  }
  const StackBlock b = a.blocks[lo_b];
  const long t = g - b.cum;
  const int T = a.T;
  float acc = (a.mode == VP_STACK_AVG) ? 0.f : -INFINITY;
  int cnt = 0;
  if (b.n_regular > 0) {
    long hi = (t - a.blind_l >= 0) ? (t - a.blind_l) / a.step : -1;
    const long lo_num = t - T + a.blind_r;
    const long lo = (lo_num < 0) ? 0 : lo_num / a.step + 1;
    if (hi > b.n_regular - 1) hi = b.n_regular - 1;
    for (long i = lo; i <= hi; ++i) {
      const float v = a.pred[((b.w0 + i) * a.n_out + c) * T + (t - i * a.step)];
      if (v != v) continue;
      acc = (a.mode == VP_STACK_AVG) ? acc + v : fmaxf(acc, v);
      ++cnt;
    }
  }
  if (b.has_tail) {
    const long j = t - (b.N - T);
    if (j >= a.blind_l && j < T - a.blind_r) {
      const float v = a.pred[((b.w0 + b.n_regular) * a.n_out + c) * T + j];
      if (v == v) {
        acc = (a.mode == VP_STACK_AVG) ? acc + v : fmaxf(acc, v);
        ++cnt;
      }
    }
  }
  float r = NAN;
  if (cnt > 0) r = (a.mode == VP_STACK_AVG) ? acc / (float)cnt : acc;
  a.out[b.off + (long)c * b.N + t] = r;
}

int launch_stack_multi(const StackMultiArgs& a, hipStream_t stream) {
  if (a.total <= 0) return 0;
  dim3 grid((unsigned)((a.total + 255) / 256), a.n_out, 1);
  hipLaunchKernelGGL(stack_multi_kernel, grid, dim3(256), 0, stream, a);
  return 0;
}

int launch_stack(const StackArgs& a, hipStream_t stream) {
  dim3 grid((unsigned)((a.N + 255) / 256), a.n_out, 1);
  if (a.N + a.T < (1L << 31) && a.step > 0 && (long)a.n_regular * a.step < (1L << 31))
    hipLaunchKernelGGL(stack_kernel<int>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(stack_kernel<long>, grid, dim3(256), 0, stream, a);
  return 0;
}

// ---------------------------------------------------------------------------------------
// A8: trigger_onset + peak.  For thr_off <= thr_on every maximal run of samples > thr_off
// that holds a sample > thr_on yields one trigger: on = first sample > thr_on in the run,
// off = last sample of the run, peak = first argmax over [on, off].
// The owner of a run is the workgroup whose chunk holds its END; one wavefront walks the run
// backwards 64 coalesced samples at a time (ballot finds the run start and the earliest sample
// > thr_on), then takes a strided max/argmax over [on, off] with a wave reduction.
// Triggers are appended with one atomic each and sorted on the host.
// ---------------------------------------------------------------------------------------
// 1024 samples per workgroup, their run ends as 16-bit offsets: 1 KB of LDS and <= 32 registers per lane, so that the scan's
// waves find room on a CU that a 1024-thread forward workgroup of another device context fills (pn_window_kernel leaves 2 KB
// of LDS and 32 registers per SIMD lane): the post-processing of step k then runs UNDER the forward pass of step k + 1
// instead of holding its CUs for itself at the boundary between two launches
constexpr int SCAN_CHUNK = 1024;  // samples per workgroup

// One launch: every workgroup lists the run ENDS inside its 2048-sample chunk in LDS (phase 1,
// one thread per 8 samples), then its four wavefronts walk those runs backwards through memory
// (phase 2) -- a run may start in an earlier chunk, only its end decides who owns it.
// I = int when every row is shorter than 2^31 - 2048 samples (launch_pick): 32-bit positions keep the kernel within 32 registers
template <class I>
__device__ __forceinline__ void trigger_scan_body(const PickArgs& a) {
  __shared__ int n_ends;
  __shared__ unsigned short ends[SCAN_CHUNK / 2 + 2];  // a run end needs a sample above and a next sample below: at most every other one
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const I n = (I)a.n;
  const I c0 = (I)blockIdx.x * SCAN_CHUNK;
  if ((long)blockIdx.x * SCAN_CHUNK >= a.n) return;
  if (tid == 0) n_ends = 0;
  __syncthreads();
  {  // all samples of the chunk requested before the first is looked at: one memory round trip instead of eight
    constexpr int PER = SCAN_CHUNK / 256;
    float v[PER], nx[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const I t = c0 + tid + 256 * k;
      v[k] = t < n ? a.trace[t] : -INFINITY;
      nx[k] = t + 1 < n ? a.trace[t + 1] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (v[k] > a.thr_off && !(nx[k] > a.thr_off)) ends[atomicAdd(&n_ends, 1)] = (unsigned short)(tid + 256 * k);
  }
  __syncthreads();
  const int ne = n_ends;
  for (int r = wave; r < ne; r += 4) {
    const I off = c0 + (int)ends[r];
    I on = -1;
    // four 64-sample blocks per trip, their reads in flight together: a Detection run (thr_off = thr / 2) is thousands of
    // samples long, and with one dependent 64-sample read per trip the kernel took 22 us on an EQTransformer step (14 now;
    // requesting the next trip ahead of the test changed nothing more)
    bool done = false;
    for (I pos = off; pos >= 0 && !done; pos -= 256) {
      float vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const I idx = pos - 64 * u - lane;
        vv[u] = (idx >= 0) ? a.trace[idx] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (done) continue;
        const I p0 = pos - 64 * u;
        if (p0 < 0) {
          done = true;
          continue;
        }
        const I idx = p0 - lane;
        const float v = vv[u];
        const bool above = (idx >= 0) && (v > a.thr_off);
        const unsigned long long broken = __ballot(!above);
        const int k = broken ? __ffsll((long long)broken) - 1 : 64;  // lanes [0, k) lie inside the run
        const unsigned long long hit = __ballot(lane < k && v > a.thr_on);
        if (hit) on = p0 - (63 - __clzll((long long)hit));
        if (k < 64) done = true;
      }
    }
    if (on < 0) continue;
    float best = -INFINITY;
    I arg = off;
    for (I i0 = on + lane; i0 <= off; i0 += 256) {  // first argmax: a lane meets its samples in increasing order
      float vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = (i0 + 64 * u <= off) ? a.trace[i0 + 64 * u] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (vv[u] > best) {
          best = vv[u];
          arg = i0 + 64 * u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const I oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) {
        best = ob;
        arg = oa;
      }
    }
    if (lane == 0) {
      const int slot = atomicAdd(a.count, 1);
      if (slot < a.cap) {
        a.on[slot] = on;
        a.off[slot] = off;
        a.peak[slot] = arg;
        a.value[slot] = best;
      }
    }
  }
}

template <class I>
__global__ __launch_bounds__(256) void trigger_scan_kernel(const PickBatch batch) { trigger_scan_body<I>(batch.a[blockIdx.y]); }
template <class I>
__global__ __launch_bounds__(256) void trigger_scan_table_kernel(const PickArgs* rows) { trigger_scan_body<I>(rows[blockIdx.y]); }
constexpr long SCAN_INT_MAX = 2147483647L - 4096;  // rows up to here are scanned with 32-bit positions

int launch_pick_table(const PickArgs* rows, int n_rows, long n_max, hipStream_t stream) {
  if (n_rows <= 0 || n_max <= 0) return 0;
  const dim3 grid((unsigned)((n_max + SCAN_CHUNK - 1) / SCAN_CHUNK), n_rows);
  if (n_max <= SCAN_INT_MAX) {
    hipLaunchKernelGGL(trigger_scan_table_kernel<int>, grid, dim3(256), 0, stream, rows);
  } else {
    hipLaunchKernelGGL(trigger_scan_table_kernel<long>, grid, dim3(256), 0, stream, rows);
  }
  return 0;
}

// All rows of one classify call in one launch (blockIdx.y = row).
int launch_pick(const PickBatch& b, hipStream_t stream) {
  long n_max = 0;
  for (int i = 0; i < b.n; ++i) n_max = (b.a[i].n > n_max) ? b.a[i].n : n_max;
  if (b.n <= 0 || n_max <= 0) return 0;
  const dim3 grid((unsigned)((n_max + SCAN_CHUNK - 1) / SCAN_CHUNK), b.n);
  if (n_max <= SCAN_INT_MAX) {
    hipLaunchKernelGGL(trigger_scan_kernel<int>, grid, dim3(256), 0, stream, b);
  } else {
    hipLaunchKernelGGL(trigger_scan_kernel<long>, grid, dim3(256), 0, stream, b);
  }
  return 0;
}

// Result hand-off without the copy engine: one workgroup copies the counters and the found triggers
// from the device result block into host-mapped pinned memory, then re-arms the counters for the next
// scan.  (A hipMemcpyAsync here would park the stream behind an SDMA round trip before the next
// batch's kernels may start.)
__global__ __launch_bounds__(256) void publish_kernel(char* dev, char* host, int n_specs, int cap, long header,
                                                      long per_spec) {
  int* cnt = reinterpret_cast<int*>(dev);
  int* hcnt = reinterpret_cast<int*>(host);
  const int tid = threadIdx.x;
  const long L = cap > 0 ? cap : 1;
  // a result block of a spec is [on | off | peak : int64 x L each][value : float x L]: the first m entries of each of the four
  // arrays as 4-byte words, one flat index space (few registers: this workgroup, too, runs beside a forward workgroup)
  for (int i = 0; i < n_specs; ++i) {
    const int found = cnt[2 * i];
    const int m = found < cap ? found : cap;
    const unsigned* src = reinterpret_cast<const unsigned*>(dev + header + per_spec * i);
    unsigned* dst = reinterpret_cast<unsigned*>(host + header + per_spec * i);
    for (int w = tid; w < 7 * m; w += 256) {
      const int arr = w / (2 * m);                                       // 0, 1, 2: the int64 arrays (2 m words each); 3: the values
      const long o = arr < 3 ? 2 * L * arr + (w - 2 * m * arr) : 6 * L + (w - 6 * m);
      dst[o] = src[o];
    }
    if (tid == 0) {
      hcnt[2 * i] = found;
      hcnt[2 * i + 1] = 0;
    }
  }
  __syncthreads();
  if (tid < 2 * n_specs) cnt[tid] = 0;
}

// The same hand-off for many rows: one workgroup per row.
__global__ __launch_bounds__(256) void publish_table_kernel(char* dev, char* host, int cap, long header, long per_row) {
  const int i = blockIdx.x, tid = threadIdx.x;
  int* cnt = reinterpret_cast<int*>(dev);
  int* hcnt = reinterpret_cast<int*>(host);
  const int found = cnt[2 * i];
  const int m = found < cap ? found : cap;
  const int64_t* s_on = reinterpret_cast<const int64_t*>(dev + header + per_row * i);
  int64_t* d_on = reinterpret_cast<int64_t*>(host + header + per_row * i);
  const long L = cap > 0 ? cap : 1;
  for (int k = tid; k < m; k += 256) {
    d_on[k] = s_on[k];
    d_on[L + k] = s_on[L + k];
    d_on[2 * L + k] = s_on[2 * L + k];
    reinterpret_cast<float*>(d_on + 3 * L)[k] = reinterpret_cast<const float*>(s_on + 3 * L)[k];
  }
  __syncthreads();
  if (tid == 0) {
    hcnt[2 * i] = found;
    hcnt[2 * i + 1] = 0;
    cnt[2 * i] = 0;  // re-armed for the next call
    cnt[2 * i + 1] = 0;
  }
}

int launch_publish_table(char* dev, char* host, int n_rows, int cap, long header, long per_row, hipStream_t stream) {
  if (n_rows <= 0) return 0;
  hipLaunchKernelGGL(publish_table_kernel, dim3(n_rows), dim3(256), 0, stream, dev, host, cap, header, per_row);
  return 0;
}

int launch_publish(char* dev, char* host, int n_specs, int cap, long header, long per_spec, hipStream_t stream) {
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(256), 0, stream, dev, host, n_specs, cap, header, per_spec);
  return 0;
}

// Debug: one wavefront per row scans the two margins of the row; anything but +0.0 / -0.0 is counted
// (a NaN or a denormal left by a stray store would corrupt the padding of every later convolution).
__global__ __launch_bounds__(256) void halo_check_kernel(const float* base, long rows, int ls, int L, int* bad) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const unsigned* row = reinterpret_cast<const unsigned*>(base + r * ls);
  int n = 0;
  for (int i = lane; i < ls; i += 64) {
    const bool margin = i < HALO || i >= HALO + L;
    if (margin && (row[i] & 0x7fffffffu) != 0u) ++n;
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if (lane == 0 && n) atomicAdd(bad, n);
}

int launch_halo_check(const float* rows_base, long rows, int ls, int L, int* bad, hipStream_t stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(halo_check_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, rows_base, rows, ls, L, bad);
  return 0;
}

// Host mirror of ObsPy's trigger_onset for host-resident traces (general thresholds).
int pick_host(const float* x, int64_t n, float thr_on, float thr_off, int64_t* on, int64_t* off, int64_t* peak,
              float* value, int cap, int* n_found) {
  int found = 0;
  int64_t i = 0;
  // State machine equivalent to the deque pairing in obspy.signal.trigger.trigger_onset:
  // a trigger opens at the first sample > thr_on and closes at the last sample of the run
  // of consecutive samples > thr_off that contains (or follows) it.
  while (i < n) {
    if (!(x[i] > thr_on)) {
      ++i;
      continue;
    }
    const int64_t s0 = i;
    int64_t s1;
    if (x[i] > thr_off) {
      int64_t k = i;
      while (k + 1 < n && x[k + 1] > thr_off) ++k;
      s1 = k;
    } else {
      // thr_off > thr_on corner: the next run end of (x > thr_off) after s0, or the last one
      int64_t k = i + 1;
      while (k < n && !(x[k] > thr_off)) ++k;
      if (k >= n) break;
      while (k + 1 < n && x[k + 1] > thr_off) ++k;
      s1 = k;
    }
    float best = x[s0];
    int64_t arg = s0;
    for (int64_t k = s0 + 1; k <= s1; ++k) {
      if (x[k] > best) {  // NaN compares false, like np.argmax on the > comparisons used upstream
        best = x[k];
        arg = k;
      }
    }
    if (found < cap) {
      on[found] = s0;
      off[found] = s1;
      peak[found] = arg;
      value[found] = best;
    }
    ++found;
    i = s1 + 1;
  }
  *n_found = found;
  return VP_OK;
}

}  // namespace vp

namespace vp {

// ---------------------------------------------------------------------------------------
// Per-window trigger/peak extraction on pre-cut windows (the reference's own evaluation
// protocol, volpick/model/eval_taks0.py:46-56,96-142): for window b, channel `row`, samples
// [lo_b, hi_b): trigger_onset(prob, thr_on, thr_off) and per trigger max / first argmax.
// One wavefront per window; lane l scans the l-th 1/64 of the range for run ENDS, walks each
// run backwards (bounded by lo_b) and appends (peak index relative to lo_b, value) to the
// window's slot list.  Lists are unordered; the host sorts the few entries per window.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void window_pick_kernel(const WindowPickArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float* x = a.prob + ((long)b * a.n_rows + a.row) * a.T;
  int lo = a.lo ? a.lo[b] : 0, hi = a.hi ? a.hi[b] : a.T;
  lo = lo < 0 ? 0 : lo;
  hi = hi > a.T ? a.T : hi;
  const int len = hi - lo;
  if (len <= 0) return;  // counts are zeroed by the caller (hipMemsetAsync) before the launch
  const int seg = (len + 63) / 64;
  const int s0 = lo + lane * seg, s1 = (s0 + seg < hi) ? s0 + seg : hi;
  for (int t = s0; t < s1; ++t) {
    if (!(x[t] > a.thr_off)) continue;
    if (t + 1 < hi && x[t + 1] > a.thr_off) continue;  // not a run end
    int on = -1;
    for (int s = t; s >= lo && x[s] > a.thr_off; --s)
      if (x[s] > a.thr_on) on = s;
    if (on < 0) continue;
    float best = x[on];
    int arg = on;
    for (int k = on + 1; k <= t; ++k)
      if (x[k] > best) {
        best = x[k];
        arg = k;
      }
    const int slot = atomicAdd(&a.count[b], 1);
    if (slot < a.K) {
      a.peak[(long)b * a.K + slot] = arg - lo;
      a.value[(long)b * a.K + slot] = best;
    }
  }
}

int launch_window_pick(const WindowPickArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(window_pick_kernel, dim3((a.B + 3) / 4), dim3(256), 0, stream, a);
  return 0;
}

}  // namespace vp
