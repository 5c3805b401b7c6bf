// How fast can every CU stream the SAME weights out of L2 (the bottom PhaseNet layers pull 606 KB per window and run at
// 16 B/clk/CU)?  256 workgroups x 16 waves read one buffer with dword / dwordx2 / dwordx4 loads per lane and several
// loads in flight per wave.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/micro_stream.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W, int DEPTH, int NWV>
__global__ __launch_bounds__(NWV * 64) void k(const float* __restrict__ src, float* out, int n_floats, int reps) {
  const int tid = threadIdx.x;
  float acc = 0.f;
  const int per_iter = NWV * 64 * W * DEPTH;
  for (int r = 0; r < reps; ++r) {
    for (int base = 0; base + per_iter <= n_floats; base += per_iter) {
      if constexpr (W == 1) {
        float v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = src[base + d * NWV * 64 + tid];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
      } else if constexpr (W == 2) {
        f2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const f2*>(src + base + (d * NWV * 64 + tid) * 2);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
      } else {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const f4*>(src + base + (d * NWV * 64 + tid) * 4);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
      }
    }
  }
  out[blockIdx.x * NWV * 64 + tid] = acc;
}

template <int W, int DEPTH, int NWV>
void run(const float* src, float* out, int n_floats) {
  const int reps = 20;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, DEPTH, NWV>), dim3(256), dim3(NWV * 64), 0, 0, src, out, n_floats, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const int per_iter = NWV * 64 * W * DEPTH;
  const double bytes = (double)(n_floats / per_iter) * per_iter * 4.0 * reps;
  const double us = ms * 1e3;
  printf("dwordx%d  depth %2d  waves %2d: %8.1f us per %d passes  %6.1f B/clk/CU at 2.1 GHz  (%.2f TB/s chip-wide)\n", W, DEPTH, NWV,
         us, reps, bytes / (us * 1e-6) / 2.1e9, bytes * 256 / (us * 1e-6) * 1e-12);
}

int main() {
  const int n = 606 * 1024 / 4;
  float *src, *out;
  (void)hipMalloc(&src, n * 4);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMemset(src, 0, n * 4);
  run<1, 1, 16>(src, out, n);
  run<1, 4, 16>(src, out, n);
  run<1, 8, 16>(src, out, n);
  run<1, 16, 16>(src, out, n);
  run<2, 4, 16>(src, out, n);
  run<2, 8, 16>(src, out, n);
  run<4, 2, 16>(src, out, n);
  run<4, 4, 16>(src, out, n);
  run<4, 8, 16>(src, out, n);
  run<1, 8, 8>(src, out, n);
  run<4, 4, 8>(src, out, n);
  run<4, 4, 4>(src, out, n);
  return 0;
}
