// Micro-benchmark of conv_lds<> for the heavy layers of the fused PhaseNet core kernel:
// sweeps NB (n-tiles per item), wave count and the hand-pipelined vs compiler-scheduled K loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I volpick_amd/csrc -I include tools/micro/micro_layers.hip -o /tmp/micro2 && /tmp/micro2
#include <hip/hip_runtime.h>

#include <cstdio>

#include "conv_lds.h"

namespace vp {
void set_error(const char*, ...) {}
}  // namespace vp
using namespace vp;

template <int S, int B>
struct RStore {
  float* img;
  int L;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)L) img[co * S + B + t] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < L; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + B + t] = v; }
};

template <class L, int SI, int SO, int COLS, int LOUT, bool PIPE, int NWV>
__global__ __launch_bounds__(NWV * 64) void k(const float* afrag, const float* bias, float* out, int reps) {
  extern __shared__ float4 raw[];
  float* lds = (float*)raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40000; i += NWV * 64) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  constexpr int OUT0 = 26000;
  for (int r = 0; r < reps; ++r) {
    RStore<SO, 4> st{lds + OUT0, LOUT};
    conv_lds<L, SI, 4, SI, 4, PIPE>(lds, lds + L::CIN1 * SI, afrag, bias, COLS, st, wave, NWV, lane);
    __syncthreads();
  }
  if (tid == 0) out[blockIdx.x] = lds[OUT0 + 5];
}

template <class L, int SI, int SO, int COLS, int LOUT, bool PIPE, int NWV>
float run(const float* af, const float* bs, float* out) {
  const int reps = 40;
  const size_t lds = 40448 * 4;
  auto fn = k<L, SI, SO, COLS, LOUT, PIPE, NWV>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(256), dim3(NWV * 64), lds, 0, af, bs, out, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(256), dim3(NWV * 64), lds, 0, af, bs, out, reps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
  return ms * 1e3f / reps;
}

#define ROW(NAME, CIN1, CIN2, COUT, P, TAPS, SN, IOFF, OOFF, NB, SI, SO, COLS, LOUT, NWV)                          \
  {                                                                                                                \
    using L = LdsLayer<CIN1, CIN2, COUT, P, TAPS, SN, IOFF, OOFF, NB, 1>;                                          \
    printf("%-9s NB=%d waves=%2d  pipe %6.2f  nopipe %6.2f us\n", NAME, NB, NWV,                                    \
           run<L, SI, SO, COLS, LOUT, true, NWV>(af, bs, out), run<L, SI, SO, COLS, LOUT, false, NWV>(af, bs, out)); \
  }

int main() {
  float *af, *bs, *out;
  hipMalloc(&af, 4 << 20);
  hipMalloc(&bs, 4096);
  hipMalloc(&out, 1 << 20);
  hipMemset(af, 0, 4 << 20);
  hipMemset(bs, 0, 4096);
  // ideal (MFMA issue at 4 SIMDs, cycles): u0same/u1same/u2same 21.5k, u0T 8.2k, d4same 7.2k; 1k cycles ~ 0.42-0.48 us
  printf("--- u2same (32->16, 751 cols), pure-MFMA floor ~10.3 us\n");
  ROW("u2same", 16, 16, 16, 1, 7, 1, -3, 0, 3, 784, 784, 751, 751, 16)
  ROW("u2same", 16, 16, 16, 1, 7, 1, -3, 0, 4, 784, 784, 751, 751, 12)
  ROW("u2same", 16, 16, 16, 1, 7, 1, -3, 0, 6, 784, 784, 751, 751, 8)
  ROW("u2same", 16, 16, 16, 1, 7, 1, -3, 0, 6, 784, 784, 751, 751, 16)
  printf("--- u1same (64->32, 188 cols)\n");
  ROW("u1same", 32, 32, 32, 1, 7, 1, -3, 0, 2, 240, 240, 188, 188, 16)
  ROW("u1same", 32, 32, 32, 1, 7, 1, -3, 0, 3, 240, 240, 188, 188, 16)
  ROW("u1same", 32, 32, 32, 1, 7, 1, -3, 0, 3, 240, 240, 188, 188, 8)
  ROW("u1same", 32, 32, 32, 1, 7, 1, -3, 0, 4, 240, 240, 188, 188, 8)
  ROW("u1same", 32, 32, 32, 1, 7, 1, -3, 0, 6, 240, 240, 188, 188, 4)
  printf("--- u0same (128->64, 47 cols)\n");
  ROW("u0same", 64, 64, 64, 1, 7, 1, -3, 0, 1, 80, 80, 47, 47, 16)
  ROW("u0same", 64, 64, 64, 1, 7, 1, -3, 0, 3, 80, 80, 47, 47, 16)
  ROW("u0same", 64, 64, 64, 1, 7, 1, -3, 0, 3, 80, 80, 47, 47, 4)
  printf("--- u0T (128->64 convT, 13 cols)\n");
  ROW("u0T", 128, 0, 64, 4, 2, 1, -1, -1, 1, 48, 80, 13, 47, 16)
  ROW("u0T", 128, 0, 64, 4, 2, 1, -1, -1, 1, 48, 80, 13, 47, 8)
  printf("--- d4same (64->128, 12 cols)\n");
  ROW("d4same", 64, 0, 128, 1, 7, 1, -3, 0, 1, 48, 48, 12, 12, 16)
  ROW("d4same", 64, 0, 128, 1, 7, 1, -3, 0, 1, 48, 48, 12, 12, 8)
  printf("--- d1same (8->16, 751 cols)\n");
  ROW("d1same", 8, 0, 16, 1, 7, 1, -3, 0, 3, 784, 784, 751, 751, 16)
  ROW("d1same", 8, 0, 16, 1, 7, 1, -3, 0, 6, 784, 784, 751, 751, 8)
  return 0;
}
