// Micro-benchmark: one conv_lds layer (PhaseNet up2.same geometry: 32 -> 16 channels, k7, 751 columns)
// repeated inside a persistent workgroup, with ablations (no A loads / no B reads) to attribute the
// cycles of the fused core kernel.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I volpick_amd/csrc -I include tools/micro/micro_layer.hip -o /tmp/micro && /tmp/micro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "conv_lds.h"

namespace vp {
void set_error(const char*, ...) {}
}  // namespace vp
using namespace vp;

constexpr int S = 784, IB_ = 4, T1 = 751;

struct RangeStoreT {
  float* img;
  int L;
  __device__ __forceinline__ void operator()(int co, int t, float v) const {
    if ((unsigned)t < (unsigned)L) img[co * S + IB_ + t] = v;
  }
  __device__ __forceinline__ bool all_valid(int t0, int t1) const { return t0 >= 0 && t1 < L; }
  __device__ __forceinline__ void unchecked(int co, int t, float v) const { img[co * S + IB_ + t] = v; }
};

// VAR 0: conv_lds PIPE=true, 4: conv_lds PIPE=false, 1: no A loads, 2: no B reads, 3: neither
template <int VAR, int NB, int NWV>
__global__ __launch_bounds__(NWV * 64) void k(const float* afrag, const float* bias, float* out, int reps) {
  extern __shared__ float4 raw[];
  float* lds = (float*)raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40000; i += NWV * 64) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  using L = LdsLayer<16, 16, 16, 1, 7, 1, -3, 0, NB, 1>;
  float acc_keep = 0.f;
  for (int r = 0; r < reps; ++r) {
    if constexpr (VAR == 0 || VAR == 4) {
      RangeStoreT st{lds + 26000, T1};
      conv_lds<L, S, IB_, S, IB_, VAR == 0>(lds, lds + 16 * S, afrag, bias, T1, st, wave, NWV, lane);
    } else {
      const int NT = (T1 + 15) >> 4, NBLK = (NT + NB - 1) / NB, items = L::MT * NBLK;
      const int g = lane >> 4, n = lane & 15;
      for (int item = wave; item < items; item += NWV) {
        const int mt = item % L::MT, nblk = item / L::MT, colb = nblk * NB * 16;
        f32x4 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = f32x4{0, 0, 0, 0};
        const float* ap = afrag + (long)mt * L::CB * 7 * 64 + lane;
        const float* bp1 = lds + g * S + IB_ + (colb + n) - 3;
#pragma unroll 1
        for (int cb = 0; cb < L::CB; ++cb) {
#pragma unroll
          for (int tap = 0; tap < 7; ++tap) {
            float av = (VAR == 1 || VAR == 3) ? (float)(lane + tap + cb) : ap[(cb * 7 + tap) * 64];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              float bv = (VAR == 2 || VAR == 3) ? (float)(lane - j + cb) : bp1[cb * 4 * S + j * 16 + tap];
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc_keep += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
      }
    }
    __syncthreads();
  }
  if (acc_keep == 123.456f) out[tid] = acc_keep;
  if (tid == 0) out[blockIdx.x] = lds[26000 + 5];
}

template <int VAR, int NB, int NWV>
float run(const float* af, const float* bs, float* out, int reps) {
  const size_t lds = 40448 * 4;
  hipFuncSetAttribute((const void*)k<VAR, NB, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR, NB, NWV>), dim3(256), dim3(NWV * 64), lds, 0, af, bs, out, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR, NB, NWV>), dim3(256), dim3(NWV * 64), lds, 0, af, bs, out, reps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
  return ms * 1e3f / reps;  // us per layer pass
}

int main() {
  float *af, *bs, *out;
  hipMalloc(&af, 1 << 20);
  hipMalloc(&bs, 4096);
  hipMalloc(&out, 1 << 20);
  hipMemset(af, 0, 1 << 20);
  hipMemset(bs, 0, 4096);
  const int reps = 50;
  // ideal: 48 n-tiles x 56 K-steps = 2688 MFMAs x 32 cyc / 4 SIMDs = 21.5k cycles = 9.0 us @ 2.39 GHz
  printf("up2.same layer, us per pass (MFMA-bound ideal ~9.0 us)\n");
  printf("NB=3 16 waves: pipe %.2f  nopipe %.2f  noA %.2f  noB %.2f  pureMFMA %.2f\n", run<0, 3, 16>(af, bs, out, reps),
         run<4, 3, 16>(af, bs, out, reps), run<1, 3, 16>(af, bs, out, reps), run<2, 3, 16>(af, bs, out, reps),
         run<3, 3, 16>(af, bs, out, reps));
  printf("NB=6  8 waves: pipe %.2f  nopipe %.2f  noA %.2f  noB %.2f  pureMFMA %.2f\n", run<0, 6, 8>(af, bs, out, reps),
         run<4, 6, 8>(af, bs, out, reps), run<1, 6, 8>(af, bs, out, reps), run<2, 6, 8>(af, bs, out, reps),
         run<3, 6, 8>(af, bs, out, reps));
  printf("NB=4 12 waves: pipe %.2f  nopipe %.2f  noA %.2f  noB %.2f  pureMFMA %.2f\n", run<0, 4, 12>(af, bs, out, reps),
         run<4, 4, 12>(af, bs, out, reps), run<1, 4, 12>(af, bs, out, reps), run<2, 4, 12>(af, bs, out, reps),
         run<3, 4, 12>(af, bs, out, reps));
  printf("NB=2 16 waves: pipe %.2f  nopipe %.2f  pureMFMA %.2f\n", run<0, 2, 16>(af, bs, out, reps),
         run<4, 2, 16>(af, bs, out, reps), run<3, 2, 16>(af, bs, out, reps));
  return 0;
}
