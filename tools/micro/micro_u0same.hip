// Where do the cycles of the PhaseNet up0.same layer (128 -> 64 channels, k7, 47 columns: 4 items of 3 n-tiles, one wave
// per SIMD, 229 KB of weights per pass) go?  The deep-prefetch K loop of conv_lds with its operands ablated:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I volpick_amd/csrc -I include tools/micro/micro_u0same.hip -o /tmp/mu && /tmp/mu
#include <hip/hip_runtime.h>

#include <cstdio>

#include "conv_lds.h"

namespace vp {
void set_error(const char*, ...) {}
}  // namespace vp
using namespace vp;

using L = LdsLayer<64, 64, 64, 1, 7, 1, -3, 0, 3, 1>;
constexpr int S = 80, B = 4, COLS = 47;

// VAR 0: A from L2 + B from LDS (as shipped)   1: A constant   2: B constant   3: both constant
// VAR 4: as 0, but the B fragments of the NEXT tap are read before the MFMAs of the current one (explicit one-tap lead)
template <int VAR>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ afrag, float* out, int reps) {
  extern __shared__ float4 raw[];
  float* lds = (float*)raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40000; i += 1024) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  const int g = lane >> 4, n = lane & 15;
  float keep = 0.f;
  for (int r = 0; r < reps; ++r) {
    if (wave < 4) {
      const int mt = wave;
      f32x4 acc[3] = {};
      const float* ap = afrag + (long)mt * L::CB * 7 * 64 + lane;
      const float* bp1 = lds + g * S + B + n - 3;
      const float* bp2 = lds + 64 * S + g * S + B + n - 3;
      float a[4][7];
      auto load_a = [&](float (&av)[7], int cb) {
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) av[tap] = (VAR == 1 || VAR == 3) ? (float)(lane + tap + cb) : ap[(cb * 7 + tap) * 64];
      };
      auto bval = [&](const float* bp, int j, int tap, int cb) { return (VAR == 2 || VAR == 3) ? (float)(lane - j + tap + cb) : bp[j * 16 + tap]; };
      auto mac = [&](const float (&av)[7], int cb) {
        const float* bp = (cb < 16) ? bp1 + cb * 4 * S : bp2 + (cb - 16) * 4 * S;
        if constexpr (VAR == 4) {
          float b0[3], b1[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) b0[j] = bp[j * 16];
#pragma unroll
          for (int tap = 0; tap < 7; ++tap) {
            if (tap + 1 < 7) {
#pragma unroll
              for (int j = 0; j < 3; ++j) ((tap & 1) ? b0 : b1)[j] = bp[j * 16 + tap + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], ((tap & 1) ? b1 : b0)[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int tap = 0; tap < 7; ++tap) {
            float bv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bv[j] = bval(bp, j, tap, cb);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], bv[j], acc[j], 0, 0, 0);
          }
        }
      };
      load_a(a[0], 0);
      load_a(a[1], 1);
      load_a(a[2], 2);
#pragma unroll 1
      for (int cb = 0; cb < L::CB; cb += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cb + u + 3 < L::CB) load_a(a[(u + 3) & 3], cb + u + 3);
          mac(a[u], cb + u);
        }
      }
      keep += acc[0][0] + acc[1][1] + acc[2][2];
    }
    __syncthreads();
  }
  if (lane == 0) out[blockIdx.x * 16 + wave] = keep;
}

template <int VAR>
void run(const char* name, const float* af, float* out) {
  const int reps = 40;
  const size_t lds = 40448 * 4;
  auto fn = k<VAR>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(1024), lds, 0, af, out, reps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-44s %7.2f us per pass (MFMA floor 21.5 k cycles = 9.0 us at 2.4 GHz)\n", name, ms * 1e3f / reps);
}

int main() {
  float *af, *out;
  (void)hipMalloc(&af, 4 << 20);
  (void)hipMalloc(&out, 1 << 20);
  (void)hipMemset(af, 0, 4 << 20);
  run<0>("A from L2, B from LDS (shipped loop)", af, out);
  run<1>("A constant, B from LDS", af, out);
  run<2>("A from L2, B constant", af, out);
  run<3>("A constant, B constant", af, out);
  run<4>("A from L2, B one tap ahead (pinned)", af, out);
  return 0;
}
