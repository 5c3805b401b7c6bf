// Micro-benchmark: direct convolution on the VALU (v_pk_fma_f32, weights in SGPRs, per-lane sliding window in
// registers) for the narrow level-0 PhaseNet layers (8 output channels), against the fp32 vector peak.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/micro_valu.hip -o /tmp/micro_valu && /tmp/micro_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// out[co][t] = relu(bias[co] + sum_ci sum_k w[co][ci][k] * in[ci][t + k - 3]); lane owns R consecutive t.
// Image row: logical sample t at column IB + t, IB = 3, so that the window of lane l starts 16-byte aligned.
template <int CIN, int R, int S>
__device__ __forceinline__ void conv_valu8(const float* __restrict__ img, const f2* __restrict__ w2, int t0, f2 (&acc)[4][R]) {
  constexpr int NV = (R + 6 + 3) / 4;
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    float win[NV * 4];
    const f4* p = reinterpret_cast<const f4*>(img + ci * S + t0);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f4 q = p[v];
      win[4 * v] = q.x, win[4 * v + 1] = q.y, win[4 * v + 2] = q.z, win[4 * v + 3] = q.w;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f2 w = w2[(ci * 7 + k) * 4 + c];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f2 x = {win[r + k], win[r + k]};
          acc[c][r] = __builtin_elementwise_fma(x, w, acc[c][r]);
        }
      }
    }
  }
}

template <int CIN, int R, int NTH>
__global__ __launch_bounds__(NTH) void k(const f2* __restrict__ w2, const float* __restrict__ bias, float* out, int reps) {
  constexpr int TILE = NTH * R, S = TILE + 16;
  __shared__ float img[CIN * S];
  __shared__ float oimg[8 * S];
  const int tid = threadIdx.x;
  for (int i = tid; i < CIN * S; i += NTH) img[i] = 0.001f * (i % 97);
  __syncthreads();
  float keep = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    f2 acc[4][R];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[c][r] = f2{bias[2 * c], bias[2 * c + 1]};
    conv_valu8<CIN, R, S>(img, w2, R * tid, acc);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (R == 4) {
        f4 lo = {fmaxf(acc[c][0].x, 0.f), fmaxf(acc[c][1].x, 0.f), fmaxf(acc[c][2].x, 0.f), fmaxf(acc[c][3].x, 0.f)};
        f4 hi = {fmaxf(acc[c][0].y, 0.f), fmaxf(acc[c][1].y, 0.f), fmaxf(acc[c][2].y, 0.f), fmaxf(acc[c][3].y, 0.f)};
        *reinterpret_cast<f4*>(oimg + (2 * c) * S + 4 + R * tid) = lo;
        *reinterpret_cast<f4*>(oimg + (2 * c + 1) * S + 4 + R * tid) = hi;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          oimg[(2 * c) * S + 4 + R * tid + r] = fmaxf(acc[c][r].x, 0.f);
          oimg[(2 * c + 1) * S + 4 + R * tid + r] = fmaxf(acc[c][r].y, 0.f);
        }
      }
    }
    __syncthreads();
    keep += oimg[(rep & 7) * S + 4 + tid];
    __syncthreads();
  }
  out[blockIdx.x * NTH + tid] = keep;
}

template <int CIN, int R, int NTH>
void run(const char* name, int wgs_per_cu) {
  std::vector<float> w(CIN * 7 * 8, 0.01f), b(8, 0.1f);
  float *dw, *db, *dout;
  hipMalloc(&dw, w.size() * 4);
  hipMalloc(&db, 32);
  const int grid = 256 * wgs_per_cu;
  hipMalloc(&dout, (size_t)grid * NTH * 4);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 32, hipMemcpyHostToDevice);
  const int reps = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CIN, R, NTH>), dim3(grid), dim3(NTH), 0, 0, (const f2*)dw, db, dout, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 8 * CIN * 7 * (double)(NTH * R) * reps * grid;
  printf("%-28s grid %5d: %.3f ms  %.1f TFLOP/s\n", name, grid, ms, flop / ms * 1e-9);
  hipFree(dw), hipFree(db), hipFree(dout);
}

int main() {
  run<8, 4, 256>("same 8->8 R=4 256thr x1", 1);
  run<8, 4, 256>("same 8->8 R=4 256thr x2", 2);
  run<8, 4, 256>("same 8->8 R=4 256thr x4", 4);
  run<8, 2, 256>("same 8->8 R=2 256thr x4", 4);
  run<8, 8, 256>("same 8->8 R=8 256thr x2", 2);
  run<16, 4, 256>("same 16->8 R=4 256thr x2", 2);
  run<3, 4, 256>("inc 3->8 R=4 256thr x4", 4);
  return 0;
}
