// Issue-rate probes on gfx950: v_pk_fma_f32 (VGPR / SGPR operand / op_sel broadcast), v_fma_f32, v_mfma_f32_16x16x4_f32,
// and MFMA waves + packed-FMA waves sharing a CU (do the two pipes add up?).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/micro_peak.hip -o /tmp/micro_peak && /tmp/micro_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int INNER = 64;  // unrolled instructions per accumulator group

// MODE 0: pk_fma vgpr; 1: pk_fma sgpr weight; 2: pk_fma sgpr + op_sel broadcast of src0; 3: v_fma_f32; 4: mfma;
// 5: waves 0..NW/2-1 mfma, the rest pk_fma (mode 2)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int reps, float sw0, float sw1) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (MODE == 4) || (MODE == 5 && wave < (int)(blockDim.x >> 7));  // first half of the waves: one or more per SIMD
  if (MODE == 5 && !do_mfma) reps *= 8;  // a packed FMA issues in 4 cycles, an MFMA occupies its pipe for 32
  float keep = 0.f;
  if (do_mfma) {
    f4 acc[4] = {};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < INNER / 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      }
    }
    keep = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if (MODE == 3) {
    float acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = j;
    float x = threadIdx.x * 0.001f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < INNER / 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "s"(sw0));
    }
    for (int j = 0; j < 16; ++j) keep += acc[j];
  } else {
    f2 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f2{(float)j, 1.f};
    f2 x = {threadIdx.x * 0.001f, 0.5f};
    f2 wv = {sw0, sw1};
    f2 ws = {sw0, sw1};
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < INNER / 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "v"(wv));
          if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "s"(ws));
          if (MODE == 2 || MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(x), "s"(ws));
        }
    }
    for (int j = 0; j < 16; ++j) keep += acc[j].x + acc[j].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

template <int MODE>
void run(const char* name, int nth, int wgs_per_cu) {
  float* dout;
  const int grid = 256 * wgs_per_cu;
  hipMalloc(&dout, (size_t)grid * nth * 4);
  const int reps = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(nth), 0, 0, dout, reps, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double waves = (double)grid * nth / 64;
  double flop;
  if (MODE == 4) flop = waves * reps * INNER * 2048.0;
  else if (MODE == 3) flop = waves * reps * INNER * 128.0;
  else if (MODE == 5) flop = waves / 2 * reps * INNER * (2048.0 + 8 * 256.0);
  else flop = waves * reps * INNER * 256.0;
  printf("%-34s %4d thr x%d/CU: %8.3f ms  %7.1f TFLOP/s", name, nth, wgs_per_cu, ms, flop / ms * 1e-9);
  if (MODE == 5) printf("  (mfma part %.1f, pk part %.1f)", waves / 2 * reps * INNER * 2048.0 / ms * 1e-9, waves / 2 * reps * INNER * 8 * 256.0 / ms * 1e-9);
  printf("\n");
  hipFree(dout);
}

int main() {
  for (int nth : {256, 512, 1024}) {
    run<0>("pk_fma vgpr", nth, 1);
    run<1>("pk_fma sgpr", nth, 1);
    run<2>("pk_fma sgpr op_sel_hi bcast", nth, 1);
    run<3>("v_fma_f32 sgpr", nth, 1);
    run<4>("mfma 16x16x4 f32", nth, 1);
    run<5>("half mfma waves, half pk waves", nth, 1);
  }
  run<0>("pk_fma vgpr", 256, 2);
  run<5>("half mfma waves, half pk waves", 512, 2);
  return 0;
}
