// How fast does ONE wave issue v_mfma_f32_16x16x32_bf16 when consecutive instructions rotate over NACC accumulators?
// (and two waves per SIMD).  hipcc --offload-arch=gfx950 -O3 -o exp/mfma_chain_probe tools/micro/mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU_PER>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 au = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u), bu = au;
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  float f = threadIdx.x * 1e-3f, f2 = 1.0001f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU_PER; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f) : "v"(f2));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int VALU_PER>
void run(int waves) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  probe<NACC, VALU_PER><<<256, waves * 64>>>(out, cyc, 10);
  probe<NACC, VALU_PER><<<256, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, 256 * waves * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256 * waves; ++i) m += (double)h[i];
  m /= 256.0 * waves;
  printf("accumulators %d  valu/mfma %d  waves/WG %d (per SIMD %.1f): %.1f cycles per MFMA of one wave, %.1f per SIMD\n", NACC, VALU_PER, waves,
         waves / 4.0, m / (iters * 24.0), m / (iters * 24.0) / (waves > 4 ? waves / 4.0 : 1.0));
  hipFree(out), hipFree(cyc);
}

int main() {
  for (int waves : {4, 8}) {
    run<1, 0>(waves), run<2, 0>(waves), run<3, 0>(waves), run<4, 0>(waves), run<6, 0>(waves);
    run<1, 2>(waves), run<2, 2>(waves), run<3, 2>(waves), run<2, 3>(waves), run<2, 4>(waves);
  }
  return 0;
}
