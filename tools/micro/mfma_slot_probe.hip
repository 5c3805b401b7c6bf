// One wave per SIMD (or two): a K-step "slot" of the ResCNN kernel as a bare pattern -- 12 v_mfma_f32_16x16x32_bf16 over two
// alternating accumulators, NDS ds_read_b128 of fragments two slots ahead, NV independent vector instructions.  Cycles per slot.
// hipcc --offload-arch=gfx950 -O3 -w -o exp/mfma_slot_probe tools/micro/mfma_slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NDS, int NV>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  __shared__ uint4 img[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) img[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  const int lane = threadIdx.x & 63;
  const uint4* p = img + (lane >> 4) * 64 + (lane & 15) + (threadIdx.x >> 6) * 256;
  uint4 b[3][6];
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < 6; ++j) b[k][j] = p[j * 16 + k];
  uint4 au = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  bf16x8 a = __builtin_bit_cast(bf16x8, au);
  float f[4] = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f};
  const float f2 = 1.0001f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NDS; ++j) b[(s + 2) % 3][j] = p[j * 16 + s + (it & 1) * 1024];
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v & 3]) : "v"(f2));
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b[s][t % 3]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b[s][3 + t % 3]), acc1, 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f[0] + f[1] + f[2] + f[3];
  for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NDS, int NV>
void run(int waves) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 1000;
  probe<NDS, NV><<<256, waves * 64>>>(out, cyc, 10);
  probe<NDS, NV><<<256, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, 256 * waves * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256 * waves; ++i) m += (double)h[i];
  m /= 256.0 * waves;
  printf("ds_read_b128 %d  valu %2d per slot of 12 MFMAs, waves/SIMD %.0f: %.1f cycles per slot of one wave (12 x 16 = 192)\n", NDS, NV, waves / 4.0,
         m / (iters * 3.0));
  hipFree(out), hipFree(cyc);
}

int main() {
  for (int waves : {4, 8}) {
    run<0, 0>(waves), run<6, 0>(waves), run<0, 12>(waves), run<0, 24>(waves), run<6, 12>(waves), run<6, 24>(waves), run<6, 36>(waves), run<6, 48>(waves);
  }
  return 0;
}
