// What v_mfma_f32_16x16x32_bf16 sustains in the loop shape of the bf16-piece kernels (conv_b3.h: b3c_mac_tiles): per
// (K-step, n-tile) pair six MFMAs chained through one accumulator -- (w piece, x piece) = (2,0) (1,1) (0,2) (1,0) (0,1) (0,0)
// -- the A pieces in registers, the three B pieces of a pair read from LDS (chunk planes, one ds_read_b128 each) two
// pairs ahead.  Variants of the same MFMA count:
//   0  B from REGISTERS (no LDS reads): the pipe's own ceiling for chains of dependent MFMAs
//   1  B from LDS, no epilogue                                           = the loop alone
//   2  B from LDS + the epilogue of the kernels per n-tile: ReLU, split into three pieces, three ds_write_b64
//   3  as 1 with the images [column][64 + 8 channels] (2-way conflicted reads)
//   4  B from LDS, the 6 MFMAs of TWO pairs interleaved (two accumulators)
//   5  v_mfma_f32_32x32x16_bf16: the same products as 32 GEMM rows x 32 columns (two m-tiles of the 16-row form = the (phase,
//      channel) rows of a 2-phase 16-channel or a 32-channel stage), the 32-row A pieces in registers (twice the registers per
//      K-step), a (K-step, 32-column n-tile) pair = 2 x 6 MFMAs over two K = 16 halves, its B pieces six ds_read_b128 (per
//      FLOP half the reads of the 16x16x32 form), no epilogue
//   6  as 5 + the epilogue per 32-column n-tile: a lane's 16 accumulator rows are four quads of consecutive channels ->
//      ReLU, split, three ds_write_b64 per quad
// Variants 5 / 6 count in the same unit: cycles per 16,384-FLOP MFMA EQUIVALENT and SIMD (a 32x32x16 instruction = 2).
// Every variant with 4 and 8 waves per workgroup (one / two per SIMD), one workgroup per CU, for ~0.1 s.
// Prints cycles per MFMA and SIMD (16 = the issue rate), the shader clock held, TFLOP/s (fp32-equivalent: 16384 per six).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/micro_b3_mfma.hip -o /tmp/micro_b3 && /tmp/micro_b3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int STEPS = 5, NB = 6, NC = 400, C = 32;      // stage 3 of the decoder: 5 taps x 32 channels, 6 n-tiles per block
constexpr int CHS = NC * 8, PS = (C / 8) * CHS;        // bf16 per chunk plane / piece
constexpr int CS_PAD = C + 8, PS_PAD = NC * CS_PAD;    // the padded layout of variant 3

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ afrag, float* out, int reps, unsigned long long* clk) {
  extern __shared__ uint4 lds4[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * PS_PAD / 8; i += blockDim.x) lds4[i] = make_uint4(0x3c003c00u + i, 0x3b803b80u, 0x3c803c80u, 0x3d003d00u);
  __syncthreads();
  const int g = lane >> 4, n = lane & 15;
  uint4 a[STEPS * 3];
#pragma unroll
  for (int i = 0; i < STEPS * 3; ++i) a[i] = afrag[i * 64 + lane];
  const int colb = (wave % 4) * 96;
  const unsigned short* p = (MODE == 3) ? lds + (colb + n) * CS_PAD + 8 * g : lds + g * CHS + (colb + n) * 8;
  unsigned short* wimg = lds + ((wave % 4) * 96 + n) * 8 + (g >> 1) * CHS + (g & 1) * 4;  // epilogue target (variant 2): own columns
  float keep = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    constexpr int PAIRS = STEPS * NB, AHEAD = 2, NBUF = 3;
    uint4 b[NBUF][3];
    auto load_b = [&](const int i) {
      const int s = i % STEPS, j = i / STEPS;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        if (MODE == 0)
          b[i % NBUF][pc] = make_uint4(0x3c003c00u + i, 0x3b803b80u + pc, 0x3c803c80u, 0x3d003d00u);
        else if (MODE == 3)
          b[i % NBUF][pc] = *reinterpret_cast<const uint4*>(p + pc * PS_PAD + (s + j * 16) * CS_PAD);
        else
          b[i % NBUF][pc] = *reinterpret_cast<const uint4*>(p + pc * PS + (s + j * 16) * 8);
      }
    };
    constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
    if constexpr (MODE == 4) {
      load_b(0);
      load_b(1);
#pragma unroll
      for (int j = 0; j < NB; j += 2) {
        f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          // pairs (s, j) and (s, j + 1): buffers 0 / 1 hold them, 2 is the one being filled -- a simple two-deep scheme
          uint4 b0[3], b1[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            b0[pc] = *reinterpret_cast<const uint4*>(p + pc * PS + (s + j * 16) * 8);
            b1[pc] = *reinterpret_cast<const uint4*>(p + pc * PS + (s + (j + 1) * 16) * 8);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[s * 3 + WP[t]]), __builtin_bit_cast(bf8, b0[XP[t]]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[s * 3 + WP[t]]), __builtin_bit_cast(bf8, b1[XP[t]]), acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        keep += acc0[0] + acc1[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) load_b(i);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          const int i = j * STEPS + s;
          if (i + AHEAD < PAIRS) load_b(i + AHEAD);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 6; ++t)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[s * 3 + WP[t]]), __builtin_bit_cast(bf8, b[i % NBUF][XP[t]]), acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 2) {
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[q], 0.f);
          const unsigned h0 = pack2(v[0], v[1]), h1 = pack2(v[2], v[3]);
          const float r0 = v[0] - __uint_as_float(h0 << 16), r1 = v[1] - __uint_as_float(h0 & 0xffff0000u);
          const float r2 = v[2] - __uint_as_float(h1 << 16), r3 = v[3] - __uint_as_float(h1 & 0xffff0000u);
          const unsigned m0 = pack2(r0, r1), m1 = pack2(r2, r3);
          const unsigned l0 = pack2(r0 - __uint_as_float(m0 << 16), r1 - __uint_as_float(m0 & 0xffff0000u));
          const unsigned l1 = pack2(r2 - __uint_as_float(m1 << 16), r3 - __uint_as_float(m1 & 0xffff0000u));
          unsigned short* q = wimg + j * 16 * 8 + 2 * CHS;  // chunk planes 2 / 3: not the ones read
          *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(q + PS) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(q + 2 * PS) = make_uint2(l0, l1);
        } else {
          keep += acc[0] + acc[3];
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (keep == 1.2345e-30f) out[0] = keep;
  if (tid == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}


// ---- the 32x32x16 form --------------------------------------------------------------------------------------------------
// B fragment of v_mfma_f32_32x32x16_bf16: lane l supplies column l % 32, k = 8 (l / 32) .. + 7 -> with the chunk planes
// [piece][8-channel chunk][column][8 channels] one ds_read_b128 per piece and K = 16 half: chunk 2 h + l / 32, column colb + l % 32.
constexpr int NB32 = 3;  // 32-column n-tiles per block: the same 96 columns per wave
template <int MODE>
__global__ __launch_bounds__(512) void k32(const uint4* __restrict__ afrag, float* out, int reps, unsigned long long* clk) {
  extern __shared__ uint4 lds4[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * PS_PAD / 8; i += blockDim.x) lds4[i] = make_uint4(0x3c003c00u + i, 0x3b803b80u, 0x3c803c80u, 0x3d003d00u);
  __syncthreads();
  const int kb = lane >> 5, n = lane & 31;
  uint4 a[STEPS * 2 * 3];  // [K-step][K = 16 half][piece]: 32 rows x 16 k per register quad
#pragma unroll
  for (int i = 0; i < STEPS * 2 * 3; ++i) a[i] = afrag[(i % (STEPS * 3)) * 64 + lane];
  const int colb = (wave % 4) * 96;
  const unsigned short* p = lds + kb * CHS + (colb + n) * 8;
  unsigned short* wimg = lds + 2 * CHS + (colb + n) * 8;  // epilogue target: chunk planes 2 / 3 (not the ones read)
  float keep = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    constexpr int PAIRS = STEPS * NB32, AHEAD = 1, NBUF = 2;
    uint4 b[NBUF][2][3];
    auto load_b = [&](const int i) {
      const int s = i % STEPS, j = i / STEPS;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[i % NBUF][h][pc] = *reinterpret_cast<const uint4*>(p + pc * PS + 2 * h * CHS + (s + j * 32) * 8);
    };
    constexpr int WP[6] = {2, 1, 0, 1, 0, 0}, XP[6] = {0, 1, 2, 0, 1, 0};
    load_b(0);
#pragma unroll
    for (int j = 0; j < NB32; ++j) {
      f16v acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int i = j * STEPS + s;
        if (i + AHEAD < PAIRS) load_b(i + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 6; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[(s * 2 + h) * 3 + WP[t]]),
                                                         __builtin_bit_cast(bf8, b[i % NBUF][h][XP[t]]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MODE == 6) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {  // rows 8 blk + 4 (lane / 32) + 0 .. 3 of column n: one quad of channels
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc[4 * blk + q], 0.f);
          const unsigned h0 = pack2(v[0], v[1]), h1 = pack2(v[2], v[3]);
          const float r0 = v[0] - __uint_as_float(h0 << 16), r1 = v[1] - __uint_as_float(h0 & 0xffff0000u);
          const float r2 = v[2] - __uint_as_float(h1 << 16), r3 = v[3] - __uint_as_float(h1 & 0xffff0000u);
          const unsigned m0 = pack2(r0, r1), m1 = pack2(r2, r3);
          const unsigned l0 = pack2(r0 - __uint_as_float(m0 << 16), r1 - __uint_as_float(m0 & 0xffff0000u));
          const unsigned l1 = pack2(r2 - __uint_as_float(m1 << 16), r3 - __uint_as_float(m1 & 0xffff0000u));
          unsigned short* q = wimg + j * 32 * 8 + (blk & 1) * CHS + kb * 4;  // chunk = blk: two of them stand in for the four
          *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(q + PS) = make_uint2(m0, m1);
          *reinterpret_cast<uint2*>(q + 2 * PS) = make_uint2(l0, l1);
        }
      } else {
        keep += acc[0] + acc[15];
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (keep == 1.2345e-30f) out[0] = keep;
  if (tid == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int MODE>
void run32(const char* name, const uint4* af, float* out, unsigned long long* clk, int waves) {
  const int reps = 2000, grid = 256, lds_bytes = 3 * PS_PAD * 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k32<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k32<MODE>, dim3(grid), dim3(64 * waves), lds_bytes, 0, af, out, 50, clk);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k32<MODE>, dim3(grid), dim3(64 * waves), lds_bytes, 0, af, out, reps, clk);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < grid; ++i) cyc += (double)h[2 * i], wall += (double)h[2 * i + 1] / 100e6;
  cyc /= grid, wall /= grid;
  const double equiv_per_wave = (double)reps * STEPS * NB32 * 12 * 2;  // a 32x32x16 instruction = two 16,384-FLOP equivalents
  const double per_simd = cyc / (equiv_per_wave * (waves / 4));
  const double tflops = 256.0 * waves * equiv_per_wave / 6.0 * 16384.0 / wall / 1e12;
  printf("%-62s %d waves/SIMD: %5.1f cycles per MFMA equivalent and SIMD  clock %.2f GHz  %6.1f TFLOP/s fp32-equivalent\n", name, waves / 4,
         per_simd, cyc / wall / 1e9, tflops);
}

template <int MODE>
void run(const char* name, const uint4* af, float* out, unsigned long long* clk, int waves) {
  const int reps = 2000, grid = 256, lds_bytes = 3 * PS_PAD * 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), lds_bytes, 0, af, out, 50, clk);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), lds_bytes, 0, af, out, reps, clk);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < grid; ++i) cyc += (double)h[2 * i], wall += (double)h[2 * i + 1] / 100e6;
  cyc /= grid, wall /= grid;
  const double mfma_per_wave = (double)reps * STEPS * NB * 6;
  const double per_simd = cyc / (mfma_per_wave * (waves / 4));
  const double tflops = 256.0 * waves * mfma_per_wave / 6.0 * 16384.0 / wall / 1e12;
  printf("%-62s %d waves/SIMD: %5.1f cycles per MFMA and SIMD  clock %.2f GHz  %6.1f TFLOP/s fp32-equivalent\n", name, waves / 4, per_simd,
         cyc / wall / 1e9, tflops);
}

int main() {
  uint4* af;
  float* out;
  unsigned long long* clk;
  hipMalloc(&af, STEPS * 3 * 64 * 16);
  hipMalloc(&out, 64);
  hipMalloc(&clk, 2 * 256 * 8);
  std::vector<unsigned> ha(STEPS * 3 * 64 * 4);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = 0x3c003b80u + (unsigned)(i % 97) * 0x00010001u;
  hipMemcpy(af, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  for (int waves : {4, 8}) {
    run<0>("B from registers", af, out, clk, waves);
    run<1>("B from LDS (chunk planes), no epilogue", af, out, clk, waves);
    run<2>("B from LDS + split / three ds_write_b64 per n-tile", af, out, clk, waves);
    run<3>("B from LDS ([column][C + 8]: 2-way conflicts), no epilogue", af, out, clk, waves);
    run<4>("B from LDS, two accumulators interleaved", af, out, clk, waves);
    run32<5>("32x32x16: B from LDS (chunk planes), no epilogue", af, out, clk, waves);
    run32<6>("32x32x16: B from LDS + split / three ds_write_b64 per quad", af, out, clk, waves);
  }
  return 0;
}
