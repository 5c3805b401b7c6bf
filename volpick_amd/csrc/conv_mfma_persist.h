// Persistent form of conv_mfma_kernel for the long, low-channel layers (EQT encoder head, decoder
// tail): one workgroup walks TPW consecutive tiles of a window.  The packed A fragments of each wave's
// m-tiles are loaded ONCE into registers (K is only 12-40 steps here, so an L2 round trip per tile is a
// large fraction of the tile's MFMA time), the K loop is fully unrolled and only touches LDS, and the
// next tile's input rows are fetched into registers while the current tile computes.
// Same arithmetic, same k-order and the same ConvCfg / ConvArgs / packed weights as conv_mfma_kernel;
// epilogues: EPI_STORE (aligned), EPI_POOL2, EPI_HEAD.
#pragma once
#include "conv_mfma.h"

namespace vp {

template <class C, int TPW>
__global__ __launch_bounds__(256) void conv_mfma_persist_kernel(const ConvArgs a, const int n_tiles) {
  static_assert(C::EPI == EPI_STORE || C::EPI == EPI_POOL2 || C::EPI == EPI_HEAD, "epilogue not supported here");
  static_assert(C::EPI != EPI_STORE || C::OUT_OFF % 4 == 0, "aligned store path only");
  extern __shared__ float4 lds_raw[];
  float* lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int win = blockIdx.y;
  const int set = win / a.win_per_set;
  const int tile_lo = blockIdx.x * TPW, tile_hi = (tile_lo + TPW < n_tiles) ? tile_lo + TPW : n_tiles;
  constexpr int STEP = (C::EPI == EPI_HEAD) ? C::TN - 8 : C::TN;
  constexpr int COL_SHIFT = (C::EPI == EPI_HEAD) ? -4 : 0;
  const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
  const int g = lane >> 4, n = lane & 15;

  // ---- weights of this wave: A fragments and biases in registers, once per workgroup -----------
  constexpr int KS = C::CB * C::TAPS;
  float areg[C::MW][KS], biasv[C::MW][4];
  {
    const float* ap = a.afrag + (long)set * a.afrag_set_stride + (long)(wm * C::MW) * KS * 64 + lane;
    const float* bias = a.bias + (long)set * a.bias_set_stride;
#pragma unroll
    for (int i = 0; i < C::MW; ++i) {
#pragma unroll
      for (int k = 0; k < KS; ++k) areg[i][k] = ap[((long)i * KS + k) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) biasv[i][r] = bias[((wm * C::MW + i) * 16 + 4 * g + r) / C::P];
    }
  }

  // ---- register prefetch of a tile's input rows ------------------------------------------------
  constexpr int TOT = C::CINP * C::W4, N_IT = (TOT + 255) / 256;
  float4 pf[N_IT];
  const float* s1w = a.src1 + (long)win * a.ws1;
  const float* s2w = (C::CIN2 > 0) ? a.src2 + (long)win * a.ws2 : nullptr;
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    const int a0 = HALO + C::SN * (tile * STEP + COL_SHIFT) + C::IN_OFF_F4;  // multiple of 4 by construction
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int idx = tid + k * 256;
      pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < TOT) {
        const int c = idx / C::W4, q = idx - c * C::W4;
        if (c < C::CIN1) {
          pf[k] = *reinterpret_cast<const float4*>(s1w + a0 + (long)c * a.ls1 + 4 * q);
        } else if (c < C::CIN) {
          pf[k] = *reinterpret_cast<const float4*>(s2w + a0 + (long)(c - C::CIN1) * a.ls2 + 4 * q);
        }
      }
    }
  };
  fetch(tile_lo);

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int col0 = tile * STEP + COL_SHIFT;
#pragma unroll
    for (int k = 0; k < N_IT; ++k) {
      const int idx = tid + k * 256;
      if (idx < TOT) {
        const int c = idx / C::W4, q = idx - c * C::W4;
        *reinterpret_cast<float4*>(lds + c * C::S + 4 * q) = pf[k];
      }
    }
    __syncthreads();
    if (tile + 1 < tile_hi) fetch(tile + 1);

    // ---- MFMA: K fully unrolled, A from registers, B from LDS at immediate offsets ---------------
    f32x4 acc[C::MW][C::NW];
#pragma unroll
    for (int i = 0; i < C::MW; ++i)
#pragma unroll
      for (int j = 0; j < C::NW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bp = lds + g * C::S + (wn * C::NW * 16 + n) * C::SN + C::SHIFT;
#pragma unroll
    for (int cb = 0; cb < C::CB; ++cb)
#pragma unroll
      for (int tap = 0; tap < C::TAPS; ++tap)
#pragma unroll
        for (int i = 0; i < C::MW; ++i)
#pragma unroll
          for (int j = 0; j < C::NW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][cb * C::TAPS + tap],
                                                             bp[cb * 4 * C::S + j * 16 * C::SN + tap], acc[i][j], 0, 0, 0);
    __syncthreads();  // all B reads done; the LDS image is reused as the output staging tile

    // ---- epilogue 1: bias (+ReLU), D fragments -> LDS [COUT][OS] -----------------------------
#pragma unroll
    for (int i = 0; i < C::MW; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wm * C::MW + i) * 16 + 4 * g + r;
        const int co = m / C::P, p = m - co * C::P;
#pragma unroll
        for (int j = 0; j < C::NW; ++j) {
          float v = acc[i][j][r] + biasv[i][r];
          if (C::RELU) v = fmaxf(v, 0.f);
          if constexpr (C::EPI == EPI_HEAD) {  // outside the signal the head must see zero padding
            const int tg = C::P * (col0 + (wn * C::NW + j) * 16 + n) + p;
            if (tg < 0 || tg >= a.l_out) v = 0.f;
          }
          lds[co * C::OS + C::P * ((wn * C::NW + j) * 16 + n) + p] = v;
        }
      }
    }
    __syncthreads();

    // ---- epilogue 2 ---------------------------------------------------------------------------
    const int t0 = C::P * col0 + C::OUT_OFF;  // global output index of staged column 0
    if constexpr (C::EPI == EPI_STORE) {
      float* d = a.dst + (long)win * a.wsd + a.dst_halo;
      for (int idx = tid; idx < C::COUT * (C::OW / 4); idx += 256) {
        const int co = idx / (C::OW / 4), q = idx - co * (C::OW / 4);
        const int t = t0 + 4 * q;
        if (t < a.l_out) {
          float4 v = *reinterpret_cast<const float4*>(lds + co * C::OS + 4 * q);
          if (t + 1 >= a.l_out) v.y = 0.f;  // keep the right margin zero
          if (t + 2 >= a.l_out) v.z = 0.f;
          if (t + 3 >= a.l_out) v.w = 0.f;
          *reinterpret_cast<float4*>(d + (long)co * a.lsd + t) = v;
        }
      }
    } else if constexpr (C::EPI == EPI_POOL2) {
      float* d = a.dst + (long)win * a.wsd + a.dst_halo;
      for (int idx = tid; idx < C::COUT * (C::OW / 2); idx += 256) {
        const int co = idx / (C::OW / 2), q = idx - co * (C::OW / 2);
        const int t = t0 + 2 * q;
        if (t < a.l_out) {
          float v = lds[co * C::OS + 2 * q];
          if (t + 1 < a.l_out) v = fmaxf(v, lds[co * C::OS + 2 * q + 1]);
          d[(long)co * a.lsd + (t >> 1)] = v;
        }
      }
    } else {  // EPI_HEAD: see conv_mfma_kernel
      const int b = win - set * a.win_per_set;
      float* y = a.dst + ((long)b * 3 + set) * a.l_out;
      const float* w = a.e0 + set * 88;
      const float bias_h = a.e1[set];
      constexpr int N_THR = (C::P * (C::TN - 8) + 8 - 5 + 3) / 4;
      if (tid < N_THR) {
        float hacc[4] = {bias_h, bias_h, bias_h, bias_h};
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          float v[16];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 x4 = *reinterpret_cast<const float4*>(lds + ci * C::OS + 4 * tid + 4 * q4);
            v[4 * q4] = x4.x;
            v[4 * q4 + 1] = x4.y;
            v[4 * q4 + 2] = x4.z;
            v[4 * q4 + 3] = x4.w;
          }
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            const float wk = w[ci * 11 + k];
#pragma unroll
            for (int o = 0; o < 4; ++o) hacc[o] = fmaf(wk, v[o + k], hacc[o]);
          }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int col = 4 * tid + 5 + o;
          const int t = C::P * col0 + col;
          if (col >= 8 && col < 8 + C::P * (C::TN - 8) && t < a.l_out) y[t] = 1.f / (1.f + expf(-hacc[o]));
        }
      }
    }
    __syncthreads();  // the staged tile is overwritten by the next tile's input rows
  }
}

template <class C, int TPW>
int launch_conv_persist(const ConvArgs& a, int cols, hipStream_t stream) {
  constexpr int STEP = (C::EPI == EPI_HEAD) ? C::TN - 8 : C::TN;
  const int n_tiles = (cols + STEP - 1) / STEP;
  dim3 grid((n_tiles + TPW - 1) / TPW, a.n_windows, 1);
  hipLaunchKernelGGL((conv_mfma_persist_kernel<C, TPW>), grid, dim3(256), C::LDS_FLOATS * sizeof(float), stream, a,
                     n_tiles);
  return 0;
}

}  // namespace vp
