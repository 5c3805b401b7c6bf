// Host-side weight preparation for conv_mfma_kernel: fold the per-output-channel scale
// (BatchNorm in eval mode) into the weights, lay the weights out as the polyphase GEMM
// A-matrix, and pack it in v_mfma_f32_16x16x4_f32 A-operand lane order.
#include "conv_mfma.h"

namespace vp {

// Conv1d weight W[cout][cin][K] (torch layout), stride s, P output phases per row block.
// Row m = co*P + p computes out[co][P*n + p]; tap' = k + s*p  (conv_mfma.h header).
std::vector<float> amat_conv(const float* W, int cout, int cin, int K, int stride, int P, int cinp,
                             const float* row_scale) {
  const int taps = K + stride * (P - 1);
  std::vector<float> A((size_t)cout * P * taps * cinp, 0.f);
  for (int co = 0; co < cout; ++co) {
    const float s = row_scale ? row_scale[co] : 1.f;
    for (int p = 0; p < P; ++p) {
      float* row = &A[(size_t)(co * P + p) * taps * cinp];
      for (int k = 0; k < K; ++k) {
        const int tap = k + stride * p;
        for (int ci = 0; ci < cin; ++ci) row[tap * cinp + ci] = s * W[((size_t)co * cin + ci) * K + k];
      }
    }
  }
  return A;
}

// ConvTranspose1d weight Wt[cin][cout][7], stride 4: full output index o = 4*n + p gets
// x[n - j] * Wt[ci][co][p + 4j], j in {0,1}.  With IN_OFF = -1 the kernel's tap' reads
// x[n + tap' - 1], so tap' = 1 - j.
std::vector<float> amat_convT_k7s4(const float* Wt, int cin, int cout, int cinp, const float* row_scale) {
  const int K = 7, P = 4, taps = 2;
  std::vector<float> A((size_t)cout * P * taps * cinp, 0.f);
  for (int co = 0; co < cout; ++co) {
    const float s = row_scale ? row_scale[co] : 1.f;
    for (int p = 0; p < P; ++p) {
      float* row = &A[(size_t)(co * P + p) * taps * cinp];
      for (int j = 0; j < 2; ++j) {
        const int k = p + 4 * j;
        if (k >= K) continue;
        const int tap = 1 - j;
        for (int ci = 0; ci < cin; ++ci) row[tap * cinp + ci] = s * Wt[((size_t)ci * cout + co) * K + k];
      }
    }
  }
  return A;
}

// Upsample(2, nearest) + Conv1d(K odd, pad K/2) folded into a 2-phase filter on the un-upsampled
// input: u[i] = x[i >> 1], so y[2n+p] = sum_k W[k] u[2n+p+k-pad] = sum_d (sum_{k: floor((p+k-pad)/2)=d} W[k]) x[n+d].
// Rows m = co*2 + p; tap' = d - dmin with dmin = floor(-pad/2); the kernel uses IN_OFF = dmin.
std::vector<float> amat_upconv(const float* W, int cout, int cin, int K, int cinp) {
  auto fdiv2 = [](int x) { return (x >= 0) ? x / 2 : -((-x + 1) / 2); };
  const int pad = K / 2, dmin = fdiv2(-pad), dmax = fdiv2(K - pad), taps = dmax - dmin + 1;
  std::vector<float> A((size_t)cout * 2 * taps * cinp, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int p = 0; p < 2; ++p) {
      float* row = &A[(size_t)(co * 2 + p) * taps * cinp];
      for (int k = 0; k < K; ++k) {
        const int tap = fdiv2(p + k - pad) - dmin;
        for (int ci = 0; ci < cin; ++ci) row[tap * cinp + ci] += W[((size_t)co * cin + ci) * K + k];
      }
    }
  return A;
}

// [M][taps*cinp] -> [MT][CB][taps][64]: lane l of K-step (cb, tap) holds
// A[mt*16 + (l & 15)][tap*cinp + cb*4 + (l >> 4)].
std::vector<float> pack_afrag(const std::vector<float>& amat, int M, int cinp, int taps) {
  const int MT = M / 16, CB = cinp / 4;
  std::vector<float> out((size_t)M * cinp * taps);
  for (int mt = 0; mt < MT; ++mt)
    for (int cb = 0; cb < CB; ++cb)
      for (int tap = 0; tap < taps; ++tap)
        for (int l = 0; l < 64; ++l) {
          const int m = mt * 16 + (l & 15), ci = cb * 4 + (l >> 4);
          out[(((size_t)mt * CB + cb) * taps + tap) * 64 + l] = amat[(size_t)m * taps * cinp + tap * cinp + ci];
        }
  return out;
}

}  // namespace vp
