// EQTransformer bottleneck kernels (sequence length 47, 16 channels): BiLSTM blocks,
// additive-attention transformer blocks and the P/S pick branches (LSTM + banded attention).
// (The 8->1 sigmoid heads run in the epilogue of the last decoder conv, conv_mfma.h EPI_HEAD.)  SURVEY.md §8a row A5, Appendix A.4/B.2.
//
// These stages are latency-bound (47 sequential LSTM steps; 47x47x32 tanh per attention)
// and hold <2 % of the model's FLOPs, so they are plain VALU code: one workgroup per
// window, activations in LDS, wavefront shuffles / readlanes for the recurrent state and
// the softmax-style row reductions.
#pragma once
#include "vp_common.h"

namespace vp {

constexpr int EQT_T = 47;  // bottleneck sequence length (6000 / 2^7)
constexpr int EQT_H = 16;  // hidden size / channels

struct LstmWeights {  // one direction, torch layout, gate order i,f,g,o
  const float* w_ih;  // [64][CIN]
  const float* w_hh;  // [64][16]
  const float* b;     // [64] = b_ih + b_hh
};

struct BiLstmArgs {
  const float* src;  // [B][CIN][ls]
  int ls_src;
  long ws_src;
  float* dst;        // [B][16][ls]
  int ls_dst;
  long ws_dst;
  LstmWeights fwd, bwd;
  const float* wc;   // folded conv1x1+BN: [16][32]
  const float* bc;   // [16]
};

struct AttnWeights {
  const float* Wt;  // [16][32]
  const float* Wx;  // [16][32]
  const float* bh;  // [32]
  const float* Wa;  // [32]
};

struct TransformerArgs {
  const float* src;  // [B][16][ls]
  int ls_src;
  long ws_src;
  float* dst;        // [B][16][ls]
  int ls_dst;
  long ws_dst;
  float* up;         // optional: decoder input, decoder input rows [B][16][ls_up] (set 0)
  int ls_up;
  long ws_up;
  AttnWeights att;
  const float *g1, *b1, *g2, *b2;  // LayerNormalization gamma/beta [16]; contiguous: g1[0..64) = g1 | b1 | g2 | b2
  const float* w1;   // [128][16]
  const float* bb1;  // [128]
  const float* w2;   // [16][128]
  const float* bb2;  // [16]
  float attn_eps, ln_eps;
};

struct PickBranchArgs {
  const float* src;  // transformer output [B][16][ls]
  int ls_src;
  long ws_src;
  float* up;         // decoder input sets 1..2: [3B][16][ls_up]
  int ls_up;
  long ws_up;
  int B;
  LstmWeights lstm[2];
  AttnWeights att[2];
  float attn_eps;
  int width;         // band width (3)
};

// The three BiLSTM blocks, the two transformer blocks and the two pick branches of one window as ONE launch
// (eqt_mid_kernel): the six launches it replaces are a chain of latency-bound kernels (47 sequential LSTM steps four
// times over) that the three device contexts cannot hide behind each other's MFMA work — measured: with them removed
// from the pipeline a 256-window step takes 510 instead of 615 us.
struct MidArgs {
  BiLstmArgs lstm[3];     // lstm[0].src: ResCNN output (64 channels); the others chain through LDS
  TransformerArgs tr[2];
  PickBranchArgs pick;
  unsigned long long* clk;  // optional debug: 8 shader-clock stamps per window (start, after each of the six stages, end)
  int B;                    // windows (the two-window workgroups clamp their last window to it)
};
// teams of eight waves: two windows per 1024-thread workgroup (128 CUs for a batch of 256; the default of rounds 3-5), or
// one_window_per_workgroup: the 512-thread form
int launch_eqt_mid(const MidArgs& a, int B, hipStream_t s, bool one_window_per_workgroup);
// default since round 6: teams of four waves, FOUR windows per 1024-thread workgroup (64 CUs for a batch of 256), eqt_mid4.hip;
// bit-identical to the forms above
int launch_eqt_mid4(const MidArgs& a, int B, hipStream_t s);

int launch_bilstm(const BiLstmArgs& a, int cin, int B, hipStream_t s);
int launch_transformer(const TransformerArgs& a, int B, hipStream_t s);
int launch_pick_branch(const PickBranchArgs& a, hipStream_t s);

}  // namespace vp
