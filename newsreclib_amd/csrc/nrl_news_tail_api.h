// Host-side interface of the fused news-encoder back half (kernels: nrl_news_tail.h, compiled in nrl_news_tail.hip so that
// an edit of the kernels does not recompile the C ABI's translation unit).
#pragma once
#include "nrl_common.h"

namespace nrl {

constexpr int NT_FB = 19;            // feature blocks of y (D = 300 + the ones column)
constexpr int NT_KB = 10;            // k-blocks of 32 plane slots of `o` (19 block columns)
constexpr int NT_QB = 13;            // query blocks (Q <= 208)
constexpr int NT_KS = 10;            // k-steps of phase 2 (pairs of feature blocks)
struct NewsTailArgs {
  const unsigned char* o_planes;  // (hi, lo) planes of `o` over the real rows, 19 block columns (head-permuted + ones at slot D)
  const uint16_t* img_o;          // W_o image, reduction in plane-slot order, b_o at slot D (rp_jobs_add_kperm with bias)
  const uint16_t* img_a;          // W_a image, reduction in kappa order, b_a at feature D (rp_jobs_add_kappa)
  const float* q_a;               // (Q)
  int64_t n_news;
  int L, D, Q;
  Dropout drop2;
  float* out;                     // (n_news, D)
  unsigned char* y_planes;        // training: post-dropout y as planes over the real rows (19 block columns, ones at D), or null
  float* t;                       // training, optional: tanh output (n_news * L, Q) for pool_bwd_pre; null when the backward
                                  // recomputes it (news_tail_bwd_kernel)
  float* w;                       // training: pooling weights (n_news * L), or null
  // evaluation only: the short-first news list of the fused front half (NewsFusedArgs::perm / n_short, nrl_news_fused.h).  A
  // short news has ONE distinct token row from token 15 on, so its wave computes token block 0 only and takes the logit and
  // the y values of tokens 16 .. L - 1 from token 15.  Both null: off.
  const int32_t* perm = nullptr;
  const int32_t* n_short = nullptr;
};

struct NewsTailBwdArgs {
  const unsigned char* y_planes;  // the forward's y planes (19 block columns, ones at feature D)
  const float* w;                 // (n_news * L) pooling weights
  const float* d_out;             // (n_news, D)
  const uint16_t* img_a;          // the forward's W_a image (kappa order, b_a at feature D)
  const uint16_t* img_ad;         // W_a^T image: 19 feature blocks, reduction = queries in kappa order (7 k-blocks)
  const float* q_a;               // (Q)
  int64_t n_news;
  int L, D, Q;
  Dropout drop2;
  unsigned char* dpre_planes;     // out: (Q + 15) / 16 block columns per 16-row block
  unsigned char* dy_planes;       // out: 19 block columns
  float* dq_a;                    // (Q), accumulated
};

constexpr int NT_QS = 7;           // k-steps of phase C (pairs of query blocks)
constexpr int NT_DROW = 320;       // floats per news of the staged d_out (zero-padded past D)

// (the fused tail applies where the fused front half does: L <= 32, D = 300 = 15 heads x 20)
static inline bool news_tail_geometry_ok(int L, int D, int Q, int heads) {
  return L >= 1 && L <= 32 && D == 300 && heads == 15 && Q % 4 == 0 && Q > 0 && Q <= 16 * NT_QB;
}
static inline bool news_tail_bwd_geometry_ok(int L, int D, int Q, int heads) {
  return news_tail_geometry_ok(L, D, Q, heads) && Q < 16 * NT_QB && Q <= 256;   // (a free query row for d_out)
}

// out-projection + dropout + additive attention + pooling of n_news news in one launch
int news_tail_fwd(const NewsTailArgs& a, hipStream_t st);
// additive-attention backward (tanh recomputed from the y planes): d_pre / dy planes, dq_a
int news_tail_bwd(const NewsTailBwdArgs& a, hipStream_t st);

}  // namespace nrl
