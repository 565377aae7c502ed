// Host-side interface of the fused back half of the NRMS user encoder's forward (kernel: nrl_user_tail.hip).
#pragma once
#include "nrl_common.h"

namespace nrl {

struct UserTailArgs {
  const float* o;          // (groups * H, D) attention output rows
  const uint16_t* img_o;   // forward image of W_o (rp_weight_image_kernel: N = D columns over K = D, no bias row)
  const uint16_t* img_a;   // forward image of W_a (N = Q over K = D)
  int nblk_o, nblk_a;      // column blocks of the two images
  const float *b_o, *b_a, *q_a;
  int64_t groups;          // users
  int H, D, Q;             // rows per user (<= 64), 300, query width (<= 208)
  Dropout drop2;           // dropout of y (flat index row * D + col; p = 0: thresh 0)
  float *y, *t, *w, *out;  // (M, D), (M, Q), (M), (groups, D)
};

struct UserTailBwdArgs {
  const float* d_out;      // (groups, D)
  const float* y;          // (M, D) post-dropout out-projection rows
  float* t;                // (M, Q): tanh output in, d_pre out (in place: the additive-attention weight gradient's operand)
  const float* w;          // (M) pooling weights
  const float* q_a;
  const uint16_t* img_ad;  // dgrad image of W_a: N = D columns over K = Q (7 k-blocks)
  const uint16_t* img_od;  // dgrad image of W_o: N = D over K = D
  int nblk_ad, nblk_od;
  int64_t groups;
  int H, D, Q;
  Dropout drop2;
  float *dy, *d_o, *dq_a;  // (M, D), (M, D), (Q) accumulated
};

bool user_tail_bwd_ok(int64_t groups, int H, int D, int Q, int nblk_ad, int nblk_od, int kb_ad, int kb_od);
int user_tail_bwd(const UserTailBwdArgs& a, hipStream_t st);
bool user_tail_ok(int64_t groups, int H, int D, int Q, int nblk_o, int nblk_a);
int user_tail_fwd(const UserTailArgs& a, hipStream_t st);

}  // namespace nrl
