// Host-callable launchers of the non-GEMM kernels (wavefront-level attention, additive-attention
// pooling, dense batching, scorer, loss, Adam).  Definitions in nrl_kernels.hip.
#pragma once
#include "nrl_common.h"

namespace nrl {

// Addressing of one family of tiny attentions inside a packed (rows, 3D) q|k|v buffer.
// group = outer * heads + head; element (group, s, d) of q lives at
//   qkv[outer * q_outer + s * q_seq + head * dh + d],   k at + D, v at + 2D
// and of the output at o[outer * o_outer + s * o_seq + head * dh + d].
// News encoder (text.py:228-229): outer = news row, s = token.  User encoder (user/nrms.py:34-36,
// seq-first quirk): outer = history slot, s = USER index.
struct AttnGeom {
  int64_t q_outer, q_seq, o_outer, o_seq;
  int64_t groups;
  int heads, S, D, dh;
  float scale;  // 1/sqrt(dh), applied to q before QK^T as torch does
};

bool attn_head_dim_supported(int dh);
// o (rows, D); lse (groups, S) may be null in inference
int attn_fwd(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream);
// the same forward on the fp32 matrix cores (flash attention, nrl_attn_mfma.hip); attn_fwd routes S >= 64 here
int attn_fwd_mfma(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream);
int attn_bwd_mfma(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                  const AttnGeom& G, hipStream_t stream);
// the same pair for (by default) 32 <= S <= 128, dh = 20 on the bf16 matrix cores with (hi, lo) split operands (nrl_attn_x3.hip): the NRMS
// user encoder's across-users attention under the bf16x3 engine (the exact-fp32 engine keeps the fp32 kernels)
bool attn_x3_ok(const AttnGeom& G);
int attn_fwd_x3(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream);
int attn_bwd_x3(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, const AttnGeom& G,
                hipStream_t stream);
// attn_fwd_x3 with the in-projection inside: q|k|v of a group computed from its input rows (x + outer * x_outer + r * x_seq) and the
// per-head weight image (rp_jobs_add_qkv_heads); `qkv` (packed rows for the backward) may be null
bool attn_x3_proj_ok(const AttnGeom& G, int nblk, int kblocks);
int attn_fwd_x3_proj(const float* x, int64_t x_outer, int64_t x_seq, const uint16_t* img, int nblk, float* qkv, float* o,
                     float* lse, const AttnGeom& G, hipStream_t stream);
// dqkv (rows, 3D) fully overwritten
int attn_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
             const AttnGeom& G, hipStream_t stream);

// additive attention tail (attention.py:37-40): a = t . q_a; w = softmax_S(a); out = sum w y.
// rows of group g are [g*S, (g+1)*S).  w (M) is saved for backward.
int pool_fwd(const float* t, const float* q_a, const float* y, int64_t groups, int S, int Q, int D,
             float* w, float* out, hipStream_t stream);
// turns t (tanh outputs) IN PLACE into d_pre = da * q_a * (1 - t^2) and adds dq_a; with `dpre_planes` d_pre is written
// there instead, as (hi, lo) bf16 fragment-block planes over the rows ((Q + 15) / 16 block columns), and t is not touched;
// with `y_planes` (and y == nullptr) y is read from its planes ((D + 16) / 16 block columns: the fused news tail)
int pool_bwd_pre(const float* d_out, const float* y, const float* w, float* t_dpre,
                 const float* q_a, float* dq_a, int64_t groups, int S, int Q, int D,
                 hipStream_t stream, void* dpre_planes = nullptr, const void* y_planes = nullptr);

int to_dense_fwd(const float* x, const int64_t* offsets, int64_t B, int64_t max_len, int D,
                 float* dense, hipStream_t stream);
int to_dense_bwd(const float* d_dense, const int64_t* offsets, int64_t B, int64_t max_len, int D,
                 int64_t n_rows, float* d_x, hipStream_t stream);

// late fusion (nrms_module.py:243-248): user[b] = sum_h hist[b, h, :] / size[b] over the zero-padded dense
// history; _bwd: d_hist[b, h, :] = d_user[b, :] / size[b] for every slot h
int hist_mean_fwd(const float* hist, const int64_t* offsets, int64_t B, int64_t max_len, int D, float* user,
                  hipStream_t stream);
int hist_mean_bwd(const float* d_user, const int64_t* offsets, int64_t B, int64_t max_len, int D, float* d_hist,
                  hipStream_t stream);

int dot_scores_fwd(const float* user, const float* cand, int64_t B, int64_t C, int D, float* scores,
                   hipStream_t stream);
int dot_scores_bwd(const float* d_scores, const float* user, const float* cand, int64_t B, int64_t C,
                   int D, float* d_user, float* d_cand, hipStream_t stream);
int supcon_loss_fwd_bwd(const float* scores, const float* y, const int64_t* sizes, int64_t B, int64_t C,
                        float temperature, float grad_scale, float* loss, float* d_scores, hipStream_t stream);
int ce_loss_fwd_bwd(const float* scores, const float* y, int64_t B, int64_t C, float grad_scale,
                    float* loss, float* d_scores, hipStream_t stream);

int adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
              double eps, int64_t step, float grad_scale, int zero_grad, hipStream_t stream);
// lazy row-wise Adam of a (rows, dim) table (nrl_kernels.hip): mark the rows a step touches; advance marked rows / a slice /
// all rows to step `upto0` with zero gradients and, with_grad, apply step upto0 + 1 with their gradient rows
int adam_rows_mark(const int64_t* ids, int64_t n, int64_t rows, int32_t* mark, int64_t step, hipStream_t stream);
int adam_rows_advance(float* p, float* g, float* m, float* v, int64_t rows, int dim, int32_t* last, const int32_t* mark,
                      int32_t* status, int64_t stride, int64_t offset, int64_t upto0, int with_grad, double lr, double b1,
                      double b2, double eps, float grad_scale, hipStream_t stream, const int32_t* excl = nullptr,
                      int32_t excl_tag = 0);

int embedding_gather(const float* table, const int64_t* ids, int64_t n_ids, int D, float* out,
                     hipStream_t stream);
// ... for any nn.Embedding: dim <= 1024, the row `padding_idx` (< 0: none) skipped
int embedding_grad_any(const float* dx, const int64_t* ids, const int64_t* order, int64_t n_rows, int D, int64_t padding_idx,
                       float* d_table, hipStream_t stream);
// d_table[ids[p]] += dx[p] for p in id-sorted order (`order` = argsort(ids)); id 0 skipped
int embedding_grad_sorted(const float* dx, const int64_t* ids, const int64_t* order, int64_t n_rows, int D,
                          float* d_table, hipStream_t stream, const int32_t* cidx = nullptr);
// live (id != 0) token positions in position order: scratch of live_compact_ints(n) int32 -> list (n_live entries), cidx (n),
// *n_live (device scalar); two small launches
size_t live_compact_ints(int64_t n);
int live_compact(const int64_t* ids, int64_t n, int32_t* scratch, const int32_t** list, const int32_t** cidx,
                 const int32_t** n_live, hipStream_t stream);
int dropout_mask(uint8_t* keep, int64_t n, Dropout d, hipStream_t stream);

// ---- LSTUR path ------------------------------------------------------------------------------
// out[i] = table[ids[i]] * m; row_mode 0: elementwise dropout with flat index i*D + col (the CNN text
// encoder's post-embedding dropout, text.py:165-166); row_mode 1: one multiplier per row i
// (nn.Dropout2d on (1, B, D), user/lstur.py:71); drop.thresh == 0: bit-exact lookup
int embedding_rows_fwd(const float* table, const int64_t* ids, int64_t n_ids, int D, Dropout drop, int row_mode,
                       float* out, hipStream_t stream);
// d_table[ids[i]] += d_out[i] * m(i) (row multiplier), id 0 skipped
int embedding_rows_bwd(const float* d_out, const int64_t* ids, int64_t n_ids, int D, Dropout drop, float* d_table,
                       hipStream_t stream);
// dst (Bd, A, D) = src (A, Bd, D) with the two leading axes swapped
int transpose01(const float* src, int64_t A, int64_t Bd, int D, float* dst, hipStream_t stream);
// one GRU step (torch nn.GRU cell, gates r|z|n): gi (B, 3Hd) = x W_ih^T + b_ih, gh (B, 3Hd) = h W_hh^T
// WITHOUT its bias (b_hh (3Hd) is added here), h_prev (B, Hd) -> h_new; rows with t >= len[b] keep h_prev.
// gates (B, 3Hd) <- (r, z, n) (may alias gi) and ghn (B, Hd) <- gh_n + b_hn when non-null (saved for
// backward).  gh is cleared after use (it is the atomic accumulator of the next step's split-K GEMM).
int gru_gate_fwd(const float* gi, float* gh, const float* b_hh, const float* h_prev, const int64_t* len, int t,
                 int64_t B, int Hd, float* gates, float* ghn, float* h_new, hipStream_t stream);
// adjoint of one step: dh (B, Hd) holds dL/dh_t on entry and the DIRECT part z * dh of dL/dh_{t-1} on
// exit (the recurrent part dgh W_hh is added by the caller's GEMM); dgi (may alias gates) / dgh (B, 3Hd)
int gru_gate_bwd(const float* gates, const float* ghn, const float* h_prev, const int64_t* len, int t, int64_t B,
                 int Hd, float* dh, float* dgi, float* dgh, hipStream_t stream);

}  // namespace nrl
