/*
 * newsreclib_amd.h -- C ABI of the MI355X (gfx950) NRMS hot path.
 *
 * This is the drop-in boundary under the reference's operator API (plug point 3):
 * the Python modules `news_encoder` / `user_encoder` / `click_predictor` of
 * `newsreclib_amd.nrms_module.NRMSModule` call these entry points through ctypes
 * (newsreclib_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - C linkage, plain pointers and sizes; no C++ or torch types cross the boundary.
 *   - every pointer except NrlBlockParams/NrlBlockGrads structs themselves is a DEVICE pointer on
 *     the current device; the caller (PyTorch's caching allocator) owns every buffer incl. the
 *     workspace; the library never allocates device memory and never synchronises.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.
 *   - return value 0 = success, negative = error (NRL_E_*); nrl_last_error() returns a
 *     thread-local message.  Nothing throws.
 *   - all floating point is IEEE fp32 ("dtype": "f32"); token ids / offsets are int64.
 *   - reference citations are relative to andreeaiana/newsreclib.
 */
#ifndef NEWSRECLIB_AMD_H
#define NEWSRECLIB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRL_ABI_VERSION 16

#define NRL_OK 0
#define NRL_E_INVALID (-1)   /* bad argument (shape / alignment / null) */
#define NRL_E_WORKSPACE (-2) /* workspace too small */
#define NRL_E_HIP (-3)       /* a HIP runtime call failed */

/* One "multi-head self-attention + additive attention" block.  Field <-> reference state_dict key
 * (prefix `news_encoder.text_encoders.title.` or `user_encoder.`):
 *   in_proj_weight  multihead_attention.in_proj_weight   (3D, D) rows [Wq; Wk; Wv]
 *   in_proj_bias    multihead_attention.in_proj_bias     (3D)
 *   out_proj_weight multihead_attention.out_proj.weight  (D, D)
 *   out_proj_bias   multihead_attention.out_proj.bias    (D)
 *   att_weight      additive_attention.linear.weight     (Q, D)
 *   att_bias        additive_attention.linear.bias       (Q)
 *   att_query       additive_attention.query             (Q)
 * PyTorch Linear layout (out, in) row-major, used in place (no re-layout, no copies).
 * Replaces: nn.MultiheadAttention + AdditiveAttention as wired in text.py:199-220 (MHSAAddAtt)
 * and user/nrms.py:23-30 (UserEncoder). */
typedef struct NrlBlockParams {
  const float* in_proj_weight;
  const float* in_proj_bias;
  const float* out_proj_weight;
  const float* out_proj_bias;
  const float* att_weight;
  const float* att_bias;
  const float* att_query;
  int32_t embed_dim;  /* D, multiple of 4 */
  int32_t num_heads;  /* D / num_heads in {16, 20, 32, 48, 64} */
  int32_t query_dim;  /* Q, multiple of 4 */
  int32_t gemm_engine; /* projection engine of THIS call: 0 = process default (nrl_set_gemm_engine),
                        * 1 = exact fp32, 2 = bf16x3.  A backward must be given the value its forward ran under
                        * (the bf16 weight planes in the workspace exist only under bf16x3). */
  int32_t options;     /* kernel-selection switches of THIS call: 0 = the process defaults (nrl_set_option / NRL_*
                        * environment), otherwise NRL_OPTIONS_EXPLICIT | mask with the bit order of nrl_get_options().  The
                        * switches choose the private formats of the workspace: a backward must be given the word its
                        * forward ran under (a host stores NRL_OPTIONS_EXPLICIT | nrl_get_options() at the forward). */
} NrlBlockParams;
#define NRL_OPTIONS_EXPLICIT 0x40000000

/* Gradient accumulators, same shapes as NrlBlockParams; kernels ADD into them (callers zero
 * them, or pass the persistent .grad / flat DP gradient buffer to accumulate in place). */
typedef struct NrlBlockGrads {
  float* in_proj_weight;
  float* in_proj_bias;
  float* out_proj_weight;
  float* out_proj_bias;
  float* att_weight;
  float* att_bias;
  float* att_query;
} NrlBlockGrads;

int nrl_abi_version(void);
/* 32 hex digits: hash of the sources the library was built from (newsreclib_amd/_build.py source_hash()); a host that has
 * the sources beside the library compares the two instead of trusting file times. */
const char* nrl_build_id(void);
const char* nrl_last_error(void);

/* ---- projection GEMM engine: the process DEFAULT, used by calls whose params carry gemm_engine == 0 and by the
 * entry points whose params have no such field (the same entry points serve both engines) ------
 *   0 = exact fp32: v_mfma_f32_16x16x4_f32, bitwise an fmaf chain (the reference's arithmetic type)
 *   1 = "bf16x3" (default): fp32 operands split a = hi + lo (two bf16), three bf16 MFMAs per product
 *       (hi*hi + hi*lo + lo*hi) with fp32 accumulation; ~2^-16 relative per product, scores within
 *       1e-4 of the fp32 path (contract 1e-3), ~1.3-2x faster GEMMs.  Attention, softmax, pooling,
 *       loss and Adam are fp32 in both engines. */
/* DEPRECATED as an interface (ABI v16): process-global mutable state, which SURVEY section 8(b) asks the boundary not to have.
 * Hosts set NrlBlockParams.gemm_engine per call (newsreclib_amd/ops.py captures it at every forward and hands it to the
 * backward); the setter stays for A/B scripts, the test fixture that runs the suite under both engines, and the entry points
 * whose params struct has no engine field. */
int nrl_set_gemm_engine(int32_t engine);
int nrl_get_gemm_engine(void);

/* ---- kernel-selection switches for A/B measurements and the equivalence tests.  All paths compute the same function;
 * they differ in the kernels used and in the PRIVATE formats of the workspace.  The switches of a call are
 * NrlBlockParams.options (per call, thread-safe); nrl_set_option changes the process DEFAULTS that options == 0 refers
 * to (the environment variables NRL_<NAME>=0/1 set the same defaults at load time).  Bit order of the mask:
 *   0 "news_fused"      gather + in-projection + token attention of the news encoder in one kernel (bf16x3 engine,
 *                       L <= 32, D = 20 * heads in [288, 316])
 *   1 "news_fused_bwd"  RETIRED (ABI v14): reserved, always 0 -- nrl_set_option(.., 1) and an options word with the bit set are
 *                       rejected.  (q|k|v recomputed inside the attention backward: measured slower, kernel in tools/experimental/)
 *   2 "news_attn_mfma"  token-attention backward of the fused news path on the matrix cores from head-major q|k|v slabs
 *   3 "news_planes"     x / dqkv of that path as pre-split bf16 fragment-block planes (DMA-only weight gradient)
 *   4 "news_od_planes"  o / dy of that path as planes too
 *   5 "news_aa_planes"  y and d_pre as planes for the additive-attention GEMMs
 *   6 "wgrad_2step"     split-K partial tiles stored and reduced in a second kernel instead of atomics
 *   7 "wgrad_ws"        wave-specialised kernel for the 900-row weight gradient
 *   8 "rowpanel"        row-panel kernel for the N <= 320 projections
 *   9 "x3_dma"          LDS-DMA staged tiled GEMMs
 *  10 "news_tail"       out-projection + dropout + additive attention + pooling of the fused news path in ONE kernel
 *  11 "news_tail_bwd"   additive-attention backward of that path in ONE kernel that recomputes tanh from the y planes
 *  12 "user_fork"       (default OFF, measured slower) user-encoder backward: the in-projection dgrad and the three weight gradients side by side on two
 *                       library-internal streams (forked from and joined back into the caller's stream inside the call)
 *  13 "news_fork"       (default ON since ABI v16; off before) news-encoder backward: the additive-attention and out-projection weight gradients on a
 *                       library-internal stream beside the activation-gradient chain of phase 1 (forked after the tail
 *                       backward, joined before the phase-1 call returns; a phase-2 call then runs only the in-projection one)
 *  14 "news_qkv_planes" token-attention backward of the fused news path: q|k|v / d_o split once into (hi, lo) bf16 planes in LDS, operand
 *                       fragments read from them (bit-identical to the kernel that builds each fragment from fp32)
 *  15 "news_pad_share"  (ABI v13) evaluation forward of the fused news path (nothing saved, p_drop == 0): the run of padding tokens
 *                       from token 15 on is ONE row (identical embedding row, no dropout), so a news whose tokens 15 .. L - 1 are
 *                       all the padding id is computed on its first 16 token rows only; bit-identical to computing every row
 *  16 "news_tail_od"    RETIRED (ABI v14): reserved, always 0 (the out-projection's activation gradient inside the fused tail backward:
 *                       measured slower, code in tools/experimental/)
 *  17 "user_proj"       (ABI v13) the NRMS user encoder's in-projection inside its across-users attention kernel (bf16x3 engine,
 *                       32 <= users <= 128 per call, D = 20 * heads in [288, 316]) instead of a separate GEMM launch
 * Entry points whose params struct has no `options` field run under the process defaults: their forward and backward
 * must see the same defaults (newsreclib_amd/ops*.py compare nrl_get_options() at both). */
/* DEPRECATED as an interface (ABI v16), for the same reason as nrl_set_gemm_engine: per-call NrlBlockParams.options is the
 * path; the setter stays for A/B measurement scripts and equivalence tests. */
int nrl_set_option(const char* name, int32_t value);
/* Bit mask of the process-default switch values (bit order above). */
int32_t nrl_get_options(void);

/* ---- measurement hook (bench.py "roofline"): HIP-event timing of the dominant kernel of the step, recorded on the launch
 * stream.  The ProfScope wraps the news-encoder forward's first launch only: under the default bf16x3 engine the fused
 * kernel (embedding gather + dropout + in-projection + per-head token attention, nrl_news_fused.h), counted as
 * 2*M*3D*D + 4*M*L*D FLOPs per launch; under the exact-fp32 engine the in-projection GEMM with the fused gather,
 * 2*M*3D*D.  Diagnostic state, process-wide: enable it around a measurement pass, not in a timed region. */
int nrl_prof_enable(int32_t on);
int nrl_prof_read(double* total_ms, int64_t* launches, double* total_flops);

/* ---- dropout keep-mask specification (normative statement: oracle/nrms_oracle.py) ----------
 * keep(i) = lowbias32(i * 0x9E3779B1 + key) >= floor(p * 2^32), i = row * D + col (32-bit).
 * Replaces nn.Dropout at text.py:220,225,230 (torch's RNG stream is not reproducible on device). */
uint32_t nrl_dropout_key(uint64_t seed, uint32_t stream);
int nrl_dropout_mask(uint8_t* keep, int64_t n_elems, double p, uint64_t seed, uint32_t stream,
                     void* stream_handle);

/* ---- token-id grouping for the embedding-table gradient (embedding_dense_backward, text.py:215-217,224) -----
 * order (n + 1) int64 <- order[0 .. n): the positions 0..n-1 of the flat id vector grouped by ASCENDING id (counting sort
 * over the vocabulary: LDS-merged histogram, scan, scatter; ids must lie in [0, vocab), vocab <= 2^20; the order inside
 * one id's run is unspecified); order[n]: the number of positions whose id is 0 -- they come first, so order[order[n] .. n)
 * lists the LIVE positions (the padding id's embedding row has no gradient).  This is the `sorted_positions` argument of
 * the *_encoder_bwd entry points; nrl_news_encoder_bwd reads the count too (ABI v11: it computes the gradient of the
 * gathered rows for the live positions only), so a host that sorts by other means appends it. */
size_t nrl_sort_positions_workspace_bytes(int64_t n, int64_t vocab);
int nrl_sort_positions(const int64_t* ids, int64_t n, int64_t vocab, int64_t* order, void* ws, size_t ws_bytes,
                       void* stream);

/* ---- news encoder: MHSAAddAtt.forward, text.py:222-236 (behind NewsEncoder.forward,
 * news.py:134-160) ------------------------------------------------------------------------------
 * ids (N, L) int64 -> out (N, D).  Embedding gather (bit-exact; id 0 is an ordinary row,
 * text.py:215-217) fused into the in-projection GEMM's A-tile loader; dropout stream `stream0`
 * after the gather and `stream0 + 1` after the attention out-projection when p_drop > 0.
 * `save_for_backward` != 0 keeps activations in `ws` for nrl_news_encoder_bwd. */
size_t nrl_news_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim,
                                        int32_t num_heads, int32_t query_dim);
int nrl_news_encoder_fwd(const NrlBlockParams* p, const float* emb_table, int64_t vocab,
                         const int64_t* ids, int64_t n_news, int32_t seq_len, double p_drop,
                         uint64_t seed, uint32_t stream0, int32_t save_for_backward, float* out,
                         void* ws, size_t ws_bytes, void* stream);
/* ---- evaluation forwards from a per-token q|k|v table (ABI v16) --------------------------------------------------
 * Validation / test re-encode every news of every impression under FROZEN weights (rec_dataset.py:98-121,
 * nrms_module.py:398-535), and without dropout the q|k|v rows of a token position depend on its token id alone
 * (text.py:224,229).  nrl_token_table_build runs the in-projection ONCE per vocabulary id -- plus the weight images the
 * back half needs -- into a caller-owned buffer; nrl_news_encoder_fwd_table is MHSAAddAtt.forward (text.py:222-236,
 * evaluation mode) from it: ids (N, L) -> out (N, D), the per-head 32 x 64 q|k|v image gathered by token id instead of
 * computed.  A table row holds the bits the projection of nrl_news_encoder_fwd produces for that id, so `out` is EQUAL
 * (torch.equal) to that call's with p_drop = 0, save_for_backward = 0.  The buffer is valid for exactly the parameter values
 * (embedding table, in/out projection, additive-attention linear) it was built from: the CALLER rebuilds it after any
 * update; `att_query` is read from `p` at every call.  bf16x3 engine, fused-news-encoder geometry only
 * (nrl_token_table_supported; D = 300, 15 heads, L <= 32, Q <= 208); nrl_token_table_bytes returns 0 outside it. */
int32_t nrl_token_table_supported(int32_t seq_len, int32_t embed_dim, int32_t num_heads, int32_t query_dim);
size_t nrl_token_table_bytes(int64_t vocab, int32_t embed_dim, int32_t num_heads, int32_t query_dim);
int nrl_token_table_build(const NrlBlockParams* p, const float* emb_table, int64_t vocab, void* table, size_t table_bytes,
                          void* stream);
size_t nrl_news_encoder_fwd_table_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t num_heads);
int nrl_news_encoder_fwd_table(const NrlBlockParams* p, const void* table, size_t table_bytes, int64_t vocab,
                               const int64_t* ids, int64_t n_news, int32_t seq_len, float* out, void* ws, size_t ws_bytes,
                               void* stream);
/* Backward of the above (autograd of text.py:222-236 incl. embedding_dense_backward with
 * padding_idx=0: rows of id 0 receive no gradient).  d_out (N, D).  Adds into `g` and into
 * d_emb_table (vocab, D).  `ws` must be the workspace the forward filled.
 * emb_table: the table the forward read (unchanged since); not read by any current path (the recomputing backward that
 * re-gathered rows from it was retired in ABI v14) -- may be NULL.
 * sorted_positions: optional (may be NULL) output of nrl_sort_positions over the flat (N*L) id vector: N*L + 1 entries,
 * the positions by ascending id and then the number of id-0 positions.  When given, the table gradient is reduced in
 * id-sorted order (one atomic per (id, 64-row segment)) instead of one atomic per element -- frequent tokens otherwise
 * serialise on their row -- and (fused path) the gradient of the gathered rows is computed for the LIVE positions only:
 * the padding id's rows, ~60 % of a MIND title batch, have no table gradient and nothing else reads theirs.
 * phase: 0 = whole backward; 1 = activation-gradient chain + table gradient only; 2 = the three
 * weight/bias-gradient GEMMs only (call 1 then 2: a data-parallel caller starts the all-reduce of the
 * table gradient, >96 % of the bytes, between them so it overlaps the weight-gradient GEMMs). */
int nrl_news_encoder_bwd(const NrlBlockParams* p, const NrlBlockGrads* g, const float* emb_table,
                         float* d_emb_table, int64_t vocab, const int64_t* ids, const int64_t* sorted_positions,
                         int64_t n_news, int32_t seq_len, double p_drop, uint64_t seed,
                         uint32_t stream0, const float* d_out, int32_t phase, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- user encoder: nrms UserEncoder.forward, user/nrms.py:32-41 --------------------------------
 * hist (B, H, D) -> out (B, D).  Reproduces the reference's seq-first nn.MultiheadAttention call:
 * attention runs across the B rows of dim 0 for each slot of dim 1 (SURVEY.md headline fact 3), then
 * additive attention pools over dim 1.  With p_drop > 0 a dropout (stream0) is applied to the input
 * and another (stream0 + 1) between attention and pooling: that is exactly the tail of the PLM text
 * encoder, PLM.forward text.py:92-99, called with hist = last_hidden_state (N_news, L, D).
 * input_dropout = 0 applies only the second one: the long-term branch of the CenNewsRec user encoder
 * (user/cen_news_rec.py:66-72). */
size_t nrl_user_encoder_workspace_bytes(int64_t batch, int64_t hist_len, int32_t embed_dim,
                                        int32_t num_heads, int32_t query_dim);
int nrl_user_encoder_fwd(const NrlBlockParams* p, const float* hist, int64_t batch,
                         int64_t hist_len, double p_drop, uint64_t seed, uint32_t stream0,
                         int32_t input_dropout, int32_t save_for_backward, float* out, void* ws,
                         size_t ws_bytes, void* stream);
int nrl_user_encoder_bwd(const NrlBlockParams* p, const NrlBlockGrads* g, const float* hist,
                         int64_t batch, int64_t hist_len, double p_drop, uint64_t seed,
                         uint32_t stream0, int32_t input_dropout, const float* d_out, float* d_hist,
                         void* ws, size_t ws_bytes, void* stream);
/* The same backward in two parts (ABI v12).  phase 1: everything up to d_hist (and the additive-attention query gradient);
 * phase 2: the three weight gradients, which only read what phase 1 left in `ws` (d_out / d_hist are not touched and may be
 * NULL) -- they are off the path into the news-encoder backward, so a caller may issue them on another stream, ordered after
 * phase 1, and join before the optimizer; phase 0 = nrl_user_encoder_bwd. */
int nrl_user_encoder_bwd_phase(const NrlBlockParams* p, const NrlBlockGrads* g, const float* hist,
                               int64_t batch, int64_t hist_len, double p_drop, uint64_t seed,
                               uint32_t stream0, int32_t input_dropout, const float* d_out,
                               float* d_hist, int32_t phase, void* ws, size_t ws_bytes, void* stream);

/* ---- to_dense_batch (torch_geometric 2.3.0; call sites nrms_module.py:233,237,277-284) ---------
 * x (N, D) + offsets (B+1) int64 (prefix sums of the sorted assignment vector) ->
 * dense (B, max_len, D), zero-filled past each group's length.  _bwd is the adjoint gather. */
int nrl_to_dense_batch_fwd(const float* x, const int64_t* offsets, int64_t batch, int64_t max_len,
                           int32_t dim, float* dense, void* stream);
int nrl_to_dense_batch_bwd(const float* d_dense, const int64_t* offsets, int64_t batch,
                           int64_t max_len, int32_t dim, int64_t n_rows, float* d_x, void* stream);

/* ---- late fusion (late_fusion=True): user vector = mean of the clicked-news vectors, nrms_module.py:243-248 /
 * lstur_module.py:295-296: hist (B, max_len, D) zero-padded dense history, offsets (B+1) as above ->
 * user[b] = sum_h hist[b, h] / (offsets[b+1] - offsets[b]).  _bwd writes d_hist (B, max_len, D). */
int nrl_hist_mean_fwd(const float* hist, const int64_t* offsets, int64_t batch, int64_t max_len,
                      int32_t dim, float* user, void* stream);
int nrl_hist_mean_bwd(const float* d_user, const int64_t* offsets, int64_t batch, int64_t max_len,
                      int32_t dim, float* d_hist, void* stream);

/* ---- click predictor: DotProduct.forward, click_predictor.py:9-11 as called at
 * nrms_module.py:251-253: user (B, D), cand (B, C, D) -> scores (B, C) -------------------------- */
int nrl_dot_scores_fwd(const float* user, const float* cand, int64_t batch, int64_t n_cand,
                       int32_t dim, float* scores, void* stream);
int nrl_dot_scores_bwd(const float* d_scores, const float* user, const float* cand, int64_t batch,
                       int64_t n_cand, int32_t dim, float* d_user, float* d_cand, void* stream);

/* ---- loss: CrossEntropyLoss()(scores, y_true) with float targets, nrms_module.py:287-288 --------
 * loss = mean_b(-sum_c y * log_softmax(scores)_c); also writes d_scores = grad_scale * dloss/dscores
 * (pass grad_scale = 1 for plain backward, 1/world_size to pre-average for data parallel). */
int nrl_ce_loss_fwd_bwd(const float* scores, const float* y_true, int64_t batch, int64_t n_cand,
                        float grad_scale, float* loss, float* d_scores, void* stream);

/* ---- loss: SupConLoss over the score matrix, losses.py:6-40 as called at nrms_module.py:289-318 (on
 * pytorch-metric-learning 2.2.0: GenericPairLoss.mat_based_loss, AvgNonZeroReducer).  Row b: positives = slots
 * with y_true != 0, negatives = its other real candidates (slot < cand_sizes[b]); x = scores / temperature;
 * loss_b = -mean_p(x_p - logsumexp over the row's real candidates); loss = mean of the loss_b > 0 (0 when the
 * batch holds no positive or no negative pair).  Also writes d_scores = grad_scale * dloss/dscores (may be NULL). */
int nrl_supcon_loss_fwd_bwd(const float* scores, const float* y_true, const int64_t* cand_sizes, int64_t batch,
                            int64_t n_cand, float temperature, float grad_scale, float* loss, float* d_scores,
                            void* stream);

/* ---- optimizer: torch.optim.Adam(lr) dense step over a flat buffer, configs/model/nrms.yaml:49-52,
 * abstract_recommender.py:96.  `step` is the 1-based step count.  grad_scale multiplies g first
 * (1/world_size after a sum all-reduce).  zero_grad != 0 clears g after use. */
int nrl_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step, float grad_scale,
                  int32_t zero_grad, void* stream);

/* ---- optimizer, lazy form for the embedding table (ABI v13).  Same arithmetic, same bits as nrl_adam_step: a row whose
 * gradient is zero evolves by a deterministic fp32 recurrence in (p, m, v, step), so its update may be applied later -- before
 * the row is next gathered, when it next receives a gradient, or when the state is exported.  `last_step[rows]` (int32, zeros
 * at step 0) records how far each row has been advanced; `mark[rows]` (int32) names the rows a step touches; `status[1]` is set
 * to 1 if a row was found more than 127 steps behind (the window of bias corrections a call carries) -- the caller's rolling
 * flush must keep every lag below that.
 *   nrl_adam_rows_mark     mark[id] = step for every id of the step's batch (ids outside [0, rows) are ignored)
 *   nrl_adam_rows_advance  candidate rows r = offset + j * stride (< rows); with `mark`: only those with mark[r] == upto_step + 1.
 *                          Each is advanced from last_step[r] to upto_step with zero gradients; with_grad != 0: then step
 *                          upto_step + 1 is applied with its gradient row (times grad_scale), the gradient row is cleared and
 *                          last_step[r] = upto_step + 1.  (catch-up before a forward: mark, upto_step = t - 1, with_grad 0;
 *                          update after the backward: mark, upto_step = t - 1, with_grad 1; flush: mark NULL, with_grad 0;
 *                          with_grad 2, mark NULL: "scan" -- every candidate row whose gradient row has a non-zero element is
 *                          updated, all-zero rows stay lazy: the update after a DENSE all-reduce, which leaves no list of rows.)
 *                          exclude_mark (ABI v14; NULL: none): rows with exclude_mark[r] == exclude_tag are skipped -- the EARLY
 *                          catch-up of step t + 1 (mark = the next batch's marks, upto_step = t, with_grad 0) issued on another
 *                          stream while step t runs must leave alone the rows step t owns (exclude_mark = step t's marks,
 *                          exclude_tag = t).  One rank only: a row outside step t's batch has a zero gradient at step t. */
int nrl_adam_rows_mark(const int64_t* ids, int64_t n_ids, int64_t rows, int32_t* mark, int64_t step, void* stream);
int nrl_adam_rows_advance(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t rows, int32_t dim,
                          int32_t* last_step, const int32_t* mark, int32_t* status, int64_t stride, int64_t offset,
                          int64_t upto_step, int32_t with_grad, double lr, double beta1, double beta2, double eps,
                          float grad_scale, const int32_t* exclude_mark, int32_t exclude_tag, void* stream);

/* ---- transformer-body glue (ABI v13; config 4, text.py:89-109: every layer of the PLM body ends its attention and its
 * feed-forward block with LayerNorm(dropout(dense_out) + residual)).  One launch each way instead of dropout + add + layer norm:
 *   fwd: z = x * keep / (1 - p) + residual;  y = (z - mean) * rstd * gamma + beta.  z_save / mean_save / rstd_save: all three
 *        (training) or all NULL (evaluation).  The keep mask is the library's counter-based one over the element index
 *        row * dim + col (seed, stream0), so the backward re-evaluates it instead of reading a stored mask.
 *   bwd: d_residual = LayerNorm backward of d_y; d_x = d_residual * keep / (1 - p) (d_x may be NULL when p_drop == 0: the two
 *        are equal); d_gamma / d_beta (both or neither; NULL for a frozen LayerNorm) are ACCUMULATED.
 * dim % 4 == 0, dim <= 2048, rows * dim < 2^32. */
int nrl_dropout_add_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, int64_t rows,
                                  int32_t dim, float eps, double p_drop, uint64_t seed, uint32_t stream0, float* z_save,
                                  float* mean_save, float* rstd_save, float* y, void* stream);
int nrl_dropout_add_layernorm_bwd(const float* d_y, const float* z_saved, const float* gamma, const float* mean_saved,
                                  const float* rstd_saved, int64_t rows, int32_t dim, double p_drop, uint64_t seed,
                                  uint32_t stream0, float* d_x, float* d_residual, float* d_gamma, float* d_beta, void* stream);

/* =================================================================================================
 * LSTUR path (BASELINE config 5): CNN text encoder, row-masked embedding
 * lookups (category / long-term user vector) and the GRU user encoder.
 * ================================================================================================= */

/* CNN + additive-attention text encoder.  Field <-> reference state_dict key (prefix
 * `news_encoder.text_encoders.<attr>.`, ONE module shared by title and abstract, news.py:69-79):
 *   conv_weight  cnn.weight                        (F, 1, W, D) contiguous = (F, W*D), k = t*D + d
 *   conv_bias    cnn.bias                          (F)
 *   att_weight   additive_attention.linear.weight  (Q, F)
 *   att_bias     additive_attention.linear.bias    (Q)
 *   att_query    additive_attention.query          (Q)
 * Replaces nn.Conv2d + F.relu + AdditiveAttention as wired in text.py:146-176 (CNNAddAtt). */
typedef struct NrlCnnParams {
  const float* conv_weight;
  const float* conv_bias;
  const float* att_weight;
  const float* att_bias;
  const float* att_query;
  int32_t embed_dim;   /* D, multiple of 4 */
  int32_t num_filters; /* F, multiple of 4 */
  int32_t window;      /* W, odd, <= 7 (padding (W-1)/2 as text.py:152) */
  int32_t query_dim;   /* Q, multiple of 4 */
} NrlCnnParams;

typedef struct NrlCnnGrads { /* accumulators, kernels ADD */
  float* conv_weight;
  float* conv_bias;
  float* att_weight;
  float* att_bias;
  float* att_query;
} NrlCnnGrads;

/* CNNAddAtt.forward, text.py:163-176: ids (N, L) int64 -> out (N, F).
 * x = dropout(emb[ids]) (stream0, flat index over (N, L, D)); c = dropout(relu(conv(x))) (stream0 + 1,
 * flat index over (N, L, F)); out = additive attention over the L tokens.  The convolution runs as
 * ONE GEMM with K = W*D over overlapping rows of x (no im2col buffer).
 * save_for_backward != 0: the workspace then holds everything nrl_cnn_encoder_bwd reads (including, under the
 * bf16x3 engine at F % 16 == 12, F <= 304, Q <= 224, the (hi, lo) planes of c its additive-attention weight gradient is
 * fed from); a backward after a forward with save_for_backward == 0 is undefined. */
size_t nrl_cnn_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim,
                                       int32_t num_filters, int32_t window, int32_t query_dim);
int nrl_cnn_encoder_fwd(const NrlCnnParams* p, const float* emb_table, int64_t vocab,
                        const int64_t* ids, int64_t n_news, int32_t seq_len, double p_drop,
                        uint64_t seed, uint32_t stream0, int32_t save_for_backward, float* out,
                        void* ws, size_t ws_bytes, void* stream);
/* Backward (autograd of text.py:163-176 incl. embedding_dense_backward, padding_idx = 0).  d_out (N, F).
 * Adds into `g` and d_emb_table.  sorted_positions as in nrl_news_encoder_bwd (may be NULL; N*L + 1 entries: the
 * count of id-0 positions in the last one -- the convolution's activation gradient runs over the live positions only). */
int nrl_cnn_encoder_bwd(const NrlCnnParams* p, const NrlCnnGrads* g, float* d_emb_table,
                        int64_t vocab, const int64_t* ids, const int64_t* sorted_positions,
                        int64_t n_news, int32_t seq_len, double p_drop, uint64_t seed,
                        uint32_t stream0, const float* d_out, void* ws, size_t ws_bytes,
                        void* stream);

/* CNNMHSAAddAtt.forward, text.py:291-309 (CenNewsRec): ids (N, L) -> out (N, F).
 *   x = dropout(emb[ids]) (stream0); c = dropout(relu(conv1d(x))) (stream0 + 1, padding (W-1)/2);
 *   self-attention over the L tokens of each news on the F conv features, dropout (stream0 + 2), additive
 *   attention.  `cp` supplies conv_weight / conv_bias and the dims (its att_* fields are ignored); conv_weight
 *   in the (F, W*D) layout of NrlCnnParams (k = t*D + d; nn.Conv1d stores (F, D, W): permute on the host);
 *   `bp` is the attention block on embed_dim = F. */
size_t nrl_cnn_mhsa_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim,
                                            int32_t num_filters, int32_t window, int32_t num_heads,
                                            int32_t query_dim);
int nrl_cnn_mhsa_encoder_fwd(const NrlCnnParams* cp, const NrlBlockParams* bp, const float* emb_table,
                             int64_t vocab, const int64_t* ids, int64_t n_news, int32_t seq_len,
                             double p_drop, uint64_t seed, uint32_t stream0, int32_t save_for_backward,
                             float* out, void* ws, size_t ws_bytes, void* stream);
int nrl_cnn_mhsa_encoder_bwd(const NrlCnnParams* cp, const NrlCnnGrads* cg, const NrlBlockParams* bp,
                             const NrlBlockGrads* bg, float* d_emb_table, int64_t vocab,
                             const int64_t* ids, const int64_t* sorted_positions, int64_t n_news,
                             int32_t seq_len, double p_drop, uint64_t seed, uint32_t stream0,
                             const float* d_out, void* ws, size_t ws_bytes, void* stream);

/* nn.Embedding(padding_idx=0) lookup followed by a ROW mask: out[i] = table[ids[i]] * m(i),
 * m(i) in {0, 1/(1-p)} from dropout stream `stream_id` with flat index i (whole rows dropped).
 * p_row = 0: plain lookup = LinearEncoder.forward, category.py:72-73.  p_row > 0: the long-term user
 * vector of LSTUR, user/lstur.py:70-71 (nn.Dropout2d on a (1, B, D) tensor drops whole users).
 * _bwd adds d_out[i] * m(i) into d_table[ids[i]], skipping id 0 (padding_idx). */
int nrl_embedding_rows_fwd(const float* table, const int64_t* ids, int64_t n_ids, int32_t dim,
                           double p_row, uint64_t seed, uint32_t stream_id, float* out,
                           void* stream);
int nrl_embedding_rows_bwd(const float* d_out, const int64_t* ids, int64_t n_ids, int32_t dim,
                           double p_row, uint64_t seed, uint32_t stream_id, float* d_table,
                           void* stream);

/* Single-layer nn.GRU.  Field <-> reference state_dict key (prefix `user_encoder.gru.`):
 *   weight_ih  weight_ih_l0 (3*Hd, Din) rows [W_ir; W_iz; W_in]      bias_ih  bias_ih_l0 (3*Hd)
 *   weight_hh  weight_hh_l0 (3*Hd, Hd)  rows [W_hr; W_hz; W_hn]      bias_hh  bias_hh_l0 (3*Hd) */
typedef struct NrlGruParams {
  const float* weight_ih;
  const float* weight_hh;
  const float* bias_ih;
  const float* bias_hh;
  int32_t input_dim;  /* Din, multiple of 4 */
  int32_t hidden_dim; /* Hd, multiple of 4 */
} NrlGruParams;

typedef struct NrlGruGrads { /* accumulators, kernels ADD */
  float* weight_ih;
  float* weight_hh;
  float* bias_ih;
  float* bias_hh;
} NrlGruGrads;

/* gru(pack_padded_sequence(hist, lengths, batch_first=True, enforce_sorted=False), h0)[1], i.e. the
 * hidden state of every sequence after its own last valid step: user/lstur.py:74-83.
 * hist (B, T, Din), lengths (B) int64 in [1, T] (validated by the caller, as pack_padded_sequence
 * does on the host), h0 (B, Hd) or NULL for zeros ("con" variant, lstur.py:85) -> out (B, Hd).
 * The input projection of all steps is one GEMM; the recurrence runs one small GEMM + one gate kernel
 * per step, t = 0 .. T-1, rows with t >= lengths[b] frozen. */
size_t nrl_gru_workspace_bytes(int64_t batch, int64_t max_len, int32_t input_dim, int32_t hidden_dim);
int nrl_gru_fwd(const NrlGruParams* p, const float* hist, const int64_t* lengths, const float* h0,
                int64_t batch, int64_t max_len, int32_t save_for_backward, float* out, void* ws,
                size_t ws_bytes, void* stream);
/* Backward through time.  d_out (B, Hd) -> d_hist (B, T, Din) overwritten, d_h0 (B, Hd) overwritten
 * (may be NULL); adds into `g`. */
int nrl_gru_bwd(const NrlGruParams* p, const NrlGruGrads* g, const float* hist,
                const int64_t* lengths, const float* h0, int64_t batch, int64_t max_len,
                const float* d_out, float* d_hist, float* d_h0, void* ws, size_t ws_bytes,
                void* stream);

/* =================================================================================================
 * Generic blocks of the sibling recommenders (NAML: news-view combination, user encoder, category encoder).
 * ================================================================================================= */

/* AdditiveAttention.forward, layers/attention.py:24-42: y (G, S, D) -> out (G, D),
 *   t = tanh(y W_a^T + b_a); w = softmax_S(t . q_a); out = sum_s w_s y_s   (no mask).
 * Field <-> state_dict key: att_weight linear.weight (Q, D), att_bias linear.bias (Q), att_query query (Q). */
typedef struct NrlAddAttParams {
  const float* att_weight;
  const float* att_bias;
  const float* att_query;
  int32_t dim;       /* D, multiple of 4 */
  int32_t query_dim; /* Q, multiple of 4 */
} NrlAddAttParams;

typedef struct NrlAddAttGrads { /* accumulators, kernels ADD */
  float* att_weight;
  float* att_bias;
  float* att_query;
} NrlAddAttGrads;

size_t nrl_additive_attention_workspace_bytes(int64_t groups, int64_t len, int32_t dim, int32_t query_dim);
int nrl_additive_attention_fwd(const NrlAddAttParams* p, const float* y, int64_t groups, int64_t len,
                               int32_t save_for_backward, float* out, void* ws, size_t ws_bytes,
                               void* stream);
/* d_out (G, D) -> d_y (G, S, D) overwritten; adds into `g`.  `ws` must be the forward's workspace. */
int nrl_additive_attention_bwd(const NrlAddAttParams* p, const NrlAddAttGrads* g, const float* y,
                               int64_t groups, int64_t len, const float* d_out, float* d_y, void* ws,
                               size_t ws_bytes, void* stream);

/* act(A W^T + bias): nn.Linear followed by an activation (0 none, 1 tanh, 2 relu), e.g. the category
 * encoder's F.relu(self.linear(x)), category.py:78-80.  a (M, K), w (N, K), c (M, N); N, K multiples of 4.
 * _bwd: d_c (M, N) -> d_a (M, K) overwritten (may be NULL), d_w / d_bias ADDED; c = the forward output. */
size_t nrl_linear_act_workspace_bytes(int64_t m, int32_t n, int32_t k);
int nrl_linear_act_fwd(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k,
                       int32_t act, float* c, void* ws, size_t ws_bytes, void* stream);
int nrl_linear_act_bwd(const float* a, const float* w, const float* c, const float* d_c, int64_t m,
                       int32_t n, int32_t k, int32_t act, float* d_a, float* d_w, float* d_bias,
                       void* ws, size_t ws_bytes, void* stream);

/* nn.MultiheadAttention(x, x, x) with batch_first=False on its own (no pooling): x (seq, batch, D) -> out
 * (seq, batch, D), attention over the `seq` axis for every (batch column, head).  Replaces the call at
 * user/mins.py:55-57 (which feeds (B, H, D), so seq = users, batch = history slots -- the NRMS quirk).
 * `scale` multiplies q before q k^T; 0 selects torch's 1/sqrt(D / num_heads).  A caller that zero-pads the
 * heads to a supported head dim (MINS: 50 -> 64) passes the un-padded 1/sqrt(50) here. */
typedef struct NrlMhaParams {
  const float* in_proj_weight;  /* (3D, D) rows [q; k; v] */
  const float* in_proj_bias;    /* (3D) */
  const float* out_proj_weight; /* (D, D) */
  const float* out_proj_bias;   /* (D) */
  int32_t embed_dim;            /* D, multiple of 4; D / num_heads in {16, 20, 32, 48, 64} */
  int32_t num_heads;
  float scale;
  int32_t gemm_engine;          /* as NrlBlockParams.gemm_engine */
} NrlMhaParams;

typedef struct NrlMhaGrads {
  float* in_proj_weight;
  float* in_proj_bias;
  float* out_proj_weight;
  float* out_proj_bias;
} NrlMhaGrads;

size_t nrl_mha_workspace_bytes(int64_t seq, int64_t batch, int32_t embed_dim, int32_t num_heads);
int nrl_mha_fwd(const NrlMhaParams* p, const float* x, int64_t seq, int64_t batch, int32_t save_for_backward,
                float* out, void* ws, size_t ws_bytes, void* stream);
/* d_out (seq, batch, D) -> d_x overwritten; adds into `g`.  `ws` must be the forward's workspace. */
int nrl_mha_bwd(const NrlMhaParams* p, const NrlMhaGrads* g, const float* x, int64_t seq, int64_t batch,
                const float* d_out, float* d_x, void* ws, size_t ws_bytes, void* stream);

/* ---- building blocks exported for unit parity tests and reuse ------------------------------- */
/* nn.Embedding lookup alone (bit-exact), text.py:224. */
int nrl_embedding_gather(const float* table, const int64_t* ids, int64_t n_ids, int32_t dim,
                         float* out, void* stream);
/* ABI v14: nn.Linear fused with the exact GELU of a BERT-family feed-forward block (HF RobertaIntermediate / RobertaOutput inside
 * self.plm_model(**text), text.py:89).  bf16x3 engine, wide side >= 256 (nrl_linear_gelu_supported(n_wide) != 0); workspace and
 * image_ready as nrl_linear_fwd_img / _bwd_img (nrl_linear_workspace_bytes(n, k)).
 *   nrl_linear_gelu_fwd_img    h (m, n) = a W^T + bias (the GELU's input, kept for the backward), g (m, n) = gelu(h) = h Phi(h)
 *   nrl_linear_dgrad_gelu_img  for the projection that CONSUMED g -- c = g W^T (+ bias), W (n, k), g (m, k):
 *                              d_pre (m, k) = (d_c W) * gelu'(pre), i.e. the gradient at the GELU's input straight from the epilogue */
int nrl_linear_gelu_fwd_img(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k, float* h, float* g,
                            void* ws, size_t ws_bytes, int32_t image_ready, void* stream);
int nrl_linear_dgrad_gelu_img(const float* w, const float* d_c, const float* pre, int64_t m, int32_t n, int32_t k, float* d_pre,
                              void* ws, size_t ws_bytes, int32_t image_ready, void* stream);
int32_t nrl_linear_gelu_supported(int32_t n_wide);
/* ABI v15: the query / key / value projections of a transformer layer (HF RobertaSelfAttention.query / .key / .value inside
 * self.plm_model(**text), text.py:89) -- three nn.Linear (n, k) of ONE input -- as one GEMM each way, and activation gradients that
 * land on a residual stream with the other branch's gradient added in the epilogue.  bf16x3 engine; n and k multiples of 256 with
 * at most 12 column panels (nrl_linear3_supported(n, k) != 0); workspace nrl_linear3_workspace_bytes(n, k), image_ready as in
 * nrl_linear_fwd_img (one image of the three weights per direction).
 *   nrl_linear3_fwd_img       c (3, m, n): c[q] = a (m, k) W_q^T + b_q
 *   nrl_linear3_dgrad_img     d_a (m, k) = add + sum_q d_c[q] W_q, d_c (3, m, n) stacked like c; add (m, k) or NULL
 *   nrl_linear_dgrad_add_img  d_a (m, k) = add + d_c (m, n) W (n, k) (k >= 256; workspace nrl_linear_workspace_bytes(n, k))
 * The weight gradients stay three nrl_linear_bwd_img calls (d_a == NULL). */
int32_t nrl_linear3_supported(int32_t n, int32_t k);
size_t nrl_linear3_workspace_bytes(int32_t n, int32_t k);
int nrl_linear3_fwd_img(const float* a, const float* w0, const float* w1, const float* w2, const float* b0, const float* b1,
                        const float* b2, int64_t m, int32_t n, int32_t k, float* c, void* ws, size_t ws_bytes, int32_t image_ready,
                        void* stream);
int nrl_linear3_dgrad_img(const float* d_c, const float* w0, const float* w1, const float* w2, int64_t m, int32_t n, int32_t k,
                          const float* add, float* d_a, void* ws, size_t ws_bytes, int32_t image_ready, void* stream);
int nrl_linear_dgrad_add_img(const float* d_c, const float* w, int64_t m, int32_t n, int32_t k, const float* add, float* d_a,
                             void* ws, size_t ws_bytes, int32_t image_ready, void* stream);
/* ABI v14: embedding_dense_backward of ANY nn.Embedding (the word / position / token-type tables of the PLM body, text.py:89):
 * d_table[ids[p]] += d_out[p] for the n_ids positions in the id-sorted order `sorted_positions` (nrl_sort_positions: n_ids + 1
 * entries, the last one unused here), one atomic per (id, 64-position segment, element) instead of one per element; the row
 * `padding_idx` (< 0: none) receives nothing (nn.Embedding(padding_idx=...)).  dim <= 1024. */
int nrl_embedding_grad(const float* d_out, const int64_t* ids, const int64_t* sorted_positions, int64_t n_ids, int32_t dim,
                       int64_t padding_idx, float* d_table, void* stream);
/* C(M,N) = A(M,K) * W(N,K)^T + bias  (nn.Linear) through the selected GEMM engine -- the same code path
 * (tile choice, LDS-DMA staging) the encoders' forward projections take.  The bf16x3 engine needs
 * nrl_linear_workspace_bytes(n, k) of workspace for the split weight planes; ws == NULL forces the exact
 * fp32 engine. */
size_t nrl_linear_workspace_bytes(int32_t n, int32_t k);
int nrl_linear_fwd(const float* a, const float* w, const float* bias, int64_t m, int32_t n,
                   int32_t k, float* c, void* ws, size_t ws_bytes, void* stream);
/* Backward of nrl_linear_fwd (nn.Linear inside a third-party stack -- the PLM body of text.py:89-109 -- on this library's
 * matrix-core engines): d_a (m, k) = d_c (m, n) W; d_w (n, k) += d_c^T a; d_bias (n) += colsum(d_c).  d_a == NULL skips
 * the activation gradient; d_w == d_bias == NULL skips the weight gradient (a frozen layer whose input still needs a
 * gradient).  n, k multiples of 4; workspace as nrl_linear_workspace_bytes(n, k) (bf16x3 engine, d_a != NULL). */
int nrl_linear_bwd(const float* a, const float* w, const float* d_c, int64_t m, int32_t n, int32_t k, float* d_a,
                   float* d_w, float* d_bias, void* ws, size_t ws_bytes, void* stream);
/* The same two calls for a weight whose matrix-core images the caller keeps across calls (ABI v12): with image_ready != 0,
 * `ws` already holds the images a previous call of the SAME entry point built for this (w, n, k) under the same engine and
 * switches -- the forward's in one buffer, the backward's (the transposed weight, for d_a) in another -- and the build is
 * skipped.  For FROZEN weights only (the PLM body's layers 0-7, text.py:69-73): nothing here can tell that `w` changed.
 * Applies to the wide (n >= 256, resp. k >= 256) bf16x3 panel path; elsewhere the flag is ignored and the images rebuilt. */
int nrl_linear_fwd_img(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k,
                       float* c, void* ws, size_t ws_bytes, int32_t image_ready, void* stream);
int nrl_linear_bwd_img(const float* a, const float* w, const float* d_c, int64_t m, int32_t n, int32_t k, float* d_a,
                       float* d_w, float* d_bias, void* ws, size_t ws_bytes, int32_t image_ready, void* stream);

/* ---- scaled dot-product attention of a transformer body: the self-attention inside `self.plm_model(**text)`, text.py:89
 * (HF RobertaSelfAttention -> F.scaled_dot_product_attention), ABI v12.  q, k, v, out (and their gradients): (n_batch, seq_len,
 * num_heads * head_dim) fp32, contiguous -- the projections' outputs as they are, no head transpose; key_keep (n_batch, seq_len)
 * uint8, 1 = the key takes part (the tokenizer's attention_mask), or NULL; `scale` multiplies q k^T; attention-probability
 * dropout with probability p_drop under the counter-based mask spec of nrl_dropout_mask: element (batch b, head h, query i,
 * key j) has flat index ((b * num_heads + h) * 128 + i) * 128 + j (uint32 arithmetic) in stream `stream0` of `seed`.
 * lse (n_batch * num_heads, seq_len): log-sum-exp of each query's scores, written by _fwd (may be NULL in inference), read by
 * _bwd.  bf16x3 arithmetic (hi, lo split operands on the bf16 matrix cores); seq_len <= 128, head_dim == 64
 * (nrl_sdpa_supported); anything else stays on the framework's attention. */
int32_t nrl_sdpa_supported(int64_t n_batch, int32_t seq_len, int32_t num_heads, int32_t head_dim);
int nrl_sdpa_fwd(const float* q, const float* k, const float* v, const uint8_t* key_keep, int64_t n_batch, int32_t seq_len,
                 int32_t num_heads, int32_t head_dim, float scale, double p_drop, uint64_t seed, uint32_t stream0, float* out,
                 float* lse, void* stream);
int nrl_sdpa_bwd(const float* q, const float* k, const float* v, const uint8_t* key_keep, const float* out, const float* d_out,
                 const float* lse, int64_t n_batch, int32_t seq_len, int32_t num_heads, int32_t head_dim, float scale,
                 double p_drop, uint64_t seed, uint32_t stream0, float* dq, float* dk, float* dv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEWSRECLIB_AMD_H */
