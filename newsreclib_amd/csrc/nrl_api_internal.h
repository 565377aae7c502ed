// Internals shared by the translation units of the C ABI (nrl_api.hip: news / user encoders, scorer, loss, Adam;
// nrl_api_lstur.hip: CNN encoders, row embeddings, GRU; nrl_api_blocks.hip: standalone additive attention, linear, MHA):
// engine / option state (ONE instance, defined in nrl_api.hip), tile choices, the GEMM dispatch helpers and the shared
// "MHSA + additive attention" block.  Everything here is static / template: each unit instantiates what it uses.
#pragma once
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <utility>
#include <vector>

#include "nrl_gemm.h"
#include "nrl_gemm_bf16x3.h"
#include "nrl_gemm_bf16x3_dma.h"
#include "nrl_rowpanel.h"
#include "nrl_gemm_ws.h"
#include "nrl_news_fused.h"
#include "nrl_news_tail_api.h"
#include "nrl_user_tail.h"
#include "nrl_wgrad_planes.h"
#include "nrl_kernels.h"
#include "nrl_conv.h"
#include "nrl_gru_fused.h"

namespace nrl {

// ---- optional HIP-event timing of the dominant kernel (bench.py's roofline line) ----------------
// Events are recorded on the launch stream right around the in-projection GEMM launch (no sync);
// nrl_prof_read() synchronises the recorded events and sums their elapsed times.
struct ProfState {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  double total_ms = 0.0, total_flops = 0.0;
  int64_t launches = 0;
};
extern ProfState g_prof;                 // (nrl_api.hip)

struct ProfScope {
  hipStream_t st;
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  bool active;
  ProfScope(hipStream_t s, double flops) : st(s), active(g_prof.on && flops > 0.0) {
    if (!active) return;
    if (!g_prof.pool.empty()) {
      ev = g_prof.pool.back();
      g_prof.pool.pop_back();
    } else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) {
      active = false;
      return;
    }
    g_prof.total_flops += flops;
    g_prof.launches += 1;
    (void)hipEventRecord(ev.first, st);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(ev.second, st);
    g_prof.pending.push_back(ev);
  }
};

// Tile shapes (WM, WN, TM, TN), picked per GEMM from tools/gemm_probe.hip measurements at the
// B=128 shapes (profiles/r01_gemm_probe.txt):
//   NRL_TILE   8 waves, 128 x 160: N = 300 -> 2 column tiles (320), N = 900 -> 6 (960)
//   NRL_TILE_Q 8 waves, 128 x 208: the additive-attention projection, N = Q = 200 in one tile
//   NRL_TILE_W 4 waves,  64 x 160: the small weight-gradient outputs (300 x 301, 200 x 301)
#define NRL_TILE 4, 2, 2, 5
#define NRL_TILE_Q 8, 1, 1, 13
#define NRL_TILE_W 2, 2, 2, 5

// bf16x3 engine (nrl_gemm_bf16x3.h): with the MFMA time cut ~5x the GEMMs are staging/epilogue
// bound, so the big forward/dgrad GEMMs take 256-row tiles (profiles/r01_gemm_bf16x3_probe.txt)
// (last number: prefetch depth flag DEEP -- two staging register sets only where they fit without spills)
#define X3_TILE_BIG 4, 2, 4, 5, 0   // 256 x 160
#define X3_TILE 4, 2, 2, 5, 1       // 128 x 160 (small M)
#define X3_TILE_Q 4, 2, 2, 7, 1     // 128 x 224 (Q = 200 in one tile)
#define X3_TILE_W 2, 2, 2, 5, 0     //  64 x 160 weight gradients
// LDS-DMA staged variant (nrl_gemm_bf16x3_dma.h; WM, WN, TM, TN, ring depth) for plain / windowed fp32 A:
// 4-wave 128 x 160 workgroups with a 2-deep ring = 72 KB LDS, so TWO workgroups share a CU and their
// split / MFMA phases interleave (profiles/r01_gemm_x3_dma_probe.txt: dgrad N=300 K=900 0.60 -> 0.45 ms)
#define X3_DMA_TILE 2, 2, 4, 5, 2
#define X3_DMA_TILE_Q 4, 2, 2, 7, 2  // 128 x 224, 8 waves: Q = 200 in one column tile (0.32 -> 0.24 ms)

enum { ENGINE_F32 = 0, ENGINE_BF16X3 = 1 };
// Engine selection.  The per-call choice (NrlBlockParams.gemm_engine / NrlMhaParams.gemm_engine: 1 = f32,
// 2 = bf16x3) wins; 0 means "the process default" (nrl_set_gemm_engine).  The value a call runs under is
// fixed at its entry (EngineScope) and kept thread-local, so a concurrent nrl_set_gemm_engine on another
// thread cannot change it halfway through a call.
extern std::atomic<int> g_default_engine;   // (defined in nrl_api.hip: one instance for the three ABI units)
extern thread_local int t_engine;
static inline int cur_engine() { return t_engine >= 0 ? t_engine : g_default_engine.load(std::memory_order_relaxed); }
struct EngineScope {
  int prev;
  explicit EngineScope(int per_call) : prev(t_engine) {
    t_engine = per_call > 0 ? per_call - 1 : g_default_engine.load(std::memory_order_relaxed);
  }
  ~EngineScope() { t_engine = prev; }
};
static inline bool engine_field_ok(int v) { return v >= 0 && v <= 2; }
// ---- kernel-selection switches ------------------------------------------------------------------------------------
// They choose between kernels that compute the same thing (A/B measurements, equivalence tests) and with it the PRIVATE
// formats of a call's workspace, so a backward must run under the switches of its forward.  The value a call runs under
// is fixed at its entry (OptScope) and thread-local: NrlBlockParams.options carries it per call (forward and backward of
// one module get the same word, two modules in one process may differ); options == 0 means "the process defaults", which
// start from the NRL_* environment and change through nrl_set_option.
//   news_fused      NRL_NEWS_FUSED=0       gather + in-projection + token attention as separate kernels (nrl_news_fused.h)
//   news_fused_bwd  (RETIRED in ABI v14: the bit is reserved and always 0)  q|k|v recomputed inside the matrix-core attention
//                                          backward: 1.69 ms against 0.82; the kernel lives in tools/experimental/nrl_news_fused_bwd.h
//   news_attn_mfma  NRL_NEWS_ATTN_MFMA=0   q|k|v saved as packed rows, attn_bwd_small (fp32 VALU) instead of the slabs +
//                                          news_attn_bwd_kernel
//   news_planes     NRL_NEWS_PLANES=0      x / dqkv stay fp32 (no fragment-block planes, nrl_wgrad_planes.h)
//   news_od_planes  NRL_NEWS_OD_PLANES=0   o / dy stay fp32 rows
//   news_aa_planes  NRL_NEWS_AA_PLANES=0   the additive-attention GEMMs read fp32 y / d_pre
//   wgrad_2step     NRL_WGRAD_2STEP=0      split-K partial tiles added with atomics instead of stored + reduced
//   wgrad_ws        NRL_WGRAD_WS=0         the large weight gradient on the register-staged kernel (nrl_gemm_ws.h otherwise)
//   rowpanel        NRL_ROWPANEL=0         the narrow (N <= 320) projections on the tiled kernels
//   x3_dma          NRL_X3_DMA=0           every bf16x3 GEMM on the register-staged kernel
//   news_tail       NRL_NEWS_TAIL=0        out-projection / additive attention / pooling of the fused news path on the
//                                          row-panel GEMMs + pool_fwd instead of ONE kernel (nrl_news_tail.h)
//   news_tail_bwd   NRL_NEWS_TAIL_BWD=0    additive-attention backward on pool_bwd_pre + the row-panel GEMM (the forward tail
//                                          then also saves the tanh output) instead of ONE kernel that recomputes tanh
//   user_fork       NRL_USER_FORK=1        (default OFF) the user encoder's in-projection dgrad and its three weight gradients
//                                          side by side on two internal streams instead of one after the other -- measured
//                                          SLOWER on one box (3.36-3.40 vs 3.31 ms per step at B = 128): the cross-stream
//                                          event waits cost more than the ~75 us of small launches they overlap
//   news_fork       NRL_NEWS_FORK=0        (default ON since round 6) the two back-half weight gradients of the fused news path (additive attention,
//                                          out-projection) on an internal side stream beside the HBM-write-bound chain
//                                          out-projection dgrad -> token-attention backward -> in-projection dgrad -> table
//                                          gradient of phase 1, joined before the call returns
//   news_qkv_planes NRL_NEWS_QKV_PLANES=0  token-attention backward builds every operand fragment from its fp32 image
//                                          (news_attn_bwd_kernel) instead of splitting q|k|v / d_o once into LDS planes
//                                          and reading fragments (news_attn_bwd_p_kernel; bit-identical output)
//   news_pad_share  NRL_NEWS_PAD_SHARE=0   evaluation forward of the fused news path computes every token row of every news,
//                                          instead of ONE row for the run of padding tokens from token 15 on (no dropout:
//                                          identical rows; news_classify_kernel + the SHARE shapes of news_fused_fwd_kernel /
//                                          news_tail_fwd_kernel; bit-identical output)
//   news_tail_od    (RETIRED in ABI v14: the bit is reserved and always 0)  the out-projection's activation gradient as phase D of
//                                          the fused tail backward: 337 -> 488 us for the kernel against the 135 us launch it replaced;
//                                          tools/experimental/nrl_news_tail_phase_d.inc
//   user_proj       NRL_USER_PROJ=0        the user encoder's in-projection as its own tiled GEMM launch (+ the weight split in
//                                          front of it) instead of inside the across-users attention kernel (ua_fwd_proj_kernel)
enum {
  O_NEWS_FUSED = 0, O_NEWS_FUSED_BWD, O_NEWS_ATTN_MFMA, O_NEWS_PLANES, O_NEWS_OD_PLANES, O_NEWS_AA_PLANES, O_WGRAD_2STEP,
  O_WGRAD_WS, O_ROWPANEL, O_X3_DMA, O_NEWS_TAIL, O_NEWS_TAIL_BWD, O_USER_FORK, O_NEWS_FORK, O_NEWS_QKV_PLANES, O_NEWS_PAD_SHARE, O_NEWS_TAIL_OD, O_USER_PROJ, O_COUNT
};
// bits whose kernels were moved out of the library (ABI v14; tools/experimental/): kept in the mask's layout, always 0
static inline bool opt_retired(int bit) { return bit == O_NEWS_FUSED_BWD || bit == O_NEWS_TAIL_OD; }
extern std::atomic<uint32_t> g_opt_default;  // (nrl_api.hip)
extern thread_local int64_t t_opts;
static inline bool opt(int bit) {
  const uint32_t m = t_opts >= 0 ? (uint32_t)t_opts : g_opt_default.load(std::memory_order_relaxed);
  return (m >> bit) & 1u;
}
struct OptScope {
  int64_t prev;
  explicit OptScope(int32_t per_call) : prev(t_opts) {
    t_opts = (per_call & NRL_OPTIONS_EXPLICIT) ? (int64_t)(per_call & ((1 << O_COUNT) - 1))
                                               : (int64_t)g_opt_default.load(std::memory_order_relaxed);
  }
  ~OptScope() { t_opts = prev; }
};
static inline bool options_field_ok(int32_t v) {
  return v == 0 || ((v & NRL_OPTIONS_EXPLICIT) && (v & ~(NRL_OPTIONS_EXPLICIT | ((1 << O_COUNT) - 1))) == 0 &&
                    (v & ((1 << O_NEWS_FUSED_BWD) | (1 << O_NEWS_TAIL_OD))) == 0);
}

static int wgrad_splits(int64_t rows_out, int cols_out, int64_t K, int bm, int bn) {
  const int64_t tiles = ceil_div(rows_out, bm) * ceil_div(cols_out, bn);
  int64_t s = ceil_div(2048, tiles);
  const int64_t max_s = ceil_div(K, 8 * GEMM_BK);
  if (s > max_s) s = max_s;
  return (int)(s < 1 ? 1 : s);
}

struct BlockShape {
  int64_t M;       // rows entering the block (N*L for news, B*H for the user encoder)
  int D, Q, heads, dh;
  int64_t pool_groups;  // output rows
  int pool_len;         // rows per output row
  AttnGeom geom;
  int64_t pad_rows = 0; // news encoder: token rows padded to 32 per news (fragment-block planes), else 0
  bool aa_planes = false;  // fused news path: y also / d_pre only as planes (BlockWs::yp, tp) for the additive-attention GEMMs
  bool tail_bwd = false;   // ... and the additive-attention backward recomputes tanh in one kernel (no t buffer)
  bool tail = false;       // fused news path: the forward's back half ran as ONE kernel (nrl_news_tail.h): y exists only as planes
  bool od_planes = false;  // fused news path: `o` and `dy` are (hi, lo) bf16 fragment-block planes over the real rows (19
                           // block columns at D = 300; `o` in the head-permuted feature order), not fp32 rows
  bool forked = false;     // phase 1 ran (or, in a phase-2 call, has run) the back-half weight gradients on the side stream
};

struct BlockWs {
  float *x, *qkv, *o, *y, *t, *w, *lse, *dy, *dqkv, *d_o;
  float *yp, *tp;    // news path only: bf16 planes of y (19 block columns at D = 300) and of d_pre (13 at Q = 200), else null
  uint16_t* planes;  // bf16 hi/lo planes of the three weights (bf16x3 engine)
  uint16_t* rp;      // fragment-ordered weight images of the row-panel GEMMs (nrl_rowpanel.h)
};

// the five narrow projections of the block that run on the row-panel kernel: forward out-projection and
// additive-attention linear, and the three activation-gradient GEMMs
struct BlockRp {
  RpImage out_f, att_f, att_d, out_d, in_d, in_heads, in_d_hp, out_f_perm, tail_o, tail_a, tail_ad;
  bool on = false;
};
static bool block_rp_ok(int D, int Q) { return opt(O_ROWPANEL) && rp_nblk_supported(D) && rp_nblk_supported(Q); }
static size_t block_rp_elems(int D, int Q) {
  if (!block_rp_ok(D, Q)) return 0;
  const int nd = rp_nblk_for(D), nq = rp_nblk_for(Q);
  return rp_image_elems(nd, rp_kblocks(D, false)) * 2      // out fwd (N = D, K = D), out dgrad (N = D, K = D)
         + rp_image_elems(nq, rp_kblocks(D, false))         // att fwd (N = Q, K = D)
         + rp_image_elems(nd, rp_kblocks(Q, false))         // att dgrad (N = D, K = Q)
         + rp_image_elems(nd, rp_kblocks(3 * D, false))     // in dgrad (N = D, K = 3D)
         + rp_image_elems((D / 20) * 4, NF_KB)              // per-head q|k|v image of the fused news encoder
         + rp_image_elems(nd, (D / 20) * 2)                 // its in-projection dgrad over head planes (K' = heads * 64)
         + rp_image_elems(nd, rp_kblocks(D + 32, false))    // its out-projection forward over the head-permuted `o` planes
         + rp_image_elems(NT_FB, NT_KB) + rp_image_elems(NT_QB, NT_KS)    // the two images of the fused tail (nrl_news_tail.h)
         + rp_image_elems(NT_FB, NT_QS);                                   // ... and W_a^T for its backward
}

static size_t plane_elems(int D, int Q) {
  return split_weight_elems(3 * D, D) + split_weight_elems(D, D) + split_weight_elems(Q, D);
}

// q|k|v / dqkv: packed rows (M, 3D), or -- dh = 20, the fused news path -- 64 floats per (token, head): fp32 head-major
// slabs / head planes over the real rows, or (hi, lo) bf16 fragment-block planes over the padded rows
static size_t qkv_elems(int64_t M, int D, int heads, int64_t pad_rows) {
  const size_t packed = (size_t)M * 3 * D, slabs = (size_t)(pad_rows > M ? pad_rows : M) * heads * 64;
  return (D == heads * 20 && slabs > packed) ? slabs : packed;
}
// x: post-dropout rows (M, D), or their fragment-block planes: 20 column blocks x (hi, lo) x 2 bytes = 320 floats per padded row
static size_t x_elems(int64_t M, int D, int64_t pad_rows) {
  const size_t rows = (size_t)M * D, planes = (size_t)pad_rows * 320;
  return planes > rows ? planes : rows;
}

// o / dy: fp32 rows (M, D), or (news path) planes over the real rows rounded up to 32: 16 floats per (row, block column)
static size_t od_elems(int64_t M, int D, int heads, int64_t pad_rows) {
  const size_t rows = (size_t)M * D;
  if (pad_rows <= 0 || D != heads * 20) return rows;
  const size_t planes = (size_t)((M + 31) / 32 * 32) * (size_t)(heads + (heads + 3) / 4) * 16;
  return planes > rows ? planes : rows;
}

static size_t block_ws_floats(int64_t M, int D, int Q, int heads, bool with_x, int64_t pad_rows = 0) {
  auto al = [](size_t n) { return align_up(n, 64); };
  size_t n = 0;
  if (with_x) n += al(x_elems(M, D, pad_rows));
  n += al(qkv_elems(M, D, heads, pad_rows)) * 2;  // qkv, dqkv
  n += al(od_elems(M, D, heads, pad_rows)) * 2 + al((size_t)M * D) * 2;      // o, dy | y, d_o (later dx)
  if (pad_rows > 0 && D == heads * 20)                                       // y planes, d_pre planes
    n += al((size_t)((M + 31) / 32 * 32) * ((D + 16) / 16) * 16) + al((size_t)((M + 31) / 32 * 32) * ((Q + 15) / 16) * 16);
  n += al((size_t)M * Q);          // t / d_pre
  n += al((size_t)M);              // w
  n += al((size_t)M * heads);      // lse
  n += al((plane_elems(D, Q) + 1) / 2);  // bf16 weight planes
  n += al((block_rp_elems(D, Q) + 1) / 2);
  return n;
}

static int carve_ws(void* ws, size_t ws_bytes, const BlockShape& s, bool with_x, BlockWs* out) {
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < block_ws_floats(s.M, s.D, s.Q, s.heads, with_x, s.pad_rows) * sizeof(float)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes,
              block_ws_floats(s.M, s.D, s.Q, s.heads, with_x, s.pad_rows) * sizeof(float));
    return NRL_E_WORKSPACE;
  }
  float* p = (float*)ws;
  auto take = [&](size_t n) { float* r = p; p += align_up(n, 64); return r; };
  out->x = with_x ? take(x_elems(s.M, s.D, s.pad_rows)) : nullptr;
  out->qkv = take(qkv_elems(s.M, s.D, s.heads, s.pad_rows));
  out->dqkv = take(qkv_elems(s.M, s.D, s.heads, s.pad_rows));
  out->o = take(od_elems(s.M, s.D, s.heads, s.pad_rows));
  out->y = take((size_t)s.M * s.D);
  out->dy = take(od_elems(s.M, s.D, s.heads, s.pad_rows));
  out->yp = out->tp = nullptr;
  if (s.pad_rows > 0 && s.D == s.heads * 20) {
    out->yp = take((size_t)((s.M + 31) / 32 * 32) * ((s.D + 16) / 16) * 16);
    out->tp = take((size_t)((s.M + 31) / 32 * 32) * ((s.Q + 15) / 16) * 16);
  }
  out->d_o = take((size_t)s.M * s.D);
  out->t = take((size_t)s.M * s.Q);
  out->w = take((size_t)s.M);
  out->lse = take((size_t)s.M * s.heads);
  out->planes = reinterpret_cast<uint16_t*>(take((plane_elems(s.D, s.Q) + 1) / 2));
  out->rp = reinterpret_cast<uint16_t*>(take((block_rp_elems(s.D, s.Q) + 1) / 2));
  return NRL_OK;
}

static int check_params(const NrlBlockParams* p) {
  NRL_REQUIRE(p != nullptr, "params struct is null");
  NRL_REQUIRE(p->in_proj_weight && p->in_proj_bias && p->out_proj_weight && p->out_proj_bias &&
                  p->att_weight && p->att_bias && p->att_query, "null parameter pointer");
  NRL_REQUIRE(engine_field_ok(p->gemm_engine), "gemm_engine must be 0 (default), 1 (f32) or 2 (bf16x3)");
  NRL_REQUIRE(options_field_ok(p->options), "options must be 0 (process defaults) or NRL_OPTIONS_EXPLICIT | switch mask");
  NRL_REQUIRE(p->embed_dim > 0 && p->embed_dim % 4 == 0, "embed_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->query_dim > 0 && p->query_dim % 4 == 0, "query_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->num_heads > 0 && p->embed_dim % p->num_heads == 0, "embed_dim must be divisible by num_heads");
  NRL_REQUIRE(attn_head_dim_supported(p->embed_dim / p->num_heads),
              "head dim %d unsupported (16, 20, 32, 48, 64)", p->embed_dim / p->num_heads);
  NRL_REQUIRE((((uintptr_t)p->in_proj_weight | (uintptr_t)p->out_proj_weight | (uintptr_t)p->att_weight) & 15) == 0,
              "weight matrices must be 16-byte aligned");
  return NRL_OK;
}

struct BlockPlanes {
  SplitWeight in, out, att;
  BlockRp rp;
};

// carve (and, in the forward, fill) the bf16 planes of the block's three weights
static int block_planes(const NrlBlockParams* P, const BlockShape& s, const BlockWs& w, bool fill,
                        BlockPlanes* bp, hipStream_t st, int fused_heads = 0, int proj_heads = 0) {
  const int D = s.D, Q = s.Q;
  uint16_t* p = w.planes;
  const float* ws[3] = {P->in_proj_weight, P->out_proj_weight, P->att_weight};
  const int ns[3] = {3 * D, D, Q};
  SplitWeight* outs[3] = {&bp->in, &bp->out, &bp->att};
  // row-panel images (bf16x3 engine): carved always, built by the forward in ONE small launch
  bp->rp.on = block_rp_ok(D, Q) && cur_engine() == ENGINE_BF16X3;
  for (int i = 0; i < 3; ++i) {
    // planes nobody reads are not built: with the row-panel kernels on, the out-projection / additive-attention
    // planes; with the fused news encoder, the in-projection ones too (its dgrad runs on the row-panel image)
    // (proj_heads: the user encoder's in-projection inside its attention kernel, ua_fwd_proj_kernel -- per-head image instead)
    const bool needed = i == 0 ? !((fused_heads > 0 || proj_heads > 0) && bp->rp.on) : !bp->rp.on;
    if (fill && needed) {
      NRL_TRY(split_weight(ws[i], ns[i], D, p, outs[i], st));
    } else {
      *outs[i] = split_weight_view(p, ns[i], D);
    }
    p += split_weight_elems(ns[i], D);
  }
  if (bp->rp.on) {
    const int nd = rp_nblk_for(D), nq = rp_nblk_for(Q);
    uint16_t* q = w.rp;
    RpImageJobs jobs;
    rp_jobs_init(&jobs);
    auto add = [&](RpImage* im, const float* src, int64_t sn, int64_t sk, int N, int K, int nblk) {
      im->img = q; im->nblk = nblk; im->kblocks = rp_kblocks(K, false);
      if (fill) rp_jobs_add(&jobs, src, sn, sk, N, K, nullptr, q, nblk);
      q += rp_image_elems(nblk, im->kblocks);
    };
    add(&bp->rp.out_f, P->out_proj_weight, D, 1, D, D, nd);       // y = o W_o^T: element (n, k) = W_o[n][k]
    add(&bp->rp.att_f, P->att_weight, D, 1, Q, D, nq);            // t = y W_a^T
    add(&bp->rp.att_d, P->att_weight, 1, D, D, Q, nd);            // dy = d_pre W_a: (n, k) = W_a[k][n]
    add(&bp->rp.out_d, P->out_proj_weight, 1, D, D, D, nd);       // d_o = dy W_o
    add(&bp->rp.in_d, P->in_proj_weight, 1, D, D, 3 * D, nd);     // dx = dqkv W_in
    if (fused_heads > 0) {                                          // nrl_news_fused.h
      bp->rp.in_heads.img = q; bp->rp.in_heads.nblk = fused_heads * 4; bp->rp.in_heads.kblocks = NF_KB;
      if (fill) rp_jobs_add_qkv_heads(&jobs, P->in_proj_weight, D, P->in_proj_bias, q, fused_heads, D / fused_heads);
      q += rp_image_elems(fused_heads * 4, NF_KB);
      // dx = dqkv W_in with dqkv in head planes (KCSlab): reduction index head * 64 + c
      bp->rp.in_d_hp.img = q; bp->rp.in_d_hp.nblk = nd; bp->rp.in_d_hp.kblocks = fused_heads * 2;
      if (fill) rp_jobs_add_kheads(&jobs, P->in_proj_weight, 1, D, D, fused_heads, D / fused_heads, q, nd);
      q += rp_image_elems(nd, fused_heads * 2);
      // y = o W_o^T with `o` in head-permuted planes (KCPlanesG): reduction index = plane slot
      bp->rp.out_f_perm.img = q; bp->rp.out_f_perm.nblk = nd;
      bp->rp.out_f_perm.kblocks = (16 * (fused_heads + (fused_heads + 3) / 4) + 31) / 32;
      if (fill) rp_jobs_add_kperm(&jobs, P->out_proj_weight, D, 1, D, fused_heads, q, nd);
      q += rp_image_elems(nd, rp_kblocks(D + 32, false));
      // fused tail (nrl_news_tail.h): W_o over the plane slots with b_o at the ones slot; W_a in kappa order with b_a
      if (opt(O_NEWS_TAIL) && news_tail_geometry_ok(32, D, Q, fused_heads)) {
        bp->rp.tail_o.img = q; bp->rp.tail_o.nblk = NT_FB; bp->rp.tail_o.kblocks = NT_KB;
        if (fill) rp_jobs_add_kperm(&jobs, P->out_proj_weight, D, 1, D, fused_heads, q, NT_FB, P->out_proj_bias);
        q += rp_image_elems(NT_FB, NT_KB);
        bp->rp.tail_a.img = q; bp->rp.tail_a.nblk = NT_QB; bp->rp.tail_a.kblocks = NT_KS;
        if (fill) rp_jobs_add_kappa(&jobs, P->att_weight, D, 1, Q, D, P->att_bias, q, NT_QB, NT_KS);
        q += rp_image_elems(NT_QB, NT_KS);
        // backward: dy^T = W_a^T d_pre^T, element (n = feature, k = query) = W_a[k][n], queries in kappa order
        bp->rp.tail_ad.img = q; bp->rp.tail_ad.nblk = NT_FB; bp->rp.tail_ad.kblocks = NT_QS;
        if (fill && news_tail_bwd_geometry_ok(32, D, Q, fused_heads)) rp_jobs_add_kappa(&jobs, P->att_weight, 1, D, D, Q, nullptr, q, NT_FB, NT_QS);
        q += rp_image_elems(NT_FB, NT_QS);
      }
    } else if (proj_heads > 0) {                                    // nrl_attn_x3.hip: ua_fwd_proj_kernel
      bp->rp.in_heads.img = q; bp->rp.in_heads.nblk = proj_heads * 4; bp->rp.in_heads.kblocks = NF_KB;
      if (fill) rp_jobs_add_qkv_heads(&jobs, P->in_proj_weight, D, P->in_proj_bias, q, proj_heads, D / proj_heads);
      q += rp_image_elems(proj_heads * 4, NF_KB);
    }
    if (fill) NRL_TRY(rp_jobs_launch(jobs, st));
  }
  return NRL_OK;
}

// epilogues with a `stream` switch can write large outputs with the streaming hint (nrl_gemm.h).  OFF by default: the
// row-panel GEMMs are not store-bound (whole step 4.20 ms either way at B = 128); NRL_STREAM_MB=<n> streams outputs of
// >= n MB for A/B runs.
template <class T, class = void>
struct HasStream : std::false_type {};
template <class T>
struct HasStream<T, std::void_t<decltype(std::declval<T&>().stream)>> : std::true_type {};
template <class Epi>
static Epi with_stream(Epi e, int64_t M, int N) {
  static const int64_t thresh = [] {
    const char* env = getenv("NRL_STREAM_MB");
    const int64_t mb = env != nullptr ? atoll(env) : 0;
    return mb <= 0 ? (int64_t)1 << 62 : mb << 20;
  }();
  if constexpr (HasStream<Epi>::value) e.stream = (M * N * (int64_t)sizeof(float) >= thresh) ? 1 : 0;
  return e;
}

template <class AOp, class Epi>
static int rp_dispatch(const AOp& a, const RpImage& b, const Epi& epi_in, int64_t M, int N, int K, hipStream_t st) {
  const Epi epi = with_stream(epi_in, M, N);
  // Few rows (the user encoder: M = B * H = 50 workgroups of 128 rows at B = 128): 16-row panels -- twice the workgroups, half
  // the MFMA work per wave and k-block; the kernel's time there is one panel's k-loop latency (NRL_RP_HALF_ROWS=0 disables)
  static const int64_t half_rows = [] { const char* e = getenv("NRL_RP_HALF_ROWS"); return e ? atoll(e) : (int64_t)16384; }();
  if constexpr (std::is_same<AOp, KCPlain>::value) {
    if (M <= half_rows) {
      switch (b.nblk) {
        case 13: return launch_rp_gemm<13, 4, 0, 1>(a, b, epi, M, N, K, st);
        case 19: return launch_rp_gemm<19, 4, 0, 1>(a, b, epi, M, N, K, st);
        case 20: return launch_rp_gemm<20, 4, 0, 1>(a, b, epi, M, N, K, st);
      }
    }
  }
  switch (b.nblk) {
    case 13: return launch_rp_gemm<13>(a, b, epi, M, N, K, st);
    case 19: return launch_rp_gemm<19>(a, b, epi, M, N, K, st);
    case 20: return launch_rp_gemm<20>(a, b, epi, M, N, K, st);
  }
  set_error("row-panel GEMM: no instantiation for %d column blocks", b.nblk);
  return NRL_E_INVALID;
}

static bool big_tiles(int64_t M, int N) { return ceil_div(M, 256) * ceil_div(N, 160) >= 512; }

// C = epi(A W^T): nn.Linear forward.  W (N, K) fp32 in place / its bf16 planes.
template <class AOp, class Epi>
static int gemm_fwd(const AOp& a, const float* W, const SplitWeight& sw, const Epi& epi_in, int64_t M, int N, int K,
                    bool q_tile, hipStream_t st, const RpImage* rp = nullptr) {
  const Epi epi = with_stream(epi_in, M, N);
  if (cur_engine() == ENGINE_BF16X3) {
    if constexpr (std::is_same<AOp, KCPlain>::value)
      if (rp != nullptr && rp->img != nullptr) return rp_dispatch(a, *rp, epi, M, N, K, st);
    const KCSplit b{sw.hi, sw.lo, sw.ld, N};
    if constexpr (!std::is_same<AOp, KCGather>::value) if (opt(O_X3_DMA)) {
      // (the gathered operand keeps the register-staged kernel: its dropout hash would be re-evaluated by
      // every wave column at fragment-read time)
      if (q_tile && N <= 224) return launch_gemm_bf16x3_dma<X3_DMA_TILE_Q>(a, b, epi, M, N, K, st);
      return launch_gemm_bf16x3_dma<X3_DMA_TILE>(a, b, epi, M, N, K, st);
    }
    if (q_tile && N <= 224) return launch_gemm_bf16x3<X3_TILE_Q>(a, b, epi, M, N, K, 1, st);
    if (big_tiles(M, N)) return launch_gemm_bf16x3<X3_TILE_BIG>(a, b, epi, M, N, K, 1, st);
    return launch_gemm_bf16x3<X3_TILE>(a, b, epi, M, N, K, 1, st);
  }
  const KCPlain b{W, K, N};
  if (q_tile && N <= 208) return launch_gemm<NRL_TILE_Q>(a, b, epi, M, N, K, 1, st);
  return launch_gemm<NRL_TILE>(a, b, epi, M, N, K, 1, st);
}

// dX = epi(dY W): dY (M, Nw), W (Nw, Kw) -> (M, Kw)
template <class Epi>
static int gemm_dgrad(const float* dy, const float* W, const SplitWeight& sw, const Epi& epi_in, int64_t M, int Nw,
                      int Kw, hipStream_t st, const RpImage* rp = nullptr) {
  const Epi epi = with_stream(epi_in, M, Kw);
  const KCPlain a{dy, Nw, M};
  if (cur_engine() == ENGINE_BF16X3) {
    if (rp != nullptr && rp->img != nullptr) return rp_dispatch(a, *rp, epi, M, Kw, Nw, st);
    const KCSplit b{sw.hi_t, sw.lo_t, sw.ld_t, Kw};
    if (opt(O_X3_DMA)) return launch_gemm_bf16x3_dma<X3_DMA_TILE>(a, b, epi, M, Kw, Nw, st);
    if (big_tiles(M, Kw)) return launch_gemm_bf16x3<X3_TILE_BIG>(a, b, epi, M, Kw, Nw, 1, st);
    return launch_gemm_bf16x3<X3_TILE>(a, b, epi, M, Kw, Nw, 1, st);
  }
  return launch_gemm<NRL_TILE>(a, RCPlain{W, Kw, Kw, 0}, epi, M, Kw, Nw, 1, st);
}

// dW (I, J) += dY^T X, db (I) += colsum(dY): dY (M, I), X (M, J); split-K over M
static int gemm_wgrad(const float* dy, int I, const float* x, int J, float* dW, float* db, int64_t M,
                      hipStream_t st, float* scratch = nullptr, size_t scratch_floats = 0) {
  const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
  const EpiAtomicWB epi{dW, J, db, J};
  if (cur_engine() == ENGINE_BF16X3) {
    // ~1.6k rows per k-split (profiles/r01_gemm_bf16x3_probe.txt): the split's operand slices are
    // re-read by all of its tiles from ONE XCD's L2 (split -> XCD mapping in the kernel)
    auto splits = [&](int bm) {
      const int64_t tiles = ceil_div(I, bm) * ceil_div(J + 1, 160);
      int64_t sp = ceil_div(M, 1664);
      if (sp * tiles < 512) sp = ceil_div(512, tiles);
      const int64_t max_s = ceil_div(M, 256);
      return (int)(sp > max_s ? max_s : (sp < 1 ? 1 : sp));
    };
    if (I > 512 && opt(O_WGRAD_WS)) {
      // k-splits of the wave-specialised kernel: a multiple of 8 (split -> XCD mapping), chosen by a two-term cost in k-tile units,
      // rounds of 256 workgroups x (k-tiles per split + the epilogue's atomics, ~12 k-tiles' worth: profiles/r05_wgrad_ws_probe.txt --
      // 768 x 768 over 38400 rows: 16 splits 0.172 ms, 32 splits 0.195 ms; 3072 x 768: 8 splits 0.675, 32 splits 0.732).
      // NRL_WGRAD_WS_SPLITS forces a count (A/B runs).
      static const int forced = [] { const char* e = getenv("NRL_WGRAD_WS_SPLITS"); return e ? atoi(e) : 0; }();
      int ws_splits = forced;
      if (ws_splits <= 0) {
        const int64_t tiles = ceil_div(I, 256) * ceil_div(J + 1, 160), ktiles = ceil_div(M, 32);
        int64_t best = -1;
        for (int sp = 8; sp <= 64; sp += 8) {
          if (sp > 8 && ktiles / sp < 8) break;         // (at least 8 k-tiles per split)
          const int64_t cost = ceil_div(tiles * sp, 256) * (ceil_div(ktiles, sp) + 12);
          if (best < 0 || cost < best) best = cost, ws_splits = sp;
        }
      }
      return launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, epi, I, J + 1, M, ws_splits, st, scratch, scratch_floats);
    }
    if (I > 512) return launch_gemm_bf16x3<X3_TILE_BIG>(a, b, epi, I, J + 1, M, splits(256), st);
    // small outputs: LDS-DMA staged, transposition at the fragment read (0.35 -> 0.30 ms at 300 x 300;
    // the 900-row gradient is faster register-staged, profiles/r01_gemm_x3_dma_probe.txt)
    if (opt(O_X3_DMA)) return launch_gemm_bf16x3_dma_tn<2, 2, 2, 5, 2>(a, b, epi, I, J + 1, M, splits(64), st, scratch, scratch_floats);
    return launch_gemm_bf16x3<X3_TILE_W>(a, b, epi, I, J + 1, M, splits(64), st);
  }
  if (I > 512) return launch_gemm<NRL_TILE>(a, b, epi, I, J + 1, M, wgrad_splits(I, J + 1, M, 128, 160), st);
  return launch_gemm<NRL_TILE_W>(a, b, epi, I, J + 1, M, wgrad_splits(I, J + 1, M, 64, 160), st);
}

static int block_fwd_tail(const NrlBlockParams* P, const BlockShape& s, const BlockWs& w, const BlockPlanes& bp,
                          Dropout drop2, float* out, hipStream_t st);

// 64 <= S <= 128 with dh = 20 under the bf16x3 engine: the (hi, lo) split matrix-core attention of nrl_attn_x3.hip (the NRMS
// user encoder at 64 .. 128 users per rank); everything else -- and the exact-fp32 engine -- on attn_fwd / attn_bwd.  Forward
// and backward of a call agree on it: the engine travels with the call, the geometry is the call's.
static inline bool block_attn_x3(const BlockShape& s) {
  // (NRL_ATTN_X3_MIN_S: A/B runs of the lower edge; below it the vector-ALU kernels pack 4-8 groups into a workgroup.  32 since
  //  round 4: with the in-projection inside the kernel (user_proj) the configs[0] step, S = B = 32, is 1.065 vs 1.093 ms; round 3,
  //  without it: 1.18 vs 1.19 at a threshold of 17, not worth a routing rule then)
  static const int min_s = [] { const char* e = getenv("NRL_ATTN_X3_MIN_S"); return e ? atoi(e) : 32; }();
  return cur_engine() == ENGINE_BF16X3 && s.geom.S >= min_s && attn_x3_ok(s.geom);
}

// forward of the shared block given an A-operand accessor for the in-projection
template <class AOp>
static int block_fwd(const NrlBlockParams* P, const AOp& a_in, const BlockShape& s, const BlockWs& w,
                     Dropout drop2, bool save, bool prof_in_proj, float* out, hipStream_t st) {
  const int D = s.D;
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  BlockPlanes bp;
  // user_proj: the in-projection inside the across-users attention kernel (plain input rows, the x3 attention's geometry, the
  // per-head image's k-blocks): no in-projection GEMM launch, no weight-split launch
  bool proj = false;
  if constexpr (std::is_same<AOp, KCPlain>::value) {
    proj = opt(O_USER_PROJ) && cur_engine() == ENGINE_BF16X3 && block_rp_ok(D, s.Q) && block_attn_x3(s) && a_in.ld == D &&
           D == s.heads * 20 && attn_x3_proj_ok(s.geom, s.heads * 4, NF_KB) && s.geom.q_outer % 3 == 0 && s.geom.q_seq % 3 == 0;
  }
  NRL_TRY(block_planes(P, s, w, cur_engine() == ENGINE_BF16X3, &bp, st, 0, proj ? s.heads : 0));
  if constexpr (std::is_same<AOp, KCPlain>::value) {
    if (proj) {
      NRL_TRY(attn_fwd_x3_proj(a_in.p, s.geom.q_outer / 3, s.geom.q_seq / 3, bp.rp.in_heads.img, bp.rp.in_heads.nblk,
                               save ? w.qkv : nullptr, w.o, save ? w.lse : nullptr, s.geom, st));
      return block_fwd_tail(P, s, w, bp, drop2, out, st);
    }
  }
  // q|k|v = x W_in^T + b_in           (text.py:229 / user/nrms.py:34; torch in-projection)
  {
    ProfScope prof(st, prof_in_proj ? 2.0 * (double)s.M * 3.0 * D * D : 0.0);
    NRL_TRY(gemm_fwd(a_in, P->in_proj_weight, bp.in, EpiLinear{w.qkv, 3 * D, P->in_proj_bias, 0, nodrop, 3 * D},
                     s.M, 3 * D, D, false, st));
  }
  // per (group, head): softmax(q k^T / sqrt(dh)) v
  if (block_attn_x3(s)) NRL_TRY(attn_fwd_x3(w.qkv, w.o, save ? w.lse : nullptr, s.geom, st));
  else NRL_TRY(attn_fwd(w.qkv, w.o, save ? w.lse : nullptr, s.geom, st));
  return block_fwd_tail(P, s, w, bp, drop2, out, st);
}

// out-projection -> dropout -> additive attention, from the attention output w.o
static int block_fwd_tail(const NrlBlockParams* P, const BlockShape& s, const BlockWs& w, const BlockPlanes& bp,
                          Dropout drop2, float* out, hipStream_t st) {
  const int D = s.D, Q = s.Q;
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  // one user (<= 64 pooled rows) per workgroup: out-projection, tanh-linear and pooling in ONE launch (nrl_user_tail.hip);
  // writes the y / t / w the row-panel launches below would, so the backward is the same either way
  if (!s.od_planes && !s.aa_planes && bp.rp.on &&
      user_tail_ok(s.pool_groups, s.pool_len, D, Q, bp.rp.out_f.nblk, bp.rp.att_f.nblk) && s.pool_groups * s.pool_len == s.M) {
    UserTailArgs a;
    a.o = w.o; a.img_o = bp.rp.out_f.img; a.img_a = bp.rp.att_f.img; a.nblk_o = bp.rp.out_f.nblk; a.nblk_a = bp.rp.att_f.nblk;
    a.b_o = P->out_proj_bias; a.b_a = P->att_bias; a.q_a = P->att_query; a.groups = s.pool_groups; a.H = s.pool_len; a.D = D;
    a.Q = Q; a.drop2 = drop2; a.y = w.y; a.t = w.t; a.w = w.w; a.out = out;
    return user_tail_fwd(a, st);
  }
  // y = dropout(o W_o^T + b_o)        (out-projection, text.py:229-230)
  const int ncb_y = (D + 16) / 16;                       // y planes: D features + the ones column
  unsigned char* const ypl = reinterpret_cast<unsigned char*>(w.yp);
  if (s.od_planes) {
    // `o` arrives as head-permuted (hi, lo) planes from the fused forward: no split, reduction over plane slots
    const int ncb = s.heads + (s.heads + 3) / 4;
    const KCPlanesG a_o{reinterpret_cast<const unsigned char*>(w.o), s.M, ncb};
    const EpiLinear epi{w.y, D, P->out_proj_bias, 0, drop2, D};
    if (s.aa_planes) {
      if (s.M % 32 != 0)   // rows past M in the last 32-row k-tile of the weight gradient
        NRL_HIP(hipMemsetAsync(ypl + (s.M / 32) * 2 * ncb_y * 1024, 0, (size_t)2 * ncb_y * 1024, st));
      NRL_TRY(rp_dispatch(a_o, bp.rp.out_f_perm, EpiLinearPlanes{epi, ypl, ncb_y}, s.M, D, 16 * ncb, st));
    } else {
      NRL_TRY(rp_dispatch(a_o, bp.rp.out_f_perm, epi, s.M, D, 16 * ncb, st));
    }
  } else {
    NRL_TRY(gemm_fwd(KCPlain{w.o, D, s.M}, P->out_proj_weight, bp.out,
                     EpiLinear{w.y, D, P->out_proj_bias, 0, drop2, D}, s.M, D, D, false, st,
                     bp.rp.on ? &bp.rp.out_f : nullptr));
  }
  // t = tanh(y W_a^T + b_a)           (attention.py:34)
  if (s.aa_planes) {
    NRL_TRY(rp_dispatch(KCPlanesG{ypl, s.M, ncb_y}, bp.rp.att_f, EpiLinear{w.t, Q, P->att_bias, 1, nodrop, Q}, s.M, Q, D, st));
  } else {
    NRL_TRY(gemm_fwd(KCPlain{w.y, D, s.M}, P->att_weight, bp.att, EpiLinear{w.t, Q, P->att_bias, 1, nodrop, Q},
                     s.M, Q, D, true, st, bp.rp.on ? &bp.rp.att_f : nullptr));
  }
  // w = softmax(t . q_a); out = sum w y   (attention.py:37-40)
  NRL_TRY(pool_fwd(w.t, P->att_query, w.y, s.pool_groups, s.pool_len, Q, D, w.w, out, st));
  return NRL_OK;
}

// Backward of the shared block, in two phases so that a data-parallel caller can start the
// all-reduce of the (large) input-side gradient while the weight gradients are still being computed:
//   phase 1: everything on the activation-gradient chain down to d(qkv)  (the caller then runs the
//            in-projection dgrad -> table gradient / d_hist)
//   phase 2: the three weight(+bias)-gradient GEMMs, which only READ saved activations/gradients.
// A side stream of the caller's (nrl_api.hip: ForkSet) for work that is off the activation-gradient chain
struct SideFork {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};

// The generic planes weight gradient as 8-wave workgroups (round 6: two waves per SIMD over the same tile and LDS stages; at B = 128
// 0.173 -> 0.133 ms for the out-projection shape, 0.190 -> 0.167 for the additive-attention one, tools/wp_probe.hip).
// NRL_WGRAD_PLANES_WAVES=4 restores the one-wave-per-SIMD shape of rounds 2-5 for all three planes weight gradients (A/B runs).
static inline int wgrad_planes_waves() {
  static const int v = [] { const char* e = getenv("NRL_WGRAD_PLANES_WAVES"); return e ? atoi(e) : 8; }();
  return v;
}
template <int TM, int TN, class Epi>
static int wgrad_planes_g_dispatch(const void* a, int ncb_a, const void* b, int ncb_b, int64_t rows, int64_t m_valid, int n_valid,
                                   const Epi& epi, int nsplit, hipStream_t st, float* scratch) {
  if (wgrad_planes_waves() == 4) return launch_wgrad_planes_g<TM, TN, 4>(a, ncb_a, b, ncb_b, rows, m_valid, n_valid, epi, nsplit, st, scratch);
  return launch_wgrad_planes_g<TM, TN, 8>(a, ncb_a, b, ncb_b, rows, m_valid, n_valid, epi, nsplit, st, scratch);
}

// The weight gradients of the back half from planes: dW_a += d_pre^T y, dW_o += dy^T o (+ biases).  Partial tiles of the
// two-step split-K go to `scratch` (scratch_floats of it), or straight to atomics when it is too small.
static int block_wgrads_back_half(const NrlBlockGrads* G, const BlockShape& s, const BlockWs& w, hipStream_t st,
                                  float* scratch, size_t scratch_floats) {
  const int D = s.D, Q = s.Q;
  auto scratch_for = [&](size_t need) -> float* { return need <= scratch_floats ? scratch : nullptr; };
  {
    // d_pre (pool_bwd_pre) and y (out-projection epilogue, with its ones column) as planes over the same rows
    static const int sp = [] { const char* e = getenv("NRL_WGRAD_PLANES_AA_SPLITS"); return e ? atoi(e) : 128; }();
    NRL_TRY((wgrad_planes_g_dispatch<7, 5>(w.tp, (Q + 15) / 16, w.yp, (D + 16) / 16, (s.M + 31) / 32 * 32, Q, D + 1,
                                         EpiAtomicWB{G->att_weight, D, G->att_bias, D}, sp, st,
                                         scratch_for(wgrad_planes_g_scratch_floats(7, 5, (Q + 15) / 16, (D + 16) / 16, sp)))));
  }
  {
    // both operands are planes over the same (real) rows: DMA + transpose-read + MFMA only (wgrad_planes_g_kernel)
    static const int sp = [] { const char* e = getenv("NRL_WGRAD_PLANES_G_SPLITS"); return e ? atoi(e) : 64; }();
    const int ncb_o = s.heads + (s.heads + 3) / 4, ncb_dy = (D + 15) / 16;
    NRL_TRY((wgrad_planes_g_dispatch<5, 5>(w.dy, ncb_dy, w.o, ncb_o, (s.M + 31) / 32 * 32, D, 16 * ncb_o,
                                         EpiAtomicWBPerm{G->out_proj_weight, D, G->out_proj_bias, s.heads}, sp, st,
                                         scratch_for(wgrad_planes_g_scratch_floats(5, 5, ncb_dy, ncb_o, sp)))));
  }
  return NRL_OK;
}

// news_fork: the two weight gradients above depend on nothing past the tail backward and are matrix-core work over operands
// that are read-only from then on; the chain they would otherwise wait behind (out-projection dgrad -> token-attention
// backward -> in-projection dgrad -> table gradient) is HBM-write-bound.  With a side stream they run beside it; their
// split-K partial tiles go to the tanh buffer (unused on the fused-tail path) because the q|k|v slabs are still live.
static inline bool news_fork_on(const BlockShape& s) {
  return s.tail_bwd && s.aa_planes && s.od_planes && opt(O_NEWS_FORK) && opt(O_WGRAD_2STEP);
}

static int block_bwd_phase1(const NrlBlockParams* P, const NrlBlockGrads* G, const BlockShape& s, const BlockWs& w,
                            const BlockPlanes& bp, Dropout drop2, const float* d_out, hipStream_t st,
                            bool attention_elsewhere = false, const SideFork* side = nullptr) {
  const int D = s.D, Q = s.Q;
  // additive attention backward: t -> d_pre in place, dq_a
  const int ncb_q = (Q + 15) / 16;
  unsigned char* const tpl = reinterpret_cast<unsigned char*>(w.tp);
  if (s.aa_planes && s.M % 32 != 0)
    NRL_HIP(hipMemsetAsync(tpl + (s.M / 32) * 2 * ncb_q * 1024, 0, (size_t)2 * ncb_q * 1024, st));
  if (s.tail_bwd) {
    // tanh recomputed from the y planes, d_pre / dq_a / dy in ONE kernel (nrl_news_tail.h)
    const int ncb = (D + 15) / 16;
    unsigned char* dyp = reinterpret_cast<unsigned char*>(w.dy);
    if (s.M % 32 != 0)
      NRL_HIP(hipMemsetAsync(dyp + (s.M / 32) * 2 * ncb * 1024, 0, (size_t)2 * ncb * 1024, st));
    NewsTailBwdArgs b;
    b.y_planes = reinterpret_cast<const unsigned char*>(w.yp); b.w = w.w; b.d_out = d_out; b.img_a = bp.rp.tail_a.img;
    b.img_ad = bp.rp.tail_ad.img; b.q_a = P->att_query; b.n_news = s.pool_groups; b.L = s.pool_len; b.D = D; b.Q = Q;
    b.drop2 = drop2; b.dpre_planes = tpl; b.dy_planes = dyp; b.dq_a = G->att_query;
    NRL_TRY(news_tail_bwd(b, st));
    if (side != nullptr && side->s != nullptr && news_fork_on(s)) {
      NRL_HIP(hipEventRecord(side->fork, st));
      NRL_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
      NRL_TRY(block_wgrads_back_half(G, s, w, side->s, w.t, (size_t)s.M * Q));
      NRL_HIP(hipEventRecord(side->join, side->s));      // the caller makes `st` wait for it at the end of phase 1
    }
    // d_o = dy W_o
    NRL_TRY(rp_dispatch(KCPlanesG{dyp, s.M, ncb}, bp.rp.out_d, EpiStore{w.d_o, D}, s.M, D, D, st));
    if (!attention_elsewhere) {
    if (block_attn_x3(s)) NRL_TRY(attn_bwd_x3(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
    else NRL_TRY(attn_bwd(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
  }
    return NRL_OK;
  }
  // one user per workgroup: pooling backward, additive-attention dgrad and out-projection dgrad in ONE launch (nrl_user_tail.hip)
  if (!s.tail && !s.od_planes && !s.aa_planes && bp.rp.on && s.pool_groups * s.pool_len == s.M &&
      user_tail_bwd_ok(s.pool_groups, s.pool_len, D, Q, bp.rp.att_d.nblk, bp.rp.out_d.nblk, bp.rp.att_d.kblocks, bp.rp.out_d.kblocks)) {
    UserTailBwdArgs a;
    a.d_out = d_out; a.y = w.y; a.t = w.t; a.w = w.w; a.q_a = P->att_query; a.img_ad = bp.rp.att_d.img; a.img_od = bp.rp.out_d.img;
    a.nblk_ad = bp.rp.att_d.nblk; a.nblk_od = bp.rp.out_d.nblk; a.groups = s.pool_groups; a.H = s.pool_len; a.D = D; a.Q = Q;
    a.drop2 = drop2; a.dy = w.dy; a.d_o = w.d_o; a.dq_a = G->att_query;
    NRL_TRY(user_tail_bwd(a, st));
    if (!attention_elsewhere) {
      if (block_attn_x3(s)) NRL_TRY(attn_bwd_x3(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
      else NRL_TRY(attn_bwd(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
    }
    return NRL_OK;
  }
  NRL_TRY(pool_bwd_pre(d_out, s.tail ? nullptr : w.y, w.w, w.t, P->att_query, G->att_query, s.pool_groups, s.pool_len, Q, D, st,
                       s.aa_planes ? tpl : nullptr, s.tail ? w.yp : nullptr));
  if (s.od_planes) {
    // dy = (d_pre W_a + w * d_out) * dropout2, written ONCE as (hi, lo) planes: its only readers are the two GEMMs below
    const int ncb = (D + 15) / 16;
    unsigned char* dyp = reinterpret_cast<unsigned char*>(w.dy);
    if (s.M % 32 != 0)   // the weight gradient reads whole 32-row k-tiles: rows past M in the last one must be zero
      NRL_HIP(hipMemsetAsync(dyp + (s.M / 32) * 2 * ncb * 1024, 0, (size_t)2 * ncb * 1024, st));
    const EpiPoolBwdPlanes epi_dy{EpiPoolBwd{nullptr, D, w.w, d_out, s.pool_len, drop2}, dyp, ncb};
    if (s.aa_planes) NRL_TRY(rp_dispatch(KCPlanesG{tpl, s.M, ncb_q}, bp.rp.att_d, epi_dy, s.M, D, Q, st));
    else NRL_TRY(rp_dispatch(KCPlain{w.t, Q, s.M}, bp.rp.att_d, epi_dy, s.M, D, Q, st));
    // d_o = dy W_o
    NRL_TRY(rp_dispatch(KCPlanesG{dyp, s.M, ncb}, bp.rp.out_d, EpiStore{w.d_o, D}, s.M, D, D, st));
  } else {
    // dy = (d_pre W_a + w * d_out) * dropout2
    NRL_TRY(gemm_dgrad(w.t, P->att_weight, bp.att, EpiPoolBwd{w.dy, D, w.w, d_out, s.pool_len, drop2}, s.M, Q, D, st,
                       bp.rp.on ? &bp.rp.att_d : nullptr));
    // d_o = dy W_o
    NRL_TRY(gemm_dgrad(w.dy, P->out_proj_weight, bp.out, EpiStore{w.d_o, D}, s.M, D, D, st,
                       bp.rp.on ? &bp.rp.out_d : nullptr));
  }
  // attention backward -> dqkv
  if (!attention_elsewhere) {
    if (block_attn_x3(s)) NRL_TRY(attn_bwd_x3(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
    else NRL_TRY(attn_bwd(w.qkv, w.o, w.d_o, w.lse, w.dqkv, s.geom, st));
  }
  return NRL_OK;
}

// The in-projection weight gradient of the fused news path from planes (dW_in += dqkv^T x): its own function because, under
// news_fork, it too is issued from phase 1 -- as soon as the token-attention backward has produced dqkv, on the internal stream
// that carries the two back-half weight gradients (behind them), beside the live-row dgrad and the table gradient: with it ALL
// weight gradients of the news encoder run beside its activation-gradient chain.  Measured on one box (profiles/r06_ab.txt, three
// alternating runs each): phase 2 (NRL_NEWS_FORK_IN=0) 2.712 ms, its own second stream (=1) 2.681, behind the back-half ones
// (=2, the default) 2.673.  Partial tiles go to the q|k|v slabs, dead once the attention backward has run.
static inline int news_fork_in_mode() {     // 0: phase 2 (off); 1: its own internal stream; 2: behind the back-half weight gradients on theirs
  static const int v = [] { const char* e = getenv("NRL_NEWS_FORK_IN"); return e != nullptr ? atoi(e) : 2; }();
  return v;
}
static inline bool news_fork_in_on(const BlockShape& s, bool bf16_planes) {
  return news_fork_in_mode() != 0 && bf16_planes && news_fork_on(s) && s.forked;
}
static int block_wgrad_in_planes(const NrlBlockGrads* G, const float* x_rows, const BlockShape& s, const BlockWs& w, hipStream_t st) {
  const int D = s.D;
  const size_t scratch_avail = opt(O_WGRAD_2STEP) ? qkv_elems(s.M, s.D, s.heads, s.pad_rows) : 0;
  static const int wp_splits = [] { const char* e = getenv("NRL_WGRAD_PLANES_SPLITS"); return e ? atoi(e) : 32; }();
  const EpiAtomicWBHeads epi_w{G->in_proj_weight, D, G->in_proj_bias, D, s.heads, s.dh};
  float* const sc_w = wgrad_planes_scratch_floats(s.heads, 20, wp_splits) <= scratch_avail ? w.qkv : nullptr;
  // round 6: two waves per SIMD over the same tile (8-wave workgroups): 0.37 -> 0.33-0.34 ms at B = 128 (tools/wp_probe.hip,
  // profiles/r06_wgrad_planes_probe.txt); NRL_WGRAD_PLANES_WAVES=4 restores the one-wave-per-SIMD shape (A/B runs)
  if (wgrad_planes_waves() == 4) return launch_wgrad_planes<0, 4>(w.dqkv, x_rows, s.pool_groups, s.heads, 20, D + 1, epi_w, wp_splits, st, sc_w);
  return launch_wgrad_planes<0, 8>(w.dqkv, x_rows, s.pool_groups, s.heads, 20, D + 1, epi_w, wp_splits, st, sc_w);
}

static int block_bwd_phase2(const NrlBlockGrads* G, const float* x_rows, const BlockShape& s, const BlockWs& w,
                            hipStream_t st, bool dqkv_head_planes = false, bool bf16_planes = false) {
  const int D = s.D, Q = s.Q;
  // partial tiles of the planes weight gradients: the q|k|v slabs are dead once the attention backward has run
  const size_t scratch_avail = (bf16_planes && opt(O_WGRAD_2STEP)) ? qkv_elems(s.M, s.D, s.heads, s.pad_rows) : 0;
  auto scratch_for = [&](size_t need) -> float* { return need <= scratch_avail ? w.qkv : nullptr; };
  // the fp32-fed weight gradients (user encoder; the fallbacks of the news path) reduce their splits the same way, the packed
  // q|k|v rows being dead by now as well (attention backward and in-projection dgrad have run)
  float* const sc = opt(O_WGRAD_2STEP) && !dqkv_head_planes && !bf16_planes ? w.qkv : nullptr;
  const size_t sc_n = sc != nullptr ? qkv_elems(s.M, s.D, s.heads, s.pad_rows) : 0;
  const bool forked = news_fork_on(s) && s.forked;      // the two back-half weight gradients ran beside phase 1
  if (!forked && s.aa_planes && s.od_planes) {
    NRL_TRY(block_wgrads_back_half(G, s, w, st, w.qkv, scratch_avail));
  } else if (!forked) {
    // dW_a += d_pre^T y ; db_a += colsum(d_pre)     (y is the post-dropout activation)
    if (s.aa_planes) {
      static const int sp = [] { const char* e = getenv("NRL_WGRAD_PLANES_AA_SPLITS"); return e ? atoi(e) : 128; }();
      NRL_TRY((wgrad_planes_g_dispatch<7, 5>(w.tp, (Q + 15) / 16, w.yp, (D + 16) / 16, (s.M + 31) / 32 * 32, Q, D + 1,
                                           EpiAtomicWB{G->att_weight, D, G->att_bias, D}, sp, st,
                                           scratch_for(wgrad_planes_g_scratch_floats(7, 5, (Q + 15) / 16, (D + 16) / 16, sp)))));
    } else {
      NRL_TRY(gemm_wgrad(w.t, Q, w.y, D, G->att_weight, G->att_bias, s.M, st, sc, sc_n));
    }
    // dW_o += dy^T o ; db_o += colsum(dy)
    if (s.od_planes) {
      static const int sp = [] { const char* e = getenv("NRL_WGRAD_PLANES_G_SPLITS"); return e ? atoi(e) : 64; }();
      const int ncb_o = s.heads + (s.heads + 3) / 4, ncb_dy = (D + 15) / 16;
      NRL_TRY((wgrad_planes_g_dispatch<5, 5>(w.dy, ncb_dy, w.o, ncb_o, (s.M + 31) / 32 * 32, D, 16 * ncb_o,
                                           EpiAtomicWBPerm{G->out_proj_weight, D, G->out_proj_bias, s.heads}, sp, st,
                                           scratch_for(wgrad_planes_g_scratch_floats(5, 5, ncb_dy, ncb_o, sp)))));
    } else {
      NRL_TRY(gemm_wgrad(w.dy, D, w.o, D, G->out_proj_weight, G->out_proj_bias, s.M, st, sc, sc_n));
    }
  }
  // dW_in += dqkv^T x ; db_in += colsum(dqkv)
  if (bf16_planes) {
    // both operands pre-split by their producers: pure DMA + transpose-read + MFMA kernel (nrl_wgrad_planes.h)
    if (news_fork_in_on(s, bf16_planes)) return NRL_OK;     // issued from phase 1, beside the live-row dgrad (nrl_news_encoder_bwd)
    return block_wgrad_in_planes(G, x_rows, s, w, st);
  }
  if (dqkv_head_planes) {
    // dqkv in head planes (news_attn_bwd_kernel): 64 output rows per head, remapped to [Wq; Wk; Wv] rows on the way out
    static const int ws_splits = [] { const char* e = getenv("NRL_WGRAD_WS_SPLITS"); return e ? atoi(e) : 32; }();
    const int Ip = s.heads * 64;
    int64_t sp = s.M >= (int64_t)ws_splits * 512 ? ws_splits : ceil_div(s.M, 1664);
    if (sp < 1) sp = 1;
    return launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(RCSlab{w.dqkv, Ip}, RCPlain{x_rows, D, D, 1},
                                                EpiAtomicWBHeads{G->in_proj_weight, D, G->in_proj_bias, D, s.heads, s.dh},
                                                Ip, D + 1, s.M, (int)sp, st);
  }
  NRL_TRY(gemm_wgrad(w.dqkv, 3 * D, x_rows, D, G->in_proj_weight, G->in_proj_bias, s.M, st, sc, sc_n));
  return NRL_OK;
}

static int check_grads(const NrlBlockGrads* g) {
  NRL_REQUIRE(g != nullptr && g->in_proj_weight && g->in_proj_bias && g->out_proj_weight &&
                  g->out_proj_bias && g->att_weight && g->att_bias && g->att_query,
              "null gradient pointer");
  return NRL_OK;
}

// the fused front half applies to the reference's news-encoder geometry under the bf16x3 engine
static bool news_fused_on(const BlockShape& s, int L) {
  return opt(O_NEWS_FUSED) && cur_engine() == ENGINE_BF16X3 && block_rp_ok(s.D, s.Q) && s.dh == 20 &&
         news_fused_ok(L, s.D, s.heads);
}

// ... and its back half as one kernel too, when `o` arrives as planes and the y / d_pre planes exist
static bool news_tail_on(const BlockShape& s, int L, const BlockWs& w) {
  return opt(O_NEWS_TAIL) && s.od_planes && opt(O_NEWS_AA_PLANES) && w.yp != nullptr && news_fused_ok(L, s.D, s.heads) && news_tail_geometry_ok(L, s.D, s.Q, s.heads);
}

static bool news_tail_bwd_on(const BlockShape& s, int L, const BlockWs& w) {
  return news_tail_on(s, L, w) && opt(O_NEWS_TAIL_BWD) && w.tp != nullptr && news_tail_bwd_geometry_ok(L, s.D, s.Q, s.heads);
}

// token rows padded to 32 per news: only where the fragment-block planes of the fused news path exist (D = 20 heads within
// NF_KB k-blocks); any other geometry keeps plain (M, D) rows and pays nothing for the padding
static int64_t news_pad_rows(int64_t n_news, int L, int D, int heads) {
  return (heads > 0 && news_fused_ok(L, D, heads)) ? n_news * 32 : 0;
}

static BlockShape news_shape(const NrlBlockParams* p, int64_t n_news, int L) {
  BlockShape s;
  s.D = p->embed_dim; s.Q = p->query_dim; s.heads = p->num_heads; s.dh = s.D / s.heads;
  s.M = n_news * L;
  s.pad_rows = news_pad_rows(n_news, L, s.D, s.heads);
  s.pool_groups = n_news; s.pool_len = L;
  s.geom.q_outer = (int64_t)L * 3 * s.D; s.geom.q_seq = 3 * s.D;
  s.geom.o_outer = (int64_t)L * s.D; s.geom.o_seq = s.D;
  s.geom.groups = n_news * s.heads; s.geom.heads = s.heads; s.geom.S = L; s.geom.D = s.D;
  s.geom.dh = s.dh; s.geom.scale = 1.0f / sqrtf((float)s.dh);
  return s;
}

static BlockShape user_shape(const NrlBlockParams* p, int64_t B, int64_t H) {
  BlockShape s;
  s.D = p->embed_dim; s.Q = p->query_dim; s.heads = p->num_heads; s.dh = s.D / s.heads;
  s.M = B * H;
  s.pool_groups = B; s.pool_len = (int)H;
  // seq-first quirk: the "sequence" is the user axis (stride H rows), the "batch" the slot axis
  s.geom.q_outer = 3 * s.D; s.geom.q_seq = H * 3 * s.D;
  s.geom.o_outer = s.D; s.geom.o_seq = H * s.D;
  s.geom.groups = H * s.heads; s.geom.heads = s.heads; s.geom.S = (int)B; s.geom.D = s.D;
  s.geom.dh = s.dh; s.geom.scale = 1.0f / sqrtf((float)s.dh);
  return s;
}

}  // namespace nrl
