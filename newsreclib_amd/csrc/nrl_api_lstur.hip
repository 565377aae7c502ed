// C ABI of the LSTUR path (include/newsreclib_amd.h, second half): CNN text encoder, row-masked
// embedding lookups, GRU user encoder.  Its own translation unit; the engine / option state, tile choices and GEMM
// helpers it shares with nrl_api.hip come from nrl_api_internal.h.

#include "nrl_api_internal.h"

namespace nrl {

// ---- conv weight planes for the bf16x3 dgrad: plane[d][t'*F + f] = Wc[f][(W-1-t')*D + d] ------
__global__ void split_conv_weight_t_kernel(const float* __restrict__ w, int F, int D, int W, int Kp,
                                           uint16_t* __restrict__ hi) {
  const int64_t total = (int64_t)D * Kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i / Kp), k = (int)(i % Kp);
    float v = 0.f;
    if (k < W * F) {
      const int tr = k / F, f = k - tr * F;
      v = w[((int64_t)f * W + (W - 1 - tr)) * D + d];
    }
    const __bf16 h = (__bf16)v;
    const int64_t pos = split_pos(d, k, 2 * Kp);  // interleaved (hi | lo) k-tiles, see KCSplit
    hi[pos] = __builtin_bit_cast(unsigned short, h);
    hi[pos + 32] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
  }
}

struct CnnShape {
  int64_t N, M;
  int L, D, F, W, Q, pad;
};

struct CnnWs {
  float *x, *c, *t, *w, *dc, *dx;
  uint16_t *planes_conv, *planes_att, *planes_conv_t;
  uint16_t* rp;  // row-panel weight images (nrl_rowpanel.h)
  unsigned char *xpl, *dpl;   // x / dc as fragment-block planes over 32 padded rows per news (conv weight gradient), or null
  float* wsc;                 // its split-K partial tiles
  unsigned char *cpl, *tpl;   // c (+ its ones column) / d_pre as fragment-block planes over the real rows (cnn_aa_planes), or null
};

// The convolution weight gradient from planes (wgrad_planes_conv_kernel, nrl_wgrad_planes.h): window 3, a news inside one
// 32-row k-tile with at least one zero row after it.  NRL_CONV_WGRAD_PLANES=0 keeps the fp32-fed kernel.
static int cnn_conv_planes_splits() {
  static const int sp = [] { const char* e = getenv("NRL_CONV_WGRAD_SPLITS"); const int v = e ? atoi(e) : 24; return v < 1 ? 1 : (v > 512 ? 512 : v); }();
  return sp;
}
static bool cnn_conv_planes_ok(const CnnShape& s) {
  static const bool on = [] { const char* e = getenv("NRL_CONV_WGRAD_PLANES"); return !(e != nullptr && e[0] == '0'); }();
  // (F <= 320: two 160-filter tiles.  At 400 filters -- NAML, CenNewsRec -- a third, 20 % empty tile and the wider dc conversion
  //  make the planes form the slower one: NAML step 9.53 vs 9.16 ms, profiles/r04_ab.txt)
  return on && s.W == 3 && s.L <= 63 && s.F <= 320 && s.N < (1LL << 31);
}
static int cnn_conv_kt(const CnnShape& s) { return s.L <= 31 ? 1 : 2; }   // k-tiles of 32 padded rows per news
static int cnn_ncb_x(const CnnShape& s) { return ((s.D + 16) / 16 + 1) & ~1; }   // + the ones column; EVEN: a tap of the tap-padded
                                                                                 // reduction index is whole 32-wide k-blocks (KCWindowPlanes)
static int cnn_ncb_dc(const CnnShape& s) { return ((s.F + 15) / 16 + 1) & ~1; }   // EVEN, as cnn_ncb_x

static bool cnn_rp_ok(const CnnShape& s);
// x ONLY as planes (round 5): the lookup writes the planes the weight gradient reads, the convolution forward reads them through
// KCWindowPlanes against a tap-padded weight image -- no fp32 x, no conversion launch.  Forward and backward evaluate the same
// predicate (shape, engine, switches).  NRL_CONV_X_PLANES=0 keeps the fp32 rows + planes_from_rows (A/B).
static bool cnn_x_planes_on(const CnnShape& s) {
  static const bool on = [] { const char* e = getenv("NRL_CONV_X_PLANES"); return !(e != nullptr && e[0] == '0'); }();
  return on && cnn_conv_planes_ok(s) && cnn_rp_ok(s) && cur_engine() == ENGINE_BF16X3;
}

// ... and dc (the convolution output's gradient) ONLY as planes: written by the additive-attention dgrad's epilogue
// (EpiPoolBwdNewsPlanes), read by the convolution's activation gradient over the live rows (KCWindowLivePlanes, tap-padded
// transposed image `conv_dp`) and by the weight gradient.  The backward falls back to fp32 dc when it has no sorted positions.
// NRL_CONV_DC_PLANES=0: fp32 dc + planes_from_rows (A/B).
static bool cnn_dc_planes_on(const CnnShape& s) {
  static const bool on = [] { const char* e = getenv("NRL_CONV_DC_PLANES"); return !(e != nullptr && e[0] == '0'); }();
  static const bool live = [] { const char* e = getenv("NRL_LIVE_ROWS"); return !(e != nullptr && e[0] == '0'); }();
  return on && live && cnn_x_planes_on(s);
}

// Round 6, the additive attention of the convolution encoder on planes, as the NRMS block has had it since round 3 (news_aa_planes):
// the convolution's epilogue writes c ALSO as (hi, lo) planes over the real rows with a ones column (EpiLinearPlanes), the pooling
// backward writes d_pre ONLY as planes; the tanh projection and the additive-attention dgrad read their A operand from them without
// a split, and the weight gradient dW_a += d_pre^T c (+ the bias from the ones column) is the DMA-fed planes kernel
// (wgrad_planes_g<7, 5>, 8-wave workgroups) instead of the fp32-fed gemm_bf16x3_dma_tn (263 us per call at B = 128, the planes kernel
// 86 us on the same shape in the NRMS step).  Widths as there: F % 16 == 12 (room for the ones column), Q <= 224.  Forward and
// backward evaluate the same predicate; a backward without sorted positions (no dc planes) is refused the planes.
// NRL_CNN_AA_PLANES=0: the fp32 operands (A/B runs).
static bool cnn_aa_planes_on(const CnnShape& s) {
  static const bool on = [] { const char* e = getenv("NRL_CNN_AA_PLANES"); return !(e != nullptr && e[0] == '0'); }();
  return on && cnn_dc_planes_on(s) && (s.F & 15) == 12 && s.F <= 304 && s.Q <= 224 && s.Q % 4 == 0 && opt(O_NEWS_AA_PLANES);
}
static int cnn_ncb_c(const CnnShape& s) { return (s.F + 16) / 16; }
static int cnn_ncb_q(const CnnShape& s) { return (s.Q + 15) / 16; }
static size_t cnn_row_planes_bytes(int64_t M, int ncb) { return (size_t)((M + 31) / 32) * 2 * ncb * 1024; }

// conv forward / dgrad and additive-attention forward / dgrad on the row-panel kernel (bf16x3, widths <= 320)
struct CnnRp {
  RpImage conv_f, conv_d, att_f, att_d, conv_dp;
  bool on = false;
};
static bool cnn_rp_ok(const CnnShape& s) {
  return opt(O_ROWPANEL) && rp_nblk_supported(s.D) && rp_nblk_supported(s.F) && rp_nblk_supported(s.Q);
}
static size_t cnn_rp_elems(const CnnShape& s) {
  if (!cnn_rp_ok(s)) return 0;
  return rp_image_elems(rp_nblk_for(s.F), rp_kblocks(s.W * 16 * cnn_ncb_x(s), false)) + rp_image_elems(rp_nblk_for(s.D), rp_kblocks(s.W * s.F, false)) +
         rp_image_elems(rp_nblk_for(s.Q), rp_kblocks(s.F, false)) + rp_image_elems(rp_nblk_for(s.F), rp_kblocks(s.Q, false)) +
         rp_image_elems(rp_nblk_for(s.D), rp_kblocks(s.W * 16 * cnn_ncb_dc(s), false));
}
// carve (and, in the forward, build in ONE launch) the four images
static int cnn_rp_images(const NrlCnnParams* p, const CnnShape& s, const CnnWs& w, bool fill, CnnRp* r, hipStream_t st) {
  r->on = cnn_rp_ok(s) && cur_engine() == ENGINE_BF16X3;
  if (!r->on) return NRL_OK;
  uint16_t* q = w.rp;
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  auto add = [&](RpImage* im, const float* src, int64_t sn, int64_t sk, int N, int K, bool conv_t) {
    const int nblk = rp_nblk_for(N);
    im->img = q; im->nblk = nblk; im->kblocks = rp_kblocks(K, false);
    if (fill) {
      RpImageJob* J = rp_jobs_add(&jobs, src, sn, sk, N, K, nullptr, q, nblk);
      if (conv_t) { J->conv_f = s.F; J->conv_w = s.W; }
    }
    q += rp_image_elems(nblk, im->kblocks);
  };
  if (cnn_x_planes_on(s)) {
    // c = window(x planes) Wc^T over the tap-padded reduction index k' = tap * (16 ncb_x) + d
    const int nblk = rp_nblk_for(s.F), dp = 16 * cnn_ncb_x(s);
    r->conv_f.img = q; r->conv_f.nblk = nblk; r->conv_f.kblocks = rp_kblocks(s.W * dp, false);
    if (fill) rp_jobs_add_kpad(&jobs, p->conv_weight, (int64_t)s.W * s.D, 1, s.F, s.W, s.D, dp, q, nblk);
    q += rp_image_elems(nblk, r->conv_f.kblocks);
  } else {
    add(&r->conv_f, p->conv_weight, (int64_t)s.W * s.D, 1, s.F, s.W * s.D, false);   // c = window(x) Wc^T
  }
  add(&r->conv_d, p->conv_weight, 0, 0, s.D, s.W * s.F, true);                     // dx = window'(dc) Wc (taps reversed)
  if (p->att_weight != nullptr) {
    add(&r->att_f, p->att_weight, s.F, 1, s.Q, s.F, false);                        // t = c W_a^T
    add(&r->att_d, p->att_weight, 1, s.F, s.F, s.Q, false);                        // dc = d_pre W_a
  }
  if (cnn_dc_planes_on(s)) {
    // dx = window'(dc planes) Wc over k' = t' * (16 ncb_dc) + f (taps reversed, tap-padded)
    const int nblk = rp_nblk_for(s.D), fp = 16 * cnn_ncb_dc(s);
    r->conv_dp.img = q; r->conv_dp.nblk = nblk; r->conv_dp.kblocks = rp_kblocks(s.W * fp, false);
    if (fill) {
      RpImageJob* J = rp_jobs_add_kpad(&jobs, p->conv_weight, 0, 0, s.D, s.W, s.F, fp, q, nblk);
      J->conv_f = s.F; J->conv_w = s.W;
    }
    q += rp_image_elems(nblk, r->conv_dp.kblocks);
  }
  if (fill) NRL_TRY(rp_jobs_launch(jobs, st));
  return NRL_OK;
}

static size_t conv_t_plane_elems(int D, int F, int W) { return (size_t)2 * D * ((W * F + 31) / 32 * 32); }

static size_t cnn_ws_floats(const CnnShape& s) {
  auto al = [](size_t n) { return align_up(n, 64); };
  const size_t slack_x = (size_t)s.W * s.D, slack_c = (size_t)s.W * s.F;
  size_t n = 0;
  n += al((size_t)s.M * s.D + 2 * slack_x) * 2;  // x, dx (dx needs no slack; kept symmetric)
  n += al((size_t)s.M * s.F + 2 * slack_c) * 2;  // c, dc
  n += al((size_t)s.M * s.Q);                    // t / d_pre
  n += al((size_t)s.M);                          // w
  n += al((split_weight_elems(s.F, s.W * s.D) + 1) / 2);
  n += al((split_weight_elems(s.Q, s.F) + 1) / 2);
  n += al((conv_t_plane_elems(s.D, s.F, s.W) + 1) / 2);
  n += al((cnn_rp_elems(s) + 1) / 2);
  if (cnn_conv_planes_ok(s)) {
    n += al(planes_from_rows_bytes(s.N, cnn_ncb_x(s), 2 * cnn_conv_kt(s)) / 4) + al(planes_from_rows_bytes(s.N, cnn_ncb_dc(s), 2 * cnn_conv_kt(s)) / 4);
    n += al(wgrad_planes_conv_scratch_floats(5, 2, cnn_ncb_dc(s), cnn_ncb_x(s), cnn_conv_planes_splits()));
    if ((s.F & 15) == 12 && s.F <= 304 && s.Q <= 224)      // (sized by shape alone: the switches are read per call)
      n += al(cnn_row_planes_bytes(s.M, cnn_ncb_c(s)) / 4) + al(cnn_row_planes_bytes(s.M, cnn_ncb_q(s)) / 4);
  }
  return n;
}

static int cnn_carve(void* ws, size_t ws_bytes, const CnnShape& s, CnnWs* o) {
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < cnn_ws_floats(s) * sizeof(float)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, cnn_ws_floats(s) * sizeof(float));
    return NRL_E_WORKSPACE;
  }
  float* p = (float*)ws;
  auto take = [&](size_t n) { float* r = p; p += align_up(n, 64); return r; };
  const size_t slack_x = (size_t)s.W * s.D, slack_c = (size_t)s.W * s.F;
  o->x = take((size_t)s.M * s.D + 2 * slack_x) + slack_x;
  o->dx = take((size_t)s.M * s.D + 2 * slack_x) + slack_x;
  o->c = take((size_t)s.M * s.F + 2 * slack_c) + slack_c;
  o->dc = take((size_t)s.M * s.F + 2 * slack_c) + slack_c;
  o->t = take((size_t)s.M * s.Q);
  o->w = take((size_t)s.M);
  o->planes_conv = reinterpret_cast<uint16_t*>(take((split_weight_elems(s.F, s.W * s.D) + 1) / 2));
  o->planes_att = reinterpret_cast<uint16_t*>(take((split_weight_elems(s.Q, s.F) + 1) / 2));
  o->planes_conv_t = reinterpret_cast<uint16_t*>(take((conv_t_plane_elems(s.D, s.F, s.W) + 1) / 2));
  o->rp = reinterpret_cast<uint16_t*>(take((cnn_rp_elems(s) + 1) / 2));
  o->xpl = o->dpl = nullptr;
  o->wsc = nullptr;
  o->cpl = o->tpl = nullptr;
  if (cnn_conv_planes_ok(s)) {
    o->xpl = reinterpret_cast<unsigned char*>(take(planes_from_rows_bytes(s.N, cnn_ncb_x(s), 2 * cnn_conv_kt(s)) / 4));
    o->dpl = reinterpret_cast<unsigned char*>(take(planes_from_rows_bytes(s.N, cnn_ncb_dc(s), 2 * cnn_conv_kt(s)) / 4));
    o->wsc = take(wgrad_planes_conv_scratch_floats(5, 2, cnn_ncb_dc(s), cnn_ncb_x(s), cnn_conv_planes_splits()));
    if ((s.F & 15) == 12 && s.F <= 304 && s.Q <= 224) {
      o->cpl = reinterpret_cast<unsigned char*>(take(cnn_row_planes_bytes(s.M, cnn_ncb_c(s)) / 4));
      o->tpl = reinterpret_cast<unsigned char*>(take(cnn_row_planes_bytes(s.M, cnn_ncb_q(s)) / 4));
    }
  }
  return NRL_OK;
}

static int cnn_check(const NrlCnnParams* p, int64_t n_news, int L, CnnShape* s) {
  NRL_REQUIRE(p != nullptr && p->conv_weight && p->conv_bias && p->att_weight && p->att_bias && p->att_query,
              "null CNN parameter pointer");
  NRL_REQUIRE(p->embed_dim > 0 && p->embed_dim % 4 == 0, "embed_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->num_filters > 0 && p->num_filters % 4 == 0, "num_filters must be a positive multiple of 4");
  NRL_REQUIRE(p->query_dim > 0 && p->query_dim % 4 == 0, "query_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->window >= 1 && p->window <= 7 && p->window % 2 == 1, "window must be odd and <= 7");
  NRL_REQUIRE(n_news >= 0 && L > 0, "bad news batch shape");
  NRL_REQUIRE((((uintptr_t)p->conv_weight | (uintptr_t)p->att_weight) & 15) == 0,
              "weight matrices must be 16-byte aligned");
  s->N = n_news; s->L = L; s->M = n_news * L;
  s->D = p->embed_dim; s->F = p->num_filters; s->W = p->window; s->Q = p->query_dim;
  s->pad = (p->window - 1) / 2;
  NRL_REQUIRE(s->M * (int64_t)(s->D > s->F ? s->D : s->F) < (1LL << 32), "dropout index space is 32-bit");
  return NRL_OK;
}

static SplitWeight planes_view(uint16_t* p, int N, int K) { return split_weight_view(p, N, K); }

// C = epi(A B) with A any k-contiguous fp32 accessor and B given both as fp32 k-major accessor (f32
// engine) and as pre-split planes [N][Kp] (bf16x3 engine)
template <class AOp, class BRc, class Epi>
static int gemm_any(const AOp& a, const BRc& b_rc, const uint16_t* hi, const uint16_t* lo, int64_t ldp,
                    const Epi& epi, int64_t M, int N, int K, hipStream_t st, const RpImage* rp = nullptr) {
  if (cur_engine() == ENGINE_BF16X3) {
    if (rp != nullptr && rp->img != nullptr) return rp_dispatch(a, *rp, epi, M, N, K, st);
    const KCSplit b{hi, lo, ldp, N};
    return launch_gemm_bf16x3_dma<X3_DMA_TILE>(a, b, epi, M, N, K, st);
  }
  return launch_gemm<NRL_TILE>(a, b_rc, epi, M, N, K, 1, st);
}

// dW (I, J) += A^T B, db (I) += colsum(A) with B any k-major accessor carrying the ones column
template <class BOp>
static int gemm_wgrad_any(const float* dy, int I, const BOp& b, int J, float* dW, float* db, int64_t M,
                          hipStream_t st) {
  const RCPlain a{dy, I, I, 0};
  const EpiAtomicWB epi{dW, J, db, J};
  if (cur_engine() == ENGINE_BF16X3) {
    auto splits = [&](int bm) {
      const int64_t tiles = ceil_div(I, bm) * ceil_div(J + 1, 160);
      int64_t sp = ceil_div(M, 1664);
      if (sp * tiles < 512) sp = ceil_div(512, tiles);
      const int64_t max_s = ceil_div(M, 256);
      return (int)(sp > max_s ? max_s : (sp < 1 ? 1 : sp));
    };
    // (the TRANSPOSED product (J + 1) x I on the wave-specialised kernel, so that its 256-row tiles cover the wide side
    //  of the 300 x 900 convolution, measured slower: 1.03 vs 0.95 ms per launch, LSTUR step 9.7 vs 9.5 ms)
    if (opt(O_X3_DMA)) return launch_gemm_bf16x3_dma_tn<2, 2, 2, 5, 2>(a, b, epi, I, J + 1, M, splits(64), st);
    return launch_gemm_bf16x3<X3_TILE_W>(a, b, epi, I, J + 1, M, splits(64), st);
  }
  return launch_gemm<NRL_TILE_W>(a, b, epi, I, J + 1, M, wgrad_splits(I, J + 1, M, 64, 160), st);
}

// dWc[f, t*D + d] += sum_m dc[m, f] x[m + t - pad, d] ; db_c += colsum(dc)
static int cnn_conv_wgrad(const CnnShape& s, const CnnWs& w, const float* dc, const float* x, int F, float* d_weight,
                          float* d_bias, hipStream_t st, bool x_planes_ready = false, bool dc_planes_ready = false) {
  const int KD = s.W * s.D;
  if (w.xpl != nullptr && cur_engine() == ENGINE_BF16X3) {
    const int ncb_x = cnn_ncb_x(s), ncb_dc = cnn_ncb_dc(s);
    const int kt = cnn_conv_kt(s);
    // (x planes: written by the forward's lookup when the convolution forward read them too; converted here otherwise)
    if (!x_planes_ready) NRL_TRY(launch_planes_from_rows(x, s.D, s.N, s.L, s.D, ncb_x, 2 * kt, true, w.xpl, st));
    if (!dc_planes_ready) NRL_TRY(launch_planes_from_rows(dc, F, s.N, s.L, F, ncb_dc, 2 * kt, false, w.dpl, st));
    // (round 6: 8-wave workgroups, two waves per SIMD over the same tile -- wgrad_planes_waves(), nrl_api_internal.h)
    const int sp = cnn_conv_planes_splits();
    if (wgrad_planes_waves() == 4) {
      if (kt == 1) return launch_wgrad_planes_conv<5, 2, 1, 4>(w.dpl, ncb_dc, w.xpl, ncb_x, s.N * 32, F, s.D, d_weight, d_bias, sp, st, w.wsc);
      return launch_wgrad_planes_conv<5, 2, 2, 4>(w.dpl, ncb_dc, w.xpl, ncb_x, s.N * 64, F, s.D, d_weight, d_bias, sp, st, w.wsc);
    }
    if (kt == 1) return launch_wgrad_planes_conv<5, 2, 1, 8>(w.dpl, ncb_dc, w.xpl, ncb_x, s.N * 32, F, s.D, d_weight, d_bias, sp, st, w.wsc);
    return launch_wgrad_planes_conv<5, 2, 2, 8>(w.dpl, ncb_dc, w.xpl, ncb_x, s.N * 64, F, s.D, d_weight, d_bias, sp, st, w.wsc);
  }
  return gemm_wgrad_any(dc, F, rc_window(x, s.D, s.L, s.pad, 1, (int64_t)KD), KD, d_weight, d_bias, s.M, st);
}

// Small-M GEMM of one GRU step, C += A B with split-K partial sums added atomically: with M = batch
// (~128 rows) there are only a handful of output tiles and the k-loop is latency bound (~1.8 us per
// 32-wide k-step measured), so K is cut into ~192-wide slices that run concurrently on other CUs.
template <class BRc>
static int gemm_step(const float* a, int64_t lda, const BRc& b_rc, const uint16_t* hi, const uint16_t* lo,
                     int64_t ldp, float* c, int64_t ldc, int64_t M, int N, int K, hipStream_t st) {
  const KCPlain A{a, lda, M};
  const EpiAtomicAdd epi{c, ldc};
  if (cur_engine() == ENGINE_BF16X3) {
    const int splits = (int)ceil_div(K, 192);
    return launch_gemm_bf16x3<X3_TILE_W>(A, KCSplit{hi, lo, ldp, N}, epi, M, N, K, splits, st);
  }
  const int splits = (int)ceil_div(K, 8 * GEMM_BK);
  return launch_gemm<NRL_TILE_W>(A, b_rc, epi, M, N, K, splits, st);
}

// ---- GRU -------------------------------------------------------------------------------------
struct GruShape {
  int64_t B, T;
  int Din, Hd;
};

struct GruWs {
  float *x_tm, *g, *gh, *ghn, *hs, *dgh, *dh, *dx_tm;
  uint16_t *planes_ih, *planes_hh;
};

static size_t gru_ws_floats(const GruShape& s) {
  auto al = [](size_t n) { return align_up(n, 64); };
  const size_t TB = (size_t)s.T * s.B;
  size_t n = 0;
  n += al(TB * s.Din) * 2;               // x_tm, dx_tm
  n += al(TB * 3 * s.Hd) * 2;            // g (gi -> gates -> dgi), dgh
  n += al((size_t)s.B * 3 * s.Hd);       // gh of the current step
  n += al(TB * s.Hd);                    // ghn
  n += al((TB + s.B) * s.Hd);            // hs[0..T]
  n += al((size_t)s.B * s.Hd);           // dh
  n += al((split_weight_elems(3 * s.Hd, s.Din) + 1) / 2);
  n += al((split_weight_elems(3 * s.Hd, s.Hd) + 1) / 2);
  return n;
}

static int gru_carve(void* ws, size_t ws_bytes, const GruShape& s, GruWs* o) {
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < gru_ws_floats(s) * sizeof(float)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, gru_ws_floats(s) * sizeof(float));
    return NRL_E_WORKSPACE;
  }
  float* p = (float*)ws;
  auto take = [&](size_t n) { float* r = p; p += align_up(n, 64); return r; };
  const size_t TB = (size_t)s.T * s.B;
  o->x_tm = take(TB * s.Din);
  o->dx_tm = take(TB * s.Din);
  o->g = take(TB * 3 * s.Hd);
  o->dgh = take(TB * 3 * s.Hd);
  o->gh = take((size_t)s.B * 3 * s.Hd);
  o->ghn = take(TB * s.Hd);
  o->hs = take((TB + s.B) * s.Hd);
  o->dh = take((size_t)s.B * s.Hd);
  o->planes_ih = reinterpret_cast<uint16_t*>(take((split_weight_elems(3 * s.Hd, s.Din) + 1) / 2));
  o->planes_hh = reinterpret_cast<uint16_t*>(take((split_weight_elems(3 * s.Hd, s.Hd) + 1) / 2));
  return NRL_OK;
}

static int gru_check(const NrlGruParams* p, int64_t B, int64_t T, GruShape* s) {
  NRL_REQUIRE(p != nullptr && p->weight_ih && p->weight_hh && p->bias_ih && p->bias_hh, "null GRU parameter pointer");
  NRL_REQUIRE(p->input_dim > 0 && p->input_dim % 4 == 0, "input_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->hidden_dim > 0 && p->hidden_dim % 4 == 0, "hidden_dim must be a positive multiple of 4");
  NRL_REQUIRE(B > 0 && T > 0, "bad GRU batch shape");
  NRL_REQUIRE((((uintptr_t)p->weight_ih | (uintptr_t)p->weight_hh) & 15) == 0, "weight matrices must be 16-byte aligned");
  s->B = B; s->T = T; s->Din = p->input_dim; s->Hd = p->hidden_dim;
  return NRL_OK;
}

}  // namespace nrl

using namespace nrl;

extern "C" {

size_t nrl_cnn_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim, int32_t num_filters,
                                       int32_t window, int32_t query_dim) {
  CnnShape s;
  s.N = n_news; s.L = seq_len; s.M = n_news * seq_len; s.D = embed_dim; s.F = num_filters; s.W = window;
  s.Q = query_dim; s.pad = (window - 1) / 2;
  return cnn_ws_floats(s) * sizeof(float);
}

int nrl_cnn_encoder_fwd(const NrlCnnParams* p, const float* emb_table, int64_t vocab, const int64_t* ids,
                        int64_t n_news, int32_t seq_len, double p_drop, uint64_t seed, uint32_t stream0,
                        int32_t save_for_backward, float* out, void* ws, size_t ws_bytes, void* stream) {
  (void)vocab;
  CnnShape s;
  NRL_TRY(cnn_check(p, n_news, seq_len, &s));
  NRL_REQUIRE(emb_table && ids && out, "cnn_encoder_fwd: null argument");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "p_drop must be in [0, 1)");
  if (s.M == 0) return NRL_OK;
  CnnWs w;
  NRL_TRY(cnn_carve(ws, ws_bytes, s, &w));
  hipStream_t st = (hipStream_t)stream;
  const Dropout drop1 = make_dropout(p_drop, seed, stream0), drop2 = make_dropout(p_drop, seed, stream0 + 1);
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  const int KD = s.W * s.D;
  SplitWeight sc = planes_view(w.planes_conv, s.F, KD), sa = planes_view(w.planes_att, s.Q, s.F);
  CnnRp rp;
  NRL_TRY(cnn_rp_images(p, s, w, true, &rp, st));
  if (cur_engine() == ENGINE_BF16X3 && !rp.on) {
    NRL_TRY(split_weight(p->conv_weight, s.F, KD, w.planes_conv, &sc, st));
    NRL_TRY(split_weight(p->att_weight, s.Q, s.F, w.planes_att, &sa, st));
  }
  if (cnn_x_planes_on(s) && rp.on && w.xpl != nullptr) {
    // x = dropout(emb[ids]) as planes only; c = dropout(relu(conv(x) + b)) straight from them
    const int ncb_x = cnn_ncb_x(s), nrb = 2 * cnn_conv_kt(s);
    NRL_REQUIRE(s.M * (int64_t)s.D < (1LL << 32), "dropout index space is 32-bit");
    NRL_TRY(launch_embedding_rows_planes(emb_table, ids, s.N, s.L, s.D, ncb_x, nrb, drop1, w.xpl, st));
    const KCWindowPlanes a{w.xpl, s.M, s.L, nrb, ncb_x, s.W, s.pad};
    const EpiLinear epi_c{w.c, s.F, p->conv_bias, 2, drop2, s.F};
    if (save_for_backward != 0 && cnn_aa_planes_on(s) && w.cpl != nullptr) {   // (an evaluation forward has no use for the planes)
      // c as fp32 rows (the pooling kernels, the ReLU gate of the backward) AND as planes with the ones column
      const int ncb_c = cnn_ncb_c(s);
      if (s.M % 32 != 0)   // rows past M in the last 32-row k-tile of the weight gradient
        NRL_HIP(hipMemsetAsync(w.cpl + (s.M / 32) * 2 * ncb_c * 1024, 0, (size_t)2 * ncb_c * 1024, st));
      NRL_TRY(rp_dispatch(a, rp.conv_f, EpiLinearPlanes{epi_c, w.cpl, ncb_c}, s.M, s.F, s.W * 16 * ncb_x, st));
      // t = tanh(c W_a^T + b_a) from the planes
      NRL_TRY(rp_dispatch(KCPlanesG{w.cpl, s.M, ncb_c}, rp.att_f, EpiLinear{w.t, s.Q, p->att_bias, 1, nodrop, s.Q}, s.M, s.Q, s.F, st));
      NRL_TRY(pool_fwd(w.t, p->att_query, w.c, s.N, s.L, s.Q, s.F, w.w, out, st));
      return NRL_OK;
    }
    NRL_TRY(rp_dispatch(a, rp.conv_f, epi_c, s.M, s.F, s.W * 16 * ncb_x, st));
  } else {
  // x = dropout(emb[ids])                                  (text.py:165-166)
  NRL_TRY(embedding_rows_fwd(emb_table, ids, s.M, s.D, drop1, 0, w.x, st));
  // c = dropout(relu(conv(x) + b))                         (text.py:169-171), K = W*D over overlapping rows
  {
    const KCWindow a{w.x, s.M, s.D, s.L, s.W, s.pad};
    const EpiLinear epi{w.c, s.F, p->conv_bias, 2, drop2, s.F};
    if (cur_engine() == ENGINE_BF16X3) {
      NRL_TRY(gemm_any(a, KCPlain{p->conv_weight, KD, s.F}, sc.hi, sc.lo, sc.ld, epi, s.M, s.F, KD, st,
                       rp.on ? &rp.conv_f : nullptr));
    } else {
      NRL_TRY(launch_gemm<NRL_TILE>(a, KCPlain{p->conv_weight, KD, s.F}, epi, s.M, s.F, KD, 1, st));
    }
  }
  }
  // t = tanh(c W_a^T + b_a); w = softmax(t . q_a); out = sum w c      (attention.py:34-40)
  NRL_TRY(gemm_fwd(KCPlain{w.c, s.F, s.M}, p->att_weight, sa, EpiLinear{w.t, s.Q, p->att_bias, 1, nodrop, s.Q}, s.M,
                   s.Q, s.F, true, st, rp.on ? &rp.att_f : nullptr));
  NRL_TRY(pool_fwd(w.t, p->att_query, w.c, s.N, s.L, s.Q, s.F, w.w, out, st));
  return NRL_OK;
}

int nrl_cnn_encoder_bwd(const NrlCnnParams* p, const NrlCnnGrads* g, float* d_emb_table, int64_t vocab,
                        const int64_t* ids, const int64_t* sorted_positions, int64_t n_news, int32_t seq_len,
                        double p_drop, uint64_t seed, uint32_t stream0, const float* d_out, void* ws,
                        size_t ws_bytes, void* stream) {
  (void)vocab;
  CnnShape s;
  NRL_TRY(cnn_check(p, n_news, seq_len, &s));
  NRL_REQUIRE(g && g->conv_weight && g->conv_bias && g->att_weight && g->att_bias && g->att_query,
              "null CNN gradient pointer");
  NRL_REQUIRE(d_emb_table && ids && d_out, "cnn_encoder_bwd: null argument");
  if (s.M == 0) return NRL_OK;
  CnnWs w;
  NRL_TRY(cnn_carve(ws, ws_bytes, s, &w));
  hipStream_t st = (hipStream_t)stream;
  const Dropout drop1 = make_dropout(p_drop, seed, stream0), drop2 = make_dropout(p_drop, seed, stream0 + 1);
  const int KF = s.W * s.F;
  const SplitWeight sa = planes_view(w.planes_att, s.Q, s.F);
  CnnRp rp;
  NRL_TRY(cnn_rp_images(p, s, w, false, &rp, st));        // built by the forward; weights unchanged since
  const bool x_pl = cnn_x_planes_on(s) && rp.on && w.xpl != nullptr;
  const bool dc_pl = x_pl && cnn_dc_planes_on(s) && sorted_positions != nullptr && w.dpl != nullptr;
  // (the forward wrote the c planes under cnn_aa_planes_on alone; a backward that cannot use them still has c and t as fp32)
  const bool aa_pl = dc_pl && cnn_aa_planes_on(s) && w.cpl != nullptr && w.tpl != nullptr;
  const int ncb_q = cnn_ncb_q(s);
  // additive attention backward: t -> d_pre (in place, or as planes only), dq_a
  if (aa_pl && s.M % 32 != 0) NRL_HIP(hipMemsetAsync(w.tpl + (s.M / 32) * 2 * ncb_q * 1024, 0, (size_t)2 * ncb_q * 1024, st));
  NRL_TRY(pool_bwd_pre(d_out, w.c, w.w, w.t, p->att_query, g->att_query, s.N, s.L, s.Q, s.F, st, aa_pl ? w.tpl : nullptr));
  // dc = (d_pre W_a + w * d_out) * dropout2 * [c > 0]       (pre-ReLU gradient)
  if (dc_pl) {
    const EpiPoolBwdNewsPlanes epi{EpiPoolBwd{nullptr, s.F, w.w, d_out, s.L, drop2, w.c}, w.dpl, cnn_ncb_dc(s), 2 * cnn_conv_kt(s), s.L, s.F};
    NRL_TRY(launch_planes_zero_pad_rows(w.dpl, s.N, s.L, cnn_ncb_dc(s), 2 * cnn_conv_kt(s), st));
    if (aa_pl) NRL_TRY(rp_dispatch(KCPlanesG{w.tpl, s.M, ncb_q}, rp.att_d, epi, s.M, s.F, s.Q, st));
    else
    NRL_TRY(rp_dispatch(KCPlain{w.t, s.Q, s.M}, rp.att_d, epi, s.M, s.F, s.Q, st));
  } else {
  NRL_TRY(gemm_dgrad(w.t, p->att_weight, sa, EpiPoolBwd{w.dc, s.F, w.w, d_out, s.L, drop2, w.c}, s.M, s.Q, s.F, st,
                     rp.on ? &rp.att_d : nullptr));
  }
  // dW_a += d_pre^T c ; db_a += colsum(d_pre)
  if (aa_pl) {
    // both operands as planes over the same rows (the ones column of the c planes gives the bias gradient); split-K partial tiles
    // in the fp32 dc buffer, which the planes path never writes
    static const int sp = [] { const char* e = getenv("NRL_WGRAD_PLANES_AA_SPLITS"); return e ? atoi(e) : 128; }();
    const int ncb_c = cnn_ncb_c(s);
    const size_t need = wgrad_planes_g_scratch_floats(7, 5, ncb_q, ncb_c, sp);
    float* const sc = opt(O_WGRAD_2STEP) && need <= (size_t)s.M * s.F ? w.dc : nullptr;
    NRL_TRY((wgrad_planes_g_dispatch<7, 5>(w.tpl, ncb_q, w.cpl, ncb_c, (s.M + 31) / 32 * 32, s.Q, s.F + 1,
                                         EpiAtomicWB{g->att_weight, s.F, g->att_bias, s.F}, sp, st, sc)));
  } else {
  NRL_TRY(gemm_wgrad(w.t, s.Q, w.c, s.F, g->att_weight, g->att_bias, s.M, st));
  }
  NRL_TRY(cnn_conv_wgrad(s, w, w.dc, w.x, s.F, g->conv_weight, g->conv_bias, st, x_pl, dc_pl));
  // dx[m, d] = dropout1 * sum_{t', f} dc[m + t' - pad', f] Wc[f, (W-1-t')*D + d]
  {
    const KCWindow a{w.dc, s.M, s.F, s.L, s.W, s.W - 1 - s.pad};
    const RCConvT b_rc{p->conv_weight, s.F, s.D, s.W, (int64_t)s.D};
    const int Kp = (KF + 31) / 32 * 32;
    uint16_t* hi = w.planes_conv_t;
    uint16_t* lo = hi + 32;
    const RpImage* rpd = rp.on ? &rp.conv_d : nullptr;
    if (cur_engine() == ENGINE_BF16X3 && !rp.on) {
      const int64_t total = (int64_t)s.D * Kp;
      hipLaunchKernelGGL(split_conv_weight_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                         p->conv_weight, s.F, s.D, s.W, Kp, hi);
      NRL_LAUNCH_CHECK();
    }
    static const bool live_env = [] { const char* e = getenv("NRL_LIVE_ROWS"); return !(e != nullptr && e[0] == '0'); }();
    if (sorted_positions != nullptr && rpd != nullptr && live_env) {
      // dx for the LIVE token rows only (id != 0), compact in position order: the padding id has no table gradient and
      // nothing else reads dx.  The list of live positions goes into the tanh / d_pre buffer (dead by now).
      NRL_REQUIRE((size_t)s.M * s.Q >= live_compact_ints(s.M), "cnn_encoder_bwd: scratch for the live-row list");
      const int32_t *list = nullptr, *cidx = nullptr, *n_live = nullptr;
      NRL_TRY(live_compact(ids, s.M, reinterpret_cast<int32_t*>(w.t), &list, &cidx, &n_live, st));
      if (dc_pl) {
        const int ncb_dc = cnn_ncb_dc(s);
        const KCWindowPlanes ap{w.dpl, s.M, s.L, 2 * cnn_conv_kt(s), ncb_dc, s.W, s.W - 1 - s.pad};
        NRL_TRY(rp_dispatch(KCWindowLivePlanes{ap, list, n_live}, rp.conv_dp, EpiDxLive{w.dx, s.D, drop1, list}, s.M, s.D,
                            s.W * 16 * ncb_dc, st));
      } else {
      NRL_TRY(rp_dispatch(KCWindowLive{a, list, n_live}, *rpd, EpiDxLive{w.dx, s.D, drop1, list}, s.M, s.D, KF, st));
      }
      NRL_TRY(embedding_grad_sorted(w.dx, ids, sorted_positions, s.M, s.D, d_emb_table, st, cidx));
    } else if (sorted_positions != nullptr) {
      NRL_TRY(gemm_any(a, b_rc, hi, lo, 2 * Kp, EpiLinear{w.dx, s.D, nullptr, 0, drop1, s.D}, s.M, s.D, KF, st, rpd));
      NRL_TRY(embedding_grad_sorted(w.dx, ids, sorted_positions, s.M, s.D, d_emb_table, st));
    } else {
      NRL_TRY(gemm_any(a, b_rc, hi, lo, 2 * Kp, EpiScatter{d_emb_table, ids, s.D, drop1}, s.M, s.D, KF, st, rpd));
    }
  }
  return NRL_OK;
}

// ---- CNN + multi-head self-attention + additive attention text encoder (CenNewsRec) ----------------------
size_t nrl_cnn_mhsa_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim, int32_t num_filters,
                                            int32_t window, int32_t num_heads, int32_t query_dim) {
  CnnShape s;
  s.N = n_news; s.L = seq_len; s.M = n_news * seq_len; s.D = embed_dim; s.F = num_filters; s.W = window;
  s.Q = query_dim; s.pad = (window - 1) / 2;
  return (cnn_ws_floats(s) + align_up(block_ws_floats(s.M, num_filters, query_dim, num_heads, false), 64)) * sizeof(float);
}

static int cnn_mhsa_carve(const NrlCnnParams* cp, const NrlBlockParams* bp, int64_t n_news, int L, void* ws,
                          size_t ws_bytes, CnnShape* cs, CnnWs* cw, BlockShape* bs, BlockWs* bw) {
  NRL_REQUIRE(cp != nullptr && cp->conv_weight && cp->conv_bias, "null CNN parameter pointer");
  NRL_REQUIRE(cp->embed_dim > 0 && cp->embed_dim % 4 == 0 && cp->num_filters > 0 && cp->num_filters % 4 == 0,
              "embed_dim / num_filters must be positive multiples of 4");
  NRL_REQUIRE(cp->window >= 1 && cp->window <= 7 && cp->window % 2 == 1, "window must be odd and <= 7");
  NRL_TRY(check_params(bp));
  NRL_REQUIRE(bp->embed_dim == cp->num_filters, "the attention block runs on num_filters features");
  NRL_REQUIRE(n_news >= 0 && L > 0, "bad news batch shape");
  cs->N = n_news; cs->L = L; cs->M = n_news * L; cs->D = cp->embed_dim; cs->F = cp->num_filters; cs->W = cp->window;
  cs->Q = bp->query_dim; cs->pad = (cp->window - 1) / 2;
  NRL_REQUIRE(cs->M * (int64_t)(cs->D > cs->F ? cs->D : cs->F) < (1LL << 32), "dropout index space is 32-bit");
  *bs = news_shape(bp, n_news, L);
  bs->pad_rows = 0;   // (the padded fragment-block planes belong to the fused NRMS news path only)
  const size_t cnn_bytes = cnn_ws_floats(*cs) * sizeof(float);
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < cnn_bytes + block_ws_floats(cs->M, cs->F, cs->Q, bp->num_heads, false) * sizeof(float)) {
    set_error("workspace too small: %zu bytes", ws_bytes);
    return NRL_E_WORKSPACE;
  }
  NRL_TRY(cnn_carve(ws, cnn_bytes, *cs, cw));
  NRL_TRY(carve_ws((unsigned char*)ws + cnn_bytes, ws_bytes - cnn_bytes, *bs, false, bw));
  return NRL_OK;
}

int nrl_cnn_mhsa_encoder_fwd(const NrlCnnParams* cp, const NrlBlockParams* bp, const float* emb_table, int64_t vocab,
                             const int64_t* ids, int64_t n_news, int32_t seq_len, double p_drop, uint64_t seed,
                             uint32_t stream0, int32_t save_for_backward, float* out, void* ws, size_t ws_bytes,
                             void* stream) {
  (void)vocab;
  CnnShape cs; CnnWs cw; BlockShape bs; BlockWs bw;
  NRL_TRY(cnn_mhsa_carve(cp, bp, n_news, seq_len, ws, ws_bytes, &cs, &cw, &bs, &bw));
  const EngineScope engine_scope(bp->gemm_engine);
  const OptScope opt_scope(bp->options);
  NRL_REQUIRE(emb_table && ids && out, "cnn_mhsa_encoder_fwd: null argument");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "p_drop must be in [0, 1)");
  if (cs.M == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const Dropout d1 = make_dropout(p_drop, seed, stream0), d2 = make_dropout(p_drop, seed, stream0 + 1),
                d3 = make_dropout(p_drop, seed, stream0 + 2);
  const int KD = cs.W * cs.D;
  SplitWeight sc = planes_view(cw.planes_conv, cs.F, KD);
  if (cur_engine() == ENGINE_BF16X3) NRL_TRY(split_weight(cp->conv_weight, cs.F, KD, cw.planes_conv, &sc, st));
  // x = dropout(emb[ids]); c = dropout(relu(conv1d(x)))              (text.py:293-299)
  NRL_TRY(embedding_rows_fwd(emb_table, ids, cs.M, cs.D, d1, 0, cw.x, st));
  {
    const KCWindow a{cw.x, cs.M, cs.D, cs.L, cs.W, cs.pad};
    const EpiLinear epi{cw.c, cs.F, cp->conv_bias, 2, d2, cs.F};
    if (cur_engine() == ENGINE_BF16X3) {
      NRL_TRY(gemm_any(a, KCPlain{cp->conv_weight, KD, cs.F}, sc.hi, sc.lo, sc.ld, epi, cs.M, cs.F, KD, st));
    } else {
      NRL_TRY(launch_gemm<NRL_TILE>(a, KCPlain{cp->conv_weight, KD, cs.F}, epi, cs.M, cs.F, KD, 1, st));
    }
  }
  // self-attention over the tokens of each news, dropout, additive attention     (text.py:301-307)
  return block_fwd(bp, KCPlain{cw.c, cs.F, cs.M}, bs, bw, d3, save_for_backward != 0, false, out, st);
}

int nrl_cnn_mhsa_encoder_bwd(const NrlCnnParams* cp, const NrlCnnGrads* cg, const NrlBlockParams* bp,
                             const NrlBlockGrads* bg, float* d_emb_table, int64_t vocab, const int64_t* ids,
                             const int64_t* sorted_positions, int64_t n_news, int32_t seq_len, double p_drop,
                             uint64_t seed, uint32_t stream0, const float* d_out, void* ws, size_t ws_bytes,
                             void* stream) {
  (void)vocab;
  CnnShape cs; CnnWs cw; BlockShape bs; BlockWs bw;
  NRL_TRY(cnn_mhsa_carve(cp, bp, n_news, seq_len, ws, ws_bytes, &cs, &cw, &bs, &bw));
  const EngineScope engine_scope(bp->gemm_engine);
  const OptScope opt_scope(bp->options);
  NRL_TRY(check_grads(bg));
  NRL_REQUIRE(cg && cg->conv_weight && cg->conv_bias, "null CNN gradient pointer");
  NRL_REQUIRE(d_emb_table && ids && d_out, "cnn_mhsa_encoder_bwd: null argument");
  if (cs.M == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const Dropout d1 = make_dropout(p_drop, seed, stream0), d2 = make_dropout(p_drop, seed, stream0 + 1),
                d3 = make_dropout(p_drop, seed, stream0 + 2);
  const int KF = cs.W * cs.F, F = cs.F;
  BlockPlanes planes;
  NRL_TRY(block_planes(bp, bs, bw, false, &planes, st));
  NRL_TRY(block_bwd_phase1(bp, bg, bs, bw, planes, d3, d_out, st));
  // dc = (dqkv W_in) * dropout2 * [c > 0]   (gradient at the conv pre-activation)
  NRL_TRY(gemm_dgrad(bw.dqkv, bp->in_proj_weight, planes.in, EpiLinear{cw.dc, F, nullptr, 0, d2, F, cw.c}, cs.M, 3 * F,
                     F, st));
  NRL_TRY(block_bwd_phase2(bg, cw.c, bs, bw, st));
  // conv weight / bias gradient, conv dgrad -> dx -> table gradient (as nrl_cnn_encoder_bwd)
  NRL_TRY(cnn_conv_wgrad(cs, cw, cw.dc, cw.x, F, cg->conv_weight, cg->conv_bias, st));
  {
    const KCWindow a{cw.dc, cs.M, F, cs.L, cs.W, cs.W - 1 - cs.pad};
    const RCConvT b_rc{cp->conv_weight, F, cs.D, cs.W, (int64_t)cs.D};
    const int Kp = (KF + 31) / 32 * 32;
    uint16_t* hi = cw.planes_conv_t;
    uint16_t* lo = hi + 32;
    if (cur_engine() == ENGINE_BF16X3) {
      const int64_t total = (int64_t)cs.D * Kp;
      hipLaunchKernelGGL(split_conv_weight_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                         cp->conv_weight, F, cs.D, cs.W, Kp, hi);
      NRL_LAUNCH_CHECK();
    }
    if (sorted_positions != nullptr) {
      NRL_TRY(gemm_any(a, b_rc, hi, lo, 2 * Kp, EpiLinear{cw.dx, cs.D, nullptr, 0, d1, cs.D}, cs.M, cs.D, KF, st));
      NRL_TRY(embedding_grad_sorted(cw.dx, ids, sorted_positions, cs.M, cs.D, d_emb_table, st));
    } else {
      NRL_TRY(gemm_any(a, b_rc, hi, lo, 2 * Kp, EpiScatter{d_emb_table, ids, cs.D, d1}, cs.M, cs.D, KF, st));
    }
  }
  return NRL_OK;
}

int nrl_embedding_rows_fwd(const float* table, const int64_t* ids, int64_t n_ids, int32_t dim, double p_row,
                           uint64_t seed, uint32_t stream_id, float* out, void* stream) {
  NRL_REQUIRE(table && ids && out && n_ids >= 0 && dim > 0, "embedding_rows_fwd: bad arguments");
  NRL_REQUIRE(p_row >= 0.0 && p_row < 1.0, "p_row must be in [0, 1)");
  return embedding_rows_fwd(table, ids, n_ids, dim, make_dropout(p_row, seed, stream_id), 1, out,
                            (hipStream_t)stream);
}

int nrl_embedding_rows_bwd(const float* d_out, const int64_t* ids, int64_t n_ids, int32_t dim, double p_row,
                           uint64_t seed, uint32_t stream_id, float* d_table, void* stream) {
  NRL_REQUIRE(d_out && ids && d_table && n_ids >= 0 && dim > 0, "embedding_rows_bwd: bad arguments");
  NRL_REQUIRE(p_row >= 0.0 && p_row < 1.0, "p_row must be in [0, 1)");
  return embedding_rows_bwd(d_out, ids, n_ids, dim, make_dropout(p_row, seed, stream_id), d_table,
                            (hipStream_t)stream);
}

size_t nrl_gru_workspace_bytes(int64_t batch, int64_t max_len, int32_t input_dim, int32_t hidden_dim) {
  GruShape s{batch, max_len, input_dim, hidden_dim};
  return gru_ws_floats(s) * sizeof(float);
}

int nrl_gru_fwd(const NrlGruParams* p, const float* hist, const int64_t* lengths, const float* h0, int64_t batch,
                int64_t max_len, int32_t save_for_backward, float* out, void* ws, size_t ws_bytes, void* stream) {
  (void)save_for_backward;
  GruShape s;
  NRL_TRY(gru_check(p, batch, max_len, &s));
  NRL_REQUIRE(hist && lengths && out, "gru_fwd: null argument");
  GruWs w;
  NRL_TRY(gru_carve(ws, ws_bytes, s, &w));
  hipStream_t st = (hipStream_t)stream;
  const int Hd = s.Hd, Din = s.Din;
  const int64_t B = s.B, T = s.T;
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  SplitWeight si = planes_view(w.planes_ih, 3 * Hd, Din), sh = planes_view(w.planes_hh, 3 * Hd, Hd);
  if (cur_engine() == ENGINE_BF16X3) {
    NRL_TRY(split_weight(p->weight_ih, 3 * Hd, Din, w.planes_ih, &si, st));
    NRL_TRY(split_weight(p->weight_hh, 3 * Hd, Hd, w.planes_hh, &sh, st));
  }
  // time-major copy of the input: step t is the contiguous slab x_tm[t] (B, Din)
  NRL_TRY(transpose01(hist, B, T, Din, w.x_tm, st));
  // gi = x W_ih^T + b_ih for all steps at once
  NRL_TRY(gemm_fwd(KCPlain{w.x_tm, Din, T * B}, p->weight_ih, si, EpiLinear{w.g, 3 * Hd, p->bias_ih, 0, nodrop, 3 * Hd},
                   T * B, 3 * Hd, Din, false, st));
  if (h0 != nullptr) {
    NRL_HIP(hipMemcpyAsync(w.hs, h0, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    NRL_HIP(hipMemsetAsync(w.hs, 0, (size_t)B * Hd * sizeof(float), st));
  }
  if (cur_engine() == ENGINE_BF16X3 && gru_step_fused_ok(Hd)) {
    // the whole recurrence in one cooperative launch with W_hh resident in registers, if the grid is co-resident ...
    bool done = false;
    NRL_TRY(gru_persistent_fwd(w.hs, sh.hi, sh.ld, w.g, p->bias_hh, lengths, T, B, Hd, w.ghn,
                               reinterpret_cast<unsigned*>(w.gh), st, &done));   // (w.gh is idle on the fused path)
    // ... else one launch per step: product + gates fused (nrl_gru_fused.h)
    for (int64_t t = 0; t < T && !done; ++t) {
      float* h_prev = w.hs + t * B * Hd;
      NRL_TRY(gru_step_fused(h_prev, sh.hi, sh.ld, w.g + t * B * 3 * Hd, p->bias_hh, lengths, (int)t, B, Hd, true,
                             w.ghn + t * B * Hd, h_prev + B * Hd, st));
    }
    NRL_HIP(hipMemcpyAsync(out, w.hs + T * B * Hd, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, st));
    return NRL_OK;
  }
  NRL_HIP(hipMemsetAsync(w.gh, 0, (size_t)B * 3 * Hd * sizeof(float), st));
  for (int64_t t = 0; t < T; ++t) {
    float* h_prev = w.hs + t * B * Hd;
    float* g_t = w.g + t * B * 3 * Hd;
    // gh += h_{t-1} W_hh^T (split-K; the gate kernel adds b_hh and clears gh for the next step)
    NRL_TRY(gemm_step(h_prev, Hd, KCPlain{p->weight_hh, Hd, 3 * Hd}, sh.hi, sh.lo, sh.ld, w.gh, 3 * Hd, B, 3 * Hd, Hd,
                      st));
    NRL_TRY(gru_gate_fwd(g_t, w.gh, p->bias_hh, h_prev, lengths, (int)t, B, Hd, g_t, w.ghn + t * B * Hd,
                         h_prev + B * Hd, st));
  }
  NRL_HIP(hipMemcpyAsync(out, w.hs + T * B * Hd, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, st));
  return NRL_OK;
}

int nrl_gru_bwd(const NrlGruParams* p, const NrlGruGrads* g, const float* hist, const int64_t* lengths,
                const float* h0, int64_t batch, int64_t max_len, const float* d_out, float* d_hist, float* d_h0,
                void* ws, size_t ws_bytes, void* stream) {
  (void)hist; (void)h0;
  GruShape s;
  NRL_TRY(gru_check(p, batch, max_len, &s));
  NRL_REQUIRE(g && g->weight_ih && g->weight_hh && g->bias_ih && g->bias_hh, "null GRU gradient pointer");
  NRL_REQUIRE(lengths && d_out && d_hist, "gru_bwd: null argument");
  GruWs w;
  NRL_TRY(gru_carve(ws, ws_bytes, s, &w));
  hipStream_t st = (hipStream_t)stream;
  const int Hd = s.Hd, Din = s.Din;
  const int64_t B = s.B, T = s.T;
  const SplitWeight si = planes_view(w.planes_ih, 3 * Hd, Din), sh = planes_view(w.planes_hh, 3 * Hd, Hd);
  NRL_HIP(hipMemcpyAsync(w.dh, d_out, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, st));
  for (int64_t t = T - 1; t >= 0; --t) {
    float* g_t = w.g + t * B * 3 * Hd;
    float* dgh_t = w.dgh + t * B * 3 * Hd;
    NRL_TRY(gru_gate_bwd(g_t, w.ghn + t * B * Hd, w.hs + t * B * Hd, lengths, (int)t, B, Hd, w.dh, g_t, dgh_t, st));
    // dh_{t-1} = z * dh_t + dgh_t W_hh (split-K partial sums added onto the direct term)
    NRL_TRY(gemm_step(dgh_t, 3 * Hd, RCPlain{p->weight_hh, Hd, Hd, 0}, sh.hi_t, sh.lo_t, sh.ld_t, w.dh, Hd, B, Hd, 3 * Hd,
                      st));
  }
  if (d_h0 != nullptr)
    NRL_HIP(hipMemcpyAsync(d_h0, w.dh, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, st));
  // dX = dgi W_ih (time-major), then back to (B, T, Din)
  NRL_TRY(gemm_dgrad(w.g, p->weight_ih, si, EpiStore{w.dx_tm, Din}, T * B, 3 * Hd, Din, st));
  NRL_TRY(transpose01(w.dx_tm, T, B, Din, d_hist, st));
  // dW_ih += dgi^T X ; db_ih += colsum(dgi) ; dW_hh += dgh^T H_prev ; db_hh += colsum(dgh)
  NRL_TRY(gemm_wgrad(w.g, 3 * Hd, w.x_tm, Din, g->weight_ih, g->bias_ih, T * B, st));
  NRL_TRY(gemm_wgrad(w.dgh, 3 * Hd, w.hs, Hd, g->weight_hh, g->bias_hh, T * B, st));
  return NRL_OK;
}

}  // extern "C"
