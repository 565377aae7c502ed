// C ABI of the gfx950 NRMS hot path (declarations + contract: include/newsreclib_amd.h).
//
// Both encoders are the same "MHSA + additive attention" block (text.py:218-219 /
// user/nrms.py:27-30); they differ in how the block's input rows are produced (embedding gather
// with dropout vs. a dense history tensor), in which axis the tiny attentions run over, and in
// where the input gradient goes (scatter-add into the table vs. a dense d_hist).
#include "nrl_api_internal.h"

namespace nrl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

ProfState g_prof;
std::atomic<int> g_default_engine{ENGINE_BF16X3};
thread_local int t_engine = -1;
thread_local int64_t t_opts = -1;

static const char* const kOptName[O_COUNT] = {"news_fused", "news_fused_bwd", "news_attn_mfma", "news_planes", "news_od_planes",
                                              "news_aa_planes", "wgrad_2step", "wgrad_ws", "rowpanel", "x3_dma", "news_tail",
                                              "news_tail_bwd", "user_fork", "news_fork", "news_qkv_planes", "news_pad_share", "news_tail_od", "user_proj"};
static const char* const kOptEnv[O_COUNT] = {"NRL_NEWS_FUSED", "NRL_NEWS_FUSED_BWD", "NRL_NEWS_ATTN_MFMA", "NRL_NEWS_PLANES",
                                             "NRL_NEWS_OD_PLANES", "NRL_NEWS_AA_PLANES", "NRL_WGRAD_2STEP", "NRL_WGRAD_WS",
                                             "NRL_ROWPANEL", "NRL_X3_DMA", "NRL_NEWS_TAIL", "NRL_NEWS_TAIL_BWD", "NRL_USER_FORK",
                                             "NRL_NEWS_FORK", "NRL_NEWS_QKV_PLANES", "NRL_NEWS_PAD_SHARE", "NRL_NEWS_TAIL_OD", "NRL_USER_PROJ"};
std::atomic<uint32_t> g_opt_default{[] {
  uint32_t m = 0;
  for (int i = 0; i < O_COUNT; ++i) {
    // (off: the measured losers; news_fused_bwd and news_tail_od are retired bits, see opt_retired.  news_fork: neutral in rounds 3-5
    //  (3.02 vs 3.03 ms), ON since round 6 -- with the back-half weight gradients as 8-wave workgroups running them beside the
    //  out-projection dgrad / token-attention backward chain is worth 37-41 us of the B = 128 step, profiles/r06_ab.txt)
    const bool dflt = i != O_NEWS_FUSED_BWD && i != O_USER_FORK && i != O_NEWS_TAIL_OD;
    const char* e = getenv(kOptEnv[i]);
    bool v = e == nullptr ? dflt : (dflt ? e[0] != '0' : e[0] == '1');
    if (opt_retired(i)) v = false;           // (bits kept for the mask's layout; their kernels left the library in ABI v14)
    m |= v ? (1u << i) : 0u;
  }
  return m;
}()};

// ---- two internal side streams per (host thread, device) for the fork / join inside nrl_user_encoder_bwd -----------
// The user encoder works on B * H rows (6400 at B = 128): each of its launches occupies 13-100 of the 256 CUs and lasts one
// panel's latency, so independent ones are run side by side.  Everything forked is joined back into the caller's stream
// before the entry point returns: the caller still sees one stream-ordered, asynchronous call; no sync, no allocation after
// the first call of a thread on a device.
struct ForkSet {
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
};
static int fork_set(ForkSet** out) {
  thread_local ForkSet sets[64];
  int dev = 0;
  NRL_HIP(hipGetDevice(&dev));
  ForkSet& f = sets[dev & 63];
  if (f.fork == nullptr) {
    for (int i = 0; i < 2; ++i) {
      NRL_HIP(hipStreamCreateWithFlags(&f.s[i], hipStreamNonBlocking));
      NRL_HIP(hipEventCreateWithFlags(&f.join[i], hipEventDisableTiming));
    }
    NRL_HIP(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
  }
  *out = &f;
  return NRL_OK;
}

// ---- wide nn.Linear (N >= 256: the projections of a transformer body) on the row-panel kernel ------------------------
// The output's columns go in panels of 256 (16 column blocks, grid axis y), one fragment-ordered weight image per panel:
// the fp32 -> (hi, lo) split of an activation fragment then feeds 96 MFMAs, where the 128 x 160 tiles of the LDS-DMA
// kernel re-split every fragment for each of their N / 160 column tiles (20 times at N = 3072).
constexpr int LIN_PANEL_BLOCKS = 16;
static inline int lin_panels(int n_out) { return (n_out + 16 * LIN_PANEL_BLOCKS - 1) / (16 * LIN_PANEL_BLOCKS); }
static inline size_t lin_image_elems(int n_out, int k_red) {
  return (size_t)lin_panels(n_out) * rp_image_elems(LIN_PANEL_BLOCKS, rp_kblocks(k_red, false));
}
static inline bool lin_panels_on(int n_out) { return opt(O_ROWPANEL) && n_out >= 16 * LIN_PANEL_BLOCKS; }
// C (m, n_out) = epi(A (m, k_red) * B), element (column j, reduction i) of B = src[j * sn + i * sk]
template <class Epi>
static int linear_panels(const float* a, const float* src, int64_t sn, int64_t sk, int n_out, int k_red, const Epi& epi,
                         int64_t m, uint16_t* img, hipStream_t st, bool build = true) {
  const int P = lin_panels(n_out), kb = rp_kblocks(k_red, false), pw = 16 * LIN_PANEL_BLOCKS;
  const size_t pe = rp_image_elems(LIN_PANEL_BLOCKS, kb);
  // (build == false: `img` already holds this weight's panel images -- a frozen weight's, kept by the caller across calls)
  for (int p0 = 0; build && p0 < P; p0 += RP_MAX_JOBS) {
    RpImageJobs jobs;
    rp_jobs_init(&jobs);
    for (int p = p0; p < P && p < p0 + RP_MAX_JOBS; ++p)
      rp_jobs_add(&jobs, src + (int64_t)p * pw * sn, sn, sk, n_out - p * pw < pw ? n_out - p * pw : pw, k_red, nullptr,
                  img + (size_t)p * pe, LIN_PANEL_BLOCKS);
    NRL_TRY(rp_jobs_launch(jobs, st));
  }
  RpImage im;
  im.img = img; im.nblk = LIN_PANEL_BLOCKS; im.kblocks = kb;
  return launch_rp_gemm<LIN_PANEL_BLOCKS, 4, 0, 2>(KCPlain{a, k_red, m}, im, epi, m, n_out, k_red, st, P);
}

}  // namespace nrl

using namespace nrl;

extern "C" {

int nrl_abi_version(void) { return NRL_ABI_VERSION; }
// (nrl_build_id: nrl_build_id.hip)
const char* nrl_last_error(void) { return g_err; }

int nrl_prof_enable(int32_t on) {
  g_prof.on = on != 0;
  if (on) {
    g_prof.total_ms = g_prof.total_flops = 0.0;
    g_prof.launches = 0;
  }
  return NRL_OK;
}

int nrl_prof_read(double* total_ms, int64_t* launches, double* total_flops) {
  NRL_REQUIRE(total_ms && launches && total_flops, "prof_read: null output");
  for (auto& ev : g_prof.pending) {
    NRL_HIP(hipEventSynchronize(ev.second));
    float ms = 0.f;
    NRL_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
    g_prof.total_ms += ms;
    g_prof.pool.push_back(ev);
  }
  g_prof.pending.clear();
  *total_ms = g_prof.total_ms;
  *launches = g_prof.launches;
  *total_flops = g_prof.total_flops;
  return NRL_OK;
}

int nrl_set_gemm_engine(int32_t engine) {
  NRL_REQUIRE(engine == ENGINE_F32 || engine == ENGINE_BF16X3, "unknown GEMM engine %d", engine);
  g_default_engine.store(engine);
  return NRL_OK;
}
int nrl_get_gemm_engine(void) { return g_default_engine.load(); }

int nrl_set_option(const char* name, int32_t value) {
  NRL_REQUIRE(name != nullptr, "set_option: null name");
  for (int i = 0; i < O_COUNT; ++i)
    if (!strcmp(name, kOptName[i])) {
      NRL_REQUIRE(value == 0 || !opt_retired(i), "set_option: '%s' was retired in ABI v14 (its kernel lost its A/B and lives in "
                  "tools/experimental/; the bit is reserved and must stay 0)", name);
      if (value != 0) g_opt_default.fetch_or(1u << i);
      else g_opt_default.fetch_and(~(1u << i));
      return NRL_OK;
    }
  set_error("set_option: unknown option '%s' (news_fused, news_fused_bwd, news_attn_mfma, news_planes, news_od_planes, "
            "news_aa_planes, wgrad_2step, wgrad_ws, rowpanel, x3_dma, news_tail, news_tail_bwd, user_fork, news_fork, news_qkv_planes, "
            "news_pad_share, news_tail_od, user_proj)", name);
  return NRL_E_INVALID;
}

// bit mask of the process-default switch values, in the order nrl_set_option lists them; NRL_OPTIONS_EXPLICIT | mask is
// what a host stores at a forward and hands to the matching backward in NrlBlockParams.options
int32_t nrl_get_options(void) { return (int32_t)g_opt_default.load(); }

uint32_t nrl_dropout_key(uint64_t seed, uint32_t stream) { return dropout_key(seed, stream); }

int nrl_dropout_mask(uint8_t* keep, int64_t n_elems, double p, uint64_t seed, uint32_t stream,
                     void* stream_handle) {
  NRL_REQUIRE(keep != nullptr && n_elems >= 0 && p >= 0.0 && p < 1.0, "dropout_mask: bad arguments");
  return dropout_mask(keep, n_elems, make_dropout(p, seed, stream), (hipStream_t)stream_handle);
}

size_t nrl_news_encoder_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t embed_dim,
                                        int32_t num_heads, int32_t query_dim) {
  return block_ws_floats(n_news * seq_len, embed_dim, query_dim, num_heads, true, news_pad_rows(n_news, seq_len, embed_dim, num_heads)) *
         sizeof(float);
}

int nrl_news_encoder_fwd(const NrlBlockParams* p, const float* emb_table, int64_t vocab,
                         const int64_t* ids, int64_t n_news, int32_t seq_len, double p_drop,
                         uint64_t seed, uint32_t stream0, int32_t save_for_backward, float* out,
                         void* ws, size_t ws_bytes, void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_REQUIRE(emb_table && ids && out && vocab > 0 && n_news >= 0 && seq_len > 0, "news_encoder_fwd: bad arguments");
  NRL_REQUIRE(((uintptr_t)emb_table & 15) == 0, "embedding table must be 16-byte aligned");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  if (n_news == 0) return NRL_OK;
  const BlockShape s = news_shape(p, n_news, seq_len);
  NRL_REQUIRE(s.M * s.D < (1LL << 32), "activation too large for the 32-bit dropout index space");
  BlockWs w;
  NRL_TRY(carve_ws(ws, ws_bytes, s, true, &w));
  const Dropout d1 = make_dropout(p_drop, seed, stream0), d2 = make_dropout(p_drop, seed, stream0 + 1);
  if (news_fused_on(s, seq_len)) {
    // gather + in-projection + token attention in ONE kernel (nrl_news_fused.h): q|k|v are never read back
    hipStream_t st = (hipStream_t)stream;
    BlockPlanes bp;
    NRL_TRY(block_planes(p, s, w, true, &bp, st, s.heads));
    NewsFusedArgs a;
    a.table = emb_table; a.ids = ids; a.img = bp.rp.in_heads.img; a.n_news = n_news; a.L = seq_len; a.D = s.D;
    a.heads = s.heads; a.dh = s.dh; a.scale = s.geom.scale; a.drop1 = d1; a.o = w.o;
    const bool planes = opt(O_NEWS_ATTN_MFMA) && opt(O_NEWS_PLANES);
    BlockShape sf = s;
    sf.od_planes = planes && opt(O_NEWS_OD_PLANES);
    // (training only: in an evaluation forward the second copy of y costs more than the additive-attention GEMM saves)
    sf.aa_planes = sf.od_planes && opt(O_NEWS_AA_PLANES) && save_for_backward && w.yp != nullptr && (s.D & 15) == 12 && s.Q <= 224;
    a.o_planes = nullptr;
    if (sf.od_planes) {
      a.o_planes = reinterpret_cast<unsigned char*>(w.o);
      const int ncb = s.heads + (s.heads + 3) / 4;
      if (s.M % 32 != 0)   // rows past M in the last 32-row k-tile of the weight gradient
        NRL_HIP(hipMemsetAsync(a.o_planes + (s.M / 32) * 2 * ncb * 1024, 0, (size_t)2 * ncb * 1024, st));
    }
    a.x_save = (save_for_backward && !planes) ? w.x : nullptr;
    a.x_planes = (save_for_backward && planes) ? reinterpret_cast<unsigned char*>(w.x) : nullptr;
    a.qkv_save = save_for_backward ? w.qkv : nullptr;
    a.qkv_head_major = opt(O_NEWS_ATTN_MFMA) ? 1 : 0;
    a.lse = save_for_backward ? w.lse : nullptr;
    // evaluation (nothing saved, no dropout): the run of padding tokens from token 15 on is ONE row -- partition the news into
    // short / long (list + counts in the log-sum-exp buffer, dead in an evaluation call) and let both kernels share the row
    const bool share = !save_for_backward && p_drop == 0.0 && opt(O_NEWS_PAD_SHARE) && seq_len >= 17 && sf.od_planes &&
                       news_tail_on(sf, seq_len, w) && n_news < (1LL << 31);
    if (share) {
      int32_t* hdr = reinterpret_cast<int32_t*>(w.lse);
      NRL_TRY(launch_news_classify(ids, n_news, seq_len, hdr, hdr + 2, st));
      a.n_short = hdr; a.perm = hdr + 2;
    }
    {
      ProfScope prof(st, 2.0 * (double)s.M * 3.0 * s.D * s.D + 4.0 * (double)s.M * seq_len * s.D);
      NRL_TRY(launch_news_fused_fwd(a, st));
    }
    if (news_tail_on(sf, seq_len, w)) {
      // out-projection + dropout + additive attention + pooling in ONE kernel (nrl_news_tail.h)
      NewsTailArgs t;
      t.o_planes = a.o_planes; t.img_o = bp.rp.tail_o.img; t.img_a = bp.rp.tail_a.img; t.q_a = p->att_query;
      t.n_news = n_news; t.L = seq_len; t.D = s.D; t.Q = s.Q; t.drop2 = d2; t.out = out;
      t.y_planes = nullptr; t.t = nullptr; t.w = nullptr;
      t.perm = a.perm; t.n_short = a.n_short;
      if (save_for_backward) {
        t.y_planes = reinterpret_cast<unsigned char*>(w.yp); t.w = w.w;
        t.t = news_tail_bwd_on(sf, seq_len, w) ? nullptr : w.t;     // the fused backward recomputes the tanh output
        const int ncb_y = (s.D + 16) / 16;
        if (s.M % 32 != 0)   // rows past M in the last 32-row k-tile of the weight gradient
          NRL_HIP(hipMemsetAsync(t.y_planes + (s.M / 32) * 2 * ncb_y * 1024, 0, (size_t)2 * ncb_y * 1024, st));
      }
      return news_tail_fwd(t, st);
    }
    return block_fwd_tail(p, sf, w, bp, d2, out, st);
  }
  KCGather a_in{emb_table, ids, s.M, s.D, d1, save_for_backward ? w.x : nullptr};
  return block_fwd(p, a_in, s, w, d2, save_for_backward != 0, true, out, (hipStream_t)stream);
}

// ---- evaluation forwards from a per-token q|k|v table (ABI v16; kernels: nrl_news_fused.h) -------------------------------
// table buffer = [fragment-ordered weight images: per-head q|k|v (the build's), W_o over the plane slots, W_a in kappa order
// (the fused tail's)] [q|k|v of every vocabulary id, head-major]; every offset is a function of (vocab, D, heads, Q) only
struct TokenTable {
  size_t img_heads, img_o, img_a, qkv, bytes;     // byte offsets
};
static TokenTable token_table_layout(int64_t vocab, int D, int heads) {
  TokenTable t;
  auto al = [](size_t n) { return align_up(n, 256); };
  size_t off = 0;
  t.img_heads = off; off += al(rp_image_elems(heads * 4, NF_KB) * 2);
  t.img_o = off; off += al(rp_image_elems(NT_FB, NT_KB) * 2);
  t.img_a = off; off += al(rp_image_elems(NT_QB, NT_KS) * 2);
  t.qkv = off; off += al(news_qkv_table_floats(vocab, heads) * sizeof(float));
  t.bytes = off;
  return t;
}
static bool token_table_geometry_ok(int L, int D, int heads, int Q) {
  return news_fused_ok(L, D, heads) && news_tail_geometry_ok(L, D, Q, heads);
}

int32_t nrl_token_table_supported(int32_t seq_len, int32_t embed_dim, int32_t num_heads, int32_t query_dim) {
  return (num_heads > 0 && token_table_geometry_ok(seq_len, embed_dim, num_heads, query_dim)) ? 1 : 0;
}

size_t nrl_token_table_bytes(int64_t vocab, int32_t embed_dim, int32_t num_heads, int32_t query_dim) {
  if (vocab <= 0 || num_heads <= 0 || !token_table_geometry_ok(32, embed_dim, num_heads, query_dim)) return 0;
  if (news_qkv_table_floats(vocab, num_heads) >= (1ull << 32)) return 0;
  return token_table_layout(vocab, embed_dim, num_heads).bytes;
}

int nrl_token_table_build(const NrlBlockParams* p, const float* emb_table, int64_t vocab, void* table, size_t table_bytes,
                          void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_REQUIRE(emb_table && table && vocab > 0, "token_table_build: bad arguments");
  NRL_REQUIRE(((uintptr_t)emb_table & 15) == 0 && ((uintptr_t)table & 255) == 0, "token_table_build: the embedding table must be 16-byte, the token table 256-byte aligned");
  const int D = p->embed_dim, heads = p->num_heads, Q = p->query_dim;
  NRL_REQUIRE(token_table_geometry_ok(32, D, heads, Q), "token_table_build: geometry outside the fused news encoder (D = 300, 15 heads, Q <= 208)");
  NRL_REQUIRE(cur_engine() == ENGINE_BF16X3, "token_table_build: the bf16x3 engine only (the exact-fp32 engine has no fused news encoder)");
  const size_t need = nrl_token_table_bytes(vocab, D, heads, Q);
  NRL_REQUIRE(need > 0, "token_table_build: vocabulary too large for one table (2^32 floats)");
  if (table_bytes < need) {
    set_error("token table too small: %zu < %zu bytes", table_bytes, need);
    return NRL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const TokenTable t = token_table_layout(vocab, D, heads);
  unsigned char* base = static_cast<unsigned char*>(table);
  uint16_t* img_heads = reinterpret_cast<uint16_t*>(base + t.img_heads);
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  rp_jobs_add_qkv_heads(&jobs, p->in_proj_weight, D, p->in_proj_bias, img_heads, heads, D / heads);
  rp_jobs_add_kperm(&jobs, p->out_proj_weight, D, 1, D, heads, reinterpret_cast<uint16_t*>(base + t.img_o), NT_FB, p->out_proj_bias);
  rp_jobs_add_kappa(&jobs, p->att_weight, D, 1, Q, D, p->att_bias, reinterpret_cast<uint16_t*>(base + t.img_a), NT_QB, NT_KS);
  NRL_TRY(rp_jobs_launch(jobs, st));
  return launch_news_qkv_table(emb_table, vocab, D, heads, img_heads, reinterpret_cast<float*>(base + t.qkv), st);
}

// workspace of the table forward: the `o` planes + the short-first news list of the pad-row sharing
static size_t table_fwd_planes_bytes(int64_t n_news, int L, int heads) {
  const int64_t M = n_news * L;
  return align_up((size_t)((M + 31) / 32 * 32) * (size_t)(heads + (heads + 3) / 4) * 16 * sizeof(float), 256);
}
size_t nrl_news_encoder_fwd_table_workspace_bytes(int64_t n_news, int32_t seq_len, int32_t num_heads) {
  if (n_news <= 0 || seq_len <= 0 || num_heads <= 0) return 0;
  return table_fwd_planes_bytes(n_news, seq_len, num_heads) + align_up((size_t)(n_news + 2) * sizeof(int32_t), 256);
}

int nrl_news_encoder_fwd_table(const NrlBlockParams* p, const void* table, size_t table_bytes, int64_t vocab,
                               const int64_t* ids, int64_t n_news, int32_t seq_len, float* out, void* ws, size_t ws_bytes,
                               void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_REQUIRE(table && ids && out && vocab > 0 && n_news >= 0 && seq_len > 0, "news_encoder_fwd_table: bad arguments");
  NRL_REQUIRE(cur_engine() == ENGINE_BF16X3, "news_encoder_fwd_table: the bf16x3 engine only");
  if (n_news == 0) return NRL_OK;
  const int D = p->embed_dim, heads = p->num_heads, Q = p->query_dim;
  NRL_REQUIRE(token_table_geometry_ok(seq_len, D, heads, Q), "news_encoder_fwd_table: geometry outside the fused news encoder");
  const size_t need_t = nrl_token_table_bytes(vocab, D, heads, Q);
  NRL_REQUIRE(need_t > 0 && table_bytes >= need_t, "news_encoder_fwd_table: not a table of this vocabulary / geometry");
  NRL_REQUIRE(n_news * (int64_t)seq_len < (1LL << 31), "news_encoder_fwd_table: too many token rows for one call");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  const size_t need_w = nrl_news_encoder_fwd_table_workspace_bytes(n_news, seq_len, heads);
  if (ws_bytes < need_w) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, need_w);
    return NRL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const TokenTable t = token_table_layout(vocab, D, heads);
  const unsigned char* base = static_cast<const unsigned char*>(table);
  unsigned char* o_planes = static_cast<unsigned char*>(ws);
  int32_t* hdr = reinterpret_cast<int32_t*>(o_planes + table_fwd_planes_bytes(n_news, seq_len, heads));
  NewsTabArgs a;
  a.tab = reinterpret_cast<const float*>(base + t.qkv); a.ids = ids; a.n_news = n_news; a.vocab = vocab; a.L = seq_len;
  a.heads = heads; a.scale = 1.0f / sqrtf((float)(D / heads)); a.o_planes = o_planes;
  if (opt(O_NEWS_PAD_SHARE) && seq_len >= 17) {
    NRL_TRY(launch_news_classify(ids, n_news, seq_len, hdr, hdr + 2, st));
    a.n_short = hdr; a.perm = hdr + 2;
  }
  {
    ProfScope prof(st, 4.0 * (double)n_news * seq_len * seq_len * D);
    NRL_TRY(launch_news_tab_attn_fwd(a, st));
  }
  NewsTailArgs tl;
  tl.o_planes = o_planes; tl.img_o = reinterpret_cast<const uint16_t*>(base + t.img_o);
  tl.img_a = reinterpret_cast<const uint16_t*>(base + t.img_a); tl.q_a = p->att_query;
  tl.n_news = n_news; tl.L = seq_len; tl.D = D; tl.Q = Q; tl.drop2 = make_dropout(0.0, 0, 0); tl.out = out;
  tl.y_planes = nullptr; tl.t = nullptr; tl.w = nullptr;
  tl.perm = a.perm; tl.n_short = a.n_short;
  return news_tail_fwd(tl, st);
}

int nrl_news_encoder_bwd(const NrlBlockParams* p, const NrlBlockGrads* g, const float* emb_table,
                         float* d_emb_table, int64_t vocab, const int64_t* ids, const int64_t* sorted_positions,
                         int64_t n_news, int32_t seq_len, double p_drop, uint64_t seed, uint32_t stream0,
                         const float* d_out, int32_t phase, void* ws, size_t ws_bytes, void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_TRY(check_grads(g));
  NRL_REQUIRE(d_emb_table && ids && d_out && vocab > 0 && n_news >= 0 && seq_len > 0, "news_encoder_bwd: bad arguments");
  NRL_REQUIRE(phase >= 0 && phase <= 2, "news_encoder_bwd: phase must be 0 (all), 1 or 2");
  if (n_news == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const BlockShape s = news_shape(p, n_news, seq_len);
  BlockWs w;
  NRL_TRY(carve_ws(ws, ws_bytes, s, true, &w));
  const Dropout d1 = make_dropout(p_drop, seed, stream0), d2 = make_dropout(p_drop, seed, stream0 + 1);
  BlockPlanes bp;
  const bool slabs = news_fused_on(s, seq_len) && opt(O_NEWS_ATTN_MFMA);   // what the forward saved
  const bool planes = slabs && opt(O_NEWS_PLANES);
  BlockShape sb_ = s;
  sb_.od_planes = planes && opt(O_NEWS_OD_PLANES);
  sb_.aa_planes = sb_.od_planes && opt(O_NEWS_AA_PLANES) && w.yp != nullptr && (s.D & 15) == 12 && s.Q <= 224;
  sb_.tail = news_tail_on(sb_, seq_len, w);
  sb_.tail_bwd = sb_.tail && sb_.aa_planes && news_tail_bwd_on(sb_, seq_len, w);
  NRL_TRY(block_planes(p, s, w, false, &bp, st, slabs ? s.heads : 0));  // filled by the forward
  // news_fork: the back-half weight gradients leave phase 2 for a side stream of phase 1 (a phase-2 call of a two-phase
  // caller evaluates the same predicate and skips them)
  sb_.forked = news_fork_on(sb_);
  SideFork side;
  hipEvent_t side_in_join = nullptr;
  if (phase != 2) {
    if (sb_.forked) {
      ForkSet* fs = nullptr;
      NRL_TRY(fork_set(&fs));
      side.s = fs->s[0]; side.fork = fs->fork; side.join = fs->join[0];
    }
    NRL_TRY(block_bwd_phase1(p, g, sb_, w, bp, d2, d_out, st, slabs, sb_.forked ? &side : nullptr));
    if (slabs) {
      // token attention backward on the matrix cores from the head-major q|k|v slabs (nrl_news_fused.h)
      NewsAttnBwdArgs a;
      a.qkv_hm = w.qkv; a.d_o = w.d_o; a.lse = w.lse; a.dqkv = w.dqkv; a.n_news = n_news; a.L = seq_len; a.D = s.D;
      a.heads = s.heads; a.scale = s.geom.scale; a.hpw = 1; a.planes = planes ? 1 : 0;
      if (planes && opt(O_NEWS_QKV_PLANES)) NRL_TRY(launch_news_attn_bwd_p(a, st));   // operands split once, into LDS planes
      else NRL_TRY(launch_news_attn_bwd(a, st));
      if (news_fork_in_on(sb_, planes)) {
        // dqkv is complete: the in-projection weight gradient leaves for the second internal stream, beside the live-row dgrad and
        // the table gradient below (joined before this call returns; a phase-2 call evaluates the same predicate and skips it)
        ForkSet* fs = nullptr;
        NRL_TRY(fork_set(&fs));
        const int si = news_fork_in_mode() == 2 ? 0 : 1;
        NRL_HIP(hipEventRecord(fs->fork, st));
        NRL_HIP(hipStreamWaitEvent(fs->s[si], fs->fork, 0));
        NRL_TRY(block_wgrad_in_planes(g, w.x, sb_, w, fs->s[si]));
        NRL_HIP(hipEventRecord(fs->join[1], fs->s[si]));
        side_in_join = fs->join[1];
      }
    }
    // dx = dqkv W_in, times dropout1, added into the table rows (embedding_dense_backward)
    const KCSlab dq_hp{w.dqkv, s.M};
    const KCPlanes dq_pl{reinterpret_cast<const unsigned char*>(w.dqkv), s.pad_rows};
    if (sorted_positions != nullptr) {
      // dx is materialised (in the now dead d_o buffer) and reduced in id-sorted order: no hot-row contention
      float* dx = w.d_o;
      const EpiLinear epi{dx, s.D, nullptr, 0, d1, s.D};
      // (NRL_LIVE_ROWS=0: dx for every token row, for A/B runs)
      static const bool live_env = [] { const char* e = getenv("NRL_LIVE_ROWS"); return !(e != nullptr && e[0] == '0'); }();
      if (planes && live_env) {
        // dx only for the LIVE token rows (id != 0: the padding id has no table gradient), compact in position order; the
        // list of live positions is built here in two small launches (scratch: the log-sum-exp buffer, dead by now)
        NRL_REQUIRE((size_t)s.M * s.heads >= live_compact_ints(s.M), "news_encoder_bwd: scratch for the live-row list");
        const int32_t *list = nullptr, *cidx = nullptr, *n_live = nullptr;
        NRL_TRY(live_compact(ids, s.M, reinterpret_cast<int32_t*>(w.lse), &list, &cidx, &n_live, st));
        const KCPlanesLive dq_live{reinterpret_cast<const unsigned char*>(w.dqkv), s.pad_rows, list, n_live, seq_len};
        NRL_TRY(rp_dispatch(dq_live, bp.rp.in_d_hp, EpiDxLive{dx, s.D, d1, list}, s.M, s.D, s.heads * 64, st));
        NRL_TRY(embedding_grad_sorted(dx, ids, sorted_positions, s.M, s.D, d_emb_table, st, cidx));
      } else {
        if (planes) NRL_TRY(rp_dispatch(dq_pl, bp.rp.in_d_hp, EpiNewsRows<EpiLinear>{epi, seq_len}, s.pad_rows, s.D, s.heads * 64, st));
        else if (slabs) NRL_TRY(rp_dispatch(dq_hp, bp.rp.in_d_hp, epi, s.M, s.D, s.heads * 64, st));
        else NRL_TRY(gemm_dgrad(w.dqkv, p->in_proj_weight, bp.in, epi, s.M, 3 * s.D, s.D, st, bp.rp.on ? &bp.rp.in_d : nullptr));
        NRL_TRY(embedding_grad_sorted(dx, ids, sorted_positions, s.M, s.D, d_emb_table, st));
      }
    } else {
      const EpiScatter epi{d_emb_table, ids, s.D, d1};
      if (planes) NRL_TRY(rp_dispatch(dq_pl, bp.rp.in_d_hp, EpiNewsRows<EpiScatter>{epi, seq_len}, s.pad_rows, s.D, s.heads * 64, st));
      else if (slabs) NRL_TRY(rp_dispatch(dq_hp, bp.rp.in_d_hp, epi, s.M, s.D, s.heads * 64, st));
      else NRL_TRY(gemm_dgrad(w.dqkv, p->in_proj_weight, bp.in, epi, s.M, 3 * s.D, s.D, st, bp.rp.on ? &bp.rp.in_d : nullptr));
    }
  }
  if (phase != 2 && sb_.forked) NRL_HIP(hipStreamWaitEvent(st, side.join, 0));   // the side stream's work is part of this call
  if (side_in_join != nullptr) NRL_HIP(hipStreamWaitEvent(st, side_in_join, 0));
  if (phase != 1) NRL_TRY(block_bwd_phase2(g, w.x, sb_, w, st, slabs, planes));
  return NRL_OK;
}

size_t nrl_user_encoder_workspace_bytes(int64_t batch, int64_t hist_len, int32_t embed_dim,
                                        int32_t num_heads, int32_t query_dim) {
  return block_ws_floats(batch * hist_len, embed_dim, query_dim, num_heads, true) * sizeof(float);
}

int nrl_user_encoder_fwd(const NrlBlockParams* p, const float* hist, int64_t batch, int64_t hist_len,
                         double p_drop, uint64_t seed, uint32_t stream0, int32_t input_dropout,
                         int32_t save_for_backward, float* out, void* ws, size_t ws_bytes, void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_REQUIRE(hist && out && batch > 0 && hist_len > 0, "user_encoder_fwd: bad arguments");
  NRL_REQUIRE(((uintptr_t)hist & 15) == 0, "hist must be 16-byte aligned");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  const BlockShape s = user_shape(p, batch, hist_len);
  NRL_REQUIRE(p_drop == 0.0 || s.M * s.D < (1LL << 32), "activation too large for the 32-bit dropout index space");
  BlockWs w;
  NRL_TRY(carve_ws(ws, ws_bytes, s, true, &w));
  const Dropout d2 = make_dropout(p_drop, seed, stream0 + 1);
  if (p_drop > 0.0 && input_dropout) {
    // seq-first block with dropouts around the attention = the PLM text encoder's tail (text.py:92-96)
    const Dropout d1 = make_dropout(p_drop, seed, stream0);
    return block_fwd(p, KCGather{hist, nullptr, s.M, s.D, d1, save_for_backward ? w.x : nullptr}, s, w, d2,
                     save_for_backward != 0, false, out, (hipStream_t)stream);
  }
  // no input dropout: NRMS user encoder (p_drop = 0) / CenNewsRec long-term branch (dropout after the
  // attention only, user/cen_news_rec.py:66-70)
  return block_fwd(p, KCPlain{hist, s.D, s.M}, s, w, d2, save_for_backward != 0, false, out, (hipStream_t)stream);
}

int nrl_user_encoder_bwd(const NrlBlockParams* p, const NrlBlockGrads* g, const float* hist,
                         int64_t batch, int64_t hist_len, double p_drop, uint64_t seed, uint32_t stream0,
                         int32_t input_dropout, const float* d_out, float* d_hist, void* ws, size_t ws_bytes,
                         void* stream) {
  return nrl_user_encoder_bwd_phase(p, g, hist, batch, hist_len, p_drop, seed, stream0, input_dropout, d_out, d_hist, 0, ws,
                                    ws_bytes, stream);
}

// phase 1: everything up to d_hist (+ the additive-attention query gradient, which the pooling backward produces);
// phase 2: the three weight gradients, which only READ what phase 1 left in the workspace -- a caller may run it on another
// stream (ordered after phase 1) beside whatever consumes d_hist; phase 0: both, in that order.
int nrl_user_encoder_bwd_phase(const NrlBlockParams* p, const NrlBlockGrads* g, const float* hist,
                               int64_t batch, int64_t hist_len, double p_drop, uint64_t seed, uint32_t stream0,
                               int32_t input_dropout, const float* d_out, float* d_hist, int32_t phase, void* ws,
                               size_t ws_bytes, void* stream) {
  NRL_TRY(check_params(p));
  const EngineScope engine_scope(p->gemm_engine);
  const OptScope opt_scope(p->options);
  NRL_TRY(check_grads(g));
  NRL_REQUIRE(phase >= 0 && phase <= 2, "user_encoder_bwd: phase must be 0 (both), 1 or 2");
  NRL_REQUIRE(hist && batch > 0 && hist_len > 0 && (phase == 2 || (d_out && d_hist)), "user_encoder_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const BlockShape s = user_shape(p, batch, hist_len);
  BlockWs w;
  NRL_TRY(carve_ws(ws, ws_bytes, s, true, &w));
  const bool in_drop = p_drop > 0.0 && input_dropout;
  const Dropout d1 = make_dropout(in_drop ? p_drop : 0.0, seed, stream0), d2 = make_dropout(p_drop, seed, stream0 + 1);
  const float* x_rows = in_drop ? w.x : hist;
  if (phase == 2) return block_bwd_phase2(g, x_rows, s, w, st);
  BlockPlanes bp;
  NRL_TRY(block_planes(p, s, w, false, &bp, st));
  NRL_TRY(block_bwd_phase1(p, g, s, w, bp, d2, d_out, st));
  if (phase == 1)
    return gemm_dgrad(w.dqkv, p->in_proj_weight, bp.in, EpiLinear{d_hist, s.D, nullptr, 0, d1, s.D}, s.M, 3 * s.D, s.D, st,
                      bp.rp.on ? &bp.rp.in_d : nullptr);
  if (!opt(O_USER_FORK) || s.M > 65536) {       // (large calls -- the PLM tail -- fill the chip with every launch)
    NRL_TRY(gemm_dgrad(w.dqkv, p->in_proj_weight, bp.in, EpiLinear{d_hist, s.D, nullptr, 0, d1, s.D}, s.M, 3 * s.D,
                       s.D, st, bp.rp.on ? &bp.rp.in_d : nullptr));
    return block_bwd_phase2(g, x_rows, s, w, st);
  }
  // The in-projection dgrad (-> d_hist, the critical path into the news-encoder backward) on the caller's stream; the three
  // weight gradients, which only READ saved activations / gradients, beside it on two side streams.  Their split-K scratch
  // is three disjoint thirds of the q|k|v rows (dead once the attention backward has run).
  ForkSet* fs = nullptr;
  NRL_TRY(fork_set(&fs));
  NRL_HIP(hipEventRecord(fs->fork, st));
  for (int i = 0; i < 2; ++i) NRL_HIP(hipStreamWaitEvent(fs->s[i], fs->fork, 0));
  const size_t third = opt(O_WGRAD_2STEP) ? qkv_elems(s.M, s.D, s.heads, s.pad_rows) / 3 / 64 * 64 : 0;
  float* const sc = third > 0 ? w.qkv : nullptr;
  int rc = gemm_dgrad(w.dqkv, p->in_proj_weight, bp.in, EpiLinear{d_hist, s.D, nullptr, 0, d1, s.D}, s.M, 3 * s.D, s.D, st,
                      bp.rp.on ? &bp.rp.in_d : nullptr);
  if (rc == NRL_OK) rc = gemm_wgrad(w.dqkv, 3 * s.D, x_rows, s.D, g->in_proj_weight, g->in_proj_bias, s.M, fs->s[0], sc, third);
  if (rc == NRL_OK)
    rc = gemm_wgrad(w.dy, s.D, w.o, s.D, g->out_proj_weight, g->out_proj_bias, s.M, fs->s[1], sc ? sc + third : nullptr, third);
  if (rc == NRL_OK)
    rc = gemm_wgrad(w.t, s.Q, w.y, s.D, g->att_weight, g->att_bias, s.M, fs->s[1], sc ? sc + 2 * third : nullptr, third);
  for (int i = 0; i < 2; ++i) {                  // join, also on an error path: the side streams must not outlive the call
    NRL_HIP(hipEventRecord(fs->join[i], fs->s[i]));
    NRL_HIP(hipStreamWaitEvent(st, fs->join[i], 0));
  }
  return rc;
}

int nrl_to_dense_batch_fwd(const float* x, const int64_t* offsets, int64_t batch, int64_t max_len,
                           int32_t dim, float* dense, void* stream) {
  NRL_REQUIRE(offsets && dense && batch >= 0 && max_len >= 0 && dim > 0, "to_dense_batch_fwd: bad arguments");
  return to_dense_fwd(x, offsets, batch, max_len, dim, dense, (hipStream_t)stream);
}

int nrl_to_dense_batch_bwd(const float* d_dense, const int64_t* offsets, int64_t batch,
                           int64_t max_len, int32_t dim, int64_t n_rows, float* d_x, void* stream) {
  NRL_REQUIRE(offsets && d_dense && batch >= 0 && max_len >= 0 && dim > 0, "to_dense_batch_bwd: bad arguments");
  return to_dense_bwd(d_dense, offsets, batch, max_len, dim, n_rows, d_x, (hipStream_t)stream);
}

int nrl_hist_mean_fwd(const float* hist, const int64_t* offsets, int64_t batch, int64_t max_len, int32_t dim,
                      float* user, void* stream) {
  NRL_REQUIRE(hist && offsets && user && batch >= 0 && max_len >= 0 && dim > 0, "hist_mean_fwd: bad arguments");
  return hist_mean_fwd(hist, offsets, batch, max_len, dim, user, (hipStream_t)stream);
}

int nrl_hist_mean_bwd(const float* d_user, const int64_t* offsets, int64_t batch, int64_t max_len, int32_t dim,
                      float* d_hist, void* stream) {
  NRL_REQUIRE(d_user && offsets && d_hist && batch >= 0 && max_len >= 0 && dim > 0, "hist_mean_bwd: bad arguments");
  return hist_mean_bwd(d_user, offsets, batch, max_len, dim, d_hist, (hipStream_t)stream);
}

int nrl_dot_scores_fwd(const float* user, const float* cand, int64_t batch, int64_t n_cand,
                       int32_t dim, float* scores, void* stream) {
  NRL_REQUIRE(user && cand && scores && batch >= 0 && n_cand >= 0 && dim > 0, "dot_scores_fwd: bad arguments");
  return dot_scores_fwd(user, cand, batch, n_cand, dim, scores, (hipStream_t)stream);
}

int nrl_dot_scores_bwd(const float* d_scores, const float* user, const float* cand, int64_t batch,
                       int64_t n_cand, int32_t dim, float* d_user, float* d_cand, void* stream) {
  NRL_REQUIRE(d_scores && user && cand && d_user && d_cand && dim > 0, "dot_scores_bwd: bad arguments");
  return dot_scores_bwd(d_scores, user, cand, batch, n_cand, dim, d_user, d_cand, (hipStream_t)stream);
}

int nrl_ce_loss_fwd_bwd(const float* scores, const float* y_true, int64_t batch, int64_t n_cand,
                        float grad_scale, float* loss, float* d_scores, void* stream) {
  NRL_REQUIRE(scores && y_true && loss, "ce_loss: bad arguments");
  return ce_loss_fwd_bwd(scores, y_true, batch, n_cand, grad_scale, loss, d_scores, (hipStream_t)stream);
}

int nrl_supcon_loss_fwd_bwd(const float* scores, const float* y_true, const int64_t* cand_sizes, int64_t batch,
                            int64_t n_cand, float temperature, float grad_scale, float* loss, float* d_scores,
                            void* stream) {
  NRL_REQUIRE(scores && y_true && cand_sizes && loss, "supcon_loss: bad arguments");
  return supcon_loss_fwd_bwd(scores, y_true, cand_sizes, batch, n_cand, temperature, grad_scale, loss, d_scores,
                             (hipStream_t)stream);
}

int nrl_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, int64_t step, float grad_scale,
                  int32_t zero_grad, void* stream) {
  NRL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0, "adam_step: bad arguments");
  return adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, zero_grad,
                   (hipStream_t)stream);
}

int nrl_adam_rows_mark(const int64_t* ids, int64_t n_ids, int64_t rows, int32_t* mark, int64_t step, void* stream) {
  NRL_REQUIRE((ids || n_ids == 0) && mark && n_ids >= 0 && rows > 0, "adam_rows_mark: bad arguments");
  return adam_rows_mark(ids, n_ids, rows, mark, step, (hipStream_t)stream);
}

int nrl_adam_rows_advance(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t rows, int32_t dim,
                          int32_t* last_step, const int32_t* mark, int32_t* status, int64_t stride, int64_t offset,
                          int64_t upto_step, int32_t with_grad, double lr, double beta1, double beta2, double eps,
                          float grad_scale, const int32_t* exclude_mark, int32_t exclude_tag, void* stream) {
  return adam_rows_advance(param, grad, exp_avg, exp_avg_sq, rows, dim, last_step, mark, status, stride, offset, upto_step,
                           with_grad, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream, exclude_mark, exclude_tag);
}

int nrl_embedding_gather(const float* table, const int64_t* ids, int64_t n_ids, int32_t dim,
                         float* out, void* stream) {
  NRL_REQUIRE(table && ids && out && n_ids >= 0 && dim > 0, "embedding_gather: bad arguments");
  return embedding_gather(table, ids, n_ids, dim, out, (hipStream_t)stream);
}

// nn.Linear + exact GELU (a BERT-family feed-forward block's first half, ABI v14): h = a W^T + bias (saved), g = gelu(h)
int nrl_linear_gelu_fwd_img(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k, float* h, float* g,
                            void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(a && w && bias && h && g && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0, "linear_gelu_fwd: bad arguments");
  NRL_REQUIRE((((uintptr_t)a | (uintptr_t)w | (uintptr_t)h | (uintptr_t)g | (uintptr_t)bias) & 15) == 0, "linear_gelu_fwd: 16-byte alignment");
  NRL_REQUIRE(cur_engine() == ENGINE_BF16X3 && lin_panels_on(n), "linear_gelu_fwd: bf16x3 engine and n >= 256 (nrl_linear_gelu_supported)");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0 && ws_bytes >= nrl_linear_workspace_bytes(n, k), "linear_gelu_fwd: workspace");
  if (m == 0) return NRL_OK;
  return linear_panels(a, w, k, 1, n, k, EpiLinearGelu{h, g, n, bias}, m, (uint16_t*)ws, (hipStream_t)stream, image_ready == 0);
}

// activation gradient of a projection whose INPUT was g = gelu(pre): d_pre (m, k) = (d_c W) * gelu'(pre) in the epilogue
int nrl_linear_dgrad_gelu_img(const float* w, const float* d_c, const float* pre, int64_t m, int32_t n, int32_t k, float* d_pre,
                              void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(w && d_c && pre && d_pre && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0, "linear_dgrad_gelu: bad arguments");
  NRL_REQUIRE((((uintptr_t)w | (uintptr_t)d_c | (uintptr_t)pre | (uintptr_t)d_pre) & 15) == 0, "linear_dgrad_gelu: 16-byte alignment");
  NRL_REQUIRE(cur_engine() == ENGINE_BF16X3 && lin_panels_on(k), "linear_dgrad_gelu: bf16x3 engine and k >= 256 (nrl_linear_gelu_supported)");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0 && ws_bytes >= nrl_linear_workspace_bytes(n, k), "linear_dgrad_gelu: workspace");
  if (m == 0) return NRL_OK;
  return linear_panels(d_c, w, 1, k, k, n, EpiGeluBwd{d_pre, k, pre}, m, (uint16_t*)ws, (hipStream_t)stream, image_ready == 0);
}

// ---- three projections of one input (query / key / value of a transformer layer) as one GEMM each way, ABI v15 -----------
// forward image: 3 n output columns over k, panel p of weight p / (n / 256); backward image: k output columns over the 3 n
// concatenated reduction rows, part q of panel p at k-block q * (n / 32)
static inline bool lin3_ok(int n, int k) {
  return cur_engine() == ENGINE_BF16X3 && opt(O_ROWPANEL) && n % (16 * LIN_PANEL_BLOCKS) == 0 && k % (16 * LIN_PANEL_BLOCKS) == 0 &&
         3 * lin_panels(n) <= RP_MAX_JOBS && 3 * lin_panels(k) <= RP_MAX_JOBS;
}
int32_t nrl_linear3_supported(int32_t n, int32_t k) { return (n > 0 && k > 0 && lin3_ok(n, k)) ? 1 : 0; }
size_t nrl_linear3_workspace_bytes(int32_t n, int32_t k) {
  size_t e = lin_image_elems(3 * n, k);
  if (lin_image_elems(k, 3 * n) > e) e = lin_image_elems(k, 3 * n);
  return align_up(e * sizeof(uint16_t), 256);
}

int nrl_linear3_fwd_img(const float* a, const float* w0, const float* w1, const float* w2, const float* b0, const float* b1,
                        const float* b2, int64_t m, int32_t n, int32_t k, float* c, void* ws, size_t ws_bytes, int32_t image_ready,
                        void* stream) {
  NRL_REQUIRE(a && w0 && w1 && w2 && b0 && b1 && b2 && c && m >= 0 && n > 0 && k > 0, "linear3_fwd: bad arguments");
  NRL_REQUIRE(lin3_ok(n, k), "linear3_fwd: bf16x3 engine, n and k multiples of 256, at most 12 panels (nrl_linear3_supported)");
  NRL_REQUIRE((((uintptr_t)a | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)c) & 15) == 0,
              "linear3_fwd: 16-byte alignment");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0 && ws_bytes >= nrl_linear3_workspace_bytes(n, k), "linear3_fwd: workspace");
  if (m == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int pp = lin_panels(n), P = 3 * pp, kb = rp_kblocks(k, false), pw = 16 * LIN_PANEL_BLOCKS;
  const size_t pe = rp_image_elems(LIN_PANEL_BLOCKS, kb);
  uint16_t* img = (uint16_t*)ws;
  if (image_ready == 0) {
    const float* w[3] = {w0, w1, w2};
    RpImageJobs jobs;
    rp_jobs_init(&jobs);
    for (int p = 0; p < P; ++p)      // element (column j, reduction i) = W_q[j][i]
      rp_jobs_add(&jobs, w[p / pp] + (int64_t)(p % pp) * pw * k, k, 1, pw, k, nullptr, img + (size_t)p * pe, LIN_PANEL_BLOCKS);
    NRL_TRY(rp_jobs_launch(jobs, st));
  }
  RpImage im;
  im.img = img; im.nblk = LIN_PANEL_BLOCKS; im.kblocks = kb;
  return launch_rp_gemm<LIN_PANEL_BLOCKS, 4, 0, 2>(KCPlain{a, k, m}, im, EpiLinearParts{c, m * (int64_t)n, n, b0, b1, b2}, m, 3 * n, k, st, P);
}

// d_a (m, k) = add + sum_q d_c[q] (m, n) W_q (n, k);  d_c: the three gradients stacked (3, m, n);  add (m, k) or NULL
int nrl_linear3_dgrad_img(const float* d_c, const float* w0, const float* w1, const float* w2, int64_t m, int32_t n, int32_t k,
                          const float* add, float* d_a, void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(d_c && w0 && w1 && w2 && d_a && m >= 0 && n > 0 && k > 0, "linear3_dgrad: bad arguments");
  NRL_REQUIRE(lin3_ok(n, k), "linear3_dgrad: bf16x3 engine, n and k multiples of 256, at most 12 panels (nrl_linear3_supported)");
  NRL_REQUIRE((((uintptr_t)d_c | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)add | (uintptr_t)d_a) & 15) == 0,
              "linear3_dgrad: 16-byte alignment");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0 && ws_bytes >= nrl_linear3_workspace_bytes(n, k), "linear3_dgrad: workspace");
  if (m == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int P = lin_panels(k), kbp = n / 32, kb = 3 * kbp, pw = 16 * LIN_PANEL_BLOCKS;
  const size_t pe = rp_image_elems(LIN_PANEL_BLOCKS, kb), part = rp_image_elems(LIN_PANEL_BLOCKS, kbp);
  uint16_t* img = (uint16_t*)ws;
  if (image_ready == 0) {
    const float* w[3] = {w0, w1, w2};
    RpImageJobs jobs;
    rp_jobs_init(&jobs);
    for (int p = 0; p < P; ++p)      // element (column j, reduction i of part q) = W_q[i][j]
      for (int q = 0; q < 3; ++q)
        rp_jobs_add(&jobs, w[q] + (int64_t)p * pw, 1, k, pw, n, nullptr, img + (size_t)p * pe + (size_t)q * part, LIN_PANEL_BLOCKS);
    NRL_TRY(rp_jobs_launch(jobs, st));
  }
  RpImage im;
  im.img = img; im.nblk = LIN_PANEL_BLOCKS; im.kblocks = kb;
  const KCParts A{d_c, n, m, n, m * (int64_t)n};
  if (add != nullptr) return launch_rp_gemm<LIN_PANEL_BLOCKS, 4, 0, 2>(A, im, EpiAddStore{d_a, k, add}, m, k, 3 * n, st, P);
  return launch_rp_gemm<LIN_PANEL_BLOCKS, 4, 0, 2>(A, im, EpiStore{d_a, k}, m, k, 3 * n, st, P);
}

// d_a (m, k) = add (m, k) + d_c (m, n) W (n, k): nrl_linear_bwd_img's activation gradient landing on a residual stream
int nrl_linear_dgrad_add_img(const float* d_c, const float* w, int64_t m, int32_t n, int32_t k, const float* add, float* d_a,
                             void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(d_c && w && add && d_a && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0, "linear_dgrad_add: bad arguments");
  NRL_REQUIRE((((uintptr_t)d_c | (uintptr_t)w | (uintptr_t)add | (uintptr_t)d_a) & 15) == 0, "linear_dgrad_add: 16-byte alignment");
  NRL_REQUIRE(cur_engine() == ENGINE_BF16X3 && lin_panels_on(k), "linear_dgrad_add: bf16x3 engine and k >= 256 (nrl_linear_gelu_supported)");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0 && ws_bytes >= nrl_linear_workspace_bytes(n, k), "linear_dgrad_add: workspace");
  if (m == 0) return NRL_OK;
  return linear_panels(d_c, w, 1, k, k, n, EpiAddStore{d_a, k, add}, m, (uint16_t*)ws, (hipStream_t)stream, image_ready == 0);
}

int32_t nrl_linear_gelu_supported(int32_t n_wide) { return (cur_engine() == ENGINE_BF16X3 && lin_panels_on(n_wide)) ? 1 : 0; }

int nrl_embedding_grad(const float* d_out, const int64_t* ids, const int64_t* sorted_positions, int64_t n_ids, int32_t dim,
                       int64_t padding_idx, float* d_table, void* stream) {
  NRL_REQUIRE(d_out && ids && sorted_positions && d_table && n_ids >= 0 && dim > 0, "embedding_grad: bad arguments");
  return embedding_grad_any(d_out, ids, sorted_positions, n_ids, dim, padding_idx, d_table, (hipStream_t)stream);
}

size_t nrl_linear_workspace_bytes(int32_t n, int32_t k) {
  // bf16 planes of W and W^T (tiled kernels), or the panel images of the forward (n columns over k) / the activation
  // gradient (k columns over n), whichever is larger
  size_t e = split_weight_elems(n, k);
  if (lin_image_elems(n, k) > e) e = lin_image_elems(n, k);
  if (lin_image_elems(k, n) > e) e = lin_image_elems(k, n);
  return align_up(e * sizeof(uint16_t), 256);
}

int nrl_linear_fwd(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k,
                   float* c, void* ws, size_t ws_bytes, void* stream) {
  return nrl_linear_fwd_img(a, w, bias, m, n, k, c, ws, ws_bytes, 0, stream);
}

int nrl_linear_fwd_img(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k,
                       float* c, void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(a && w && c && m >= 0 && n > 0 && k > 0 && k % 4 == 0, "linear_fwd: bad arguments (k % 4 == 0)");
  NRL_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0, "linear_fwd: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const EpiLinear epi{c, n, bias, 0, make_dropout(0.0, 0, 0), n};
  if (ws == nullptr || cur_engine() != ENGINE_BF16X3)
    return launch_gemm<NRL_TILE>(KCPlain{a, k, m}, KCPlain{w, k, n}, epi, m, n, k, 1, st);
  NRL_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < nrl_linear_workspace_bytes(n, k)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, nrl_linear_workspace_bytes(n, k));
    return NRL_E_WORKSPACE;
  }
  if (m == 0) return NRL_OK;
  if (lin_panels_on(n)) return linear_panels(a, w, k, 1, n, k, epi, m, (uint16_t*)ws, st, image_ready == 0);   // element (n, k) = W[n][k]
  SplitWeight sw;
  NRL_TRY(split_weight(w, n, k, (uint16_t*)ws, &sw, st));
  return gemm_fwd(KCPlain{a, k, m}, w, sw, epi, m, n, k, n <= 224, st);
}

// Backward of nrl_linear_fwd: d_a (m, k) = d_c W, d_w += d_c^T a, d_bias += colsum(d_c).  d_a == NULL skips the
// activation gradient, d_w == NULL (with d_bias == NULL) the weight gradient -- a FROZEN nn.Linear inside a trainable
// stack (the PLM body: layers 0-7 frozen, their inputs still need gradients, text.py:69-73).
int nrl_linear_bwd(const float* a, const float* w, const float* d_c, int64_t m, int32_t n, int32_t k, float* d_a,
                   float* d_w, float* d_bias, void* ws, size_t ws_bytes, void* stream) {
  return nrl_linear_bwd_img(a, w, d_c, m, n, k, d_a, d_w, d_bias, ws, ws_bytes, 0, stream);
}

int nrl_linear_bwd_img(const float* a, const float* w, const float* d_c, int64_t m, int32_t n, int32_t k, float* d_a,
                       float* d_w, float* d_bias, void* ws, size_t ws_bytes, int32_t image_ready, void* stream) {
  NRL_REQUIRE(w && d_c && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0, "linear_bwd: bad arguments (n, k multiples of 4)");
  NRL_REQUIRE((d_w == nullptr) == (d_bias == nullptr), "linear_bwd: d_w and d_bias come together");
  NRL_REQUIRE(d_w == nullptr || a != nullptr, "linear_bwd: the weight gradient needs the forward's input");
  NRL_REQUIRE((((uintptr_t)a | (uintptr_t)w | (uintptr_t)d_c | (uintptr_t)d_a) & 15) == 0, "linear_bwd: operands must be 16-byte aligned");
  if (m == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  if (d_a != nullptr) {
    SplitWeight sw{};
    if (cur_engine() == ENGINE_BF16X3) {
      NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
      if (ws_bytes < nrl_linear_workspace_bytes(n, k)) {
        set_error("workspace too small: %zu < %zu bytes", ws_bytes, nrl_linear_workspace_bytes(n, k));
        return NRL_E_WORKSPACE;
      }
      if (lin_panels_on(k)) {        // d_a columns j over the reduction i: element = W[i][j]
        NRL_TRY(linear_panels(d_c, w, 1, k, k, n, EpiStore{d_a, k}, m, (uint16_t*)ws, st, image_ready == 0));
      } else {
        NRL_TRY(split_weight(w, n, k, (uint16_t*)ws, &sw, st));
        NRL_TRY(gemm_dgrad(d_c, w, sw, EpiStore{d_a, k}, m, n, k, st));
      }
    } else {
      NRL_TRY(gemm_dgrad(d_c, w, sw, EpiStore{d_a, k}, m, n, k, st));
    }
  }
  if (d_w != nullptr) NRL_TRY(gemm_wgrad(d_c, n, a, k, d_w, d_bias, m, st));
  return NRL_OK;
}

}  // extern "C"

