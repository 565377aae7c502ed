// C ABI of two generic blocks shared by the sibling recommenders (NAML / TANR / MINS ... user and news
// encoders): standalone additive attention and nn.Linear + activation.  Its own translation unit (shared internals: nrl_api_internal.h).

#include "nrl_api_internal.h"

namespace nrl {

struct AddAttWs {
  float *t, *w;
  uint16_t* planes;
};

static size_t addatt_ws_floats(int64_t M, int D, int Q) {
  auto al = [](size_t n) { return align_up(n, 64); };
  return al((size_t)M * Q) + al((size_t)M) + al((split_weight_elems(Q, D) + 1) / 2);
}

static int addatt_carve(void* ws, size_t ws_bytes, int64_t M, int D, int Q, AddAttWs* o) {
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < addatt_ws_floats(M, D, Q) * sizeof(float)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, addatt_ws_floats(M, D, Q) * sizeof(float));
    return NRL_E_WORKSPACE;
  }
  float* p = (float*)ws;
  auto take = [&](size_t n) { float* r = p; p += align_up(n, 64); return r; };
  o->t = take((size_t)M * Q);
  o->w = take((size_t)M);
  o->planes = reinterpret_cast<uint16_t*>(take((split_weight_elems(Q, D) + 1) / 2));
  return NRL_OK;
}

static int addatt_check(const NrlAddAttParams* p, int64_t groups, int64_t len) {
  NRL_REQUIRE(p != nullptr && p->att_weight && p->att_bias && p->att_query, "null additive-attention parameter");
  NRL_REQUIRE(p->dim > 0 && p->dim % 4 == 0 && p->query_dim > 0 && p->query_dim % 4 == 0,
              "dim and query_dim must be positive multiples of 4");
  NRL_REQUIRE(groups >= 0 && len > 0 && len <= 8192, "bad additive-attention shape");
  NRL_REQUIRE(((uintptr_t)p->att_weight & 15) == 0, "weight matrices must be 16-byte aligned");
  return NRL_OK;
}

// d_pre = d_c * act'(c), act' from the OUTPUT c: relu -> [c > 0], tanh -> 1 - c^2
__global__ void act_grad_kernel(const float4* __restrict__ d_c, const float4* __restrict__ c, int64_t n4, int act,
                                float4* __restrict__ d_pre) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 g = d_c[i];
    if (act != 0) {
      const float4 v = c[i];
      if (act == 2) {
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f;
        g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
      } else {
        g.x *= 1.f - v.x * v.x; g.y *= 1.f - v.y * v.y; g.z *= 1.f - v.z * v.z; g.w *= 1.f - v.w * v.w;
      }
    }
    d_pre[i] = g;
  }
}

struct MhaWs {
  float *qkv, *dqkv, *o, *d_o, *lse;
  uint16_t* planes;
};

static size_t mha_plane_elems(int D) { return split_weight_elems(3 * D, D) + split_weight_elems(D, D); }

static size_t mha_ws_floats(int64_t M, int D, int heads) {
  auto al = [](size_t n) { return align_up(n, 64); };
  return 2 * al((size_t)M * 3 * D) + 2 * al((size_t)M * D) + al((size_t)M * heads) + al((mha_plane_elems(D) + 1) / 2);
}

static int mha_carve(void* ws, size_t ws_bytes, int64_t M, int D, int heads, MhaWs* o) {
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < mha_ws_floats(M, D, heads) * sizeof(float)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, mha_ws_floats(M, D, heads) * sizeof(float));
    return NRL_E_WORKSPACE;
  }
  float* p = (float*)ws;
  auto take = [&](size_t n) { float* r = p; p += align_up(n, 64); return r; };
  o->qkv = take((size_t)M * 3 * D);
  o->dqkv = take((size_t)M * 3 * D);
  o->o = take((size_t)M * D);
  o->d_o = take((size_t)M * D);
  o->lse = take((size_t)M * heads);
  o->planes = reinterpret_cast<uint16_t*>(take((mha_plane_elems(D) + 1) / 2));
  return NRL_OK;
}

static int mha_check(const NrlMhaParams* p, int64_t S, int64_t Bt) {
  NRL_REQUIRE(p != nullptr && p->in_proj_weight && p->in_proj_bias && p->out_proj_weight && p->out_proj_bias,
              "null attention parameter");
  NRL_REQUIRE(engine_field_ok(p->gemm_engine), "gemm_engine must be 0 (default), 1 (f32) or 2 (bf16x3)");
  NRL_REQUIRE(p->embed_dim > 0 && p->embed_dim % 4 == 0, "embed_dim must be a positive multiple of 4");
  NRL_REQUIRE(p->num_heads > 0 && p->embed_dim % p->num_heads == 0, "embed_dim must be divisible by num_heads");
  NRL_REQUIRE(attn_head_dim_supported(p->embed_dim / p->num_heads), "head dim %d unsupported (16, 20, 32, 48, 64)",
              p->embed_dim / p->num_heads);
  NRL_REQUIRE(S > 0 && Bt > 0, "bad attention shape");
  NRL_REQUIRE(p->scale >= 0.f, "scale must be positive (0 = 1/sqrt(head dim))");
  NRL_REQUIRE((((uintptr_t)p->in_proj_weight | (uintptr_t)p->out_proj_weight) & 15) == 0,
              "weight matrices must be 16-byte aligned");
  return NRL_OK;
}

// x (S, Bt, D) seq-first: attention over the S axis for every (batch column, head)
static AttnGeom mha_geom(const NrlMhaParams* p, int64_t S, int64_t Bt) {
  AttnGeom g;
  const int D = p->embed_dim, dh = D / p->num_heads;
  g.q_outer = 3 * D; g.q_seq = Bt * 3 * D;
  g.o_outer = D; g.o_seq = Bt * D;
  g.groups = Bt * p->num_heads; g.heads = p->num_heads; g.S = (int)S; g.D = D; g.dh = dh;
  g.scale = p->scale > 0.f ? p->scale : 1.0f / sqrtf((float)dh);
  return g;
}

static int mha_planes(const NrlMhaParams* P, const MhaWs& w, bool fill, SplitWeight* in, SplitWeight* out,
                      hipStream_t st) {
  const int D = P->embed_dim;
  uint16_t* p = w.planes;
  if (fill) {
    NRL_TRY(split_weight(P->in_proj_weight, 3 * D, D, p, in, st));
    NRL_TRY(split_weight(P->out_proj_weight, D, D, p + split_weight_elems(3 * D, D), out, st));
  } else {
    *in = split_weight_view(p, 3 * D, D);
    *out = split_weight_view(p + split_weight_elems(3 * D, D), D, D);
  }
  return NRL_OK;
}

}  // namespace nrl

using namespace nrl;

extern "C" {

size_t nrl_additive_attention_workspace_bytes(int64_t groups, int64_t len, int32_t dim, int32_t query_dim) {
  return addatt_ws_floats(groups * len, dim, query_dim) * sizeof(float);
}

int nrl_additive_attention_fwd(const NrlAddAttParams* p, const float* y, int64_t groups, int64_t len,
                               int32_t save_for_backward, float* out, void* ws, size_t ws_bytes, void* stream) {
  (void)save_for_backward;
  NRL_TRY(addatt_check(p, groups, len));
  NRL_REQUIRE(y && out && ((uintptr_t)y & 15) == 0, "additive_attention_fwd: bad arguments");
  if (groups == 0) return NRL_OK;
  const int D = p->dim, Q = p->query_dim;
  const int64_t M = groups * len;
  AddAttWs w;
  NRL_TRY(addatt_carve(ws, ws_bytes, M, D, Q, &w));
  hipStream_t st = (hipStream_t)stream;
  SplitWeight sa = split_weight_view(w.planes, Q, D);
  if (cur_engine() == ENGINE_BF16X3) NRL_TRY(split_weight(p->att_weight, Q, D, w.planes, &sa, st));
  // t = tanh(y W_a^T + b_a); w = softmax(t . q_a); out = sum w y          (attention.py:34-40)
  NRL_TRY(gemm_fwd(KCPlain{y, D, M}, p->att_weight, sa, EpiLinear{w.t, Q, p->att_bias, 1, make_dropout(0.0, 0, 0), Q},
                   M, Q, D, true, st));
  NRL_TRY(pool_fwd(w.t, p->att_query, y, groups, (int)len, Q, D, w.w, out, st));
  return NRL_OK;
}

int nrl_additive_attention_bwd(const NrlAddAttParams* p, const NrlAddAttGrads* g, const float* y, int64_t groups,
                               int64_t len, const float* d_out, float* d_y, void* ws, size_t ws_bytes,
                               void* stream) {
  NRL_TRY(addatt_check(p, groups, len));
  NRL_REQUIRE(g && g->att_weight && g->att_bias && g->att_query, "null additive-attention gradient pointer");
  NRL_REQUIRE(y && d_out && d_y, "additive_attention_bwd: null argument");
  if (groups == 0) return NRL_OK;
  const int D = p->dim, Q = p->query_dim;
  const int64_t M = groups * len;
  AddAttWs w;
  NRL_TRY(addatt_carve(ws, ws_bytes, M, D, Q, &w));
  hipStream_t st = (hipStream_t)stream;
  const SplitWeight sa = split_weight_view(w.planes, Q, D);   // filled by the forward
  NRL_TRY(pool_bwd_pre(d_out, y, w.w, w.t, p->att_query, g->att_query, groups, (int)len, Q, D, st));
  // d_y = d_pre W_a + w * d_out
  NRL_TRY(gemm_dgrad(w.t, p->att_weight, sa, EpiPoolBwd{d_y, D, w.w, d_out, (int)len, make_dropout(0.0, 0, 0)}, M, Q,
                     D, st));
  NRL_TRY(gemm_wgrad(w.t, Q, y, D, g->att_weight, g->att_bias, M, st));
  return NRL_OK;
}

size_t nrl_linear_act_workspace_bytes(int64_t m, int32_t n, int32_t k) {
  (void)k;
  return align_up(split_weight_elems(n, k) * sizeof(uint16_t), 256) + align_up((size_t)m * n * sizeof(float), 256);
}

int nrl_linear_act_fwd(const float* a, const float* w, const float* bias, int64_t m, int32_t n, int32_t k,
                       int32_t act, float* c, void* ws, size_t ws_bytes, void* stream) {
  NRL_REQUIRE(a && w && c && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0,
              "linear_act_fwd: bad arguments (n, k multiples of 4)");
  NRL_REQUIRE(act >= 0 && act <= 2, "linear_act: act must be 0 (none), 1 (tanh) or 2 (relu)");
  NRL_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0 && ws != nullptr && ((uintptr_t)ws & 255) == 0,
              "linear_act_fwd: operands / workspace misaligned");
  if (ws_bytes < nrl_linear_act_workspace_bytes(m, n, k)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, nrl_linear_act_workspace_bytes(m, n, k));
    return NRL_E_WORKSPACE;
  }
  if (m == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  SplitWeight sw = split_weight_view((uint16_t*)ws, n, k);
  if (cur_engine() == ENGINE_BF16X3) NRL_TRY(split_weight(w, n, k, (uint16_t*)ws, &sw, st));
  return gemm_fwd(KCPlain{a, k, m}, w, sw, EpiLinear{c, n, bias, act, make_dropout(0.0, 0, 0), n}, m, n, k, n <= 224,
                  st);
}

int nrl_linear_act_bwd(const float* a, const float* w, const float* c, const float* d_c, int64_t m, int32_t n,
                       int32_t k, int32_t act, float* d_a, float* d_w, float* d_bias, void* ws, size_t ws_bytes,
                       void* stream) {
  NRL_REQUIRE(a && w && c && d_c && d_w && m >= 0 && n > 0 && k > 0 && k % 4 == 0 && n % 4 == 0,
              "linear_act_bwd: bad arguments");
  NRL_REQUIRE(act >= 0 && act <= 2, "linear_act: act must be 0 (none), 1 (tanh) or 2 (relu)");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < nrl_linear_act_workspace_bytes(m, n, k)) {
    set_error("workspace too small: %zu < %zu bytes", ws_bytes, nrl_linear_act_workspace_bytes(m, n, k));
    return NRL_E_WORKSPACE;
  }
  if (m == 0) return NRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const SplitWeight sw = split_weight_view((uint16_t*)ws, n, k);   // filled by the forward
  float* d_pre = reinterpret_cast<float*>((unsigned char*)ws + align_up(split_weight_elems(n, k) * sizeof(uint16_t), 256));
  const int64_t n4 = m * n / 4;
  hipLaunchKernelGGL(act_grad_kernel, dim3((unsigned)((n4 + 255) / 256 < 65535 * 16 ? (n4 + 255) / 256 : 65535 * 16)),
                     dim3(256), 0, st, (const float4*)d_c, (const float4*)c, n4, act, (float4*)d_pre);
  NRL_LAUNCH_CHECK();
  if (d_a != nullptr) NRL_TRY(gemm_dgrad(d_pre, w, sw, EpiStore{d_a, k}, m, n, k, st));
  NRL_TRY(gemm_wgrad(d_pre, n, a, k, d_w, d_bias, m, st));
  return NRL_OK;
}

// ---- nn.MultiheadAttention(x, x, x) (batch_first=False) on its own: MINS user encoder, user/mins.py:55-57 ----
size_t nrl_mha_workspace_bytes(int64_t seq, int64_t batch, int32_t embed_dim, int32_t num_heads) {
  if (seq <= 0 || batch <= 0 || embed_dim <= 0 || num_heads <= 0) return 0;
  return mha_ws_floats(seq * batch, embed_dim, num_heads) * sizeof(float);
}

int nrl_mha_fwd(const NrlMhaParams* p, const float* x, int64_t seq, int64_t batch, int32_t save_for_backward,
                float* out, void* ws, size_t ws_bytes, void* stream) {
  NRL_TRY(mha_check(p, seq, batch));
  const EngineScope engine_scope(p->gemm_engine);
  NRL_REQUIRE(x && out && ((uintptr_t)x & 15) == 0, "mha_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int D = p->embed_dim;
  const int64_t M = seq * batch;
  MhaWs w;
  NRL_TRY(mha_carve(ws, ws_bytes, M, D, p->num_heads, &w));
  SplitWeight in, outp;
  NRL_TRY(mha_planes(p, w, cur_engine() == ENGINE_BF16X3, &in, &outp, st));
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  NRL_TRY(gemm_fwd(KCPlain{x, D, M}, p->in_proj_weight, in, EpiLinear{w.qkv, 3 * D, p->in_proj_bias, 0, nodrop, 3 * D},
                   M, 3 * D, D, false, st));
  NRL_TRY(attn_fwd(w.qkv, w.o, save_for_backward ? w.lse : nullptr, mha_geom(p, seq, batch), st));
  return gemm_fwd(KCPlain{w.o, D, M}, p->out_proj_weight, outp, EpiLinear{out, D, p->out_proj_bias, 0, nodrop, D}, M,
                  D, D, false, st);
}

int nrl_mha_bwd(const NrlMhaParams* p, const NrlMhaGrads* g, const float* x, int64_t seq, int64_t batch,
                const float* d_out, float* d_x, void* ws, size_t ws_bytes, void* stream) {
  NRL_TRY(mha_check(p, seq, batch));
  const EngineScope engine_scope(p->gemm_engine);
  NRL_REQUIRE(g && g->in_proj_weight && g->in_proj_bias && g->out_proj_weight && g->out_proj_bias,
              "null gradient pointer");
  NRL_REQUIRE(x && d_out && d_x && (((uintptr_t)x | (uintptr_t)d_out) & 15) == 0, "mha_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int D = p->embed_dim;
  const int64_t M = seq * batch;
  MhaWs w;
  NRL_TRY(mha_carve(ws, ws_bytes, M, D, p->num_heads, &w));
  SplitWeight in, outp;
  NRL_TRY(mha_planes(p, w, false, &in, &outp, st));
  const Dropout nodrop = make_dropout(0.0, 0, 0);
  NRL_TRY(gemm_dgrad(d_out, p->out_proj_weight, outp, EpiStore{w.d_o, D}, M, D, D, st));
  NRL_TRY(attn_bwd(w.qkv, w.o, w.d_o, w.lse, w.dqkv, mha_geom(p, seq, batch), st));
  NRL_TRY(gemm_dgrad(w.dqkv, p->in_proj_weight, in, EpiLinear{d_x, D, nullptr, 0, nodrop, D}, M, 3 * D, D, st));
  NRL_TRY(gemm_wgrad(d_out, D, w.o, D, g->out_proj_weight, g->out_proj_bias, M, st));
  return gemm_wgrad(w.dqkv, 3 * D, x, D, g->in_proj_weight, g->in_proj_bias, M, st);
}

}  // extern "C"
