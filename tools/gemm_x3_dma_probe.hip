// Probe (not product code): DMA-staged bf16x3 GEMM vs the register-staged one -- equality and speed at
// the NRMS forward/dgrad shapes (M = 211200).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc -Itools tools/gemm_x3_dma_probe.hip -o tools/bin/gemm_x3_dma_probe
#include <stdarg.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "experimental/nrl_gemm_bf16x3_astat.h"

namespace nrl {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fprintf(stderr, "\n");
}
}  // namespace nrl
using namespace nrl;

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

static float *g_a, *g_w, *g_c, *g_c2, *g_bias, *g_bias2, *g_tbl, *g_x, *g_x2;
static int64_t* g_ids;
static uint16_t* g_planes;
static const int64_t M = 211200;
static int g_lda_pad = 0;  // extra floats per A row (alignment experiment)

// kind 0: dgrad (A (M,K) plain, W (K out, N in) -> transposed planes), EpiStore
// kind 1: fwd linear with bias + dropout epilogue (A plain, W (N,K))
// kind 2: fwd with gathered A + input dropout + x save
template <class F>
int with_b(int kind, int N, int K, hipStream_t st, F f) {
  SplitWeight sw;
  if (kind == 0) {
    if (split_weight(g_w, K, N, g_planes, &sw, st) != 0) return -1;
    return f(KCSplit{sw.hi_t, sw.lo_t, sw.ld_t, N});
  }
  if (split_weight(g_w, N, K, g_planes, &sw, st) != 0) return -1;
  return f(KCSplit{sw.hi, sw.lo, sw.ld, N});
}

template <int WM, int WN, int TM, int TN, int S, int ABL = 0, int PIPE = 0, int JC = TN>
int run_dma(int kind, int N, int K, float* c, float* xsave, hipStream_t st) {
  return with_b(kind, N, K, st, [&](KCSplit B) {
    if (kind == 0) return launch_gemm_bf16x3_dma<WM, WN, TM, TN, S, ABL, PIPE, JC>(KCPlain{g_a, K + g_lda_pad, M}, B, EpiStore{c, N}, M, N, K, st);
    EpiLinear e{c, N, g_bias, 0, make_dropout(0.2, 3, 1), N};
    if (kind == 1) return launch_gemm_bf16x3_dma<WM, WN, TM, TN, S, ABL, PIPE, JC>(KCPlain{g_a, K + g_lda_pad, M}, B, e, M, N, K, st);
    return launch_gemm_bf16x3_dma<WM, WN, TM, TN, S, ABL, PIPE, JC>(KCGather{g_tbl, g_ids, M, K, make_dropout(0.2, 1, 0), xsave}, B, e, M, N, K, st);
  });
}
template <int WM, int WN, int TM, int TN, int KT>
int run_astat(int kind, int N, int K, float* c, float* xsave, hipStream_t st) {
  return with_b(kind, N, K, st, [&](KCSplit B) {
    if (kind == 0) return launch_gemm_bf16x3_astat<WM, WN, TM, TN, KT>(KCPlain{g_a, K + g_lda_pad, M}, B, EpiStore{c, N}, M, N, K, st);
    EpiLinear e{c, N, g_bias, 0, make_dropout(0.2, 3, 1), N};
    if (kind == 1) return launch_gemm_bf16x3_astat<WM, WN, TM, TN, KT>(KCPlain{g_a, K + g_lda_pad, M}, B, e, M, N, K, st);
    return launch_gemm_bf16x3_astat<WM, WN, TM, TN, KT>(KCGather{g_tbl, g_ids, M, K, make_dropout(0.2, 1, 0), xsave}, B, e, M, N, K, st);
  });
}
template <int WM, int WN, int TM, int TN, int DEEP>
int run_reg(int kind, int N, int K, float* c, float* xsave, hipStream_t st) {
  return with_b(kind, N, K, st, [&](KCSplit B) {
    if (kind == 0) return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP>(KCPlain{g_a, K + g_lda_pad, M}, B, EpiStore{c, N}, M, N, K, 1, st);
    EpiLinear e{c, N, g_bias, 0, make_dropout(0.2, 3, 1), N};
    if (kind == 1) return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP>(KCPlain{g_a, K + g_lda_pad, M}, B, e, M, N, K, 1, st);
    return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP>(KCGather{g_tbl, g_ids, M, K, make_dropout(0.2, 1, 0), xsave}, B, e, M, N, K, 1, st);
  });
}

// kind 3: weight gradient dW (I, J) += dY (M, I)^T X (M, J), db via the ones-column; c must be zeroed first
static int tn_splits(int I, int J, int bm) {
  const int64_t tiles = ceil_div(I, bm) * ceil_div(J + 1, 160);
  int64_t sp = ceil_div(M, 1664);
  if (sp * tiles < 512) sp = ceil_div(512, tiles);
  const int64_t max_s = ceil_div(M, 256);
  return (int)(sp > max_s ? max_s : (sp < 1 ? 1 : sp));
}
template <int WM, int WN, int TM, int TN>
int run_reg_tn(int I, int J, float* c, hipStream_t st) {
  return launch_gemm_bf16x3<WM, WN, TM, TN, 0>(RCPlain{g_a, I, I, 0}, RCPlain{g_x, J, J, 1}, EpiAtomicWB{c, J, g_bias2, J}, I,
                                               J + 1, M, tn_splits(I, J, WM * TM * 16), st);
}
template <int WM, int WN, int TM, int TN, int S>
int run_dma_tn(int I, int J, float* c, hipStream_t st) {
  return launch_gemm_bf16x3_dma_tn<WM, WN, TM, TN, S>(RCPlain{g_a, I, I, 0}, RCPlain{g_x, J, J, 1},
                                                      EpiAtomicWB{c, J, g_bias2, J}, I, J + 1, M,
                                                      tn_splits(I, J, WM * TM * 16), st);
}

struct Case {
  std::string name;
  double flops;
  std::function<int(float*, float*, hipStream_t)> fn;
  int kind, N, K;
};

static double max_diff(const float* d0, const float* d1, size_t n) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost));
  double m = 0;
  for (size_t i = 0; i < n; ++i) m = std::max(m, (double)fabsf(a[i] - b[i]));
  return m;
}

int main(int argc, char** argv) {
  const int V = 70000;
  CK(hipMalloc(&g_a, M * 960 * 4));
  if (getenv("LDA_PAD")) g_lda_pad = atoi(getenv("LDA_PAD"));
  CK(hipMalloc(&g_w, 900 * 900 * 4));
  CK(hipMalloc(&g_c, M * 900 * 4));
  CK(hipMalloc(&g_c2, M * 900 * 4));
  CK(hipMalloc(&g_x, M * 300 * 4));
  CK(hipMalloc(&g_x2, M * 300 * 4));
  CK(hipMalloc(&g_bias, 1024 * 4));
  CK(hipMalloc(&g_bias2, 1024 * 4));
  CK(hipMalloc(&g_tbl, (size_t)V * 300 * 4));
  CK(hipMalloc(&g_ids, M * 8));
  CK(hipMalloc(&g_planes, split_weight_elems(900, 900) * 2 + 1024));
  {
    std::vector<float> h((size_t)M * 960);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(g_a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_w, h.data() + 777, 900 * 900 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_bias, h.data() + 5, 1024 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_tbl, h.data() + 99, (size_t)V * 300 * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> ids(M);
    for (auto& v : ids) { s = s * 1664525u + 1013904223u; v = (s >> 4) % V; }
    CK(hipMemcpy(g_ids, ids.data(), M * 8, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<Case> cases;
  struct Shape { const char* n; int kind, N, K; };
  const Shape shapes[] = {{"dgrad_o  N=300 K=300", 0, 300, 300}, {"dgrad_in N=300 K=900", 0, 300, 900},
                          {"fwd_out  N=300 K=300", 1, 300, 300}, {"fwd_att  N=200 K=300", 1, 200, 300},
                          {"gather   N=900 K=300", 2, 900, 300}, {"plain    N=900 K=300", 1, 900, 300}};
  for (const Shape& sh : shapes) {
    const double fl = 2.0 * M * sh.N * sh.K;
    const int kind = sh.kind, N = sh.N, K = sh.K;
#define REG(tag, ...) cases.push_back({std::string(sh.n) + " reg " tag, fl, [=](float* c, float* x, hipStream_t s) { return run_reg<__VA_ARGS__>(kind, N, K, c, x, s); }, kind, N, K});
#define DMA(tag, ...) cases.push_back({std::string(sh.n) + " dma " tag, fl, [=](float* c, float* x, hipStream_t s) { return run_dma<__VA_ARGS__>(kind, N, K, c, x, s); }, kind, N, K});
    REG("256x160 8w      ", 4, 2, 4, 5, 0)
    DMA("256x160 8w S=3  ", 4, 2, 4, 5, 3)
    DMA("256x160 8w S=2  ", 4, 2, 4, 5, 2)
    DMA("128x160 8w S=4  ", 4, 2, 2, 5, 4)
    DMA("128x160 8w S=3  ", 4, 2, 2, 5, 3)
    DMA("128x160 4w S=4  ", 2, 2, 4, 5, 4)
    DMA("128x224 8w S=3  ", 4, 2, 2, 7, 3)
    DMA("128x160 4w S=2  ", 2, 2, 4, 5, 2)
    DMA("128x160 4w S=3  ", 2, 2, 4, 5, 3)
    DMA("128x80 4w S=3   ", 4, 1, 2, 5, 3)
    DMA("64x160 4w S=2   ", 2, 2, 2, 5, 2)
    DMA("64x224 4w S=2   ", 2, 2, 2, 7, 2)
    DMA("128x112 4w S=2  ", 2, 2, 4, 7 / 2, 2)
    DMA("128x224 8w S=2  ", 4, 2, 2, 7, 2)
    if (K <= 320) {
#define AST(tag, ...) cases.push_back({std::string(sh.n) + " astat " tag, fl, [=](float* c, float* x, hipStream_t s) { return run_astat<__VA_ARGS__>(kind, N, K, c, x, s); }, kind, N, K});
      AST("96x160 6w       ", 3, 2, 2, 5, 10)
      AST("64x160 4w       ", 2, 2, 2, 5, 10)
      AST("96x160 6w (6x1) ", 6, 1, 1, 10, 10)
    }
    DMA("192x160 12w S=2 ", 6, 2, 2, 5, 2)
    DMA("256x160 16w S=2 ", 8, 2, 2, 5, 2)
    DMA("128x320 16w S=2 ", 4, 4, 2, 5, 2)
    DMA("192x160 12w S=3 ", 6, 2, 2, 5, 3)
    DMA("256x320 8w S=2 jc2", 4, 2, 4, 10, 2, 0, 0, 2)
    DMA("256x320 8w S=2 jc5", 4, 2, 4, 10, 2, 0, 0, 5)
    DMA("128x320 8w S=2 jc5", 2, 4, 4, 5, 2, 0, 0, 5)
    DMA("128x320 4w S=2 jc5", 2, 2, 4, 10, 2, 0, 0, 5)
    DMA("pipe 256x160 8w S=3", 4, 2, 4, 5, 3, 0, 1)
    DMA("pipe 128x160 4w S=3", 2, 2, 4, 5, 3, 0, 1)
    DMA("pipe 128x160 4w S=4", 2, 2, 4, 5, 4, 0, 1)
    DMA("pipe 128x160 8w S=3", 4, 2, 2, 5, 3, 0, 1)
    DMA("pipe 128x160 8w S=4", 4, 2, 2, 5, 4, 0, 1)
    DMA("pipe 64x160 4w S=3 ", 2, 2, 2, 5, 3, 0, 1)
    DMA("abl no-dma      ", 4, 2, 4, 5, 2, 1)
    DMA("abl no-split    ", 4, 2, 4, 5, 2, 2)
    DMA("abl no-mfma     ", 4, 2, 4, 5, 2, 4)
    DMA("abl no-barrier  ", 4, 2, 4, 5, 2, 8)
    DMA("abl no-dma,split", 4, 2, 4, 5, 2, 3)
    DMA("abl no-dma,mfma ", 4, 2, 4, 5, 2, 5)
    DMA("abl dma only    ", 4, 2, 4, 5, 2, 6)
  }
  {
    struct TShape { const char* n; int I, J; };
    const TShape ts[] = {{"wgrad_in 900x300", 900, 300}, {"wgrad_o  300x300", 300, 300}, {"wgrad_a  200x300", 200, 300}};
    for (const TShape& sh : ts) {
      const double fl = 2.0 * M * sh.I * sh.J;
      const int I = sh.I, J = sh.J;
#define TREG(tag, ...) cases.push_back({std::string(sh.n) + " reg " tag, fl, [=](float* c, float*, hipStream_t s) { return run_reg_tn<__VA_ARGS__>(I, J, c, s); }, 3, J, I});
#define TDMA(tag, ...) cases.push_back({std::string(sh.n) + " dma " tag, fl, [=](float* c, float*, hipStream_t s) { return run_dma_tn<__VA_ARGS__>(I, J, c, s); }, 3, J, I});
      if (I > 512) { TREG("256x160 8w      ", 4, 2, 4, 5) } else { TREG("64x160 4w       ", 2, 2, 2, 5) }
      TDMA("128x160 4w S=2  ", 2, 2, 4, 5, 2)
      TDMA("256x160 8w S=2  ", 4, 2, 4, 5, 2)
      TDMA("256x160 8w S=3  ", 4, 2, 4, 5, 3)
      TDMA("128x160 8w S=2  ", 4, 2, 2, 5, 2)
      TDMA("64x160 4w S=2   ", 2, 2, 2, 5, 2)
      TDMA("128x160 4w S=3  ", 2, 2, 4, 5, 3)
    }
  }
  if (argc > 1) {
    std::vector<Case> keep;
    for (auto& c : cases)
      if (c.name.find(argv[1]) != std::string::npos) keep.push_back(c);
    cases.swap(keep);
  }
  // equality against the first (register-staged) case of each shape
  size_t ref = 0;
  for (size_t i = 0; i < cases.size(); ++i) {
    if (cases[i].name.find(" reg ") != std::string::npos) {
      ref = i;
      continue;
    }
    if (cases[i].name.find(" abl ") != std::string::npos) continue;
    if (cases[i].kind == 3) {
      CK(hipMemset(g_c, 0, 1024 * 1024 * 4));
      CK(hipMemset(g_c2, 0, 1024 * 1024 * 4));
      CK(hipMemset(g_bias2, 0, 1024 * 4));
      CK(hipMemcpy(g_x, g_a + 12345, M * 300 * 4, hipMemcpyDeviceToDevice));
      if (cases[ref].fn(g_c, g_x, st) != 0 || cases[i].fn(g_c2, g_x2, st) != 0) return 1;
      CK(hipStreamSynchronize(st));
      const double d = max_diff(g_c, g_c2, (size_t)cases[i].K * cases[i].N);
      std::vector<float> h(16);
      CK(hipMemcpy(h.data(), g_c, 64, hipMemcpyDeviceToHost));
      printf("verify %-44s max|dma-reg| = %.3e (ref[0..2] = %.4f %.4f %.4f)\n", cases[i].name.c_str(), d, h[0], h[1], h[2]);
      continue;
    }
    CK(hipMemset(g_c, 0, M * 900 * 4));
    CK(hipMemset(g_c2, 0, M * 900 * 4));
    CK(hipMemset(g_x, 0, M * 300 * 4));
    CK(hipMemset(g_x2, 0, M * 300 * 4));
    if (cases[ref].fn(g_c, g_x, st) != 0 || cases[i].fn(g_c2, g_x2, st) != 0) return 1;
    CK(hipStreamSynchronize(st));
    const double d = max_diff(g_c, g_c2, (size_t)M * cases[i].N);
    const double dx = cases[i].kind == 2 ? max_diff(g_x, g_x2, (size_t)M * 300) : 0.0;
    printf("verify %-44s max|dma-reg| = %.3e  x-save diff %.3e\n", cases[i].name.c_str(), d, dx);
  }
  const int rounds = 5;
  std::vector<std::vector<float>> ms(cases.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r)
    for (size_t i = 0; i < cases.size(); ++i) {
      CK(hipEventRecord(e0, st));
      if (cases[i].fn(g_c, g_x, st) != 0) return 1;
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[i].push_back(t);
    }
  for (size_t i = 0; i < cases.size(); ++i) {
    std::sort(ms[i].begin(), ms[i].end());
    const float med = ms[i][ms[i].size() / 2];
    printf("%-48s median %7.3f ms  %6.1f TF\n", cases[i].name.c_str(), med, cases[i].flops / med / 1e9);
  }
  return 0;
}
