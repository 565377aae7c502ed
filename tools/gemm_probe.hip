// Tuning probe (not product code): times the fp32-MFMA GEMM template at the NRMS shapes for
// several tile configurations inside one process (interleaved rounds), prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/gemm_probe.hip -o gpurun_out/gemm_probe
#include <stdarg.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "experimental/nrl_gemm_dma.h"
#include "experimental/nrl_gemm_bf16x3_ws.h"

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

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e = (x);                                            \
    if (e != hipSuccess) {                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));       \
      exit(1);                                                     \
    }                                                              \
  } while (0)

struct Bufs {
  float *a, *w, *c, *bias, *tbl;
  int64_t* ids;
};

template <int WM, int WN, int TM, int TN, int BK>
int run_nt(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  return launch_gemm<WM, WN, TM, TN, BK>(KCPlain{b.a, K, M}, KCPlain{b.w, K, N},
                                         EpiLinear{b.c, N, b.bias, 0, make_dropout(0.0, 0, 0), N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN, int BK>
int run_gather(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  return launch_gemm<WM, WN, TM, TN, BK>(KCGather{b.tbl, b.ids, M, K, make_dropout(0.2, 1, 0), b.a},
                                         KCPlain{b.w, K, N},
                                         EpiLinear{b.c, N, b.bias, 0, make_dropout(0.0, 0, 0), N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN, int BK>
int run_nn(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {  // dgrad: A (M,K), W as (K,N)
  return launch_gemm<WM, WN, TM, TN, BK>(KCPlain{b.a, K, M}, RCPlain{b.w, N, N, 0}, EpiStore{b.c, N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN, int BK>
int run_tn(const Bufs& b, int64_t Mr, int I, int J, hipStream_t st) {  // wgrad: dW(I,J) = dY(Mr,I)^T X(Mr,J)
  const int64_t tiles = ceil_div(I, WM * TM * 16) * ceil_div(J + 1, WN * TN * 16);
  int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(2048, tiles), ceil_div(Mr, 8 * BK)));
  return launch_gemm<WM, WN, TM, TN, BK>(RCPlain{b.a, I, I, 0}, RCPlain{b.w, J, J, 1},
                                         EpiAtomicWB{b.c, J, b.bias, J}, I, J + 1, Mr, splits, st);
}

template <int WM, int WN, int TM, int TN>
int dma_nt(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  return launch_gemm_dma<WM, WN, TM, TN>(KCPlain{b.a, K, M}, KCPlain{b.w, K, N},
                                         EpiLinear{b.c, N, b.bias, 0, make_dropout(0.0, 0, 0), N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN>
int dma_gather(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  return launch_gemm_dma<WM, WN, TM, TN>(KCGather{b.tbl, b.ids, M, K, make_dropout(0.2, 1, 0), b.a},
                                         KCPlain{b.w, K, N},
                                         EpiLinear{b.c, N, b.bias, 0, make_dropout(0.0, 0, 0), N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN>
int dma_nn(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  return launch_gemm_dma<WM, WN, TM, TN>(KCPlain{b.a, K, M}, RCPlain{b.w, N, N, 0}, EpiStore{b.c, N}, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN>
int dma_tn(const Bufs& b, int64_t Mr, int I, int J, hipStream_t st) {
  const int64_t tiles = ceil_div(I, WM * TM * 16) * ceil_div(J + 1, WN * TN * 16);
  int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(2048, tiles), ceil_div(Mr, 8 * 16)));
  return launch_gemm_dma<WM, WN, TM, TN>(RCPlain{b.a, I, I, 0}, RCPlain{b.w, J, J, 1},
                                         EpiAtomicWB{b.c, J, b.bias, J}, I, J + 1, Mr, splits, st);
}

// ---- bf16x3 engine ---------------------------------------------------------------------------
static uint16_t* g_planes = nullptr;
template <int WM, int WN, int TM, int TN, int DEEP = 1, int OCC = 1, int WS = 0>
int x3_nt(const Bufs& b, int64_t M, int N, int K, hipStream_t st, bool gather = false) {
  SplitWeight sw;
  if (split_weight(b.w, N, K, g_planes, &sw, st) != 0) return -1;
  KCSplit B{sw.hi, sw.lo, sw.Kp, N};
  EpiLinear e{b.c, N, b.bias, 0, make_dropout(0.0, 0, 0), N};
  if (gather)
    return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP, OCC, WS>(KCGather{b.tbl, b.ids, M, K, make_dropout(0.2, 1, 0), b.a}, B, e, M, N, K, 1, st);
  return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP, OCC, WS>(KCPlain{b.a, K, M}, B, e, M, N, K, 1, st);
}
template <int WM, int WN, int TM, int TN, int DEEP = 1, int OCC = 1, int WS = 0>
int x3_nn(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {  // b.w is W (K rows = out, N cols = in)
  SplitWeight sw;
  if (split_weight(b.w, K, N, g_planes, &sw, st) != 0) return -1;  // W is (out=K, in=N): transposed planes [N][Kp']
  KCSplit B{sw.hi_t, sw.lo_t, sw.Np, N};
  return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP, OCC, WS>(KCPlain{b.a, K, M}, B, EpiStore{b.c, N}, M, N, K, 1, st);
}
static int g_x3_splits = 0;  // 0 = heuristic
template <int WM, int WN, int TM, int TN, int DEEP = 1, int WS = 0>
int x3_tn(const Bufs& b, int64_t Mr, int I, int J, hipStream_t st) {
  const int64_t tiles = ceil_div(I, WM * TM * 16) * ceil_div(J + 1, WN * TN * 16);
  int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(2048, tiles), ceil_div(Mr, 8 * 32)));
  if (g_x3_splits > 0) splits = g_x3_splits;
  return launch_gemm_bf16x3<WM, WN, TM, TN, DEEP, 1, WS>(RCPlain{b.a, I, I, 0}, RCPlain{b.w, J, J, 1},
                                                         EpiAtomicWB{b.c, J, b.bias, J}, I, J + 1, Mr, splits, st);
}

struct EpiNull {  // keeps the accumulators alive, writes nothing
  struct Row {};
  __device__ __forceinline__ Row row(int64_t) const { return Row{}; }
  __device__ __forceinline__ void operator()(const Row&, int64_t, int, float v) const { asm volatile("" ::"v"(v)); }
};
template <int ABL, bool NOEPI>
int run_nn_abl(const Bufs& b, int64_t M, int N, int K, hipStream_t st) {
  if (NOEPI)
    return launch_gemm<4, 2, 2, 5, 16, ABL>(KCPlain{b.a, K, M}, RCPlain{b.w, N, N, 0}, EpiNull{}, M, N, K, 1, st);
  return launch_gemm<4, 2, 2, 5, 16, ABL>(KCPlain{b.a, K, M}, RCPlain{b.w, N, N, 0}, EpiStore{b.c, N}, M, N, K, 1, st);
}

struct Case {
  std::string name;
  double flops;
  std::function<int(hipStream_t)> fn;
};

int main(int argc, char** argv) {
  const int64_t M = 211200;  // B=128: 7040 news x 30 tokens
  const int V = 70000, D = 300;
  Bufs b;
  CK(hipMalloc(&b.a, (size_t)M * 912 * 4));
  CK(hipMalloc(&b.w, (size_t)M * 304 * 4));
  CK(hipMalloc(&b.c, (size_t)M * 912 * 4));
  CK(hipMalloc(&b.bias, 4096 * 4));
  CK(hipMalloc(&b.tbl, (size_t)V * D * 4));
  CK(hipMalloc(&b.ids, (size_t)M * 8));
  {
    std::vector<float> h((size_t)M * 912);
    uint32_t s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    CK(hipMemcpy(b.a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.w, h.data(), (size_t)M * 304 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.tbl, h.data(), (size_t)V * D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.bias, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> ids(M);
    for (auto& x : ids) { s = s * 1664525u + 1013904223u; x = (s >> 4) % V; }
    CK(hipMemcpy(b.ids, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  {  // ---- correctness: LDS-DMA kernel vs the register-staged kernel on ragged shapes ----------
    const int64_t Mv = 1000 + 37;
    auto fetch = [&](float* dev, size_t n) {
      std::vector<float> h(n);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost));
      return h;
    };
    auto maxdiff = [](const std::vector<float>& x, const std::vector<float>& y) {
      double d = 0;
      for (size_t i = 0; i < x.size(); ++i) d = std::max(d, (double)fabsf(x[i] - y[i]));
      return d;
    };
    struct V { const char* name; std::function<int()> ref, dma; size_t n; bool zero; };
    float* xsave = b.a + (size_t)M * 600;  // scratch region inside b.a for the gather's x output
    Bufs bx = b; bx.a = xsave;
    CK(hipMalloc(&g_planes, (size_t)8 << 20));
    std::vector<V> vs = {
      {"x3 nt 300x300", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 300, 300, st); }, [&] { return x3_nt<4, 2, 2, 5>(b, Mv, 300, 300, st); }, (size_t)Mv * 300, false},
      {"x3 nt 200x300", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 200, 300, st); }, [&] { return x3_nt<4, 2, 2, 5>(b, Mv, 200, 300, st); }, (size_t)Mv * 200, false},
      {"x3 gather 900x300 (+dropout)", [&] { return run_gather<4, 2, 2, 5, 16>(bx, Mv, 900, 300, st); }, [&] { return x3_nt<4, 2, 2, 5>(bx, Mv, 900, 300, st, true); }, (size_t)Mv * 900, false},
      {"x3 nn 300x900", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 900, st); }, [&] { return x3_nn<4, 2, 2, 5>(b, Mv, 300, 900, st); }, (size_t)Mv * 300, false},
      {"x3 nn 300x200", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 200, st); }, [&] { return x3_nn<4, 2, 2, 5>(b, Mv, 300, 200, st); }, (size_t)Mv * 300, false},
      {"x3 tn 900x300", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 900, 300, st); }, [&] { return x3_tn<4, 2, 2, 5>(b, Mv, 900, 300, st); }, (size_t)900 * 300, true},
      {"x3rot nt 300x300", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 300, 300, st); }, [&] { return x3_nt<2, 2, 4, 5, 0, 1, 2>(b, Mv, 300, 300, st); }, (size_t)Mv * 300, false},
      {"x3rot gather 900x300 (+dropout)", [&] { return run_gather<4, 2, 2, 5, 16>(bx, Mv, 900, 300, st); }, [&] { return x3_nt<2, 2, 4, 5, 0, 1, 2>(bx, Mv, 900, 300, st, true); }, (size_t)Mv * 900, false},
      {"x3rot nn 300x200", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 200, st); }, [&] { return x3_nn<2, 2, 4, 5, 0, 1, 2>(b, Mv, 300, 200, st); }, (size_t)Mv * 300, false},
      {"x3rot tn 900x300", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 900, 300, st); }, [&] { return x3_tn<2, 2, 4, 5, 0, 2>(b, Mv, 900, 300, st); }, (size_t)900 * 300, true},
      {"x3ws nt 300x300", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 300, 300, st); }, [&] { return x3_nt<2, 2, 4, 5, 0, 1, 1>(b, Mv, 300, 300, st); }, (size_t)Mv * 300, false},
      {"x3ws gather 900x300 (+dropout)", [&] { return run_gather<4, 2, 2, 5, 16>(bx, Mv, 900, 300, st); }, [&] { return x3_nt<2, 2, 4, 5, 0, 1, 1>(bx, Mv, 900, 300, st, true); }, (size_t)Mv * 900, false},
      {"x3ws nn 300x900", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 900, st); }, [&] { return x3_nn<2, 2, 4, 5, 0, 1, 1>(b, Mv, 300, 900, st); }, (size_t)Mv * 300, false},
      {"x3ws tn 900x300", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 900, 300, st); }, [&] { return x3_tn<2, 2, 4, 5, 0, 1>(b, Mv, 900, 300, st); }, (size_t)900 * 300, true},
      {"x3 tn 200x300 (4w)", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 200, 300, st); }, [&] { return x3_tn<2, 2, 2, 5>(b, Mv, 200, 300, st); }, (size_t)200 * 300, true},
      {"nt 300x300", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 300, 300, st); }, [&] { return dma_nt<4, 2, 2, 5>(b, Mv, 300, 300, st); }, (size_t)Mv * 300, false},
      {"nt 200x300 (8x1w)", [&] { return run_nt<4, 2, 2, 5, 16>(b, Mv, 200, 300, st); }, [&] { return dma_nt<8, 1, 1, 13>(b, Mv, 200, 300, st); }, (size_t)Mv * 200, false},
      {"gather 900x300 (+dropout)", [&] { return run_gather<4, 2, 2, 5, 16>(bx, Mv, 900, 300, st); }, [&] { return dma_gather<4, 2, 2, 5>(bx, Mv, 900, 300, st); }, (size_t)Mv * 900, false},
      {"nn 300x900", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 900, st); }, [&] { return dma_nn<4, 2, 2, 5>(b, Mv, 300, 900, st); }, (size_t)Mv * 300, false},
      {"nn 300x200", [&] { return run_nn<4, 2, 2, 5, 16>(b, Mv, 300, 200, st); }, [&] { return dma_nn<4, 2, 2, 5>(b, Mv, 300, 200, st); }, (size_t)Mv * 300, false},
      {"tn 900x300", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 900, 300, st); }, [&] { return dma_tn<4, 2, 2, 5>(b, Mv, 900, 300, st); }, (size_t)900 * 300, true},
      {"tn 200x300 (4w)", [&] { return run_tn<4, 2, 2, 5, 16>(b, Mv, 200, 300, st); }, [&] { return dma_tn<2, 2, 2, 5>(b, Mv, 200, 300, st); }, (size_t)200 * 300, true},
    };
    for (auto& v : vs) {
      std::vector<float> r[2], xs[2], bias[2];
      for (int w = 0; w < 2; ++w) {
        CK(hipMemsetAsync(b.c, v.zero ? 0 : 0xFF, v.n * 4, st));
        CK(hipMemsetAsync(xsave, 0, (size_t)Mv * 300 * 4, st));
        if (v.zero) CK(hipMemsetAsync(b.bias, 0, 4096 * 4, st));
        if ((w == 0 ? v.ref() : v.dma()) != 0) { fprintf(stderr, "verify launch failed\n"); return 1; }
        r[w] = fetch(b.c, v.n);
        xs[w] = fetch(xsave, (size_t)Mv * 300);
        bias[w] = fetch(b.bias, 1024);
      }
      printf("verify %-28s max|dma-ref| = %.3e   x-save diff %.3e   bias diff %.3e\n", v.name, maxdiff(r[0], r[1]),
             maxdiff(xs[0], xs[1]), maxdiff(bias[0], bias[1]));
    }
    // restore the bias buffer content used by the timing cases
    CK(hipMemsetAsync(b.bias, 0, 4096 * 4, st));
  }
  std::vector<Case> cases;
#define ADD_CFG(tag, WM, WN, TM, TN, BK)                                                                    \
  cases.push_back({std::string("gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                       \
                   [=](hipStream_t s) { return run_gather<WM, WN, TM, TN, BK>(b, M, 900, 300, s); }});       \
  cases.push_back({std::string("nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                       \
                   [=](hipStream_t s) { return run_nt<WM, WN, TM, TN, BK>(b, M, 300, 300, s); }});           \
  cases.push_back({std::string("nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                       \
                   [=](hipStream_t s) { return run_nt<WM, WN, TM, TN, BK>(b, M, 200, 300, s); }});           \
  cases.push_back({std::string("nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                       \
                   [=](hipStream_t s) { return run_nn<WM, WN, TM, TN, BK>(b, M, 300, 900, s); }});           \
  cases.push_back({std::string("nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                       \
                   [=](hipStream_t s) { return run_nn<WM, WN, TM, TN, BK>(b, M, 300, 300, s); }});           \
  cases.push_back({std::string("tn_wgrad_in 900x300     ") + tag, 2.0 * M * 900 * 300,                       \
                   [=](hipStream_t s) { return run_tn<WM, WN, TM, TN, BK>(b, M, 900, 300, s); }});           \
  cases.push_back({std::string("tn_wgrad_o  300x300     ") + tag, 2.0 * M * 300 * 300,                       \
                   [=](hipStream_t s) { return run_tn<WM, WN, TM, TN, BK>(b, M, 300, 300, s); }});           \
  cases.push_back({std::string("tn_wgrad_a  200x300     ") + tag, 2.0 * M * 200 * 300,                       \
                   [=](hipStream_t s) { return run_tn<WM, WN, TM, TN, BK>(b, M, 200, 300, s); }});           \
  cases.push_back({std::string("nn_dgrad_a  N=300 K=200 ") + tag, 2.0 * M * 300 * 200,                       \
                   [=](hipStream_t s) { return run_nn<WM, WN, TM, TN, BK>(b, M, 300, 200, s); }});

  ADD_CFG("4x2w 2x5b bk16 (128x160 8w)", 4, 2, 2, 5, 16)
  ADD_CFG("4x2w 2x5b bk32 (128x160 8w)", 4, 2, 2, 5, 32)
  ADD_CFG("8x1w 1x13b bk16 (128x208 8w)", 8, 1, 1, 13, 16)
  ADD_CFG("4x2w 2x7b bk16 (128x224 8w)", 4, 2, 2, 7, 16)
  ADD_CFG("2x4w 4x3b bk16 (128x192 8w)", 2, 4, 4, 3, 16)
  ADD_CFG("2x4w 3x5b bk16 (96x320 8w)", 2, 4, 3, 5, 16)
  ADD_CFG("4x4w 2x5b bk16 (128x320 16w)", 4, 4, 2, 5, 16)
  ADD_CFG("2x2w 2x5b bk16 (64x160 4w)", 2, 2, 2, 5, 16)
  ADD_CFG("4x2w 1x5b bk16 (64x160 8w)", 4, 2, 1, 5, 16)

#define ADD_DMA(tag, WM, WN, TM, TN)                                                                       \
  cases.push_back({std::string("dma gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                   \
                   [=](hipStream_t s) { return dma_gather<WM, WN, TM, TN>(b, M, 900, 300, s); }});           \
  cases.push_back({std::string("dma nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                   \
                   [=](hipStream_t s) { return dma_nt<WM, WN, TM, TN>(b, M, 300, 300, s); }});               \
  cases.push_back({std::string("dma nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                   \
                   [=](hipStream_t s) { return dma_nt<WM, WN, TM, TN>(b, M, 200, 300, s); }});               \
  cases.push_back({std::string("dma nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                   \
                   [=](hipStream_t s) { return dma_nn<WM, WN, TM, TN>(b, M, 300, 900, s); }});               \
  cases.push_back({std::string("dma nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                   \
                   [=](hipStream_t s) { return dma_nn<WM, WN, TM, TN>(b, M, 300, 300, s); }});               \
  cases.push_back({std::string("dma tn_wgrad_in 900x300     ") + tag, 2.0 * M * 900 * 300,                   \
                   [=](hipStream_t s) { return dma_tn<WM, WN, TM, TN>(b, M, 900, 300, s); }});               \
  cases.push_back({std::string("dma tn_wgrad_o  300x300     ") + tag, 2.0 * M * 300 * 300,                   \
                   [=](hipStream_t s) { return dma_tn<WM, WN, TM, TN>(b, M, 300, 300, s); }});               \
  cases.push_back({std::string("dma tn_wgrad_a  200x300     ") + tag, 2.0 * M * 200 * 300,                   \
                   [=](hipStream_t s) { return dma_tn<WM, WN, TM, TN>(b, M, 200, 300, s); }});
  ADD_DMA("4x2w 2x5b (128x160 8w)", 4, 2, 2, 5)
  ADD_DMA("2x2w 2x5b (64x160 4w)", 2, 2, 2, 5)
  ADD_DMA("8x1w 1x13b (128x208 8w)", 8, 1, 1, 13)
  ADD_DMA("2x2w 4x5b (128x160 4w)", 2, 2, 4, 5)
#define ADD_X3(tag, WM, WN, TM, TN)                                                                        \
  cases.push_back({std::string("x3 gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                    \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN>(b, M, 900, 300, s, true); }});          \
  cases.push_back({std::string("x3 nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                    \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN>(b, M, 300, 300, s); }});                \
  cases.push_back({std::string("x3 nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                    \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN>(b, M, 200, 300, s); }});                \
  cases.push_back({std::string("x3 nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                    \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN>(b, M, 300, 900, s); }});                \
  cases.push_back({std::string("x3 nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                    \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN>(b, M, 300, 300, s); }});                \
  cases.push_back({std::string("x3 tn_wgrad_in 900x300     ") + tag, 2.0 * M * 900 * 300,                    \
                   [=](hipStream_t s) { return x3_tn<WM, WN, TM, TN>(b, M, 900, 300, s); }});                \
  cases.push_back({std::string("x3 tn_wgrad_o  300x300     ") + tag, 2.0 * M * 300 * 300,                    \
                   [=](hipStream_t s) { return x3_tn<WM, WN, TM, TN>(b, M, 300, 300, s); }});                \
  cases.push_back({std::string("x3 tn_wgrad_a  200x300     ") + tag, 2.0 * M * 200 * 300,                    \
                   [=](hipStream_t s) { return x3_tn<WM, WN, TM, TN>(b, M, 200, 300, s); }});
#define ADD_X3D(tag, WM, WN, TM, TN, DEEP)                                                                 \
  cases.push_back({std::string("x3deep gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, DEEP>(b, M, 900, 300, s, true); }});    \
  cases.push_back({std::string("x3deep nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, DEEP>(b, M, 300, 300, s); }});          \
  cases.push_back({std::string("x3deep nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, DEEP>(b, M, 200, 300, s); }});          \
  cases.push_back({std::string("x3deep nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, DEEP>(b, M, 300, 900, s); }});          \
  cases.push_back({std::string("x3deep nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, DEEP>(b, M, 300, 300, s); }});          \
  cases.push_back({std::string("x3deep tn_wgrad_in 900x300 S128 ") + tag, 2.0 * M * 900 * 300,               \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, DEEP>(b, M, 900, 300, s); g_x3_splits = 0; return r; }}); \
  cases.push_back({std::string("x3deep tn_wgrad_o  300x300 S128 ") + tag, 2.0 * M * 300 * 300,               \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, DEEP>(b, M, 300, 300, s); g_x3_splits = 0; return r; }});
#define ADD_X3W(tag, WM, WN, TM, TN)                                                                       \
  cases.push_back({std::string("x3ws gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                  \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 1>(b, M, 900, 300, s, true); }}); \
  cases.push_back({std::string("x3ws nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                  \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 1>(b, M, 300, 300, s); }});       \
  cases.push_back({std::string("x3ws nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                  \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 1>(b, M, 200, 300, s); }});       \
  cases.push_back({std::string("x3ws nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                  \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, 1, 1>(b, M, 300, 900, s); }});       \
  cases.push_back({std::string("x3ws nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                  \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, 1, 1>(b, M, 300, 300, s); }});       \
  cases.push_back({std::string("x3ws tn_wgrad_in 900x300 S128 ") + tag, 2.0 * M * 900 * 300,                 \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, 0, 1>(b, M, 900, 300, s); g_x3_splits = 0; return r; }}); \
  cases.push_back({std::string("x3ws tn_wgrad_o  300x300 S128 ") + tag, 2.0 * M * 300 * 300,                 \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, 0, 1>(b, M, 300, 300, s); g_x3_splits = 0; return r; }});
#define ADD_X3R(tag, WM, WN, TM, TN)                                                                       \
  cases.push_back({std::string("x3rot gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                 \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 2>(b, M, 900, 300, s, true); }}); \
  cases.push_back({std::string("x3rot nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                 \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 2>(b, M, 300, 300, s); }});       \
  cases.push_back({std::string("x3rot nt_addatt   N=200 K=300 ") + tag, 2.0 * M * 200 * 300,                 \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, 1, 2>(b, M, 200, 300, s); }});       \
  cases.push_back({std::string("x3rot nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                 \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, 1, 2>(b, M, 300, 900, s); }});       \
  cases.push_back({std::string("x3rot nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                 \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, 1, 2>(b, M, 300, 300, s); }});       \
  cases.push_back({std::string("x3rot tn_wgrad_in 900x300 S128 ") + tag, 2.0 * M * 900 * 300,                \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, 0, 2>(b, M, 900, 300, s); g_x3_splits = 0; return r; }}); \
  cases.push_back({std::string("x3rot tn_wgrad_o  300x300 S128 ") + tag, 2.0 * M * 300 * 300,                \
                   [=](hipStream_t s) { g_x3_splits = 128; int r = x3_tn<WM, WN, TM, TN, 0, 2>(b, M, 300, 300, s); g_x3_splits = 0; return r; }});
  ADD_X3R("128x160", 2, 2, 4, 5)
  ADD_X3R("64x160", 2, 2, 2, 5)
  ADD_X3W("128x160", 2, 2, 4, 5)
  ADD_X3W("64x160", 2, 2, 2, 5)
  ADD_X3W("128x224", 2, 2, 4, 7)
#define ADD_X3O(tag, WM, WN, TM, TN, OCC)                                                                  \
  cases.push_back({std::string("x3occ gather_qkv  N=900 K=300 ") + tag, 2.0 * M * 900 * 300,                 \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, OCC>(b, M, 900, 300, s, true); }});  \
  cases.push_back({std::string("x3occ nt_outproj  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                 \
                   [=](hipStream_t s) { return x3_nt<WM, WN, TM, TN, 0, OCC>(b, M, 300, 300, s); }});        \
  cases.push_back({std::string("x3occ nn_dgrad_in N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                 \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, OCC>(b, M, 300, 900, s); }});        \
  cases.push_back({std::string("x3occ nn_dgrad_o  N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                 \
                   [=](hipStream_t s) { return x3_nn<WM, WN, TM, TN, 0, OCC>(b, M, 300, 300, s); }});
  ADD_X3O("128x160 8w occ4", 4, 2, 2, 5, 4)
  ADD_X3O("128x160 8w occ3", 4, 2, 2, 5, 3)
  ADD_X3O("128x160 8w occ2", 4, 2, 2, 5, 2)
  ADD_X3O("64x160 4w occ4", 2, 2, 2, 5, 4)
  ADD_X3O("64x160 4w occ3", 2, 2, 2, 5, 3)
  ADD_X3O("128x160 4w occ2", 2, 2, 4, 5, 2)
  ADD_X3D("128x160 4w shallow", 2, 2, 4, 5, 0)
  ADD_X3D("128x160 4w deep", 2, 2, 4, 5, 1)
  ADD_X3D("64x320 4w shallow", 1, 4, 4, 5, 0)
  ADD_X3D("128x160 deep", 4, 2, 2, 5, 1)
  ADD_X3D("128x160 shallow", 4, 2, 2, 5, 0)
  ADD_X3D("256x160 deep(spill)", 4, 2, 4, 5, 1)
  ADD_X3D("256x160 shallow", 4, 2, 4, 5, 0)
  ADD_X3D("64x160 deep", 2, 2, 2, 5, 1)
  ADD_X3D("64x160 shallow", 2, 2, 2, 5, 0)
  ADD_X3D("128x224 deep", 4, 2, 2, 7, 1)
  ADD_X3D("128x224 shallow", 4, 2, 2, 7, 0)
#define ADD_X3S(S, tag, WM, WN, TM, TN)                                                                    \
  cases.push_back({std::string("x3split tn_wgrad_in 900x300 S=" #S " ") + tag, 2.0 * M * 900 * 300,         \
                   [=](hipStream_t s) { g_x3_splits = S; int r = x3_tn<WM, WN, TM, TN>(b, M, 900, 300, s); g_x3_splits = 0; return r; }}); \
  cases.push_back({std::string("x3split tn_wgrad_o  300x300 S=" #S " ") + tag, 2.0 * M * 300 * 300,         \
                   [=](hipStream_t s) { g_x3_splits = S; int r = x3_tn<WM, WN, TM, TN>(b, M, 300, 300, s); g_x3_splits = 0; return r; }}); \
  cases.push_back({std::string("x3split tn_wgrad_a  200x300 S=" #S " ") + tag, 2.0 * M * 200 * 300,         \
                   [=](hipStream_t s) { g_x3_splits = S; int r = x3_tn<WM, WN, TM, TN>(b, M, 200, 300, s); g_x3_splits = 0; return r; }});
  ADD_X3S(32, "64x160", 2, 2, 2, 5)
  ADD_X3S(64, "64x160", 2, 2, 2, 5)
  ADD_X3S(128, "64x160", 2, 2, 2, 5)
  ADD_X3S(256, "64x160", 2, 2, 2, 5)
  ADD_X3S(512, "64x160", 2, 2, 2, 5)
  ADD_X3S(64, "128x160", 4, 2, 2, 5)
  ADD_X3S(128, "128x160", 4, 2, 2, 5)
  ADD_X3S(256, "128x160", 4, 2, 2, 5)
  ADD_X3S(512, "128x160", 4, 2, 2, 5)
  ADD_X3S(128, "256x160", 4, 2, 4, 5)
  ADD_X3S(256, "256x160", 4, 2, 4, 5)
  ADD_X3S(512, "256x160", 4, 2, 4, 5)
  ADD_X3("4x2w 2x5b (128x160 8w)", 4, 2, 2, 5)
  ADD_X3("2x2w 2x5b (64x160 4w)", 2, 2, 2, 5)
  ADD_X3("4x2w 2x7b (128x224 8w)", 4, 2, 2, 7)
  ADD_X3("4x2w 4x5b (256x160 8w)", 4, 2, 4, 5)
#define ADD_ABL(tag, ABL, NOEPI)                                                                          \
  cases.push_back({std::string("abl nn N=300 K=300 ") + tag, 2.0 * M * 300 * 300,                           \
                   [=](hipStream_t s) { return run_nn_abl<ABL, NOEPI>(b, M, 300, 300, s); }});              \
  cases.push_back({std::string("abl nn N=300 K=900 ") + tag, 2.0 * M * 300 * 900,                           \
                   [=](hipStream_t s) { return run_nn_abl<ABL, NOEPI>(b, M, 300, 900, s); }});
  ADD_ABL("baseline", 0, false)
  ADD_ABL("prefetch-dist-2", 16, false)
  ADD_ABL("no-epilogue", 0, true)
  ADD_ABL("no-barrier", 1, false)
  ADD_ABL("no-lds-store", 2, false)
  ADD_ABL("no-global-load", 4, false)
  ADD_ABL("no-frag-read", 8, false)
  ADD_ABL("no-store,load", 6, false)
  ADD_ABL("no-barrier,store,load", 7, false)
  ADD_ABL("mfma-only(+epi)", 15, false)
  ADD_ABL("mfma-only no-epi", 15, true)
  if (argc > 1) {  // keep only cases whose name contains argv[1]
    std::vector<Case> keep;
    for (auto& c : cases) if (c.name.find(argv[1]) != std::string::npos) keep.push_back(c);
    cases.swap(keep);
  }
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  std::vector<std::vector<float>> ms(cases.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t i = 0; i < cases.size(); ++i) {
      if (cases[i].name.find("tn_") == 0) CK(hipMemsetAsync(b.c, 0, 912 * 304 * 4, st));
      CK(hipEventRecord(e0, st));
      if (cases[i].fn(st) != 0) { fprintf(stderr, "launch failed: %s\n", cases[i].name.c_str()); return 1; }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[i].push_back(t);
    }
  }
  for (size_t i = 0; i < cases.size(); ++i) {
    std::sort(ms[i].begin(), ms[i].end());
    const float med = ms[i][ms[i].size() / 2], mn = ms[i][0];
    printf("%-52s median %7.3f ms  %6.1f TF   (min %7.3f ms %6.1f TF)\n", cases[i].name.c_str(), med,
           cases[i].flops / med / 1e9, mn, cases[i].flops / mn / 1e9);
  }
  return 0;
}
