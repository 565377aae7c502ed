// Probe (not product code): weight-gradient GEMMs dW (I x J+1) += dY^T [X | 1], K = M = 211200 rows -- the
// register-staged / LDS-DMA kernels against the wave-specialised one (nrl_gemm_ws.h): agreement and speed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc -Itools tools/wgrad_probe.hip -o tools/bin/wgrad_probe
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "nrl_gemm_bf16x3_dma.h"
#include "nrl_gemm_ws.h"
#include "../tools/experimental/nrl_wgrad_tn2.h"

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

template <class F>
static float time_ms(F f, hipStream_t st, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

static const int64_t M = 211200;
static float *g_dy, *g_x, *g_dw1, *g_dw2, *g_db1, *g_db2;

template <class FOld, class FNew>
static void compare(const char* name, int I, int J, FOld fold, FNew fnew, hipStream_t st) {
  const size_t n = (size_t)I * J;
  CK(hipMemsetAsync(g_dw1, 0, n * 4, st));
  CK(hipMemsetAsync(g_dw2, 0, n * 4, st));
  CK(hipMemsetAsync(g_db1, 0, 4096, st));
  CK(hipMemsetAsync(g_db2, 0, 4096, st));
  fold(g_dw1, g_db1);
  fnew(g_dw2, g_db2);
  CK(hipStreamSynchronize(st));
  std::vector<float> h1(n), h2(n), b1(I), b2(I);
  CK(hipMemcpy(h1.data(), g_dw1, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), g_dw2, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b1.data(), g_db1, I * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b2.data(), g_db2, I * 4, hipMemcpyDeviceToHost));
  double md = 0, mv = 0, mb = 0;
  for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)fabs(h1[i] - h2[i])); mv = std::max(mv, (double)fabs(h1[i])); }
  for (int i = 0; i < I; ++i) mb = std::max(mb, (double)fabs(b1[i] - b2[i]));
  const float t_old = time_ms([&] { fold(g_dw1, g_db1); }, st), t_new = time_ms([&] { fnew(g_dw2, g_db2); }, st);
  const double gf = 2.0 * M * I * (J + 1) * 1e-9;
  printf("%-34s I=%3d : old %.3f ms (%.0f TF fp32-equiv)  ws %.3f ms (%.0f TF)  max|dW diff| %.3e of %.1f, |db diff| %.3e\n", name, I,
         t_old, gf / t_old, t_new, gf / t_new, md, mv, mb);
  fflush(stdout);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  CK(hipMalloc(&g_dy, (size_t)M * 900 * 4));
  CK(hipMalloc(&g_x, (size_t)M * 304 * 4));
  CK(hipMalloc(&g_dw1, (size_t)900 * 304 * 4));
  CK(hipMalloc(&g_dw2, (size_t)900 * 304 * 4));
  CK(hipMalloc(&g_db1, 4096));
  CK(hipMalloc(&g_db2, 4096));
  {
    uint32_t s = 4242;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    std::vector<float> h((size_t)M * 900);
    for (auto& v : h) v = rnd() * 0.05f;
    CK(hipMemcpy(g_dy, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)M * 304);
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(g_x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const int J = 300;
  auto splits_for = [&](int I, int bm, int bn) {
    const int64_t tiles = ceil_div(I, bm) * ceil_div(J + 1, bn);
    int64_t sp = ceil_div(M, 1664);
    if (sp * tiles < 512) sp = ceil_div(512, tiles);
    return (int)sp;
  };
  for (int I : {900, 300, 200}) {
    const RCPlain a{g_dy, I, I, 0}, b{g_x, J, J, 1};
    auto old_big = [&](float* dw, float* db) { launch_gemm_bf16x3<4, 2, 4, 5, 0>(a, b, EpiAtomicWB{dw, J, db, J}, I, J + 1, M, splits_for(I, 256, 160), st); };
    auto old_tn = [&](float* dw, float* db) { launch_gemm_bf16x3_dma_tn<2, 2, 2, 5, 2>(a, b, EpiAtomicWB{dw, J, db, J}, I, J + 1, M, splits_for(I, 64, 160), st); };
    if (I > 512) {
      for (int splits : {32, 64}) {
        char nm[64];
        snprintf(nm, sizeof nm, "ws<4,2,2,8,5> 256x160 splits=%d", splits);
        compare(nm, I, J, old_big, [&](float* dw, float* db) { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, EpiAtomicWB{dw, J, db, J}, I, J + 1, M, splits, st); }, st);
      }
    } else {
      // larger wave tiles for the transposing LDS-DMA kernel: a fragment is split by every wave that reads it, so
      // MFMAs per fragment (TM * TN / (TM + TN)) is what the VALU budget follows
      for (int splits : {32, 43, 64, 128}) {
        char nm[64];
        snprintf(nm, sizeof nm, "tn2<4,5> 128x160 splits=%d", splits);
        compare(nm, I, J, old_tn, [&](float* dw, float* db) { launch_gemm_bf16x3_tn2<4, 5>(a, b, EpiAtomicWB{dw, J, db, J}, I, J + 1, M, splits, st); }, st);
        snprintf(nm, sizeof nm, "tn2<3,5> 96x160 splits=%d", splits);
        compare(nm, I, J, old_tn, [&](float* dw, float* db) { launch_gemm_bf16x3_tn2<3, 5>(a, b, EpiAtomicWB{dw, J, db, J}, I, J + 1, M, splits, st); }, st);
      }
    }
  }
  return 0;
}
