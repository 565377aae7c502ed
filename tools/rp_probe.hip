// Probe (not product code): row-panel bf16x3 GEMM (nrl_rowpanel.h) vs the LDS-DMA tiled kernel -- agreement and
// speed at the NRMS forward / dgrad shapes (M = 211200).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/rp_probe.hip -o tools/bin/rp_probe
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "nrl_rowpanel.h"

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

static const int64_t M = 211200;

template <class F>
static float time_ms(F f, hipStream_t st, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <int NBLK, int WAVES, int DEEP = 0>
static void run_case(const char* name, int kind, int N, int K, float* a, float* w, float* bias, float* c, float* c2,
                     uint16_t* planes, uint16_t* img, hipStream_t st) {
  // kind 0: dgrad C = A (M, K) * W (K, N) with EpiStore; kind 1: forward C = A W^T (W (N, K)) + bias, dropout
  SplitWeight sw;
  RpImage im;
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  if (kind == 0) {
    split_weight(w, K, N, planes, &sw, st);
    rp_jobs_add(&jobs, w, 1, N, N, K, nullptr, img, NBLK);
  } else {
    split_weight(w, N, K, planes, &sw, st);
    rp_jobs_add(&jobs, w, K, 1, N, K, nullptr, img, NBLK);
  }
  rp_jobs_launch(jobs, st);
  im.img = img; im.nblk = NBLK; im.kblocks = rp_kblocks(K, false);
  const KCSplit B = kind == 0 ? KCSplit{sw.hi_t, sw.lo_t, sw.ld_t, N} : KCSplit{sw.hi, sw.lo, sw.ld, N};
  const KCPlain A{a, K, M};
  const EpiLinear el{c, N, bias, 0, make_dropout(0.2, 3, 1), N};
  const EpiLinear el2{c2, N, bias, 0, make_dropout(0.2, 3, 1), N};
  auto old_k = [&]() {
    if (kind == 0) launch_gemm_bf16x3_dma<2, 2, 4, 5, 2>(A, B, EpiStore{c, N}, M, N, K, st);
    else launch_gemm_bf16x3_dma<2, 2, 4, 5, 2>(A, B, el, M, N, K, st);
  };
  auto new_k = [&]() {
    if (kind == 0) launch_rp_gemm<NBLK, WAVES, DEEP>(A, im, EpiStore{c2, N}, M, N, K, st);
    else launch_rp_gemm<NBLK, WAVES, DEEP>(A, im, el2, M, N, K, st);
  };
  CK(hipMemsetAsync(c, 0, (size_t)M * N * 4, st));
  CK(hipMemsetAsync(c2, 0xFF, (size_t)M * N * 4, st));
  old_k();
  new_k();
  CK(hipStreamSynchronize(st));
  std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
  CK(hipMemcpy(h1.data(), c, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), c2, h2.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0;
  size_t nbad = 0;
  for (size_t i = 0; i < h1.size(); ++i) {
    const double d = fabs((double)h1[i] - (double)h2[i]);
    if (!(d <= 1e-30)) nbad += (d > 1e-4);
    if (d > maxd || d != d) maxd = d;
    maxv = std::max(maxv, (double)fabs(h1[i]));
  }
  const float t_old = time_ms(old_k, st), t_new = time_ms(new_k, st);
  const double gf = 2.0 * M * N * K * 1e-9;
  printf("%-28s N=%3d K=%3d NBLK=%2d WAVES=%d : tiled %.3f ms (%.0f TF fp32-equiv)  row-panel %.3f ms (%.0f TF)  max|diff| %.3e (max|c| %.2f) bad %zu\n",
         name, N, K, NBLK, WAVES, t_old, gf / t_old, t_new, gf / t_new, maxd, maxv, nbad);
  fflush(stdout);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *a, *w, *bias, *c, *c2;
  uint16_t *planes, *img;
  CK(hipMalloc(&a, (size_t)M * 900 * 4));
  CK(hipMalloc(&w, (size_t)900 * 320 * 4));
  CK(hipMalloc(&bias, 4096));
  CK(hipMalloc(&c, (size_t)M * 320 * 4));
  CK(hipMalloc(&c2, (size_t)M * 320 * 4));
  CK(hipMalloc(&planes, (size_t)8 << 20));
  CK(hipMalloc(&img, (size_t)8 << 20));
  {
    std::vector<float> h((size_t)M * 900);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw((size_t)900 * 320);
    for (auto& v : hw) v = rnd() * 0.06f;
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hb(1024);
    for (auto& v : hb) v = rnd() * 0.1f;
    CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  }
  run_case<19, 4>("out-proj fwd", 1, 300, 300, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4, 1>("out-proj fwd  DEEP", 1, 300, 300, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4, 1>("in-proj dgrad DEEP", 0, 300, 900, a, w, bias, c, c2, planes, img, st);
  run_case<13, 4, 1>("add-att fwd   DEEP", 1, 200, 300, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4, 1>("add-att dgrad DEEP", 0, 300, 200, a, w, bias, c, c2, planes, img, st);
  run_case<13, 4>("add-att fwd", 1, 200, 300, a, w, bias, c, c2, planes, img, st);
  run_case<13, 8>("add-att fwd", 1, 200, 300, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4>("in-proj dgrad", 0, 300, 900, a, w, bias, c, c2, planes, img, st);
  run_case<19, 8>("in-proj dgrad", 0, 300, 900, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4>("out-proj dgrad", 0, 300, 300, a, w, bias, c, c2, planes, img, st);
  run_case<19, 4>("add-att dgrad", 0, 300, 200, a, w, bias, c, c2, planes, img, st);
  run_case<20, 4>("N=320 fwd", 1, 320, 320, a, w, bias, c, c2, planes, img, st);
  return 0;
}
