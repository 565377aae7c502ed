// Probe (not product code): the wave-specialised weight-gradient GEMM (nrl_gemm_ws.h) at the PLM body's shapes
// (dW (I x J+1) += dY^T [X | 1], M = 38400 token rows): where a k-tile's time goes -- ablations (no epilogue / no split /
// no MFMAs / no global loads), split counts, the two-step reduction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc -Itools tools/wgrad_ws_probe.hip -o tools/bin/wgrad_ws_probe
#include <stdarg.h>

#include <algorithm>
#include <vector>

#include "nrl_gemm_ws.h"

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
#ifndef WS_CHECK_ABL
#define WS_CHECK_ABL 0
#endif

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

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int64_t M = 38400;
  float *dy, *x, *dw, *db, *scratch;
  const size_t scratch_floats = (size_t)64 << 20;
  CK(hipMalloc(&dy, (size_t)M * 4096 * 4));
  CK(hipMalloc(&x, (size_t)M * 3072 * 4));
  CK(hipMalloc(&dw, (size_t)4096 * 2560 * 4));
  CK(hipMalloc(&db, 65536));
  CK(hipMalloc(&scratch, scratch_floats * 4));
  {
    uint32_t s = 4242;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    std::vector<float> h((size_t)M * 3072);
    for (auto& v : h) v = rnd() * 0.05f;
    CK(hipMemcpy(dy, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  for (auto IJ : {std::pair<int, int>{768, 768}, {3072, 768}}) {
    const int I = IJ.first, J = IJ.second;
    const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
    const EpiAtomicWB epi{dw, J, db, J};
    const double gf = 2.0 * M * I * J * 3 * 1e-9;
    {   // agreement with the register-staged kernel (same arithmetic, other order of the atomic adds)
      float* dw2;
      CK(hipMalloc(&dw2, (size_t)I * J * 4 + 65536));
      float* db2 = dw2 + (size_t)I * J;
      CK(hipMemsetAsync(dw, 0, (size_t)I * J * 4, st));
      CK(hipMemsetAsync(db, 0, 65536, st));
      CK(hipMemsetAsync(dw2, 0, (size_t)I * J * 4 + 65536, st));
      launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, WS_CHECK_ABL>(a, b, epi, I, J + 1, M, 16, st);
      launch_gemm_bf16x3<4, 2, 4, 5, 0>(a, b, EpiAtomicWB{dw2, J, db2, J}, I, J + 1, M, 24, st);
      CK(hipStreamSynchronize(st));
      std::vector<float> h1((size_t)I * J), h2((size_t)I * J), b1(I), b2(I);
      CK(hipMemcpy(h1.data(), dw, h1.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h2.data(), dw2, h2.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b1.data(), db, I * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b2.data(), db2, I * 4, hipMemcpyDeviceToHost));
      double md = 0, mv = 0, mb = 0, mbv = 0;
      for (size_t i = 0; i < h1.size(); ++i) { md = std::max(md, (double)fabs(h1[i] - h2[i])); mv = std::max(mv, (double)fabs(h2[i])); }
      for (int i = 0; i < I; ++i) { mb = std::max(mb, (double)fabs(b1[i] - b2[i])); mbv = std::max(mbv, (double)fabs(b2[i])); }
      printf("I=%4d J=%4d agreement with the register-staged kernel: max|dW diff| %.3e of %.2f, max|db diff| %.3e of %.2f\n", I, J, md, mv, mb, mbv);
      CK(hipFree(dw2));
    }
    auto run = [&](const char* name, auto f) {
      const float t = time_ms(f, st);
      printf("I=%4d J=%4d %-44s %.3f ms (%.0f TF-bf16/s)\n", I, J, name, t, gf / t);
      fflush(stdout);
    };
    for (int sp : {8, 16, 24, 32})  {
      char nm[64];
      snprintf(nm, sizeof nm, "product, atomics, splits=%d", sp);
      run(nm, [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, epi, I, J + 1, M, sp, st); });
    }
    run("split through register pairs as before (512), cost-rule splits", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 512>(a, b, epi, I, J + 1, M, I > 768 ? 8 : 16, st); });
    run("loads and splits as two blocks (256), cost-rule splits", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 256>(a, b, epi, I, J + 1, M, I > 768 ? 8 : 16, st); });
    run("two-step reduction, splits=32", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, epi, I, J + 1, M, 32, st, scratch, scratch_floats); });
    run("no epilogue (1)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 1>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, no split (3)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 3>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, no MFMA (5)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 5>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, no global loads (9)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 9>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, no loads, no split (11)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 11>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, no loads, no MFMA (13)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 13>(a, b, epi, I, J + 1, M, 32, st); });
    run("(11) + MFMA waves read one buffer (27)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 27>(a, b, epi, I, J + 1, M, 32, st); });
    run("(11) + MFMA waves read nothing in the loop (43)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 43>(a, b, epi, I, J + 1, M, 32, st); });
    run("no epilogue, MFMA waves read nothing (33)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 33>(a, b, epi, I, J + 1, M, 32, st); });
  }
  // the three loader forms, round-robin (clocks and boxes move single measurements by +-10 %): median of 7 passes of 10 launches
  for (auto IJ : {std::pair<int, int>{768, 768}, {3072, 768}}) {
    const int I = IJ.first, J = IJ.second;
    const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
    const EpiAtomicWB epi{dw, J, db, J};
    const int sp = I > 768 ? 8 : 16;
    std::vector<float> t0, t1, t2;
    for (int pass = 0; pass < 7; ++pass) {
      t0.push_back(time_ms([&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 0>(a, b, epi, I, J + 1, M, sp, st); }, st));
      t1.push_back(time_ms([&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 512>(a, b, epi, I, J + 1, M, sp, st); }, st));
      t2.push_back(time_ms([&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 256>(a, b, epi, I, J + 1, M, sp, st); }, st));
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mn = [](std::vector<float> v) { return *std::min_element(v.begin(), v.end()); };
    printf("I=%4d J=%4d round-robin x7, %d splits: product %.3f (min %.3f) | split through register pairs (512) %.3f (min %.3f) | loads and splits as two blocks (256) %.3f (min %.3f) ms\n",
           I, J, sp, med(t0), mn(t0), med(t1), mn(t1), med(t2), mn(t2));
  }
  // a k extent that is not whole k-tiles (13200 rows = 440 news x 30 tokens) against the whole-tile form one tile shorter
  for (auto IJ : {std::pair<int, int>{768, 768}, {3072, 768}}) {
    const int I = IJ.first, J = IJ.second;
    const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
    const EpiAtomicWB epi{dw, J, db, J};
    for (int64_t Mk : {(int64_t)13184, (int64_t)13200}) {
      const float t = time_ms([&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, epi, I, J + 1, Mk, 8, st); }, st);
      printf("I=%4d J=%4d K=%lld rows (K %% 32 = %d), 8 splits: %.3f ms\n", I, J, (long long)Mk, (int)(Mk % 32), t);
    }
  }
  // per-k-tile pace of ONE workgroup per CU over 1200 k-tiles (splits = 1): 1, 16 and 256 workgroups
  for (auto IJ : {std::pair<int, int>{256, 159}, {4096, 159}, {4096, 2559}}) {
    const int I = IJ.first, J = IJ.second;
    const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
    const EpiAtomicWB epi{dw, J, db, J};
    const int blocks = (I / 256) * ((J + 1) / 160);
    auto run = [&](const char* name, auto f) {
      const float t = time_ms(f, st, 5);
      printf("%4d workgroups x 1200 k-tiles  %-52s %.3f ms = %.2f us per k-tile (120 MFMAs per SIMD: %.1f ns each)\n", blocks, name, t,
             t * 1e3 / 1200, t * 1e6 / 1200 / 120);
      fflush(stdout);
    };
    run("product", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5>(a, b, epi, I, J + 1, M, 1, st); });
    run("no MFMAs, loaders re-read k-tile 0 (133)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 133>(a, b, epi, I, J + 1, M, 1, st); });
    run("product but loaders re-read k-tile 0 (129)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 129>(a, b, epi, I, J + 1, M, 1, st); });
    run("no global loads (9)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 9>(a, b, epi, I, J + 1, M, 1, st); });
    run("no loads, no split (11)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 11>(a, b, epi, I, J + 1, M, 1, st); });
    run("no loads, no split, no LDS reads in the loop (43)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 43>(a, b, epi, I, J + 1, M, 1, st); });
    run("(43) without barriers (107)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 107>(a, b, epi, I, J + 1, M, 1, st); });
    run("no MFMAs (5)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 5>(a, b, epi, I, J + 1, M, 1, st); });
    run("no loads, no MFMAs (13)", [&] { launch_gemm_bf16x3_ws<4, 2, 2, 8, 5, 13>(a, b, epi, I, J + 1, M, 1, st); });
  }
  return 0;
}
