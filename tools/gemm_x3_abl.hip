// Ablation probe of the bf16x3 GEMM k-loop (not product code): which phase of an iteration costs the
// time?  Same shapes as the NRMS dgrads (M = 211200, N = 300, K = 300 / 900), tile 256x160 and 128x160.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/gemm_x3_abl.hip -o gpurun_out/gemm_x3_abl
#include <stdarg.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "experimental/nrl_gemm_bf16x3_abl.h"

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

struct EpiNull {
  struct Row {};
  __device__ __forceinline__ Row row(int64_t) const { return Row{}; }
  __device__ __forceinline__ void operator()(const Row&, int64_t, int, float v) const { asm volatile("" ::"v"(v)); }
};

static float *g_a, *g_w, *g_c;
static uint16_t* g_planes;

template <int WM, int WN, int TM, int TN, int DEEP, int ABL, bool NOEPI>
int run(int64_t M, int N, int K, hipStream_t st) {
  SplitWeight sw;
  if (split_weight(g_w, K, N, g_planes, &sw, st) != 0) return -1;
  KCSplit B{sw.hi_t, sw.lo_t, sw.Np, N};
  if (NOEPI) return launch_gemm_bf16x3_abl<WM, WN, TM, TN, DEEP, 1, ABL>(KCPlain{g_a, K, M}, B, EpiNull{}, M, N, K, 1, st);
  return launch_gemm_bf16x3_abl<WM, WN, TM, TN, DEEP, 1, ABL>(KCPlain{g_a, K, M}, B, EpiStore{g_c, N}, M, N, K, 1, st);
}

struct Case {
  std::string name;
  double flops;
  std::function<int(hipStream_t)> fn;
};

int main(int argc, char** argv) {
  const int64_t M = 211200;
  CK(hipMalloc(&g_a, M * 2400 * 4));
  CK(hipMalloc(&g_w, 2400 * 960 * 4));
  CK(hipMalloc(&g_c, M * 960 * 4));
  CK(hipMalloc(&g_planes, split_weight_elems(2400, 960) * 2 + 1024));
  CK(hipMemset(g_a, 0, M * 2400 * 4));
  CK(hipMemset(g_w, 0, 2400 * 960 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<Case> cases;
#define ADD(tag, WM, WN, TM, TN, DEEP, ABL, NOEPI)                                                         \
  for (int K : {300, 900, 2400})                                                                            \
    cases.push_back({std::string(tag) + " K=" + std::to_string(K), 2.0 * M * 300 * K,                        \
                     [=](hipStream_t s) { return run<WM, WN, TM, TN, DEEP, ABL, NOEPI>(M, 300, K, s); }});
#define SUITE(t, WM, WN, TM, TN, DEEP)                          \
  ADD(t " baseline          ", WM, WN, TM, TN, DEEP, 0, false)  \
  ADD(t " no-epilogue       ", WM, WN, TM, TN, DEEP, 0, true)   \
  ADD(t " no-barrier        ", WM, WN, TM, TN, DEEP, 1, true)   \
  ADD(t " no-lds-store/split", WM, WN, TM, TN, DEEP, 2, true)   \
  ADD(t " no-global-load    ", WM, WN, TM, TN, DEEP, 4, true)   \
  ADD(t " no-frag-read      ", WM, WN, TM, TN, DEEP, 8, true)   \
  ADD(t " no-mfma           ", WM, WN, TM, TN, DEEP, 16, true)  \
  ADD(t " mfma+frag only    ", WM, WN, TM, TN, DEEP, 7, true)   \
  ADD(t " mfma only         ", WM, WN, TM, TN, DEEP, 15, true)  \
  ADD(t " staging only      ", WM, WN, TM, TN, DEEP, 24, true)
  SUITE("256x160 shallow", 4, 2, 4, 5, 0)
  SUITE("128x160 deep   ", 4, 2, 2, 5, 1)
  if (argc > 1) {
    std::vector<Case> keep;
    for (auto& c : cases)
      if (c.name.find(argv[1]) != std::string::npos) keep.push_back(c);
    cases.swap(keep);
  }
  const int rounds = 5;
  std::vector<std::vector<float>> ms(cases.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r)
    for (size_t i = 0; i < cases.size(); ++i) {
      CK(hipEventRecord(e0, st));
      if (cases[i].fn(st) != 0) return 1;
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[i].push_back(t);
    }
  for (size_t i = 0; i < cases.size(); ++i) {
    std::sort(ms[i].begin(), ms[i].end());
    const float med = ms[i][ms[i].size() / 2];
    printf("%-44s median %7.3f ms  %6.1f TF\n", cases[i].name.c_str(), med, cases[i].flops / med / 1e9);
  }
  return 0;
}
