// Probe (not product code): split-K sweep of the weight-gradient GEMMs (dW (I x J+1) += dY^T X, K = M = 211200).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/wgrad_probe.hip -o tools/bin/wgrad_probe
#include <stdarg.h>

#include <vector>

#include "nrl_gemm_bf16x3_dma.h"

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

int main() {
  const int64_t M = 211200;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *dy, *x, *dw, *db;
  CK(hipMalloc(&dy, (size_t)M * 900 * 4));
  CK(hipMalloc(&x, (size_t)M * 304 * 4));
  CK(hipMalloc(&dw, (size_t)900 * 304 * 4));
  CK(hipMalloc(&db, 4096));
  CK(hipMemset(dy, 0x3c, (size_t)M * 900 * 4));
  CK(hipMemset(x, 0x3c, (size_t)M * 304 * 4));
  CK(hipMemset(dw, 0, (size_t)900 * 304 * 4));
  CK(hipMemset(db, 0, 4096));
  const int J = 300;
  for (int I : {900, 300, 200}) {
    const RCPlain a{dy, I, I, 0}, b{x, J, J, 1};
    const EpiAtomicWB epi{dw, J, db, J};
    const double gf = 2.0 * M * I * (J + 1) * 1e-9;
    for (int splits : {8, 16, 24, 32, 48, 64, 96, 127, 160, 256}) {
      float t_big = -1, t_w = -1, t_tn = -1, t_tn4 = -1;
      t_big = time_ms([&] { launch_gemm_bf16x3<4, 2, 4, 5, 0>(a, b, epi, I, J + 1, M, splits, st); }, st);
      if (I <= 512) {
        t_w = time_ms([&] { launch_gemm_bf16x3<2, 2, 2, 5, 0>(a, b, epi, I, J + 1, M, splits, st); }, st);
        t_tn = time_ms([&] { launch_gemm_bf16x3_dma_tn<2, 2, 2, 5, 2>(a, b, epi, I, J + 1, M, splits, st); }, st);
      }
      t_tn4 = time_ms([&] { launch_gemm_bf16x3_dma_tn<2, 2, 4, 5, 2>(a, b, epi, I, J + 1, M, splits, st); }, st);
      printf("I=%3d splits=%3d : reg 256x160 %.3f ms (%.0f TF)  reg 64x160 %.3f  dma_tn 64x160 %.3f  dma_tn 128x160 %.3f\n", I,
             splits, t_big, gf / t_big, t_w, t_tn, t_tn4);
      fflush(stdout);
    }
  }
  return 0;
}
