// Probe (not product code): fixed vs per-k-block cost of the row-panel GEMM (N = 300, EpiStore / EpiLinear with dropout) over K
// and over whole / fractional dispatch rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/rp_fixed_probe.hip -o tools/bin/rp_fixed_probe
#include <stdarg.h>
#include <vector>
#include "nrl_rowpanel.h"
namespace nrl {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
}
using namespace nrl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F>
static float time_ms(F f, hipStream_t st, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int64_t MMAX = 262144;
  float *a, *w, *c; uint16_t* img;
  CK(hipMalloc(&a, (size_t)MMAX * 960 * 4)); CK(hipMalloc(&w, (size_t)960 * 320 * 4)); CK(hipMalloc(&c, (size_t)MMAX * 320 * 4));
  CK(hipMalloc(&img, (size_t)8 << 20));
  CK(hipMemset(a, 0, (size_t)MMAX * 960 * 4)); CK(hipMemset(w, 0, (size_t)960 * 320 * 4));
  const int N = 300;
  for (int64_t M : {(int64_t)211200, (int64_t)196608, (int64_t)65536, (int64_t)262144}) {
    for (int K : {32, 96, 160, 300, 608, 928}) {
      RpImageJobs jobs; rp_jobs_init(&jobs);
      rp_jobs_add(&jobs, w, K, 1, N, K, nullptr, img, 19);
      rp_jobs_launch(jobs, st);
      RpImage im; im.img = img; im.nblk = 19; im.kblocks = rp_kblocks(K, false);
      const KCPlain A{a, K, M};
      const float t0 = time_ms([&] { launch_rp_gemm<19, 4>(A, im, EpiStore{c, N}, M, N, K, st); }, st);
      const float t1 = time_ms([&] { launch_rp_gemm<19, 4>(A, im, EpiLinear{c, N, nullptr, 0, make_dropout(0.2, 3, 1), N}, M, N, K, st); }, st);
      printf("M=%7lld (%.2f rounds of 512 workgroups)  K=%3d (%2d k-blocks): store %.3f ms   dropout epilogue %.3f ms\n", (long long)M,
             (double)((M + 127) / 128) / 512.0, K, im.kblocks, t0, t1);
      fflush(stdout);
    }
  }
  return 0;
}
