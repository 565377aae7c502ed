// Probe (not product code): the pre-split-operand weight gradient (nrl_wgrad_planes.h) -- agreement with an fp64 host
// reference at a small size, and speed at the NRMS in-projection shape (7040 news x 32 padded tokens, 960 x 301).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/wp_probe.hip -o tools/bin/wp_probe
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "nrl_wgrad_planes.h"

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

static uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  const int heads = 15, dh = 20, D = 300, ncb_b = 20, L = 30;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  {
    // ---- correctness at 37 news ------------------------------------------------------------------------
    const int n_news = 37;
    const int64_t n_mb = 2 * n_news, Mp = 32 * n_news;
    std::vector<float> A((size_t)Mp * heads * 64, 0.f), B((size_t)Mp * 320, 0.f);
    uint32_t s = 4242;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (int64_t m = 0; m < Mp; ++m) {
      const bool live = (m & 31) < L;
      for (int h = 0; h < heads; ++h)
        for (int c = 0; c < 3 * dh; ++c) A[(m * heads + h) * 64 + c] = live ? rnd() : 0.f;
      for (int j = 0; j < D; ++j) B[m * 320 + j] = rnd();
      B[m * 320 + D] = 1.0f;
    }
    std::vector<uint16_t> pa((size_t)heads * n_mb * 4 * 512), pb((size_t)n_mb * ncb_b * 512);
    for (int64_t m = 0; m < Mp; ++m)
      for (int h = 0; h < heads; ++h)
        for (int c = 0; c < 64; ++c) {
          const float x = A[(m * heads + h) * 64 + c];
          const uint16_t hi = bf16_rne(x), lo = bf16_rne(x - bf16_f(hi));
          const size_t blk = (((size_t)h * n_mb + m / 16) * 4 + c / 16) * 512;
          pa[blk + (m % 16) * 16 + c % 16] = hi;
          pa[blk + 256 + (m % 16) * 16 + c % 16] = lo;
        }
    for (int64_t m = 0; m < Mp; ++m)
      for (int j = 0; j < 320; ++j) {
        const float x = B[m * 320 + j];
        const uint16_t hi = bf16_rne(x), lo = bf16_rne(x - bf16_f(hi));
        const size_t blk = ((size_t)(m / 16) * ncb_b + j / 16) * 512;
        pb[blk + (m % 16) * 16 + j % 16] = hi;
        pb[blk + 256 + (m % 16) * 16 + j % 16] = lo;
      }
    uint16_t *da, *db_;
    float *dw, *dbias;
    CK(hipMalloc(&da, pa.size() * 2));
    CK(hipMalloc(&db_, pb.size() * 2));
    CK(hipMalloc(&dw, (size_t)900 * 300 * 4));
    CK(hipMalloc(&dbias, 900 * 4));
    CK(hipMemcpy(da, pa.data(), pa.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db_, pb.data(), pb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dw, 0, (size_t)900 * 300 * 4));
    CK(hipMemset(dbias, 0, 900 * 4));
    const EpiAtomicWBHeads epi{dw, D, dbias, D, heads, dh};
    if (launch_wgrad_planes(da, db_, n_news, heads, ncb_b, D + 1, epi, 5, st) != NRL_OK) return 1;
    CK(hipStreamSynchronize(st));
    std::vector<float> hw((size_t)900 * 300), hb(900);
    CK(hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), dbias, hb.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    size_t bad = 0;
    for (int part = 0; part < 3; ++part)
      for (int h = 0; h < heads; ++h)
        for (int d = 0; d < dh; ++d) {
          const int row = part * 300 + h * dh + d, c = part * dh + d;
          for (int j = 0; j <= D; ++j) {
            double ref = 0, mag = 0;
            for (int64_t m = 0; m < Mp; ++m) {
              const double a = A[(m * heads + h) * 64 + c], b = B[m * 320 + j];
              ref += a * b;
              mag += fabs(a * b);
            }
            const double got = j < D ? hw[(size_t)row * 300 + j] : hb[row];
            const double err = fabs(got - ref) / (mag + 1e-30);
            worst = std::max(worst, err);
            bad += err > 3e-5;
          }
        }
    printf("correctness (37 news, 5 splits): worst |err| / sum|a b| = %.3e, outside 3e-5: %zu\n", worst, bad);
    // round 6: the 8-wave workgroup (two waves per SIMD over the same tile) must produce the same sums
    CK(hipMemset(dw, 0, (size_t)900 * 300 * 4));
    CK(hipMemset(dbias, 0, 900 * 4));
    if (launch_wgrad_planes<0, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 5, st) != NRL_OK) return 1;
    CK(hipStreamSynchronize(st));
    std::vector<float> hw8((size_t)900 * 300), hb8(900);
    CK(hipMemcpy(hw8.data(), dw, hw8.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb8.data(), dbias, hb8.size() * 4, hipMemcpyDeviceToHost));
    double d8 = 0, mx = 0;
    for (size_t i = 0; i < hw.size(); ++i) { d8 = std::max(d8, (double)fabs(hw[i] - hw8[i])); mx = std::max(mx, (double)fabs(hw[i])); }
    for (size_t i = 0; i < hb.size(); ++i) d8 = std::max(d8, (double)fabs(hb[i] - hb8[i]));
    printf("8-wave workgroup vs 4-wave: max |difference| %.3e (|dW| max %.3e; split-K atomics reorder the sums)\n", d8, mx);
  }
  {
    // ---- speed at B = 128 -----------------------------------------------------------------------------
    const int64_t n_news = getenv("WP_NEWS") ? atoll(getenv("WP_NEWS")) : 7040, n_mb = 2 * n_news;
    uint16_t *da, *db_;
    float *dw, *dbias;
    CK(hipMalloc(&da, (size_t)heads * n_mb * 4 * 1024));
    CK(hipMalloc(&db_, (size_t)n_mb * ncb_b * 1024));
    CK(hipMalloc(&dw, (size_t)900 * 300 * 4));
    CK(hipMalloc(&dbias, 900 * 4));
    CK(hipMemset(da, 0x3c, (size_t)heads * n_mb * 4 * 1024));
    CK(hipMemset(db_, 0x3c, (size_t)n_mb * ncb_b * 1024));
    const EpiAtomicWBHeads epi{dw, D, dbias, D, heads, dh};
    auto timeit = [&](const char* name, auto fn) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) fn();
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 20; ++i) fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-40s %.3f ms\n", name, ms / 20);
    };
    timeit("32 splits", [&] { launch_wgrad_planes<0>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, 8 waves (2 per SIMD)", [&] { launch_wgrad_planes<0, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, 8 waves, no DMA in the loop", [&] { launch_wgrad_planes<1, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, 8 waves, no MFMAs", [&] { launch_wgrad_planes<2, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, again", [&] { launch_wgrad_planes<0>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, 8 waves, again", [&] { launch_wgrad_planes<0, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("64 splits, 8 waves", [&] { launch_wgrad_planes<0, 8>(da, db_, n_news, heads, ncb_b, D + 1, epi, 64, st); });
    timeit("32 splits, no DMA in the loop", [&] { launch_wgrad_planes<1>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, no MFMAs", [&] { launch_wgrad_planes<2>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    timeit("32 splits, no DMA, no MFMAs", [&] { launch_wgrad_planes<3>(da, db_, n_news, heads, ncb_b, D + 1, epi, 32, st); });
    for (int nsplit : {32, 64}) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) launch_wgrad_planes(da, db_, n_news, heads, ncb_b, D + 1, epi, nsplit, st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 20; ++i) launch_wgrad_planes(da, db_, n_news, heads, ncb_b, D + 1, epi, nsplit, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 20;
      const double fl = 2.0 * 225280.0 * 1024 * 320 * 3;
      printf("B = 128 shape, %3d splits: %.3f ms  (%.0f TF of issued bf16 MFMA work, %.1f TB/s of plane reads)\n", nsplit, ms,
             fl / ms * 1e-9, ((double)heads * n_mb * 4 * 1024 * 2 + (double)n_mb * ncb_b * 1024 * 4) / ms * 1e-9);
    }
  }
  {
    // ---- generic form (out-projection shape): A (rows x 300) and B (rows x 301 incl. a ones column) as 19 block columns each
    const int ncb = 19, I = 300, J = 300;
    {
      const int64_t rows = 32 * 41, n_mb = rows / 16;
      std::vector<float> A((size_t)rows * 304, 0.f), B((size_t)rows * 304, 0.f);
      uint32_t s = 99;
      auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
      for (int64_t m = 0; m < rows; ++m) {
        for (int c = 0; c < I; ++c) A[m * 304 + c] = rnd();
        for (int c = 0; c < J; ++c) B[m * 304 + c] = rnd();
        B[m * 304 + J] = 1.0f;
      }
      auto planes = [&](const std::vector<float>& X) {
        std::vector<uint16_t> p((size_t)n_mb * ncb * 512);
        for (int64_t m = 0; m < rows; ++m)
          for (int c = 0; c < 304; ++c) {
            const float x = X[m * 304 + c];
            const uint16_t hi = bf16_rne(x), lo = bf16_rne(x - bf16_f(hi));
            const size_t blk = ((size_t)(m / 16) * ncb + c / 16) * 512;
            p[blk + (m % 16) * 16 + c % 16] = hi;
            p[blk + 256 + (m % 16) * 16 + c % 16] = lo;
          }
        return p;
      };
      const auto pa = planes(A), pb = planes(B);
      uint16_t *da, *db_;
      float *dw, *dbias;
      CK(hipMalloc(&da, pa.size() * 2)); CK(hipMalloc(&db_, pb.size() * 2));
      CK(hipMalloc(&dw, (size_t)I * J * 4)); CK(hipMalloc(&dbias, I * 4));
      CK(hipMemcpy(da, pa.data(), pa.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(db_, pb.data(), pb.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemset(dw, 0, (size_t)I * J * 4)); CK(hipMemset(dbias, 0, I * 4));
      if (launch_wgrad_planes_g<5, 5>(da, ncb, db_, ncb, rows, I, J + 1, EpiAtomicWB{dw, J, dbias, J}, 7, st) != NRL_OK) return 1;
      CK(hipStreamSynchronize(st));
      std::vector<float> hw((size_t)I * J), hb(I);
      CK(hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), dbias, hb.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0; size_t bad = 0;
      for (int i = 0; i < I; ++i)
        for (int j = 0; j <= J; ++j) {
          double ref = 0, mag = 0;
          for (int64_t m = 0; m < rows; ++m) { const double a = A[m * 304 + i], b = B[m * 304 + j]; ref += a * b; mag += fabs(a * b); }
          const double got = j < J ? hw[(size_t)i * J + j] : hb[i];
          const double err = fabs(got - ref) / (mag + 1e-30);
          worst = std::max(worst, err); bad += err > 3e-5;
        }
      printf("generic 300 x 301 (1312 rows, 7 splits): worst |err| / sum|a b| = %.3e, outside 3e-5: %zu\n", worst, bad);
      // round 6: the 8-wave workgroups (row blocks 3, 3, 2, 2 / 4, 4, 3, 3 over the four wave rows), with and without the scratch path
      for (int variant = 0; variant < 3; ++variant) {
        CK(hipMemset(dw, 0, (size_t)I * J * 4)); CK(hipMemset(dbias, 0, I * 4));
        float* scratch = nullptr;
        if (variant == 1) CK(hipMalloc(&scratch, wgrad_planes_g_scratch_floats(5, 5, ncb, ncb, 7) * 4));
        if (variant == 2) CK(hipMalloc(&scratch, wgrad_planes_g_scratch_floats(7, 5, ncb, ncb, 7) * 4));
        int rc;
        if (variant < 2) rc = launch_wgrad_planes_g<5, 5, 8>(da, ncb, db_, ncb, rows, I, J + 1, EpiAtomicWB{dw, J, dbias, J}, 7, st, scratch);
        else rc = launch_wgrad_planes_g<7, 5, 8>(da, ncb, db_, ncb, rows, I, J + 1, EpiAtomicWB{dw, J, dbias, J}, 7, st, scratch);
        if (rc != NRL_OK) return 1;
        CK(hipStreamSynchronize(st));
        std::vector<float> hw8((size_t)I * J), hb8(I);
        CK(hipMemcpy(hw8.data(), dw, hw8.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb8.data(), dbias, hb8.size() * 4, hipMemcpyDeviceToHost));
        double d8 = 0, mx = 0;
        for (size_t i = 0; i < hw.size(); ++i) { d8 = std::max(d8, (double)fabs(hw[i] - hw8[i])); mx = std::max(mx, (double)fabs(hw[i])); }
        for (size_t i = 0; i < hb.size(); ++i) d8 = std::max(d8, (double)fabs(hb[i] - hb8[i]));
        printf("generic, 8-wave workgroup <%d, 5>%s vs 4-wave <5, 5>: max |difference| %.3e (|dW| max %.3e)\n", variant < 2 ? 5 : 7,
               variant ? " two-step reduce" : " atomics", d8, mx);
      }
    }
    {
      const int64_t rows = 211200, n_mb = rows / 16;
      uint16_t *da, *db_; float *dw, *dbias;
      CK(hipMalloc(&da, (size_t)n_mb * ncb * 1024)); CK(hipMalloc(&db_, (size_t)n_mb * ncb * 1024));
      CK(hipMalloc(&dw, (size_t)I * J * 4)); CK(hipMalloc(&dbias, I * 4));
      CK(hipMemset(da, 0x3c, (size_t)n_mb * ncb * 1024)); CK(hipMemset(db_, 0x3c, (size_t)n_mb * ncb * 1024));
      for (int nsplit : {32, 64, 128}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto fn = [&] { launch_wgrad_planes_g<5, 5>(da, ncb, db_, ncb, rows, I, J + 1, EpiAtomicWB{dw, J, dbias, J}, nsplit, st); };
        for (int i = 0; i < 5; ++i) fn();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) fn();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("generic 300 x 301 over 211200 rows, 160 x 160 tiles, %3d splits: %.3f ms\n", nsplit, ms / 20);
        auto fn8 = [&] { launch_wgrad_planes_g<5, 5, 8>(da, ncb, db_, ncb, rows, I, J + 1, EpiAtomicWB{dw, J, dbias, J}, nsplit, st); };
        for (int i = 0; i < 5; ++i) fn8();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) fn8();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   ... 8-wave workgroups:                                  %3d splits: %.3f ms\n", nsplit, ms / 20);
        auto fn7 = [&] { launch_wgrad_planes_g<7, 5>(da, ncb, db_, ncb, rows, 200, J + 1, EpiAtomicWB{dw, J, dbias, J}, nsplit, st); };
        auto fn78 = [&] { launch_wgrad_planes_g<7, 5, 8>(da, ncb, db_, ncb, rows, 200, J + 1, EpiAtomicWB{dw, J, dbias, J}, nsplit, st); };
        for (int i = 0; i < 5; ++i) fn7();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) fn7();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        float ms7 = ms / 20;
        for (int i = 0; i < 5; ++i) fn78();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) fn78();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   ... 224 x 160 tiles <7, 5> (13 of 19 A block columns):  %3d splits: %.3f ms, 8-wave %.3f ms\n", nsplit, ms7, ms / 20);
      }
    }
  }
  return 0;
}