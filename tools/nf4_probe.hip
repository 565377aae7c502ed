// Probe (not product code): the fused news-encoder forward (nrl_news_fused.h) as 8-wave workgroups (one per CU, whole-head weight
// ring) against 4-wave workgroups (two per CU, two-slot ring) at the BASELINE configs[1] shape -- time and bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/nf4_probe.hip -o tools/bin/nf4_probe
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "nrl_news_fused.h"

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
static float time_ms(F f, hipStream_t st, int reps = 30) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

static size_t diff_bytes(const void* d1, const void* d2, size_t n) {
  std::vector<unsigned char> h1(n), h2(n);
  CK(hipMemcpy(h1.data(), d1, n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), d2, n, hipMemcpyDeviceToHost));
  size_t d = 0;
  for (size_t i = 0; i < n; ++i) d += h1[i] != h2[i];
  return d;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 7040;
  const int L = argc > 2 ? atoi(argv[2]) : 30, D = 300, H = 15, V = 70000;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *table, *w, *b;
  int64_t* ids;
  uint16_t* img;
  CK(hipMalloc(&table, (size_t)V * D * 4));
  CK(hipMalloc(&w, (size_t)3 * D * D * 4));
  CK(hipMalloc(&b, (size_t)3 * D * 4));
  CK(hipMalloc(&ids, (size_t)N * L * 8));
  CK(hipMalloc(&img, rp_image_elems(H * 4, NF_KB) * 2));
  {
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    std::vector<float> h((size_t)V * D);
    for (auto& v : h) v = rnd();
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hw((size_t)3 * D * D);
    for (auto& v : hw) v = rnd() * 0.06f;
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hb(3 * D);
    for (auto& v : hb) v = rnd() * 0.1f;
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<int64_t> hi((size_t)N * L);
    for (auto& v : hi) { s = s * 1664525u + 1013904223u; const double u = (s >> 8) / 16777216.0; v = 1 + (int64_t)((V - 1) * u * u * u); }
    CK(hipMemcpy(ids, hi.data(), hi.size() * 8, hipMemcpyHostToDevice));
  }
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  rp_jobs_add_qkv_heads(&jobs, w, D, b, img, H, 20);
  rp_jobs_launch(jobs, st);
  const size_t xp_bytes = (size_t)N * 2 * 20 * 1024, op_bytes = (size_t)((N * L + 31) / 32 * 2) * 19 * 1024;
  const size_t q_bytes = (size_t)N * L * 15 * 64 * 4, l_bytes = (size_t)N * H * L * 4;
  unsigned char *xp[2], *op[2];
  float *qkv[2], *lse[2];
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&xp[i], xp_bytes)); CK(hipMalloc(&op[i], op_bytes)); CK(hipMalloc(&qkv[i], q_bytes)); CK(hipMalloc(&lse[i], l_bytes));
    CK(hipMemset(op[i], 0, op_bytes)); CK(hipMemset(xp[i], 0, xp_bytes)); CK(hipMemset(qkv[i], 0, q_bytes)); CK(hipMemset(lse[i], 0, l_bytes));
  }
  auto args = [&](int i, bool train) {
    NewsFusedArgs a;
    a.table = table; a.ids = ids; a.img = img; a.n_news = N; a.L = L; a.D = D; a.heads = H; a.dh = 20;
    a.scale = 1.0f / sqrtf(20.f); a.drop1 = make_dropout(train ? 0.2 : 0.0, 5, 0); a.o = nullptr; a.o_planes = op[i];
    a.x_save = nullptr; a.x_planes = train ? xp[i] : nullptr; a.qkv_save = train ? qkv[i] : nullptr; a.lse = train ? lse[i] : nullptr;
    a.qkv_head_major = 1;
    return a;
  };
  const double gf = 2.0 * N * L * 3.0 * D * D * 1e-9;
  auto report = [&](const char* name, float ms) { printf("%-52s %.3f ms  (%.0f TF fp32-equiv in-projection)\n", name, ms, gf / ms); fflush(stdout); };
  for (int rep = 0; rep < 3; ++rep) {
    report("eval   8 waves, one workgroup per CU", time_ms([&] { launch_news_fused_fwd<0, true>(args(0, false), st, 8); }, st));
    report("eval   4 waves, two workgroups per CU", time_ms([&] { launch_news_fused_fwd<0, true>(args(1, false), st, 4); }, st));
    report("train  8 waves (x / o planes, slabs, lse)", time_ms([&] { launch_news_fused_fwd<0, true>(args(0, true), st, 8); }, st));
    report("train  4 waves (x / o planes, slabs, lse)", time_ms([&] { launch_news_fused_fwd<0, true>(args(1, true), st, 4); }, st));
  }
  CK(hipStreamSynchronize(st));
  printf("train, 4 vs 8 waves: o planes differ in %zu bytes, x planes %zu, q|k|v slabs %zu, lse %zu\n",
         diff_bytes(op[0], op[1], op_bytes), diff_bytes(xp[0], xp[1], xp_bytes), diff_bytes(qkv[0], qkv[1], q_bytes),
         diff_bytes(lse[0], lse[1], l_bytes));
  launch_news_fused_fwd<0, true>(args(0, false), st, 8);
  launch_news_fused_fwd<0, true>(args(1, false), st, 4);
  CK(hipStreamSynchronize(st));
  printf("eval,  4 vs 8 waves: o planes differ in %zu bytes\n", diff_bytes(op[0], op[1], op_bytes));
  return 0;
}
