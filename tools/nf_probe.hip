// Probe (not product code): timing + ablations of the fused news-encoder front half (nrl_news_fused.h) at the
// BASELINE configs[1] shape (7040 news x 30 tokens, D = 300, 15 heads, V = 70000).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/nf_probe.hip -o tools/bin/nf_probe
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

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 7040;
  const int L = 30, D = 300, H = 15, V = 70000;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *table, *w, *b, *o, *x, *qkv, *lse;
  int64_t* ids;
  uint16_t* img;
  CK(hipMalloc(&table, (size_t)V * D * 4));
  CK(hipMalloc(&w, (size_t)3 * D * D * 4));
  CK(hipMalloc(&b, (size_t)3 * D * 4));
  CK(hipMalloc(&o, (size_t)N * L * D * 4));
  CK(hipMalloc(&x, (size_t)N * L * D * 4));
  CK(hipMalloc(&qkv, (size_t)N * L * 15 * 64 * 4));
  CK(hipMalloc(&lse, (size_t)N * H * L * 4));
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
  NewsFusedArgs a;
  a.table = table; a.ids = ids; a.img = img; a.n_news = N; a.L = L; a.D = D; a.heads = H; a.dh = 20;
  a.scale = 1.0f / sqrtf(20.f); a.drop1 = make_dropout(0.2, 5, 0); a.o = o;
  NewsFusedArgs as = a;
  as.x_save = x; as.x_planes = nullptr; as.o_planes = nullptr; as.qkv_save = qkv; as.lse = lse; as.qkv_head_major = 1;
  a.x_save = nullptr; a.x_planes = nullptr; a.o_planes = nullptr; a.qkv_save = nullptr; a.lse = nullptr; a.qkv_head_major = 0;
  const double gf = 2.0 * N * L * 3.0 * D * D * 1e-9;
  auto report = [&](const char* name, float ms) { printf("%-46s %.3f ms  (%.0f TF fp32-equiv in-projection)\n", name, ms, gf / ms); fflush(stdout); };
  report("eval  (no saves)", time_ms([&] { launch_news_fused_fwd<0>(a, st); }, st));
  report("train (x, q|k|v, lse saved)", time_ms([&] { launch_news_fused_fwd<0>(as, st); }, st));
  report("train streaming saves", time_ms([&] { launch_news_fused_fwd<16>(as, st); }, st));
  report("eval  streaming o stores", time_ms([&] { launch_news_fused_fwd<32>(a, st); }, st));
  report("train streaming o stores", time_ms([&] { launch_news_fused_fwd<32>(as, st); }, st));
  report("eval  no attention phase", time_ms([&] { launch_news_fused_fwd<1>(a, st); }, st));
  report("eval  no in-projection MFMAs", time_ms([&] { launch_news_fused_fwd<2>(a, st); }, st));
  report("eval  no attention, no MFMAs", time_ms([&] { launch_news_fused_fwd<3>(a, st); }, st));
  report("eval  no weight DMA", time_ms([&] { launch_news_fused_fwd<4>(a, st); }, st));
  report("eval  no stores", time_ms([&] { launch_news_fused_fwd<8>(a, st); }, st));
  report("train no stores in the head loop", time_ms([&] { launch_news_fused_fwd<8>(as, st); }, st));
  report("eval  no attention, no MFMAs, no DMA", time_ms([&] { launch_news_fused_fwd<7>(a, st); }, st));
  report("eval  (no saves), again", time_ms([&] { launch_news_fused_fwd<0>(a, st); }, st));
  report("train (x, q|k|v, lse saved), again", time_ms([&] { launch_news_fused_fwd<0>(as, st); }, st));
  report("train streaming saves, again", time_ms([&] { launch_news_fused_fwd<16>(as, st); }, st));
  // ---- the product's training form: x and o as fragment-block planes, q|k|v as head-major slabs; the head-top wait counted
  // (ABL 64, vmcnt(8): the slab stores stay in flight) against the product's vmcnt(0), and the two outputs compared bit for bit
  {
    unsigned char *xp, *op, *op2;
    float *qkv2, *lse2;
    const size_t xp_bytes = (size_t)N * 2 * 20 * 1024, op_bytes = (size_t)((N * L + 15) / 16) * 19 * 1024;
    CK(hipMalloc(&xp, xp_bytes));
    CK(hipMalloc(&op, op_bytes));
    CK(hipMalloc(&op2, op_bytes));
    CK(hipMalloc(&qkv2, (size_t)N * L * 15 * 64 * 4));
    CK(hipMalloc(&lse2, (size_t)N * H * L * 4));
    CK(hipMemset(op, 0, op_bytes));
    CK(hipMemset(op2, 0, op_bytes));
    NewsFusedArgs ap = as;
    ap.x_save = nullptr; ap.x_planes = xp; ap.o = nullptr; ap.o_planes = op;
    NewsFusedArgs ap2 = ap;
    ap2.o_planes = op2; ap2.qkv_save = qkv2; ap2.lse = lse2;
    for (int rep = 0; rep < 3; ++rep) {
      report("train planes, vmcnt(0) at head tops", time_ms([&] { launch_news_fused_fwd<0>(ap, st); }, st));
      report("train planes, counted head-top wait", time_ms([&] { launch_news_fused_fwd<64>(ap2, st); }, st));
    }
    CK(hipStreamSynchronize(st));
    // where the saves' 0.17-0.19 ms go: the same stores onto cache-resident lines (ABL 128), and no slab stores at all
    {
      NewsFusedArgs ap3 = ap2;
      for (int rep = 0; rep < 3; ++rep) {
        report("train planes, slab stores onto resident lines", time_ms([&] { launch_news_fused_fwd<128>(ap3, st); }, st));
        NewsFusedArgs ap4 = ap2;
        ap4.qkv_save = nullptr;
        report("train planes, no q|k|v save (x, o planes, lse)", time_ms([&] { launch_news_fused_fwd<0>(ap4, st); }, st));
        report("train planes (all saves)", time_ms([&] { launch_news_fused_fwd<0>(ap2, st); }, st));
        report("train planes, slab stores before the attention phase", time_ms([&] { launch_news_fused_fwd<256>(ap2, st); }, st));
        report("train planes, slab stores half / half", time_ms([&] { launch_news_fused_fwd<512>(ap2, st); }, st));
      }
      CK(hipStreamSynchronize(st));
    }
    launch_news_fused_fwd<64>(ap2, st);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned char> h1(op_bytes), h2(op_bytes);
    CK(hipMemcpy(h1.data(), op, op_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), op2, op_bytes, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < op_bytes; ++i) diff += h1[i] != h2[i];
    std::vector<float> q1((size_t)N * L * 15 * 64), q2(q1.size());
    CK(hipMemcpy(q1.data(), qkv, q1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(q2.data(), qkv2, q2.size() * 4, hipMemcpyDeviceToHost));
    size_t qdiff = 0;
    for (size_t i = 0; i < q1.size(); ++i) qdiff += memcmp(&q1[i], &q2[i], 4) != 0;
    printf("counted vs vmcnt(0): o planes differ in %zu bytes, q|k|v slabs in %zu floats\n", diff, qdiff);
    CK(hipFree(xp)); CK(hipFree(op)); CK(hipFree(op2)); CK(hipFree(qkv2)); CK(hipFree(lse2));
  }
  // ---- token-attention backward from the head-major slabs the training forward just wrote ----
  float *d_o, *dqkv;
  CK(hipMalloc(&d_o, (size_t)N * L * D * 4));
  CK(hipMalloc(&dqkv, (size_t)N * 32 * 15 * 64 * 4));      // (planes: 32 padded rows per news)
  CK(hipMemcpy(d_o, o, (size_t)N * L * D * 4, hipMemcpyDeviceToDevice));
  NewsAttnBwdArgs ab;
  ab.qkv_hm = qkv; ab.d_o = d_o; ab.lse = lse; ab.dqkv = dqkv; ab.n_news = N; ab.L = L; ab.D = D; ab.heads = H;
  ab.scale = a.scale; ab.hpw = 1; ab.planes = 0;
  auto rep2 = [&](const char* name, float ms) { printf("%-46s %.3f ms\n", name, ms); fflush(stdout); };
  rep2("attn bwd  2 waves/SIMD", time_ms([&] { launch_news_attn_bwd<2, 0>(ab, st); }, st));
  rep2("attn bwd  3 waves/SIMD", time_ms([&] { launch_news_attn_bwd<3, 0>(ab, st); }, st));
  rep2("attn bwd  plain (write-allocate) stores", time_ms([&] { launch_news_attn_bwd<2, 8>(ab, st); }, st));
  rep2("attn bwd  no arithmetic, plain stores", time_ms([&] { launch_news_attn_bwd<2, 10>(ab, st); }, st));
  {
    NewsAttnBwdArgs abp = ab;
    abp.planes = 1;                                         // the product's form: dqkv as (hi, lo) fragment-block planes
    for (int rep = 0; rep < 3; ++rep) {
      rep2("attn bwd  planes, streaming stores (product)", time_ms([&] { launch_news_attn_bwd<2, 0>(abp, st); }, st));
      rep2("attn bwd  planes, plain stores", time_ms([&] { launch_news_attn_bwd<2, 8>(abp, st); }, st));
    }
  }
  {
    // every operand split once into LDS planes (news_attn_bwd_p_kernel) against per-fragment splits: time and bits
    float* dqkv2;
    CK(hipMalloc(&dqkv2, (size_t)N * 32 * 15 * 64 * 4));
    NewsAttnBwdArgs b1 = ab, b2 = ab;
    b1.planes = 1;
    b2.planes = 1; b2.dqkv = dqkv2;
    for (int rep = 0; rep < 3; ++rep) {
      rep2("attn bwd  fragments built from fp32", time_ms([&] { launch_news_attn_bwd<2, 0>(b1, st); }, st));
      rep2("attn bwd  operands split once into LDS planes", time_ms([&] { launch_news_attn_bwd_p<2, 0>(b2, st); }, st));
    }
    rep2("attn bwd  LDS planes, 3 waves/SIMD", time_ms([&] { launch_news_attn_bwd_p<3, 0>(b2, st); }, st));
    rep2("attn bwd  LDS planes, no dqkv stores", time_ms([&] { launch_news_attn_bwd_p<2, 1>(b2, st); }, st));
    CK(hipStreamSynchronize(st));
    const size_t nb = (size_t)N * 32 * 15 * 64 * 4;
    std::vector<unsigned char> h1(nb), h2(nb);
    CK(hipMemcpy(h1.data(), dqkv, nb, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), dqkv2, nb, hipMemcpyDeviceToHost));
    size_t diff = 0, first = 0;
    for (size_t i = 0; i < nb; ++i) if (h1[i] != h2[i]) { if (!diff) first = i; ++diff; }
    printf("dqkv planes, LDS-plane kernel vs fragment-building kernel: %zu of %zu bytes differ (first at %zu)\n", diff, nb, first);
    CK(hipFree(dqkv2));
  }
  rep2("attn bwd  no dqkv stores", time_ms([&] { launch_news_attn_bwd<2, 1>(ab, st); }, st));
  rep2("attn bwd  no arithmetic", time_ms([&] { launch_news_attn_bwd<2, 2>(ab, st); }, st));
  rep2("attn bwd  no arithmetic, no stores", time_ms([&] { launch_news_attn_bwd<2, 3>(ab, st); }, st));
  rep2("attn bwd  no loads", time_ms([&] { launch_news_attn_bwd<2, 4>(ab, st); }, st));
  rep2("attn bwd  no loads, no stores", time_ms([&] { launch_news_attn_bwd<2, 5>(ab, st); }, st));
  rep2("attn bwd  nothing but LDS traffic", time_ms([&] { launch_news_attn_bwd<2, 7>(ab, st); }, st));
  return 0;
}
