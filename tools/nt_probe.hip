// Probe (not product code): correctness against a host fp64 restatement + timing / ablations of the fused news-encoder
// back half (nrl_news_tail.h) at the BASELINE configs[1] shape (7040 news x 30 tokens, D = 300, Q = 200).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/nt_probe.hip -o tools/bin/nt_probe
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "nrl_news_tail.h"

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

template <int WV>
static int run(int64_t N, int L) {
  printf("======== %d waves per workgroup ========\n", WV);
  const int D = 300, H = 15, Q = 200, NCB = 19;
  const int64_t M = N * L, Mp = (M + 31) / 32 * 32;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<float> ho((size_t)M * D), hwo((size_t)D * D), hbo(D), hwa((size_t)Q * D), hba(Q), hqa(Q);
  {
    uint32_t s = 4242;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : ho) v = rnd();
    for (auto& v : hwo) v = rnd() * 0.06f;
    for (auto& v : hbo) v = rnd() * 0.1f;
    for (auto& v : hwa) v = rnd() * 0.08f;
    for (auto& v : hba) v = rnd() * 0.1f;
    for (auto& v : hqa) v = rnd() * 0.5f;
  }
  // `o` planes in the head-permuted slot order of the fused forward (nrl_news_fused.h flush_o)
  std::vector<unsigned char> hpl((size_t)Mp * NCB * 64, 0);
  auto put = [&](int64_t m, int slot, float v) {
    const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_f(hi));
    unsigned char* blk = hpl.data() + ((m >> 4) * NCB + (slot >> 4)) * 1024 + (m & 15) * 32 + (slot & 15) * 2;
    memcpy(blk, &hi, 2);
    memcpy(blk + 512, &lo, 2);
  };
  for (int64_t m = 0; m < M; ++m) {
    for (int h = 0; h < H; ++h)
      for (int d = 0; d < 20; ++d) {
        const int slot = d < 16 ? h * 16 + d : 16 * (H + (h >> 2)) + 4 * (h & 3) + (d - 16);
        put(m, slot, ho[m * D + h * 20 + d]);
      }
    put(m, 300, 1.0f);
  }
  unsigned char *o_pl, *y_pl;
  float *wo, *bo, *wa, *ba, *qa, *out, *t, *w;
  uint16_t *img_o, *img_a;
  CK(hipMalloc(&o_pl, hpl.size()));
  CK(hipMalloc(&y_pl, hpl.size()));
  CK(hipMalloc(&wo, hwo.size() * 4));
  CK(hipMalloc(&bo, hbo.size() * 4));
  CK(hipMalloc(&wa, hwa.size() * 4));
  CK(hipMalloc(&ba, hba.size() * 4));
  CK(hipMalloc(&qa, hqa.size() * 4));
  CK(hipMalloc(&out, (size_t)N * D * 4));
  CK(hipMalloc(&t, (size_t)M * Q * 4));
  CK(hipMalloc(&w, (size_t)M * 4));
  CK(hipMalloc(&img_o, rp_image_elems(NT_FB, NT_KB) * 2));
  CK(hipMalloc(&img_a, rp_image_elems(NT_QB, NT_KS) * 2));
  CK(hipMemcpy(o_pl, hpl.data(), hpl.size(), hipMemcpyHostToDevice));
  CK(hipMemset(y_pl, 0, hpl.size()));
  CK(hipMemcpy(wo, hwo.data(), hwo.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bo, hbo.data(), hbo.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wa, hwa.data(), hwa.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ba, hba.data(), hba.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(qa, hqa.data(), hqa.size() * 4, hipMemcpyHostToDevice));
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  rp_jobs_add_kperm(&jobs, wo, D, 1, D, H, img_o, NT_FB, bo);
  rp_jobs_add_kappa(&jobs, wa, D, 1, Q, D, ba, img_a, NT_QB);
  if (rp_jobs_launch(jobs, st) != NRL_OK) return 1;

  NewsTailArgs a;
  a.o_planes = o_pl; a.img_o = img_o; a.img_a = img_a; a.q_a = qa; a.n_news = N; a.L = L; a.D = D; a.Q = Q;
  a.drop2 = make_dropout(0.2, 5, 1); a.out = out; a.y_planes = y_pl; a.t = t; a.w = w;
  NewsTailArgs ae = a;
  ae.y_planes = nullptr; ae.t = nullptr; ae.w = nullptr; ae.drop2 = make_dropout(0.0, 0, 0);

  // ---- correctness: training launch against the host restatement on a sample of news ----
  if (launch_news_tail_fwd<WV, 0>(a, st) != NRL_OK) return 1;
  CK(hipStreamSynchronize(st));
  std::vector<float> gout((size_t)N * D), gt((size_t)M * Q), gw(M);
  std::vector<unsigned char> gy(hpl.size());
  CK(hipMemcpy(gout.data(), out, gout.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gt.data(), t, gt.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gw.data(), w, gw.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gy.data(), y_pl, gy.size(), hipMemcpyDeviceToHost));
  double e_out = 0, e_t = 0, e_w = 0, e_y = 0, s_out = 0, s_y = 0;
  const int64_t sample[] = {0, 1, 7, 8, N / 2, N - 9, N - 2, N - 1};
  for (int64_t n : sample) {
    if (n < 0 || n >= N) continue;
    std::vector<double> y((size_t)L * D), av(L), tt((size_t)L * Q);
    for (int l = 0; l < L; ++l) {
      const int64_t m = n * L + l;
      for (int f = 0; f < D; ++f) {
        double acc = hbo[f];
        for (int k = 0; k < D; ++k) acc += (double)hwo[(size_t)f * D + k] * ho[m * D + k];
        acc *= a.drop2.thresh == 0u ? 1.0 : (lowbias32((uint32_t)(m * D + f) * 0x9E3779B1u + a.drop2.key) >= a.drop2.thresh ? a.drop2.scale : 0.0f);
        y[(size_t)l * D + f] = acc;
        const unsigned char* blk = gy.data() + ((m >> 4) * NCB + (f >> 4)) * 1024 + (m & 15) * 32 + (f & 15) * 2;
        uint16_t hi, lo;
        memcpy(&hi, blk, 2);
        memcpy(&lo, blk + 512, 2);
        e_y = fmax(e_y, fabs((double)bf16_f(hi) + bf16_f(lo) - acc));
        s_y = fmax(s_y, fabs(acc));
      }
      double aa = 0;
      for (int q = 0; q < Q; ++q) {
        double acc = hba[q];
        for (int f = 0; f < D; ++f) acc += (double)hwa[(size_t)q * D + f] * y[(size_t)l * D + f];
        const double tv = tanh(acc);
        tt[(size_t)l * Q + q] = tv;
        e_t = fmax(e_t, fabs(tv - gt[m * Q + q]));
        aa += tv * hqa[q];
      }
      av[l] = aa;
    }
    double mx = -1e300, sum = 0;
    for (int l = 0; l < L; ++l) mx = fmax(mx, av[l]);
    for (int l = 0; l < L; ++l) sum += exp(av[l] - mx);
    for (int l = 0; l < L; ++l) e_w = fmax(e_w, fabs(exp(av[l] - mx) / sum - gw[n * L + l]));
    for (int f = 0; f < D; ++f) {
      double acc = 0;
      for (int l = 0; l < L; ++l) acc += exp(av[l] - mx) / sum * y[(size_t)l * D + f];
      e_out = fmax(e_out, fabs(acc - gout[n * D + f]));
      s_out = fmax(s_out, fabs(acc));
    }
    // ones column of the y planes
    for (int l = 0; l < L; ++l) {
      const int64_t m = n * L + l;
      const unsigned char* blk = gy.data() + ((m >> 4) * NCB + 18) * 1024 + (m & 15) * 32 + 12 * 2;
      uint16_t hi[4], lo[4];
      memcpy(hi, blk, 8);
      memcpy(lo, blk + 512, 8);
      if (hi[0] != 0x3F80 || hi[1] || hi[2] || hi[3] || lo[0] || lo[1] || lo[2] || lo[3]) { printf("BAD ones column at row %lld\n", (long long)m); break; }
    }
  }
  printf("train vs host fp64: out %.3g (scale %.3g)  y planes %.3g (scale %.3g)  t %.3g  w %.3g\n", e_out, s_out, e_y, s_y, e_t, e_w);
  // eval launch must reproduce the training launch when the dropout is off
  {
    NewsTailArgs at = a;
    at.drop2 = make_dropout(0.0, 0, 0);
    launch_news_tail_fwd<WV, 0>(at, st);
    CK(hipStreamSynchronize(st));
    std::vector<float> g1((size_t)N * D), g2((size_t)N * D);
    CK(hipMemcpy(g1.data(), out, g1.size() * 4, hipMemcpyDeviceToHost));
    launch_news_tail_fwd<WV, 0>(ae, st);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(g2.data(), out, g2.size() * 4, hipMemcpyDeviceToHost));
    double d = 0;
    bool finite = true;
    for (size_t i = 0; i < g1.size(); ++i) { d = fmax(d, fabs((double)g1[i] - g2[i])); finite = finite && isfinite(g2[i]); }
    printf("eval vs train(p=0): max diff %.3g, all finite: %d\n", d, (int)finite);
  }

  const double gf = 2.0 * M * (300.0 * 300 + 200.0 * 300) * 1e-9;
  auto report = [&](const char* name, float ms) { printf("%-44s %.3f ms  (%.0f TF fp32-equiv)\n", name, ms, gf / ms); fflush(stdout); };
  report("eval", time_ms([&] { launch_news_tail_fwd<WV, 0>(ae, st); }, st));
  report("train (y planes, t, w saved)", time_ms([&] { launch_news_tail_fwd<WV, 0>(a, st); }, st));
  report("train no dropout hash", time_ms([&] { launch_news_tail_fwd<WV, 1>(a, st); }, st));
  report("train no stores", time_ms([&] { launch_news_tail_fwd<WV, 2>(a, st); }, st));
  report("eval  no weight DMA", time_ms([&] { launch_news_tail_fwd<WV, 4>(ae, st); }, st));
  report("eval  no phase-1 MFMAs", time_ms([&] { launch_news_tail_fwd<WV, 8>(ae, st); }, st));
  report("eval  no phase-2 MFMAs", time_ms([&] { launch_news_tail_fwd<WV, 16>(ae, st); }, st));
  report("eval  no MFMAs at all", time_ms([&] { launch_news_tail_fwd<WV, 24>(ae, st); }, st));
  report("eval  no MFMAs, no DMA", time_ms([&] { launch_news_tail_fwd<WV, 28>(ae, st); }, st));
  report("eval  no pooling reduction", time_ms([&] { launch_news_tail_fwd<WV, 32>(ae, st); }, st));
  // round 6: the upper bound of a front + tail fusion for THIS kernel -- `o` never read from HBM
  report("eval  o planes cache-resident", time_ms([&] { launch_news_tail_fwd<WV, 128>(ae, st); }, st));
  report("train o planes cache-resident", time_ms([&] { launch_news_tail_fwd<WV, 128>(a, st); }, st));
  report("train o resident, no stores", time_ms([&] { launch_news_tail_fwd<WV, 130>(a, st); }, st));
  report("eval, again", time_ms([&] { launch_news_tail_fwd<WV, 0>(ae, st); }, st));
  report("train, again", time_ms([&] { launch_news_tail_fwd<WV, 0>(a, st); }, st));
  {
    NewsTailArgs a1 = a;
    a1.t = nullptr;
    report("train without t (y planes, w saved)", time_ms([&] { launch_news_tail_fwd<WV, 0>(a1, st); }, st));
    report("train without t, plain y stores", time_ms([&] { launch_news_tail_fwd<WV, 64>(a1, st); }, st));
    report("train without t (y planes, w saved), again", time_ms([&] { launch_news_tail_fwd<WV, 0>(a1, st); }, st));
  }

  // ================= backward of the additive attention (news_tail_bwd_kernel) =================
  std::vector<float> hd((size_t)N * D);
  {
    uint32_t s2 = 99;
    for (auto& v : hd) { s2 = s2 * 1664525u + 1013904223u; v = ((s2 >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
  }
  float *d_out, *dq;
  unsigned char *dpre_pl, *dy_pl;
  uint16_t* img_ad;
  const size_t dpre_bytes = (size_t)Mp * NT_QB * 64;
  CK(hipMalloc(&d_out, hd.size() * 4));
  CK(hipMalloc(&dq, 256 * 4));
  CK(hipMalloc(&dpre_pl, dpre_bytes));
  CK(hipMalloc(&dy_pl, hpl.size()));
  CK(hipMalloc(&img_ad, rp_image_elems(NT_FB, NT_QS) * 2));
  CK(hipMemcpy(d_out, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dq, 0, 256 * 4));
  CK(hipMemset(dpre_pl, 0, dpre_bytes));
  CK(hipMemset(dy_pl, 0, hpl.size()));
  {
    RpImageJobs j2;
    rp_jobs_init(&j2);
    RpImageJob* J = rp_jobs_add_kappa(&j2, wa, 1, D, D, Q, nullptr, img_ad, NT_FB);
    if (J->kblocks != NT_QS) { printf("unexpected k-block count %d\n", J->kblocks); return 1; }
    if (rp_jobs_launch(j2, st) != NRL_OK) return 1;
  }
  launch_news_tail_fwd<WV, 0>(a, st);      // training forward: y planes + w (+ t, unused here)
  NewsTailBwdArgs b;
  b.y_planes = y_pl; b.w = w; b.d_out = d_out; b.img_a = img_a; b.img_ad = img_ad; b.q_a = qa; b.n_news = N; b.L = L;
  b.D = D; b.Q = Q; b.drop2 = a.drop2; b.dpre_planes = dpre_pl; b.dy_planes = dy_pl; b.dq_a = dq;
  const int64_t NS = N < 21 ? N : 21;        // small launch: every news checked, dq_a complete
  NewsTailBwdArgs bs = b;
  bs.n_news = NS;
  if (launch_news_tail_bwd<WV, 0>(bs, st) != NRL_OK) return 1;
  CK(hipStreamSynchronize(st));
  {
    std::vector<unsigned char> gdp(dpre_bytes), gdy(hpl.size());
    std::vector<float> gdq(256);
    CK(hipMemcpy(gdp.data(), dpre_pl, dpre_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gdy.data(), dy_pl, gdy.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(gdq.data(), dq, 256 * 4, hipMemcpyDeviceToHost));
    auto plane_val = [&](const std::vector<unsigned char>& pl, int ncb, int64_t m, int col) {
      const unsigned char* blk = pl.data() + ((m >> 4) * ncb + (col >> 4)) * 1024 + (m & 15) * 32 + (col & 15) * 2;
      uint16_t hi, lo;
      memcpy(&hi, blk, 2);
      memcpy(&lo, blk + 512, 2);
      return (double)bf16_f(hi) + bf16_f(lo);
    };
    std::vector<double> dq_ref(Q, 0.0);
    double e_dp = 0, s_dp = 0, e_dy = 0, s_dy = 0, e_dq = 0, s_dq = 0;
    for (int64_t n = 0; n < NS; ++n) {
      std::vector<double> y((size_t)L * D), tt((size_t)L * Q), av(L), c(L), wv(L), dav(L);
      for (int l = 0; l < L; ++l) {
        const int64_t m = n * L + l;
        for (int f = 0; f < D; ++f) y[(size_t)l * D + f] = plane_val(gy, NCB, m, f);    // the y the kernel read
        double aa = 0, cc = 0;
        for (int q = 0; q < Q; ++q) {
          double acc = hba[q];
          for (int f = 0; f < D; ++f) acc += (double)hwa[(size_t)q * D + f] * y[(size_t)l * D + f];
          tt[(size_t)l * Q + q] = tanh(acc);
          aa += tanh(acc) * hqa[q];
        }
        for (int f = 0; f < D; ++f) cc += (double)hd[n * D + f] * y[(size_t)l * D + f];
        av[l] = aa;
        c[l] = cc;
      }
      double mx = -1e300, sum = 0, dbar = 0;
      for (int l = 0; l < L; ++l) mx = fmax(mx, av[l]);
      for (int l = 0; l < L; ++l) sum += exp(av[l] - mx);
      for (int l = 0; l < L; ++l) { wv[l] = exp(av[l] - mx) / sum; dbar += wv[l] * c[l]; }
      for (int l = 0; l < L; ++l) dav[l] = wv[l] * (c[l] - dbar);
      for (int l = 0; l < L; ++l) {
        const int64_t m = n * L + l;
        std::vector<double> dp(Q);
        for (int q = 0; q < Q; ++q) {
          const double tv = tt[(size_t)l * Q + q];
          dp[q] = dav[l] * hqa[q] * (1.0 - tv * tv);
          dq_ref[q] += dav[l] * tv;
          e_dp = fmax(e_dp, fabs(dp[q] - plane_val(gdp, NT_QB, m, q)));
          s_dp = fmax(s_dp, fabs(dp[q]));
        }
        for (int f = 0; f < D; ++f) {
          double acc = wv[l] * hd[n * D + f];
          for (int q = 0; q < Q; ++q) acc += dp[q] * hwa[(size_t)q * D + f];
          acc *= (lowbias32((uint32_t)(m * D + f) * 0x9E3779B1u + a.drop2.key) >= a.drop2.thresh ? a.drop2.scale : 0.0f);
          e_dy = fmax(e_dy, fabs(acc - plane_val(gdy, NCB, m, f)));
          s_dy = fmax(s_dy, fabs(acc));
        }
      }
    }
    for (int q = 0; q < Q; ++q) { e_dq = fmax(e_dq, fabs(dq_ref[q] - gdq[q])); s_dq = fmax(s_dq, fabs(dq_ref[q])); }
    printf("bwd vs host fp64 (%lld news): d_pre %.3g (scale %.3g)  dy %.3g (scale %.3g)  dq_a %.3g (scale %.3g)\n", (long long)NS, e_dp,
           s_dp, e_dy, s_dy, e_dq, s_dq);
  }
  auto rep2 = [&](const char* name, float ms) { printf("%-44s %.3f ms\n", name, ms); fflush(stdout); };
  rep2("tail bwd", time_ms([&] { launch_news_tail_bwd<WV, 0>(b, st); }, st));
  rep2("tail bwd no dropout hash", time_ms([&] { launch_news_tail_bwd<WV, 1>(b, st); }, st));
  rep2("tail bwd no stores", time_ms([&] { launch_news_tail_bwd<WV, 2>(b, st); }, st));
  rep2("tail bwd no weight DMA", time_ms([&] { launch_news_tail_bwd<WV, 4>(b, st); }, st));
  rep2("tail bwd no phase-A MFMAs", time_ms([&] { launch_news_tail_bwd<WV, 8>(b, st); }, st));
  rep2("tail bwd no phase-C MFMAs", time_ms([&] { launch_news_tail_bwd<WV, 16>(b, st); }, st));
  rep2("tail bwd no MFMAs, no DMA", time_ms([&] { launch_news_tail_bwd<WV, 28>(b, st); }, st));
  rep2("tail bwd, again", time_ms([&] { launch_news_tail_bwd<WV, 0>(b, st); }, st));
  rep2("tail bwd plain stores", time_ms([&] { launch_news_tail_bwd<WV, 64>(b, st); }, st));
  rep2("tail bwd, once more", time_ms([&] { launch_news_tail_bwd<WV, 0>(b, st); }, st));
  return 0;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 7040;
  const int L = argc > 2 ? atoi(argv[2]) : 30;
  if (run<8>(N, L)) return 1;
  return run<4>(N, L);
}
