// Probe (not product code): why does a deeper weight ring not speed up the few-row row-panel GEMM?  A ring kernel with
// LOOK chunks of look-ahead, inline-asm row loads and either a counted wait (MODE 0) or vmcnt(0) (MODE 1), against
// rp_gemm_kernel<.., RB = 1> at M = 6400 (user encoder), K = 900 (29 k-blocks: per-k-block cost = slope).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/rp_ring_probe.hip -o tools/bin/rp_ring_probe
#include <stdarg.h>
#include <string.h>

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

__device__ __forceinline__ f32x4 asm_load16(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// MODE: 0 counted wait, 1 vmcnt(0); DMAON: 0 = no weight DMA at all (stale LDS: timing only); ROWS: 0 = no row loads
template <int NBLK, int WAVES, class Epi, int LOOK, int MODE, int DMAON, int SLOTS>
__global__ void __launch_bounds__(WAVES * 64, 1)
    ring_kernel(const KCPlain A, const uint16_t* __restrict__ img, const Epi epi, const int64_t M, const int N, const int K,
                const int kblocks) {
  constexpr int RB = 1;
  constexpr int CHUNK = NBLK * 2048, PIECES = 2 * NBLK, G = (PIECES + WAVES - 1) / WAVES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SLOTS * CHUNK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * (WAVES * 16);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto issue = [&](int kb) {
    if constexpr (!DMAON) return;
    const int kc = kb < kblocks ? kb : kblocks - 1;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(img) + (size_t)kc * CHUNK;
    const uint32_t dst = smem_base + (uint32_t)(kb % SLOTS) * CHUNK;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      int piece = wave + c * WAVES;
      piece = piece < PIECES ? piece : PIECES - 1;
      glds16_saddr(src + piece * 1024, lane_off, dst + (uint32_t)piece * 1024u);
    }
  };
  const int64_t rowi = m0 + wave * 16 + l15;
  const KCPlain::State st = A.init(rowi);
  auto load_rows = [&](int kb, f32x4 (&r)[2]) {
    const int kc = kb < kblocks ? kb : kblocks - 1;
    const int k = kc * 32 + 8 * g;
    r[0] = asm_load16(A.src(st, k, K));
    r[1] = asm_load16(A.src(st, k + 4, K));
  };
  auto wait_rows = [&](f32x4 (&r)[2]) {
    if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"((DMAON ? (LOOK - 1) * G : 0) + 2) : "memory");
  };
  f32x4 acc[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NPAIR = (NBLK + 1) / 2;
  auto mfma_chunk = [&](int slot, const bf16x8& ah, const bf16x8& al) {
    const unsigned char* base = smem + slot * CHUNK + lane * 16;
    auto rd = [&](int p, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * p + jj < NBLK ? 2 * p + jj : NBLK - 1;
        bh[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048);
        bl[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048 + 1024);
      }
    };
    auto mm = [&](int p, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          if (2 * p + jj < NBLK)
            acc[2 * p + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al : ah, pass == 0 ? bl[jj] : bh[jj], acc[2 * p + jj], 0, 0, 0);
    };
    bf16x8 bh0[2], bl0[2], bh1[2], bl1[2];
    rd(0, bh0, bl0);
#pragma unroll
    for (int p = 0; p < NPAIR; p += 2) {
      if (p + 1 < NPAIR) rd(p + 1, bh1, bl1);
      mm(p, bh0, bl0);
      if (p + 1 < NPAIR) {
        if (p + 2 < NPAIR) rd(p + 2, bh0, bl0);
        mm(p + 1, bh1, bl1);
      }
    }
  };
  // steady state of iteration j: [rows j + 1] [DMA j + LOOK]; rows one k-block ahead (two named sets)
  f32x4 ra[2], rb[2];
#pragma unroll
  for (int c = 0; c < LOOK - 1; ++c) issue(c);
  load_rows(0, ra);
  issue(LOOK - 1);
  auto step = [&](int kb, f32x4 (&cur)[2], f32x4 (&nxt)[2]) {
    wait_rows(cur);                    // younger than rows kb: DMA(kb + 1) .. DMA(kb + LOOK - 1) -> (LOOK - 1) G ... + nothing else
    __builtin_amdgcn_s_barrier();
    bf16x8 ah, al;
    {
      const int k = kb * 32 + 8 * g;
      float4 v0 = make_float4(cur[0][0], cur[0][1], cur[0][2], cur[0][3]), v1 = make_float4(cur[1][0], cur[1][1], cur[1][2], cur[1][3]);
      A.finish(v0, st, rowi, k, K, true);
      A.finish(v1, st, rowi, k + 4, K, true);
      rp_split8(v0, v1, ah, al);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_rows(kb + 1, nxt);
    issue(kb + LOOK);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(kb % SLOTS, ah, al);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kb = 0; kb < kblocks; kb += 2) {
    step(kb, ra, rb);
    if (kb + 1 < kblocks) step(kb + 1, rb, ra);
  }
  wait_vmcnt<0>();
  f32x4 acc2[1][NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) acc2[0][j] = acc[j];
  store_accumulators<1, NBLK>(epi, acc2, m0, 0, wave, 0, l15, g, M, N);
}
}  // namespace nrl
using namespace nrl;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F>
static float time_us(F f, hipStream_t st, int reps = 50) {
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
  return ms / reps * 1e3f;
}

template <int LOOK, int MODE, int DMAON, int SLOTS>
static float ring(const KCPlain& A, const uint16_t* img, float* c, int64_t M, int N, int K, int kblocks, hipStream_t st) {
  return time_us([&] {
    hipLaunchKernelGGL((ring_kernel<19, 4, EpiStore, LOOK, MODE, DMAON, SLOTS>), dim3((unsigned)ceil_div(M, 64)), dim3(256), 0, st, A, img,
                       EpiStore{c, N}, M, N, K, kblocks);
  }, st);
}

int main() {
  const int64_t M = 6400;
  const int N = 300;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *a, *w, *c, *c2;
  uint16_t* img;
  CK(hipMalloc(&a, (size_t)M * 900 * 4));
  CK(hipMalloc(&w, (size_t)900 * 320 * 4));
  CK(hipMalloc(&c, (size_t)M * 320 * 4));
  CK(hipMalloc(&c2, (size_t)M * 320 * 4));
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
  }
  for (int K : {300, 900}) {
    RpImage im;
    RpImageJobs jobs;
    rp_jobs_init(&jobs);
    rp_jobs_add(&jobs, w, 1, N, N, K, nullptr, img, 19);
    rp_jobs_launch(jobs, st);
    im.img = img; im.nblk = 19; im.kblocks = rp_kblocks(K, false);
    const KCPlain A{a, K, M};
    const float t_ref = time_us([&] { launch_rp_gemm<19, 4, 0, 1>(A, im, EpiStore{c, N}, M, N, K, st); }, st);
    // correctness of the counted-wait ring (bit-identical expected)
    CK(hipMemsetAsync(c2, 0xFF, (size_t)M * N * 4, st));
    hipLaunchKernelGGL((ring_kernel<19, 4, EpiStore, 3, 0, 1, 4>), dim3((unsigned)ceil_div(M, 64)), dim3(256), 0, st, A, img, EpiStore{c2, N},
                       M, N, K, im.kblocks);
    CK(hipStreamSynchronize(st));
    std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
    CK(hipMemcpy(h1.data(), c, h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), c2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h1.size(); ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("K=%d (%d k-blocks)  rp_gemm_kernel RB=1: %.1f us;  ring LOOK=3 counted differs in %zu elements\n", K, im.kblocks, t_ref, bad);
    printf("   ring LOOK=1 vmcnt(0) 4 slots : %.1f us\n", ring<1, 1, 1, 4>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring LOOK=1 vmcnt(0) 2 slots : %.1f us\n", ring<1, 1, 1, 2>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring LOOK=2 counted  3 slots : %.1f us\n", ring<2, 0, 1, 3>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring LOOK=3 counted  4 slots : %.1f us\n", ring<3, 0, 1, 4>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring LOOK=3 vmcnt(0) 4 slots : %.1f us\n", ring<3, 1, 1, 4>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring no weight DMA   2 slots : %.1f us\n", ring<1, 0, 0, 2>(A, img, c2, M, N, K, im.kblocks, st));
    printf("   ring no weight DMA   4 slots : %.1f us\n", ring<1, 0, 0, 4>(A, img, c2, M, N, K, im.kblocks, st));
    fflush(stdout);
  }
  return 0;
}
