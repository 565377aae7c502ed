// Probe (not product code): where the k-block time of the row-panel GEMM (nrl_rowpanel.h) goes.  A COPY of rp_gemm_kernel's k-loop
// (the product form: two-slot chunk ring, A rows one k-block ahead) with ablation switches, on the shapes of config 4
// (38,400 x 768 x 768 in three 256-column panels, NBLK = 16) and of the NRMS out-projection dgrad (211,200 x 304 x 320, NBLK = 19).
//   ABL: 1 = no chunk DMA inside the loop (the slots keep the first chunk), 2 = no global loads of A inside the loop,
//        4 = no MFMAs, 8 = no barrier, 16 = no fragment reads of B inside the loop (registers keep one pair), 32 = no split of A
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Inewsreclib_amd/csrc tools/rp_abl_probe.hip -o tools/bin/rp_abl_probe
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

template <int NBLK, int WAVES, int RB, int ABL, int OCC = 2>
__global__ void __launch_bounds__(WAVES * 64, OCC)
    rp_probe_kernel(const KCPlain A, const uint16_t* __restrict__ img_base, const EpiStore epi, const int64_t M, const int N, const int K,
                    const int kblocks, const int64_t panel_elems, const int panels, const int panel_group, const int64_t row_blocks) {
  int64_t rb_idx = blockIdx.x;
  int panel = (int)blockIdx.y;
  if (panel_group > 0) {
    const int64_t id = blockIdx.x;
    const int xcd = (int)(id % 8);
    const int64_t local = id / 8;
    const int64_t rx = (row_blocks + 7) / 8;
    const int64_t per_group = rx * panel_group;
    const int64_t grp = local / per_group, rem = local % per_group;
    rb_idx = (rem / panel_group) * 8 + xcd;
    panel = (int)(grp * panel_group + rem % panel_group);
    if (rb_idx >= row_blocks || panel >= panels) return;
  }
  const uint16_t* __restrict__ img = img_base + (int64_t)panel * panel_elems;
  const int n_panel0 = panel * (NBLK * 16);
  constexpr int CHUNK = NBLK * 2048;
  constexpr int PIECES = 2 * NBLK;
  constexpr int G = (PIECES + WAVES - 1) / WAVES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CHUNK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t m0 = rb_idx * (WAVES * 16 * RB);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto issue = [&](int kb, int slot) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(img) + (size_t)kb * CHUNK + lane * 16;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      int piece = wave + c * WAVES;
      piece = piece < PIECES ? piece : PIECES - 1;
      glds16_asm(src + piece * 1024, smem_base + (uint32_t)slot * CHUNK + (uint32_t)piece * 1024u);
    }
  };
  typename KCPlain::State st[RB];
  int64_t rowi[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    rowi[i] = m0 + wave * (16 * RB) + i * 16 + l15;
    st[i] = A.init(rowi[i]);
  }
  auto load_raw = [&](int kb, float4 (&r)[RB][2]) {
    const int k = kb * 32 + 8 * g;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      r[i][0] = A.load(st[i], k, K);
      r[i][1] = A.load(st[i], k + 4, K);
    }
  };
  auto convert = [&](int kb, float4 (&r)[RB][2], bf16x8 (&ah)[RB], bf16x8 (&al)[RB]) {
    const int k = kb * 32 + 8 * g;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      A.finish(r[i][0], st[i], rowi[i], k, K, true);
      A.finish(r[i][1], st[i], rowi[i], k + 4, K, true);
      if constexpr (ABL & 32) {
        ah[i] = __builtin_bit_cast(bf16x8, r[i][0]);
        al[i] = __builtin_bit_cast(bf16x8, r[i][1]);
      } else {
        rp_split8(r[i][0], r[i][1], ah[i], al[i]);
      }
    }
  };
  f32x4 acc[RB][NBLK];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NBLK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NPAIR = (NBLK + 1) / 2;
  auto read_pair = [&](const unsigned char* base, int p, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * p + jj < NBLK ? 2 * p + jj : NBLK - 1;
      bh[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048);
      bl[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048 + 1024);
    }
  };
  auto mfma_pair = [&](int p, const bf16x8 (&ah)[RB], const bf16x8 (&al)[RB], const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (2 * p + jj < NBLK)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            if constexpr (ABL & 4) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[jj]), "v"(bl[jj]));
            else acc[i][2 * p + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i] : ah[i], pass == 0 ? bl[jj] : bh[jj], acc[i][2 * p + jj], 0, 0, 0);
          }
  };
  bf16x8 kbh[2], kbl[2];      // (ABL & 16: the pair every MFMA of the loop uses)
  auto mfma_chunk = [&](int slot, const bf16x8 (&ah)[RB], const bf16x8 (&al)[RB]) {
    const unsigned char* base = smem + slot * CHUNK + lane * 16;
    if constexpr (ABL & 16) {
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) mfma_pair(p, ah, al, kbh, kbl);
      return;
    }
    bf16x8 bh0[2], bl0[2], bh1[2], bl1[2];
    read_pair(base, 0, bh0, bl0);
#pragma unroll
    for (int p = 0; p < NPAIR; p += 2) {
      if (p + 1 < NPAIR) read_pair(base, p + 1, bh1, bl1);
      mfma_pair(p, ah, al, bh0, bl0);
      if (p + 1 < NPAIR) {
        if (p + 2 < NPAIR) read_pair(base, p + 2, bh0, bl0);
        mfma_pair(p + 1, ah, al, bh1, bl1);
      }
    }
    if constexpr (!(ABL & 4)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        if (p + 1 < NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        if (2 * p + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x008, 6 * RB, 0);
        else __builtin_amdgcn_sched_group_barrier(0x008, 3 * RB, 0);
      }
    }
  };
  bf16x8 ah[RB], al[RB];
  {
    float4 r0[RB][2];
    issue(0, 0);
    if constexpr (ABL & 1) issue(kblocks > 1 ? 1 : 0, 1);
    load_raw(0, r0);
    convert(0, r0, ah, al);
  }
  if constexpr (ABL & 16) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_pair(smem + lane * 16, 0, kbh, kbl);
  }
  float4 rkeep[RB][2];
  if constexpr (ABL & 2) load_raw(0, rkeep);
  for (int kb = 0; kb < kblocks; ++kb) {
    wait_vmcnt<0>();
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    const int kn = kb + 1 < kblocks ? kb + 1 : kb;
    float4 r[RB][2];
    if constexpr (!(ABL & 1)) issue(kn, (kb + 1) & 1);
    if constexpr (ABL & 2) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        r[i][0] = rkeep[i][0]; r[i][1] = rkeep[i][1];
        asm volatile("" : "+v"(r[i][0].x), "+v"(r[i][1].x));
      }
    } else {
      load_raw(kn, r);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(kb & 1, ah, al);
    __builtin_amdgcn_sched_barrier(0);
    convert(kn, r, ah, al);
  }
  store_accumulators<RB, NBLK>(epi, acc, m0, n_panel0, wave, 0, l15, g, M, N);
}

template <int NBLK, int ABL, int OCC = 2>
static float run(const float* a, const uint16_t* img, float* c, int64_t M, int N, int K, int panels, hipStream_t st) {
  const int kb = rp_kblocks(K, false);
  const int64_t blocks = (M + 127) / 128;
  int group = panels > 1 ? (panels % 3 == 0 ? 3 : (panels % 4 == 0 ? 4 : (panels % 2 == 0 ? 2 : 0))) : 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, st));
    if (group > 0) {
      const int64_t rx = (blocks + 7) / 8, groups = (panels + group - 1) / group;
      hipLaunchKernelGGL((rp_probe_kernel<NBLK, 4, 2, ABL, OCC>), dim3((unsigned)(8 * rx * group * groups)), dim3(256), 0, st, KCPlain{a, K, M}, img,
                         EpiStore{c, N}, M, N, K, kb, (int64_t)rp_image_elems(NBLK, kb), panels, group, blocks);
    } else {
      hipLaunchKernelGGL((rp_probe_kernel<NBLK, 4, 2, ABL, OCC>), dim3((unsigned)blocks, (unsigned)panels), dim3(256), 0, st, KCPlain{a, K, M}, img,
                         EpiStore{c, N}, M, N, K, kb, (int64_t)rp_image_elems(NBLK, kb), panels, 0, blocks);
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (rep > 0 && t < best) best = t;
  }
  return best;
}

template <int NBLK>
static void shape(const char* name, int64_t M, int N, int K, hipStream_t st) {
  const int panels = (N + 16 * NBLK - 1) / (16 * NBLK), kb = rp_kblocks(K, false);
  std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
  uint32_t s = 99;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : ha) v = rnd();
  for (auto& v : hw) v = rnd() * 0.05f;
  float *a, *w, *c;
  uint16_t* img;
  CK(hipMalloc(&a, ha.size() * 4));
  CK(hipMalloc(&w, hw.size() * 4));
  CK(hipMalloc(&c, (size_t)M * N * 4));
  CK(hipMalloc(&img, (size_t)panels * rp_image_elems(NBLK, kb) * 2));
  CK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  const int pw = 16 * NBLK;
  for (int p = 0; p < panels; ++p)
    rp_jobs_add(&jobs, w + (int64_t)p * pw * K, K, 1, N - p * pw < pw ? N - p * pw : pw, K, nullptr, img + (size_t)p * rp_image_elems(NBLK, kb), NBLK);
  if (rp_jobs_launch(jobs, st) != NRL_OK) exit(1);
  CK(hipStreamSynchronize(st));
  const double gf = 2.0 * M * (double)(panels * pw) * (kb * 32) * 3 / 1e9;      // issued bf16 work (padded)
  const double wgs = (double)((M + 127) / 128) * panels;
  printf("%s: M=%lld N=%d K=%d  NBLK=%d panels=%d  %d k-blocks, %.2f rounds of 512 workgroups, MFMA floor %.3f ms at 2.5 PF\n", name, (long long)M, N, K,
         NBLK, panels, kb, wgs / 512.0, gf / 2.5e6);
#define RUN(abl, what)                                                                                          \
  {                                                                                                             \
    const float t = run<NBLK, abl>(a, img, c, M, N, K, panels, st);                                             \
    printf("  ABL %3d  %-64s %.3f ms  (%4.0f TF-bf16/s)\n", abl, what, t, gf / t);                             \
    fflush(stdout);                                                                                             \
  }
  RUN(0, "product loop")
  RUN(1, "no chunk DMA in the loop")
  RUN(2, "no global loads of A in the loop")
  RUN(3, "no DMA, no A loads")
  RUN(32, "no split of A (bits as fragments)")
  RUN(34, "no A loads, no split")
  RUN(16, "no fragment reads of B in the loop")
  RUN(8, "no barrier")
  RUN(4, "no MFMAs")
  RUN(51, "no DMA, no A loads, no B reads, no split: barrier + MFMAs")
  RUN(59, "... and no barrier: MFMAs + loop only")
#undef RUN
  CK(hipFree(a)); CK(hipFree(w)); CK(hipFree(c)); CK(hipFree(img));
}

// another panel width at another occupancy (NBLK = 12: 192-column panels, 48 KB of LDS, three workgroups per CU: 168 VGPRs, 20 B of scratch).
// ALL-ZERO operands here: the same launches run 11-14 % faster than over random data (0.121 vs 0.136 ms, 0.417 vs 0.475) -- the part's
// clocks follow the data's switching activity -- so compare inside this block only
template <int NBLK, int OCC>
static void occ_shape(const char* name, int64_t M, int N, int K, hipStream_t st) {
  const int panels = (N + 16 * NBLK - 1) / (16 * NBLK), kb = rp_kblocks(K, false);
  float *a, *w, *c;
  uint16_t* img;
  CK(hipMalloc(&a, (size_t)M * K * 4));
  CK(hipMalloc(&w, (size_t)N * K * 4));
  CK(hipMalloc(&c, (size_t)M * N * 4));
  CK(hipMalloc(&img, (size_t)panels * rp_image_elems(NBLK, kb) * 2));
  CK(hipMemset(a, 0, (size_t)M * K * 4));
  CK(hipMemset(w, 0, (size_t)N * K * 4));
  RpImageJobs jobs;
  rp_jobs_init(&jobs);
  const int pw = 16 * NBLK;
  for (int p = 0; p < panels; ++p)
    rp_jobs_add(&jobs, w + (int64_t)p * pw * K, K, 1, N - p * pw < pw ? N - p * pw : pw, K, nullptr, img + (size_t)p * rp_image_elems(NBLK, kb), NBLK);
  if (rp_jobs_launch(jobs, st) != NRL_OK) exit(1);
  CK(hipStreamSynchronize(st));
  const double gf = 2.0 * M * (double)N * K * 3 / 1e9;
  const float t = run<NBLK, 0, OCC>(a, img, c, M, N, K, panels, st);
  printf("%s: M=%lld N=%d K=%d  NBLK=%d (%d panels), %d workgroups per CU asked: %.3f ms (%4.0f TF-bf16/s of useful work)\n", name, (long long)M, N, K, NBLK, panels, OCC, t, gf / t);
  fflush(stdout);
  CK(hipFree(a)); CK(hipFree(w)); CK(hipFree(c)); CK(hipFree(img));
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  occ_shape<16, 2>("768 x 768", 38400, 768, 768, st);
  occ_shape<12, 2>("768 x 768", 38400, 768, 768, st);
  occ_shape<12, 3>("768 x 768", 38400, 768, 768, st);
  occ_shape<16, 2>("768 x 3072", 38400, 768, 3072, st);
  occ_shape<12, 2>("768 x 3072", 38400, 768, 3072, st);
  occ_shape<12, 3>("768 x 3072", 38400, 768, 3072, st);
  occ_shape<16, 2>("3072 x 768", 38400, 3072, 768, st);
  shape<16>("config-4 projection", 38400, 768, 768, st);
  shape<16>("config-4 feed-forward 2", 38400, 768, 3072, st);
  shape<19>("NRMS out-projection shape", 211200, 300, 300, st);
  return 0;
}
