// "A-stationary" bf16x3 GEMM for the short-K projections (K <= 320: every forward / dgrad GEMM of the NRMS
// block except the in-projection dgrad), gfx950.
//
//   C (M x N) = epi(A (M x K) * B^T),   A fp32 k-contiguous rows (plain / gathered), B pre-split planes
//
// With K this short the k-loop of a classic output-tiled GEMM is 7-10 iterations: prologue latency, the
// per-tile re-staging of A (fetched, dropout-hashed and split again by every one of the N / BN column
// tiles) and the epilogue dominate (in-projection forward: 0.91 ms against a 0.16 ms HBM floor).  Here a
// workgroup owns BM = 96 rows for ALL N columns:
//   * prologue: its A panel (96 x K) is fetched ONCE, finished (gather, dropout, x save), split to
//     (hi, lo) bf16 and parked in LDS for the whole kernel (96 x 320 x 2 planes x 2 B = 120 KB);
//   * main loop over (column tile, k-tile): only the weight planes stream, by LDS-DMA into a 2-deep ring
//     (2 x 20 KB) -- they are L2 resident and one k-tile of a row is one aligned 128-B line;
//     the loop body is fragment reads + 30 MFMAs per wave, no VALU conversion, no hash;
//   * the accumulators are flushed through the epilogue after the last k-tile of each column tile.
// LDS: 120 KB + 40 KB = the full 160 KB of a CU (one workgroup of 6 waves per CU).
//
// EXPERIMENT, not used by the product: bit-identical to the register-staged kernel, but NOT faster
// (profiles/r01_gemm_x3_dma_probe.txt: in-projection forward 1.01 ms vs 0.96 ms; N=300 K=300 0.25 vs 0.21 ms for
// the DMA kernel).  With the LDS full there is room for only 6 waves and a 2-deep ring, and what these loops
// need is more resident waves per SIMD (every sweep: 4 waves/CU ~0.65 ms, 8 waves/CU ~0.45-0.5 ms at the dgrad
// shape, independent of ring depth), not fewer instructions per wave.
#pragma once
#include "../../newsreclib_amd/csrc/nrl_gemm_bf16x3_dma.h"

namespace nrl {

template <int WM, int WN, int TM, int TN, int KT, class AOp, class Epi>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_bf16x3_astat_kernel(const AOp A, const KCSplit B, const Epi epi, const int64_t M, const int N, const int K,
                             const int tiles_n) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 32, S = 2;
  constexpr int PLANE_A = BM * 64;               // one plane of one k-tile
  constexpr int A_BYTES = KT * 2 * PLANE_A;
  constexpr int PLANE_B = BN * 64, STAGE = 2 * PLANE_B;
  constexpr int PB_PLANE = BN / 16, PB_TOT = 2 * PB_PLANE;
  constexpr int GB = (PB_TOT + NW - 1) / NW;
  constexpr int NCH = (BM * 8 + NT - 1) / NT;    // fp32 16-B chunks of one k-tile per thread
  static_assert(AOp::kLayout == SRC_KC, "A: fp32 k-contiguous source");
  static_assert(NT % 8 == 0, "a thread keeps its row across k-tiles");
  static_assert(A_BYTES + S * STAGE <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[A_BYTES + S * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int ntk = (K + BK - 1) / BK;             // <= KT
  const int nsteps = tiles_n * ntk;

  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- weight DMA: piece p (1 KiB = 16 rows x 64 B of one plane) of a stage; waves 0.. take GB pieces each
  auto issue = [&](int step, int buf) {
    const int nt = step / ntk, kt = step - nt * ntk;
    const uint32_t base = smem_base + (uint32_t)A_BYTES + (uint32_t)buf * (uint32_t)STAGE;
#pragma unroll
    for (int c = 0; c < GB; ++c) {
      const int piece = wave * GB + c;           // wave-uniform
      if (piece < PB_TOT) {
        const int plane = piece >= PB_PLANE ? 1 : 0;
        const int ch = (piece - plane * PB_PLANE) * 64 + lane;
        const int row = ch >> 2, pc = ch & 3;
        int64_t grow = (int64_t)nt * BN + row;
        grow = grow < B.rows ? grow : B.rows - 1;
        const int lc = pc ^ ((4 - ((row >> 2) & 3)) & 3);
        glds16_asm((plane ? B.lo : B.hi) + grow * B.ld + 2 * (kt * BK) + 8 * lc, base + (uint32_t)piece * 1024u);
      }
    }
  };
  issue(0, 0);

  // ---- prologue: the A panel, once: load -> finish (gather / dropout / x save) -> split -> LDS planes
  {
    typename AOp::State sa[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * NT;
      sa[c] = A.init(ch < BM * 8 ? m0 + (ch >> 3) : (int64_t)1 << 60);
    }
    float4 ra[KT][NCH];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = tid + c * NT;
        ra[kt][c] = A.load(sa[c], kt * BK + 4 * (ch & 7), K);
      }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = tid + c * NT;
        if (ch < BM * 8) {
          const int row = ch >> 3, kc4 = ch & 7;
          A.finish(ra[kt][c], sa[c], m0 + row, kt * BK + 4 * kc4, K, true);
          uint32_t h0, l0, h1, l1;
          split_pair(ra[kt][c].x, ra[kt][c].y, h0, l0);
          split_pair(ra[kt][c].z, ra[kt][c].w, h1, l1);
          unsigned char* a_hi = smem + kt * 2 * PLANE_A;
          const int off = row * 64 + swz(kc4 >> 1, row) * 16 + (kc4 & 1) * 8;
          *reinterpret_cast<uint2*>(a_hi + off) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(a_hi + PLANE_A + off) = make_uint2(l0, l1);
        }
      }
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int buf = 0, kt = 0, nt = 0;
  for (int step = 0; step < nsteps; ++step) {
    wait_vmcnt<0>();       // this wave's pieces of `step` have landed (ring depth 2: one group in flight)
    __syncthreads();       // ... everybody's have, and everybody is done reading the other stage
    if (step + 1 < nsteps) issue(step + 1, buf ^ 1);

    const unsigned char* a_hi = smem + kt * 2 * PLANE_A;
    const unsigned char* bb = smem + A_BYTES + buf * STAGE;
    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 16 + l15;
      const int off = row * 64 + swz(g, row) * 16;
      ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
      al[i] = *reinterpret_cast<const bf16x8*>(a_hi + PLANE_A + off);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = (wn * TN + j) * 16 + l15;
      const int off = row * 64 + swz(g, row) * 16;
      bh[j] = *reinterpret_cast<const bf16x8*>(bb + off);
      bl[j] = *reinterpret_cast<const bf16x8*>(bb + PLANE_B + off);
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i] : ah[i], pass == 0 ? bl[j] : bh[j],
                                                             acc[i][j], 0, 0, 0);
    buf ^= 1;
    if (++kt == ntk) {     // column tile finished: epilogue, fresh accumulators
      const int n0 = nt * BN;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + (wm * TM + i) * 16 + 4 * g + r;
          if (m < M) {
            const typename Epi::Row rs = epi.row(m);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int n = n0 + (wn * TN + j) * 16 + l15;
              if (n < N) epi(rs, m, n, acc[i][j][r]);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      kt = 0;
      ++nt;
    }
  }
}

// K <= 32 * KT
template <int WM, int WN, int TM, int TN, int KT, class AOp, class Epi>
int launch_gemm_bf16x3_astat(const AOp& A, const KCSplit& B, const Epi& epi, int64_t M, int N, int K,
                             hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  NRL_REQUIRE(K <= 32 * KT, "astat GEMM: K too large for the resident A panel");
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  NRL_REQUIRE(tiles_m < (1LL << 31), "gemm grid too large");
  hipLaunchKernelGGL((gemm_bf16x3_astat_kernel<WM, WN, TM, TN, KT, AOp, Epi>), dim3((unsigned)tiles_m),
                     dim3(WM * WN * 64), 0, stream, A, B, epi, M, N, K, tiles_n);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
