// Wave-specialised bf16x3 weight-gradient GEMM (gfx950):  C (I x J) += A^T B over k = data rows, both operands
// fp32 k-major (RCPlain / RCWindow accessors: element (k, r) at p[k * ld + r]), split-K over k.
//
// The register-staged kernel (nrl_gemm_bf16x3.h) gives every wave BOTH jobs -- fetch + split + transpose a
// k-tile into LDS, then run the MFMAs on it -- and only one k-tile of loads fits next to its 80 accumulators, so each
// k-tile pays a full global-load latency (matrix cores busy 20 %, SQ_WAIT 70 % of wave cycles,
// profiles/r01_x3_pmc_summary.json).  Here a workgroup is NL LOADER waves + WM x WN MFMA waves:
//   * loader waves own no accumulators: a three-deep register ring of k-tiles (56 VGPRs per stage) keeps two
//     tiles of global loads in flight; they split fp32 -> (hi, lo) bf16 ONCE per element, write the transposed
//     [row][32 k] planes (same conflict-free image as nrl_gemm_bf16x3.h) into the LDS buffer the MFMA waves will
//     read next, and never touch the matrix pipe;
//   * MFMA waves only read fragments (one ds_read_b128 per plane per block) and issue MFMAs: 8 x 5 blocks per
//     wave at 256 x 160, 120 MFMAs per k-tile against 13 fragment pairs.
// One s_barrier per k-tile hands buffers over (double-buffered planes).  A workgroup's waves are spread over the
// four SIMDs round-robin, so every SIMD hosts one loader and one MFMA wave: VALU / VMEM work and matrix work
// overlap by construction instead of by scheduling luck.
// Tile / split -> XCD mapping, epilogue (`store_accumulators`, atomic accumulation) and arithmetic are those of
// nrl_gemm_bf16x3.h: results are bit-identical up to the order of the atomic adds.
#pragma once
#include "nrl_gemm_bf16x3.h"

namespace nrl {

template <int NL, int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__((NL + WM * WN) * 64, 2)
    gemm_bf16x3_ws_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N, const int64_t K,
                          const int tiles_n, const int64_t tiles_total, const int64_t k_per_split, const int nsplit,
                          float* scratch) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 32;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;
  constexpr int BUF = 2 * (PLANE_A + PLANE_B);
  static_assert(AOp::kLayout == SRC_RC && BOp::kLayout == SRC_RC, "both operands fp32 k-major");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile rows must be multiples of 32");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  int64_t t, split;
  {
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid % 8, local = bid / 8;
    if (nsplit > 1) {
      t = local % tiles_total;
      split = (local / tiles_total) * 8 + xcd;
      if (split >= nsplit) return;
    } else {
      const int64_t q = tiles_total / 8, rem = tiles_total % 8;
      t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
      split = 0;
    }
  }
  const int64_t m0 = (t / tiles_n) * BM;
  const int n0 = (int)(t % tiles_n) * BN;
  const int64_t kbeg = split * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  if (kbeg >= kend) return;
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);
  auto kk = [&](int tt) { return kbeg + (int64_t)tt * BK; };

  if (wave < NL) {
    // ================================ loader waves ======================================================
    // wave-task = 64 output rows (16 row-quads) x the 32 k of the tile; lane = (row-quad rq = lane & 15, k-octet
    // ko = lane >> 4): a lane loads the float4 of rows 8 ko .. 8 ko + 7 for its 4 output rows (each load
    // instruction of the wave covers 4 k-rows x 256 contiguous bytes) and writes, per output row and plane, ONE
    // 16-byte chunk = its 8 consecutive k as bf16.
    constexpr int WT_A = (BM + 63) / 64, WT_B = (BN + 63) / 64;       // wave-tasks per operand
    constexpr int NT_L = (WT_A + WT_B + NL - 1) / NL;           // wave-tasks per loader wave
    const int rq = lane & 15, ko = lane >> 4;
    struct Stage {
      float4 r[NT_L][8];
    };
    // task q of this wave: operand + first output row of the lane's quad
    bool t_isb[NT_L], t_on[NT_L];
    int64_t t_row[NT_L];
#pragma unroll
    for (int q = 0; q < NT_L; ++q) {
      const int wt = wave + q * NL;
      t_on[q] = wt < WT_A + WT_B;
      t_isb[q] = wt >= WT_A;
      const int local = (t_isb[q] ? wt - WT_A : wt) * 64 + 4 * rq;
      t_row[q] = (t_isb[q] ? (int64_t)n0 : m0) + local;
      if (local >= (t_isb[q] ? BN : BM)) t_on[q] = false;       // tile rows % 64 != 0: the last wave-task is partial
    }
    auto load_tiles = [&](int64_t k0, Stage& S) {
#pragma unroll
      for (int q = 0; q < NT_L; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int64_t k = k0 + 8 * ko + e;
          S.r[q][e] = t_isb[q] ? B.load(k, t_row[q], K) : A.load(k, t_row[q], K);
        }
      }
    };
    auto store_tiles = [&](int buf, int64_t k0, Stage& S) {
      unsigned char* base = smem + buf * BUF;
#pragma unroll
      for (int q = 0; q < NT_L; ++q) {
        if (!t_on[q]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int64_t k = k0 + 8 * ko + e;
          if (t_isb[q]) B.finish(S.r[q][e], k, t_row[q], kend);
          else A.finish(S.r[q][e], k, t_row[q], kend);
        }
        const int wt = wave + q * NL;
        const int lrow = (t_isb[q] ? wt - WT_A : wt) * 64 + 4 * rq;
        unsigned char* hi_plane = base + (t_isb[q] ? 2 * PLANE_A : 0);
        unsigned char* lo_plane = hi_plane + (t_isb[q] ? PLANE_B : PLANE_A);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = lrow + j;
          const float v[8] = {(&S.r[q][0].x)[j], (&S.r[q][1].x)[j], (&S.r[q][2].x)[j], (&S.r[q][3].x)[j],
                              (&S.r[q][4].x)[j], (&S.r[q][5].x)[j], (&S.r[q][6].x)[j], (&S.r[q][7].x)[j]};
          uint32_t h[4], l[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
          const int off = row * 64 + swz(ko, row) * 16;
          *reinterpret_cast<uint4*>(hi_plane + off) = make_uint4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<uint4*>(lo_plane + off) = make_uint4(l[0], l[1], l[2], l[3]);
        }
      }
    };
    // Register ring of three k-tiles (named sets; the loop is unrolled by three so none of them is a loop-carried
    // array).  Barrier b (b = 0, 1, ..) separates "tile b is complete in buffer b & 1" from its consumption; the
    // loader writes tile b + 1 into the other buffer while the MFMA waves work on tile b.
    Stage S0, S1, S2;
    load_tiles(kk(0), S0);
    if (ntiles > 1) load_tiles(kk(1), S1);
    if (ntiles > 2) load_tiles(kk(2), S2);
    store_tiles(0, kk(0), S0);
    __builtin_amdgcn_s_barrier();                       // barrier 0: tile 0 visible
    auto step = [&](int tt, Stage& cur_next, Stage& refill) {
      // iteration tt: MFMA waves consume tile tt; this wave stages tile tt + 1 (already loaded) and re-issues the
      // set that held tile tt for tile tt + 3
      if (tt + 3 < ntiles) load_tiles(kk(tt + 3), refill);
      if (tt + 1 < ntiles) store_tiles((tt + 1) & 1, kk(tt + 1), cur_next);
      __builtin_amdgcn_s_barrier();                     // barrier tt + 1
    };
    for (int tt = 0; tt < ntiles; tt += 3) {
      step(tt, S1, S0);
      if (tt + 1 < ntiles) step(tt + 1, S2, S1);
      if (tt + 2 < ntiles) step(tt + 2, S0, S2);
    }
    return;
  }

  // ================================== MFMA waves ==========================================================
  const int mw = wave - NL;
  const int wm = mw / WN, wn = mw % WN;
  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * BUF;
    bf16x8 ah[TM], al[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 16 + l15;
      const int off = row * 64 + swz(g, row) * 16;
      ah[i] = *reinterpret_cast<const bf16x8*>(base + off);
      al[i] = *reinterpret_cast<const bf16x8*>(base + PLANE_A + off);
    }
    // one column block at a time: its three split products run over the TM independent accumulators of the
    // column (lo terms first), so consecutive MFMAs never depend on each other
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = (wn * TN + j) * 16 + l15;
      const int off = row * 64 + swz(g, row) * 16;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + off);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + PLANE_B + off);
      if (j < nvj) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            if (i < nvi)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i] : ah[i], pass == 0 ? bl : bh,
                                                                 acc[i][j], 0, 0, 0);
      }
    }
  };

  __builtin_amdgcn_s_barrier();                         // barrier 0
  for (int tt = 0; tt < ntiles; ++tt) {
    compute(tt & 1);
    __builtin_amdgcn_s_barrier();                       // barrier tt + 1
  }
  if (scratch != nullptr) {   // split-K in two steps (wgrad_reduce_kernel, nrl_gemm.h): partial tile -> scratch[split][tile]
    const EpiStore part{scratch + (split * tiles_total + t) * (BM * BN), BN};
    store_accumulators<TM, TN>(part, acc, 0, 0, wm, wn, l15, g, BM, BN);
  } else {
    store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
  }
}

template <int NL, int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi>
int launch_gemm_bf16x3_ws(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
                          hipStream_t stream, float* scratch = nullptr, size_t scratch_floats = 0) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  int64_t kps = ceil_div(ceil_div(K, splits), 32) * 32;
  splits = (int)ceil_div(K, kps);
  if (scratch != nullptr && splits > 1 && (size_t)splits * tiles_total * BM * BN > scratch_floats) {
    // fewer, longer splits if that is what fits the scratch (the two-step reduction no longer pays per split in atomics)
    const int fit = (int)(scratch_floats / ((size_t)tiles_total * BM * BN));
    if (fit >= 2 && fit * 2 >= splits) {
      kps = ceil_div(ceil_div(K, fit), 32) * 32;
      splits = (int)ceil_div(K, kps);
    }
  }
  if (splits < 2 || (size_t)splits * tiles_total * BM * BN > scratch_floats) scratch = nullptr;
  const int64_t nblocks = splits > 1 ? ceil_div(splits, 8) * 8 * tiles_total : tiles_total;
  NRL_REQUIRE(nblocks < (1LL << 31), "gemm grid too large");
  hipLaunchKernelGGL((gemm_bf16x3_ws_kernel<NL, WM, WN, TM, TN, AOp, BOp, Epi>), dim3((unsigned)nblocks),
                     dim3((NL + WM * WN) * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps, splits, scratch);
  NRL_LAUNCH_CHECK();
  if (scratch != nullptr) return launch_wgrad_reduce<BM, BN>(scratch, splits, tiles_n, tiles_total, M, N, epi, stream);
  return NRL_OK;
}

}  // namespace nrl
