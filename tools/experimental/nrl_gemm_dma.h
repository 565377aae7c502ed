// Exact-fp32 MFMA GEMM, LDS-DMA staged ("glds"): global -> LDS with global_load_lds_dwordx4, no
// VGPR round trip, no ds_write pass, no transposition.  Same operand accessors / epilogues / tile
// parameters as nrl_gemm.h.  EXPERIMENT, not used by the product: verified bit-for-tolerance equal to
// the register-staged kernel but measured 0-12 % SLOWER at the NRMS shapes (profiles/
// r01_gemm_dma_probe.txt) -- with K = 300 and fp32 MFMA the staging was not the limiter.
//
// LDS images (one k-tile = BK = 16 k-values):
//   k-contiguous operand ("KC": activations, nn.Linear weights):  [row][16 floats], 64 B per row,
//     the four 16-B chunks of a row stored at physical chunk  pc = lc ^ h[(row >> 2) & 3],
//     h = {0, 3, 2, 1}.  A lane's MFMA fragment for the whole k-tile is ONE ds_read_b128 of logical
//     chunk lc = lane >> 4 (k = 4*(lane>>4) + t feeds MFMA step t); the XOR makes the 16 rows x 4
//     chunks that one ds_read_b128 lane group touches fall on 16 distinct 16-B slots (conflict-free).
//     LDS-DMA can only write lane-linear (base + lane*16), so the swizzle is applied to each lane's
//     global SOURCE address (guide rule 21).
//   k-major operand ("RC": the in-place weight of a dgrad, both operands of a wgrad):
//     [k][rows + 4 floats]; the 16-B pad chunk per k-row keeps the image lane-linear for the DMA and
//     makes (rows + 4) == 4 (mod 8), so the b32 fragment read of k = 4g + t is conflict-free.
// Both operands use the same k <-> (lane group, step) mapping, so any permutation cancels.
//
// Out-of-range rows / k / split-K tails and the virtual ones-column are served by pointing the
// lane at 16-byte constants (zeros / {1,0,0,0}) instead of predicating the DMA.
#pragma once
#include "../../newsreclib_amd/csrc/nrl_gemm.h"

namespace nrl {

__device__ const float nrl_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ const float nrl_ones16[4] = {1.f, 0.f, 0.f, 0.f};

// ---- DMA source-address helpers for the accessors of nrl_gemm.h --------------------------------
__device__ __forceinline__ const float* dma_src(const KCPlain&, const KCPlain::State& s, int k, int K) {
  return (s.ok && k < K) ? s.ptr + k : nrl_zero16;
}
__device__ __forceinline__ const float* dma_src(const KCGather&, const KCGather::State& s, int k, int K) {
  return (s.ok && k < K) ? s.ptr + k : nrl_zero16;
}
__device__ __forceinline__ const float* dma_src(const RCPlain& a, int64_t k, int64_t r, int64_t kend) {
  if (k >= kend) return nrl_zero16;
  if (r < a.rows) return a.p + k * a.ld + r;
  return (a.ones && r == a.rows) ? nrl_ones16 : nrl_zero16;
}

// post-read transform of an A fragment (4 consecutive k of one row): dropout + save for KCGather
__device__ __forceinline__ void frag_fix(const KCPlain&, float4&, int64_t, int, int64_t, int, bool) {}
__device__ __forceinline__ void frag_fix(const KCGather& a, float4& v, int64_t m, int k, int64_t M, int K,
                                         bool save_here) {
  if (a.drop.thresh != 0u) {
    const uint32_t idx = (uint32_t)m * (uint32_t)a.dim + (uint32_t)k;
    v.x *= a.drop.mult(idx);
    v.y *= a.drop.mult(idx + 1);
    v.z *= a.drop.mult(idx + 2);
    v.w *= a.drop.mult(idx + 3);
  }
  if (save_here && a.save != nullptr && m < M && k < K)
    *reinterpret_cast<float4*>(a.save + m * (int64_t)a.dim + k) = v;
}

__device__ __forceinline__ void glds16(const float* src, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

template <int ROWS, int LAYOUT>
struct DmaTile {
  // KC: ROWS*4 chunks; RC: 16 * (ROWS/4 + 1) chunks; both rounded up to whole 64-lane pieces
  static constexpr int kChunks = LAYOUT == SRC_KC ? ROWS * 4 : 16 * (ROWS / 4 + 1);
  static constexpr int kPieces = (kChunks + 63) / 64;
  static constexpr int kFloats = kPieces * 256;
  static constexpr int kLd = ROWS + 4;  // RC row length in floats
};

template <int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_f32_dma_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N,
                        const int64_t K, const int tiles_n, const int64_t tiles_total,
                        const int64_t k_per_split) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 16;
  using TA = DmaTile<BM, AOp::kLayout>;
  using TB = DmaTile<BN, BOp::kLayout>;
  constexpr int PA = (TA::kPieces + NW - 1) / NW, PB = (TB::kPieces + NW - 1) / NW;
  __shared__ __attribute__((aligned(1024))) float smem[2 * (TA::kFloats + TB::kFloats)];
  float* const As = smem;
  float* const Bs = smem + 2 * TA::kFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  int64_t t;
  {
    const int64_t bid = blockIdx.x;
    const int64_t q = tiles_total / 8, rem = tiles_total % 8;
    const int64_t xcd = bid % 8, local = bid / 8;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int64_t m0 = (t / tiles_n) * BM;
  const int n0 = (int)(t % tiles_n) * BN;
  const bool primary = (n0 == 0) && (wn == 0);
  const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  if (kbeg >= kend) return;
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);

  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;
  const bool full = (nvi == TM) && (nvj == TN);

  // per-lane DMA assignment: piece p (64 chunks) of each operand tile, p = wave, wave + NW, ...
  auto hsw = [](int a) { return (4 - a) & 3; };  // {0, 3, 2, 1}
  typename AOp::State sa[PA];
  typename BOp::State sb[PB];
  int lca[PA], lcb[PB];  // KC: logical k-chunk this lane fetches (un-swizzled)
  if constexpr (AOp::kLayout == SRC_KC) {
#pragma unroll
    for (int c = 0; c < PA; ++c) {
      const int ch = (wave + c * NW) * 64 + lane;
      const int row = ch >> 2;
      sa[c] = A.init(row < BM ? m0 + row : (int64_t)1 << 60);
      lca[c] = (ch & 3) ^ hsw((row >> 2) & 3);
    }
  }
  if constexpr (BOp::kLayout == SRC_KC) {
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      const int ch = (wave + c * NW) * 64 + lane;
      const int row = ch >> 2;
      sb[c] = B.init(row < BN ? (int64_t)n0 + row : (int64_t)1 << 60);
      lcb[c] = (ch & 3) ^ hsw((row >> 2) & 3);
    }
  }

  auto stage = [&](int buf, int64_t k0) {
    float* as = As + buf * TA::kFloats;
    float* bs = Bs + buf * TB::kFloats;
#pragma unroll
    for (int c = 0; c < PA; ++c) {
      const int piece = wave + c * NW;  // wave-uniform
      if (piece < TA::kPieces) {
        const float* src;
        if constexpr (AOp::kLayout == SRC_KC) {
          src = dma_src(A, sa[c], (int)(k0 + 4 * lca[c]), (int)kend);
        } else {
          constexpr int CPR = BM / 4 + 1;
          const int ch = piece * 64 + lane;
          const int kk = ch / CPR, rc = ch % CPR;
          src = (kk < BK && rc < CPR - 1) ? dma_src(A, k0 + kk, m0 + 4 * rc, kend) : nrl_zero16;
        }
        glds16(src, as + piece * 256);
      }
    }
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      const int piece = wave + c * NW;
      if (piece < TB::kPieces) {
        const float* src;
        if constexpr (BOp::kLayout == SRC_KC) {
          src = dma_src(B, sb[c], (int)(k0 + 4 * lcb[c]), (int)kend);
        } else {
          constexpr int CPR = BN / 4 + 1;
          const int ch = piece * 64 + lane;
          const int kk = ch / CPR, rc = ch % CPR;
          src = (kk < BK && rc < CPR - 1) ? dma_src(B, k0 + kk, (int64_t)n0 + 4 * rc, kend) : nrl_zero16;
        }
        glds16(src, bs + piece * 256);
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, kbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int tt = 0; tt < ntiles; ++tt) {
      const int buf = tt & 1;
      const int64_t k0 = kbeg + (int64_t)tt * BK;
      if (tt + 1 < ntiles) stage(buf ^ 1, k0 + BK);  // DMA of the next tile flies under the MFMAs

      const float* as = As + buf * TA::kFloats;
      const float* bs = Bs + buf * TB::kFloats;
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + l15;
        if constexpr (AOp::kLayout == SRC_KC) {
          float4 v = *reinterpret_cast<const float4*>(as + row * 16 + 4 * (g ^ hsw((row >> 2) & 3)));
          frag_fix(A, v, m0 + row, (int)(k0 + 4 * g), M, (int)kend, primary);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[i][s] = as[(4 * g + s) * TA::kLd + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 16 + l15;
        if constexpr (BOp::kLayout == SRC_KC) {
          const float4 v = *reinterpret_cast<const float4*>(bs + row * 16 + 4 * (g ^ hsw((row >> 2) & 3)));
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = bs[(4 * g + s) * TB::kLd + row];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (FULL || i < nvi) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (FULL || j < nvj)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
      __syncthreads();
    }
  };
  if (full)
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});

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
}

template <int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi>
int launch_gemm_dma(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
                    hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  int64_t kps = ceil_div(ceil_div(K, splits), 16) * 16;
  splits = (int)ceil_div(K, kps);
  NRL_REQUIRE(tiles_total < (1LL << 31) && splits < 65536, "gemm grid too large");
  dim3 grid((unsigned)tiles_total, (unsigned)splits, 1);
  hipLaunchKernelGGL((gemm_f32_dma_kernel<WM, WN, TM, TN, AOp, BOp, Epi>), grid, dim3(WM * WN * 64), 0,
                     stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
