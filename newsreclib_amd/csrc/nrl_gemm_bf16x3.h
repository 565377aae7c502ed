// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3"), gfx950.
//
//   a = a_hi + a_lo  with a_hi = bf16(a), a_lo = bf16(a - a_hi)   (likewise b)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (the dropped a_lo*b_lo term is ~2^-16 relative)
//
// Three v_mfma_f32_16x16x32_bf16 per 16x16x32 block, fp32 accumulation (bf16 x bf16 products are exact
// in fp32).  The bf16 MFMA runs at 16x the fp32-MFMA rate, so three of them are ~5.3x faster than the
// exact v_mfma_f32_16x16x4_f32 path while keeping ~16 mantissa bits per operand: scores stay within
// 1e-4 of the fp32 reference (contract: 1e-3; tests/test_gpu_parity.py runs both engines).
//
// Same operand-accessor / epilogue / tile machinery as nrl_gemm.h.  LDS image per operand and k-tile
// (BK = 32): two planes (hi, lo) of [row][32 bf16] = 64 B per row, the four 16-B chunks of a row
// XOR-swizzled (pc = c ^ h[(row>>2)&3], h = {0,3,2,1}) so the per-block fragment read -- one
// ds_read_b128 per plane, lane (l&15, l>>4) = (row, chunk) -- is bank-conflict free.
//   * fp32 k-contiguous sources (activations, gathered embedding rows) are split while staged;
//   * nn.Linear weights are pre-split once per step into bf16 planes (split_weight_planes) and staged
//     by plain 16-B copies (`KCSplit`);
//   * fp32 k-major sources (both operands of a weight gradient) are transposed while staged: a lane
//     loads the float4s of k and k+1 for 4 rows and writes 4 packed (k, k+1) bf16 pairs per plane.
#pragma once
#include "nrl_gemm.h"

namespace nrl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { SRC_SPLIT = 2 };

__device__ __forceinline__ uint32_t pack_bf16(__bf16 a, __bf16 b) {
  return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}
// (a, b) -> packed (hi_a | hi_b << 16), (lo_a | lo_b << 16).  Written on two-element vectors so that hipcc emits ONE
// v_cvt_pk_bf16_f32 per pair and a packed subtract: 5 VALU per pair (the scalar `(__bf16)a` form costs ~10: one
// convert per element with a zero second source, shifts and an sdwa merge).  Same round-to-nearest-even values.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const bf16x2_t h = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  const bf16x2_t l = __builtin_convertvector(f32x2_t{a - ha, b - hb}, bf16x2_t);
  lo = __builtin_bit_cast(uint32_t, l);
}

// pre-split weight, zero padded past K to a multiple of 32.  The hi and lo halves of one k-tile are
// ADJACENT: row r holds [k-tile 0: 32 hi | 32 lo][k-tile 1: 32 hi | 32 lo]..., i.e. element (r, k) of the
// hi plane sits at hi[r * ld + 2 * (k - k % 32) + k % 32], ld = 2 * Kp, and lo == hi + 32.  One k-tile of a
// row is then exactly one aligned 128-B line (two half-line fetches from separate planes cost twice the
// L2 -> L1 traffic; the staging of these GEMMs is bound by exactly that, profiles/r01_bw_probe.txt).
struct KCSplit {
  static constexpr int kLayout = SRC_SPLIT;
  const uint16_t* hi;
  const uint16_t* lo;
  int64_t ld;
  int64_t rows;
  struct State {};
};

__device__ __forceinline__ int swz(int chunk, int row) { return chunk ^ ((4 - ((row >> 2) & 3)) & 3); }

// DEEP = 1: two named staging-register sets, global loads issued TWO k-tiles ahead (the k-loop is
// unrolled by two so neither set is a loop-carried array the compiler has to copy); DEEP = 0: one set,
// loads one tile ahead.
template <int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi, int DEEP = 0, int OCC = 1>
__global__ void __launch_bounds__(WM* WN * 64, OCC)
    gemm_bf16x3_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N,
                       const int64_t K, const int tiles_n, const int64_t tiles_total,
                       const int64_t k_per_split, const int nsplit) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 32;
  constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;       // bytes per plane
  constexpr int BUF = 2 * (PLANE_A + PLANE_B);              // hi+lo of both operands
  static_assert(AOp::kLayout == SRC_KC || AOp::kLayout == SRC_RC, "A: fp32 source");
  static_assert(BOp::kLayout == SRC_SPLIT || BOp::kLayout == SRC_RC, "B: pre-split weight or fp32 k-major");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  // Tile / k-split of this workgroup.  Workgroup b runs on XCD b % 8 (private 4 MiB L2 each):
  //  * no split-K: XCD-aware bijective tile order, n-tiles sharing an A row-panel share an L2;
  //  * split-K (weight gradients): ALL tiles of one k-split are given to ONE XCD, so the split's
  //    slices of both operands are fetched from HBM once and re-read from that L2 by every tile.
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
  const bool primary = (n0 == 0);
  const int64_t kbeg = split * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  if (kbeg >= kend) return;
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);

  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;
  const bool full = (nvi == TM) && (nvj == TN);

  // ---- staging assignment ------------------------------------------------------------------
  // KC fp32: chunk = (row, 4 consecutive k); 8 chunks per row
  constexpr int NCH_A = AOp::kLayout == SRC_KC ? (BM * 8 + NT - 1) / NT : 1;
  // RC fp32: wave-piece = (k-half, block of 8 row-quads); lane = (row-quad in block, k-pair in half)
  constexpr int NPC_A = (2 * (BM / 32) + NW - 1) / NW;
  constexpr int NPC_B = (2 * (BN / 32) + NW - 1) / NW;
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile rows must be multiples of 32");
  // SPLIT: chunk = (row, 8 consecutive bf16); 4 chunks per row per plane
  constexpr int NCH_B = (BN * 4 + NT - 1) / NT;

  typename AOp::State sa[NCH_A];
  if constexpr (AOp::kLayout == SRC_KC) {
#pragma unroll
    for (int c = 0; c < NCH_A; ++c) {
      const int ch = tid + c * NT;
      sa[c] = A.init(ch < BM * 8 ? m0 + (ch >> 3) : (int64_t)1 << 60);
    }
  }

  constexpr int NRA = AOp::kLayout == SRC_KC ? NCH_A : 2 * NPC_A;
  constexpr int NRB = BOp::kLayout == SRC_SPLIT ? 2 * NCH_B : 2 * NPC_B;
  struct Stage {
    float4 ra[NRA];
    uint4 rb[NRB];  // SPLIT: hi/lo 16-B chunks; RC: the two float4 of a patch, bit-cast
  };

  auto load_tiles = [&](int64_t k0, Stage& S) {
    float4* ra = S.ra;
    uint4* rb = S.rb;
    if constexpr (AOp::kLayout == SRC_KC) {
#pragma unroll
      for (int c = 0; c < NCH_A; ++c) {
        const int ch = tid + c * NT;
        ra[c] = A.load(sa[c], (int)(k0 + 4 * (ch & 7)), (int)K);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NPC_A; ++c) {
        int piece = wave + c * NW;
        piece = piece < 2 * (BM / 32) ? piece : 2 * (BM / 32) - 1;
        const int rq = (piece >> 1) * 8 + (lane & 7), kp = (piece & 1) * 8 + (lane >> 3);
        ra[2 * c] = A.load(k0 + 2 * kp, m0 + 4 * rq, K);
        ra[2 * c + 1] = A.load(k0 + 2 * kp + 1, m0 + 4 * rq, K);
      }
    }
    if constexpr (BOp::kLayout == SRC_SPLIT) {
#pragma unroll
      for (int c = 0; c < NCH_B; ++c) {
        int ch = tid + c * NT;
        ch = ch < BN * 4 ? ch : BN * 4 - 1;
        int64_t row = (int64_t)n0 + (ch >> 2);
        row = row < B.rows ? row : B.rows - 1;
        const int64_t off = row * B.ld + 2 * k0 + 8 * (ch & 3);
        rb[2 * c] = *reinterpret_cast<const uint4*>(B.hi + off);
        rb[2 * c + 1] = *reinterpret_cast<const uint4*>(B.lo + off);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NPC_B; ++c) {
        int piece = wave + c * NW;
        piece = piece < 2 * (BN / 32) ? piece : 2 * (BN / 32) - 1;
        const int rq = (piece >> 1) * 8 + (lane & 7), kp = (piece & 1) * 8 + (lane >> 3);
        const float4 v0 = B.load(k0 + 2 * kp, (int64_t)n0 + 4 * rq, K);
        const float4 v1 = B.load(k0 + 2 * kp + 1, (int64_t)n0 + 4 * rq, K);
        rb[2 * c] = __builtin_bit_cast(uint4, v0);
        rb[2 * c + 1] = __builtin_bit_cast(uint4, v1);
      }
    }
  };

  auto store_rc_patch = [&](unsigned char* hi_plane, unsigned char* lo_plane, float4 v0, float4 v1, int rq, int kp) {
    const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * rq + j;
      uint32_t h, l;
      split_pair(a0[j], a1[j], h, l);
      const int off = row * 64 + swz(kp >> 2, row) * 16 + (kp & 3) * 4;
      *reinterpret_cast<uint32_t*>(hi_plane + off) = h;
      *reinterpret_cast<uint32_t*>(lo_plane + off) = l;
    }
  };

  auto store_tiles = [&](int buf, int64_t k0, Stage& S) {
    float4* ra = S.ra;
    uint4* rb = S.rb;
    unsigned char* base = smem + buf * BUF;
    unsigned char* a_hi = base;
    unsigned char* a_lo = base + PLANE_A;
    unsigned char* b_hi = base + 2 * PLANE_A;
    unsigned char* b_lo = b_hi + PLANE_B;
    if constexpr (AOp::kLayout == SRC_KC) {
#pragma unroll
      for (int c = 0; c < NCH_A; ++c) {
        const int ch = tid + c * NT;
        if (ch < BM * 8) {
          const int row = ch >> 3, kc4 = ch & 7;
          A.finish(ra[c], sa[c], m0 + row, (int)(k0 + 4 * kc4), (int)kend, primary);
          uint32_t h0, l0, h1, l1;
          split_pair(ra[c].x, ra[c].y, h0, l0);
          split_pair(ra[c].z, ra[c].w, h1, l1);
          const int off = row * 64 + swz(kc4 >> 1, row) * 16 + (kc4 & 1) * 8;
          *reinterpret_cast<uint2*>(a_hi + off) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(a_lo + off) = make_uint2(l0, l1);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < NPC_A; ++c) {
        const int piece = wave + c * NW;
        if (piece < 2 * (BM / 32)) {
          const int rq = (piece >> 1) * 8 + (lane & 7), kp = (piece & 1) * 8 + (lane >> 3);
          A.finish(ra[2 * c], k0 + 2 * kp, m0 + 4 * rq, kend);
          A.finish(ra[2 * c + 1], k0 + 2 * kp + 1, m0 + 4 * rq, kend);
          store_rc_patch(a_hi, a_lo, ra[2 * c], ra[2 * c + 1], rq, kp);
        }
      }
    }
    if constexpr (BOp::kLayout == SRC_SPLIT) {
#pragma unroll
      for (int c = 0; c < NCH_B; ++c) {
        const int ch = tid + c * NT;
        if (ch < BN * 4) {
          const int row = ch >> 2;
          const bool ok = (int64_t)n0 + row < B.rows;
          const int off = row * 64 + swz(ch & 3, row) * 16;
          *reinterpret_cast<uint4*>(b_hi + off) = ok ? rb[2 * c] : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(b_lo + off) = ok ? rb[2 * c + 1] : make_uint4(0, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < NPC_B; ++c) {
        const int piece = wave + c * NW;
        if (piece < 2 * (BN / 32)) {
          const int rq = (piece >> 1) * 8 + (lane & 7), kp = (piece & 1) * 8 + (lane >> 3);
          float4 v0 = __builtin_bit_cast(float4, rb[2 * c]), v1 = __builtin_bit_cast(float4, rb[2 * c + 1]);
          B.finish(v0, k0 + 2 * kp, (int64_t)n0 + 4 * rq, kend);
          B.finish(v1, k0 + 2 * kp + 1, (int64_t)n0 + 4 * rq, kend);
          store_rc_patch(b_hi, b_lo, v0, v1, rq, kp);
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned char* base = smem + buf * BUF;
    bf16x8 ah[TM], al[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 16 + l15;
      const int off = row * 64 + swz(g, row) * 16;
      ah[i] = *reinterpret_cast<const bf16x8*>(base + off);
      al[i] = *reinterpret_cast<const bf16x8*>(base + PLANE_A + off);
    }
    if constexpr (OCC >= 4) {
      // register-lean order (lets two 8-wave workgroups share a CU): B fragments are fetched two
      // column blocks at a time; the 3 split products of a block are issued small-terms-first and
      // interleaved over TM x 2 independent accumulators
#pragma unroll
      for (int j0 = 0; j0 < TN; j0 += 2) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (j0 + jj < TN) {
            const int row = (wn * TN + j0 + jj) * 16 + l15;
            const int off = row * 64 + swz(g, row) * 16;
            bh[jj] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + off);
            bl[jj] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + PLANE_B + off);
          }
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if (FULL || i < nvi) {
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                if (j0 + jj < TN && (FULL || j0 + jj < nvj))
                  acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                      pass == 1 ? al[i] : ah[i], pass == 0 ? bl[jj] : bh[jj], acc[i][j0 + jj], 0, 0, 0);
              }
            }
          }
        }
      }
    } else {
      bf16x8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 16 + l15;
        const int off = row * 64 + swz(g, row) * 16;
        bh[j] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + off);
        bl[j] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + PLANE_B + off);
      }
      // small cross terms first, the dominant hi*hi term last; each pass touches all blocks, so
      // consecutive MFMAs never depend on each other
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (FULL || i < nvi) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (FULL || j < nvj)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i] : ah[i],
                                                                   pass == 0 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
  };

  Stage S0, S1;
  auto kk = [&](int tt) { return kbeg + (int64_t)tt * BK; };
  load_tiles(kbeg, S0);
  store_tiles(0, kbeg, S0);
  if (DEEP && ntiles > 1) load_tiles(kk(1), S1);
  __syncthreads();

  auto k_loop = [&](auto full_tag) {
    if constexpr (DEEP) {
      // tile tt sits in LDS buffer 0, S1 holds tile tt+1 (in flight since the previous half-step)
      for (int tt = 0; tt < ntiles; tt += 2) {
        if (tt + 2 < ntiles) load_tiles(kk(tt + 2), S0);
        compute(0, full_tag);
        if (tt + 1 < ntiles) store_tiles(1, kk(tt + 1), S1);
        __syncthreads();
        if (tt + 1 < ntiles) {
          if (tt + 3 < ntiles) load_tiles(kk(tt + 3), S1);
          compute(1, full_tag);
          if (tt + 2 < ntiles) store_tiles(0, kk(tt + 2), S0);
          __syncthreads();
        }
      }
    } else {
      for (int tt = 0; tt < ntiles; ++tt) {
        const int buf = tt & 1;
        if (tt + 1 < ntiles) load_tiles(kk(tt + 1), S0);
        compute(buf, full_tag);
        if (tt + 1 < ntiles) store_tiles(buf ^ 1, kk(tt + 1), S0);
        __syncthreads();
      }
    }
  };
  if (full)
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});

  store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
}

template <int WM, int WN, int TM, int TN, int DEEP = 0, int OCC = 1, class AOp, class BOp, class Epi>
int launch_gemm_bf16x3(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
                       hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  int64_t kps = ceil_div(ceil_div(K, splits), 32) * 32;
  splits = (int)ceil_div(K, kps);
  const int64_t nblocks = splits > 1 ? ceil_div(splits, 8) * 8 * tiles_total : tiles_total;
  NRL_REQUIRE(nblocks < (1LL << 31), "gemm grid too large");
  hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, AOp, BOp, Epi, DEEP, OCC>), dim3((unsigned)nblocks),
                     dim3(WM * WN * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps, splits);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// ---- weight preparation: W (N, K) fp32 -> interleaved planes [N][2*Kp] (for x W^T) and [K][2*Np] (for dy W)
__device__ __forceinline__ int64_t split_pos(int64_t row, int k, int ld) {  // hi position; lo is + 32
  return row * ld + 2 * (k & ~31) + (k & 31);
}
static __global__ void split_weight_kernel(const float* __restrict__ w, int N, int K, int Kp, int Np,
                                    uint16_t* __restrict__ hi, uint16_t* __restrict__ hi_t) {
  const int64_t total = (int64_t)N * Kp, total_t = (int64_t)K * Np;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + total_t;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < total) {
      const int n = (int)(i / Kp), k = (int)(i % Kp);
      const float v = k < K ? w[(int64_t)n * K + k] : 0.f;
      const __bf16 h = (__bf16)v;
      const int64_t pos = split_pos(n, k, 2 * Kp);
      hi[pos] = __builtin_bit_cast(unsigned short, h);
      hi[pos + 32] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
    } else {
      const int64_t q = i - total;
      const int k = (int)(q / Np), n = (int)(q % Np);
      const float v = n < N ? w[(int64_t)n * K + k] : 0.f;
      const __bf16 h = (__bf16)v;
      const int64_t pos = split_pos(k, n, 2 * Np);
      hi_t[pos] = __builtin_bit_cast(unsigned short, h);
      hi_t[pos + 32] = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
    }
  }
}

struct SplitWeight {  // device pointers into the workspace
  uint16_t *hi, *lo, *hi_t, *lo_t;  // lo == hi + 32, lo_t == hi_t + 32 (interleaved layout, see KCSplit)
  int N, K, Kp, Np;
  int64_t ld, ld_t;                 // 2 * Kp, 2 * Np
};
static inline size_t split_weight_elems(int N, int K) {
  const int64_t Kp = (K + 31) / 32 * 32, Np = (N + 31) / 32 * 32;
  return (size_t)(2 * ((int64_t)N * Kp + (int64_t)K * Np));
}
static inline SplitWeight split_weight_view(uint16_t* buf, int N, int K) {
  SplitWeight o;
  o.N = N; o.K = K; o.Kp = (K + 31) / 32 * 32; o.Np = (N + 31) / 32 * 32;
  o.ld = 2 * (int64_t)o.Kp; o.ld_t = 2 * (int64_t)o.Np;
  o.hi = buf; o.lo = buf + 32;
  o.hi_t = buf + (size_t)N * o.ld; o.lo_t = o.hi_t + 32;
  return o;
}
static inline int split_weight(const float* w, int N, int K, uint16_t* buf, SplitWeight* out, hipStream_t st) {
  *out = split_weight_view(buf, N, K);
  const int64_t total = (int64_t)N * out->Kp + (int64_t)K * out->Np;
  hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, N, K, out->Kp,
                     out->Np, out->hi, out->hi_t);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
