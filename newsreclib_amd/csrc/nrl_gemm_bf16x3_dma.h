// bf16x3 GEMM with LDS-DMA staging and an S-deep LDS ring (gfx950), for the forward / dgrad projections:
// A = fp32 k-contiguous rows (plain, gathered, windowed), B = pre-split bf16 weight planes (KCSplit).
//
// Why: the register-staged kernel (nrl_gemm_bf16x3.h) is bound by its staging chain, not by the matrix
// cores -- ablation at M = 211200, N = 300, K = 900 (tools/gemm_x3_abl.hip, profiles/r01_gemm_x3_ablation.txt):
// MFMA + fragment reads alone 0.22 ms, staging alone (global load -> split -> ds_write -> barrier) 0.41 ms,
// together 0.54 ms: every k-tile pays a full global-load latency because only ONE tile of loads fits in the
// VGPR budget next to the 80 accumulators.  Here the tiles travel global -> LDS by `global_load_lds_dwordx4`
// (no VGPRs), so S - 1 k-tiles are in flight per workgroup, and the fp32 -> (hi, lo) split moves to the
// fragment read (a few VALU ops per MFMA operand, hidden under the MFMAs of the other wave).
//
// Measured (tools/gemm_x3_dma_probe.hip, profiles/r01_gemm_x3_dma_probe.txt, bit-identical outputs): dgrad
// N=300 K=900 0.60 -> 0.45 ms with 4-wave 128x160 workgroups and a 2-deep ring (two workgroups per CU, so one
// wave's split phase overlaps its SIMD partner's MFMA phase); deeper rings and splitting one tile ahead in the
// MFMA shadow (PIPE) do not help further: with the DMA removed the loop runs in 0.40 ms, the DMA alone in
// 0.45 ms -- the kernel now sits on the L2 -> LDS delivery rate of this access pattern (~8 TB/s aggregate;
// 128-B aligned A rows would give another 4-10 %).
//
// The DMA is issued through inline asm: with the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of the
// next ds_read (it cannot tell the DMA's LDS destination from the fragment reads), which serialises the ring.
// The waits are explicit instead: before tile t is consumed, `s_waitcnt vmcnt((S-2) * G)` (G = DMA
// instructions per wave per tile, identical for every wave) + one barrier per k-tile.
//
// LDS image of one stage: A fp32 [BM rows][32 k] (128 B per row, the eight 16-B chunks XOR-swizzled by
// (row >> 1) & 5 so the two ds_read_b128 of a fragment are conflict-free), then the hi and lo planes of B
// [BN rows][32 bf16] with the 4-chunk swizzle of nrl_gemm_bf16x3.h.  LDS-DMA writes lane-linear, so the
// swizzle is applied to the SOURCE address each lane fetches.
#pragma once
#include "nrl_gemm_bf16x3.h"

namespace nrl {

__device__ __forceinline__ void glds16_asm(const void* src, uint32_t lds_byte_addr_uniform) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off"
      :
      : "s"(lds_byte_addr_uniform), "v"(src)
      : "memory");
}

// the same with a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit VALU add per DMA
__device__ __forceinline__ void glds16_saddr(const void* base_uniform, uint32_t lane_byte_off, uint32_t lds_byte_addr_uniform) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "s"(lds_byte_addr_uniform), "v"(lane_byte_off), "s"(base_uniform)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WM, int WN, int TM, int TN, int S, class AOp, class Epi, int ABL = 0, int PIPE = 0, int JC = TN>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_bf16x3_dma_kernel(const AOp A, const KCSplit B, const Epi epi, const int64_t M, const int N,
                           const int64_t K, const int tiles_n, const int64_t tiles_total) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 32;
  constexpr int A_BYTES = BM * 128, PLANE_B = BN * 64;
  constexpr int STAGE = A_BYTES + 2 * PLANE_B;
  constexpr int PA_TOT = BM / 8;           // 64-lane pieces of the A tile (1 KiB each)
  constexpr int PB_PLANE = BN / 16;        // pieces per B plane
  constexpr int PB_TOT = 2 * PB_PLANE;
  constexpr int GA = (PA_TOT + NW - 1) / NW, GB = (PB_TOT + NW - 1) / NW;
  constexpr int G = GA + GB;               // DMA instructions per wave per k-tile
  static_assert(AOp::kLayout == SRC_KC, "A: fp32 k-contiguous source");
  static_assert(S >= 2 && (S - 2) * G <= 63, "ring depth vs vmcnt range");
  static_assert(BN % 16 == 0 && BM % 16 == 0, "tile shape");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[S * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  int64_t t;
  {
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid % 8, local = bid / 8;
    const int64_t q = tiles_total / 8, rem = tiles_total % 8;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int64_t m0 = (t / tiles_n) * BM;
  const int n0 = (int)(t % tiles_n) * BN;
  const bool primary = (n0 == 0) && (wn == 0);
  const int ntiles = (int)((K + BK - 1) / BK);

  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;
  const bool full = (nvi == TM) && (nvj == TN);

  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- DMA assignment: wave w moves pieces w, w + NW, ... (clamped: a duplicate piece rewrites the same
  // bytes, which keeps the per-wave instruction count -- and with it the vmcnt arithmetic -- uniform)
  typename AOp::State sa[GA];
  int ka[GA];          // logical k offset (floats) of the chunk this lane fetches
  uint32_t da[GA];     // LDS byte offset of the piece inside a stage
#pragma unroll
  for (int c = 0; c < GA; ++c) {
    int piece = wave + c * NW;
    piece = piece < PA_TOT ? piece : PA_TOT - 1;
    const int ch = piece * 64 + lane;
    const int row = ch >> 3, pc = ch & 7;
    sa[c] = A.init(m0 + row);
    ka[c] = 4 * (pc ^ ((row >> 1) & 5));
    da[c] = (uint32_t)piece * 1024u;
  }
  const uint16_t* pb[GB];  // source of this lane's chunk at k0 = 0
  uint32_t db[GB];
#pragma unroll
  for (int c = 0; c < GB; ++c) {
    int piece = wave + c * NW;
    piece = piece < PB_TOT ? piece : PB_TOT - 1;
    const int plane = piece >= PB_PLANE ? 1 : 0;
    const int ch = (piece - plane * PB_PLANE) * 64 + lane;
    const int row = ch >> 2, pc = ch & 3;
    int64_t grow = (int64_t)n0 + row;
    grow = grow < B.rows ? grow : B.rows - 1;
    const int lc = pc ^ ((4 - ((row >> 2) & 3)) & 3);  // swz is an XOR: its own inverse
    pb[c] = (plane ? B.lo : B.hi) + grow * B.ld + 8 * lc;
    db[c] = (uint32_t)(A_BYTES + plane * PLANE_B) + (uint32_t)(piece - plane * PB_PLANE) * 1024u;
  }

  auto issue = [&](int tile, int buf) {
    const int k0 = tile * BK;
    const uint32_t base = smem_base + (uint32_t)buf * (uint32_t)STAGE;
#pragma unroll
    for (int c = 0; c < GA; ++c) glds16_asm(A.src(sa[c], k0 + ka[c], (int)K), base + da[c]);
#pragma unroll
    for (int c = 0; c < GB; ++c) glds16_asm(pb[c] + 2 * k0, base + db[c]);  // interleaved planes: k-tile stride 64
  };

  // fragment-row states (for `finish`: row validity, dropout, window mask, x save)
  typename AOp::State fa[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = A.init(m0 + (wm * TM + i) * 16 + l15);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // raw fp32 A fragments of one stage -> finish (row / k masks, dropout, x save) -> (hi, lo) bf16 fragments
  auto convert_a = [&](int buf, int k0, bool prim, bf16x8 (&ah)[TM], bf16x8 (&al)[TM]) {
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 16 + l15;
      const int s = (row >> 1) & 5;
      float4 v0 = *reinterpret_cast<const float4*>(base + row * 128 + ((2 * g) ^ s) * 16);
      float4 v1 = *reinterpret_cast<const float4*>(base + row * 128 + ((2 * g + 1) ^ s) * 16);
      A.finish(v0, fa[i], m0 + row, k0 + 8 * g, (int)K, prim);
      A.finish(v1, fa[i], m0 + row, k0 + 8 * g + 4, (int)K, prim);
      uint32_t h[4], l[4];
      if constexpr (ABL & 2) {  // probe only: no split
        h[0] = __float_as_uint(v0.x); h[1] = __float_as_uint(v0.y); h[2] = __float_as_uint(v0.z); h[3] = __float_as_uint(v0.w);
        l[0] = __float_as_uint(v1.x); l[1] = __float_as_uint(v1.y); l[2] = __float_as_uint(v1.z); l[3] = __float_as_uint(v1.w);
      } else {
        split_pair(v0.x, v0.y, h[0], l[0]);
        split_pair(v0.z, v0.w, h[1], l[1]);
        split_pair(v1.x, v1.y, h[2], l[2]);
        split_pair(v1.z, v1.w, h[3], l[3]);
      }
      ah[i] = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
      al[i] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    }
  };
  auto mfma_tile = [&](int buf, const bf16x8 (&ah)[TM], const bf16x8 (&al)[TM], auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const unsigned char* base = smem + buf * STAGE;
    // B fragments are fetched JC column blocks at a time (wide tiles: all TN at once would not fit next to
    // the accumulators); per accumulator the three split products keep the order lo-terms first, hi*hi last
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += JC) {
      bf16x8 bh[JC], bl[JC];
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        if (j0 + jj < TN) {
          const int row = (wn * TN + j0 + jj) * 16 + l15;
          const int off = row * 64 + swz(g, row) * 16;
          bh[jj] = *reinterpret_cast<const bf16x8*>(base + A_BYTES + off);
          bl[jj] = *reinterpret_cast<const bf16x8*>(base + A_BYTES + PLANE_B + off);
        }
      }
      if constexpr (ABL & 4) {  // probe only: no MFMA, keep the fragments alive
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) asm volatile("" ::"v"(bh[jj]), "v"(bl[jj]));
        continue;
      }
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (FULL || i < nvi) {
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
              if (j0 + jj < TN && (FULL || j0 + jj < nvj))
                acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    pass == 1 ? al[i] : ah[i], pass == 0 ? bl[jj] : bh[jj], acc[i][j0 + jj], 0, 0, 0);
            }
          }
        }
      }
    }
  };
  auto compute = [&](int buf, int k0, auto full_tag) {
    bf16x8 ah[TM], al[TM];
    convert_a(buf, k0, primary, ah, al);
    mfma_tile(buf, ah, al, full_tag);
  };

  // prologue: tiles 0 .. S-2 in flight
#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < ntiles) issue(p, p);

  auto k_loop = [&](auto full_tag) {
    int buf = 0, nbuf = S - 1;  // nbuf = buffer of tile tt + S - 1 = the one read in iteration tt - 1
    for (int tt = 0; tt < ntiles; ++tt) {
      // groups possibly in flight: tiles tt .. min(tt + S - 2, ntiles - 1); tile tt must have landed
      const int ahead = ntiles - 1 - tt;
      if (ahead >= S - 2) {
        wait_vmcnt<(S - 2) * G>();
      } else {
        wait_vmcnt<0>();
      }
      if constexpr (!(ABL & 8)) __syncthreads();
      if constexpr (!(ABL & 1))
        if (tt + S - 1 < ntiles) issue(tt + S - 1, nbuf);
      compute(buf, tt * BK, full_tag);
      buf = buf + 1 == S ? 0 : buf + 1;
      nbuf = nbuf + 1 == S ? 0 : nbuf + 1;
    }
  };
  // PIPE: the A fragments of tile tt + 1 are read and split WHILE the MFMAs of tile tt run (the split is
  // ~150 VALU ops per k-tile and wave; done up front it idles the matrix pipe for a third of the iteration).
  // Two named fragment sets alternate (loop unrolled by two, no register copies); the MFMA / VALU
  // interleave is requested from the scheduler with sched_group_barrier.
  auto k_loop_pipe = [&](auto full_tag) {
    static_assert(!PIPE || S >= 3, "pipelined mode reads two stages per iteration");
    bf16x8 ah0[TM], al0[TM], ah1[TM], al1[TM];
    if (ntiles - 1 >= S - 2) {
      wait_vmcnt<(S - 2) * G>();
    } else {
      wait_vmcnt<0>();
    }
    __syncthreads();
    convert_a(0, 0, primary, ah0, al0);
    int buf = 0;
    auto step = [&](int tt, const bf16x8 (&ahc)[TM], const bf16x8 (&alc)[TM], bf16x8 (&ahn)[TM], bf16x8 (&aln)[TM]) {
      const bool more = tt + 1 < ntiles;
      if (more) {  // tile tt + 1 must have landed; tiles tt + 2 .. may still be in flight
        if (S > 3 && ntiles - tt - 2 >= S - 3) {
          wait_vmcnt<(S > 3 ? (S - 3) * G : 0)>();
        } else {
          wait_vmcnt<0>();
        }
      }
      __syncthreads();
      const int b1 = buf + 1 == S ? 0 : buf + 1;
      const int bp = buf == 0 ? S - 1 : buf - 1;  // stage of tile tt - 1 == stage of tile tt + S - 1
      if (tt + S - 1 < ntiles) issue(tt + S - 1, bp);
      mfma_tile(buf, ahc, alc, full_tag);
      convert_a(b1, (tt + 1) * BK, primary && more, ahn, aln);  // (a stale stage on the last tile: unused)
      // pin the split to THIS block (otherwise it is sunk into the next iteration, in front of the barrier)
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(ahn[i]), "+v"(aln[i]));
      constexpr int NMF = 3 * TM * TN;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * TN + 2 * TM, 0);  // all fragment reads first
#pragma unroll
      for (int q = 0; q < NMF; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three VALU ops in its shadow
      }
      buf = b1;
    };
    for (int tt = 0; tt < ntiles; tt += 2) {
      step(tt, ah0, al0, ah1, al1);
      if (tt + 1 >= ntiles) break;
      step(tt + 1, ah1, al1, ah0, al0);
    }
  };
  if constexpr (PIPE) {
    if (full)
      k_loop_pipe(std::true_type{});
    else
      k_loop_pipe(std::false_type{});
  } else {
    if (full)
      k_loop(std::true_type{});
    else
      k_loop(std::false_type{});
  }

  store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
}

template <int WM, int WN, int TM, int TN, int S, int ABL = 0, int PIPE = 0, int JC = TN, class AOp, class Epi>
int launch_gemm_bf16x3_dma(const AOp& A, const KCSplit& B, const Epi& epi, int64_t M, int N, int64_t K,
                           hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  NRL_REQUIRE(tiles_total < (1LL << 31), "gemm grid too large");
  hipLaunchKernelGGL((gemm_bf16x3_dma_kernel<WM, WN, TM, TN, S, AOp, Epi, ABL, PIPE, JC>), dim3((unsigned)tiles_total),
                     dim3(WM * WN * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// ---------------------------------------------------------------------------------------------------
// Weight-gradient form: C (I x J) += A^T B with BOTH operands k-major fp32 (element (k, r) at p[k*ld + r]),
// split-K over k with all tiles of one k-split on one XCD (as in nrl_gemm_bf16x3.h).  The k-tile of an
// operand is 32 k-rows x ROWS floats; LDS-DMA copies it row-major ([k][ROWS], 16-B chunks XOR-swizzled by
// 4 * ((k >> 3) & 1) so the two k-groups of a 32-lane ds_read_b32 group use disjoint bank halves), and the
// TRANSPOSITION happens at the fragment read: a lane fetches its 8 consecutive k of one column with eight
// ds_read_b32 (immediate offsets s * ROWS * 4) and splits them to (hi, lo) bf16.  No VGPR staging, no
// ds_write pass.  Out-of-range rows / k tails / the virtual ones-column are served by pointing the lane at
// 16-byte constants (accessor `src`).
// ---------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int S, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_bf16x3_dma_tn_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N,
                              const int64_t K, const int tiles_n, const int64_t tiles_total,
                              const int64_t k_per_split, const int nsplit, float* scratch) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 32;
  constexpr int A_BYTES = BK * BM * 4, B_BYTES = BK * BN * 4;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CA = BM / 4, CB = BN / 4;        // 16-B chunks per k-row
  constexpr int PA_TOT = BK * CA / 64, PB_TOT = BK * CB / 64;
  constexpr int GA = (PA_TOT + NW - 1) / NW, GB = (PB_TOT + NW - 1) / NW;
  constexpr int G = GA + GB;
  static_assert(AOp::kLayout == SRC_RC && BOp::kLayout == SRC_RC, "both operands k-major fp32");
  static_assert(CA % 8 == 0 && CB % 8 == 0, "tile rows must be multiples of 32 (chunk swizzle)");
  static_assert(S >= 2 && (S - 2) * G <= 63, "ring depth vs vmcnt range");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[S * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
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

  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;
  const bool full = (nvi == TM) && (nvj == TN);

  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // DMA assignment (pieces clamped so every wave issues exactly G instructions per k-tile)
  int kra[GA], ca[GA], krb[GB], cb[GB];
  uint32_t da[GA], db[GB];
#pragma unroll
  for (int c = 0; c < GA; ++c) {
    int piece = wave + c * NW;
    piece = piece < PA_TOT ? piece : PA_TOT - 1;
    const int ch = piece * 64 + lane;
    kra[c] = ch / CA;
    ca[c] = 4 * ((ch % CA) ^ (((kra[c] >> 3) & 1) << 2));
    da[c] = (uint32_t)piece * 1024u;
  }
#pragma unroll
  for (int c = 0; c < GB; ++c) {
    int piece = wave + c * NW;
    piece = piece < PB_TOT ? piece : PB_TOT - 1;
    const int ch = piece * 64 + lane;
    krb[c] = ch / CB;
    cb[c] = 4 * ((ch % CB) ^ (((krb[c] >> 3) & 1) << 2));
    db[c] = (uint32_t)A_BYTES + (uint32_t)piece * 1024u;
  }
  auto issue = [&](int tile, int buf) {
    const int64_t k0 = kbeg + (int64_t)tile * BK;
    const uint32_t base = smem_base + (uint32_t)buf * (uint32_t)STAGE;
#pragma unroll
    for (int c = 0; c < GA; ++c) glds16_asm(A.src(k0 + kra[c], m0 + ca[c], kend), base + da[c]);
#pragma unroll
    for (int c = 0; c < GB; ++c) glds16_asm(B.src(k0 + krb[c], (int64_t)n0 + cb[c], kend), base + db[c]);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // float offset of this lane's column inside a k-row image (chunk swizzle of its k-group applied)
  const int fsw = (g & 1) << 2;
  auto frag = [&](const float* img, int rows_per_k, int col, bf16x8& hi, bf16x8& lo) {
    const float* p = img + (8 * g) * rows_per_k + ((((col >> 2) ^ fsw) << 2) | (col & 3));
    float v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = p[s * rows_per_k];
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
  };

  auto compute = [&](int buf, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const float* as = reinterpret_cast<const float*>(smem + buf * STAGE);
    const float* bs = reinterpret_cast<const float*>(smem + buf * STAGE + A_BYTES);
    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) frag(as, BM, (wm * TM + i) * 16 + l15, ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) frag(bs, BN, (wn * TN + j) * 16 + l15, bh[j], bl[j]);
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
  };

#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < ntiles) issue(p, p);

  auto k_loop = [&](auto full_tag) {
    int buf = 0, nbuf = S - 1;
    for (int tt = 0; tt < ntiles; ++tt) {
      const int ahead = ntiles - 1 - tt;
      if (ahead >= S - 2) {
        wait_vmcnt<(S - 2) * G>();
      } else {
        wait_vmcnt<0>();
      }
      __syncthreads();
      if (tt + S - 1 < ntiles) issue(tt + S - 1, nbuf);
      compute(buf, full_tag);
      buf = buf + 1 == S ? 0 : buf + 1;
      nbuf = nbuf + 1 == S ? 0 : nbuf + 1;
    }
  };
  if (full)
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});

  if (scratch != nullptr) {   // split-K in two steps (wgrad_reduce_kernel, nrl_gemm.h)
    const EpiStore part{scratch + (split * tiles_total + t) * (BM * BN), BN};
    store_accumulators<TM, TN>(part, acc, 0, 0, wm, wn, l15, g, BM, BN);
  } else {
    store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
  }
}

template <int WM, int WN, int TM, int TN, int S, class AOp, class BOp, class Epi>
int launch_gemm_bf16x3_dma_tn(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
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
  hipLaunchKernelGGL((gemm_bf16x3_dma_tn_kernel<WM, WN, TM, TN, S, AOp, BOp, Epi>), dim3((unsigned)nblocks),
                     dim3(WM * WN * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps, splits, scratch);
  NRL_LAUNCH_CHECK();
  if (scratch != nullptr) return launch_wgrad_reduce<BM, BN>(scratch, splits, tiles_n, tiles_total, M, N, epi, stream);
  return NRL_OK;
}

}  // namespace nrl
