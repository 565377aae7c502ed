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

// split_pair for operands that are NOT neighbours in registers (round 5).  The loaders' pairs are (k, k + 1) of one row -- the same
// component of two different loaded float4s.  `split_pair` builds a two-float vector for the conversion and lets hipcc pick
// v_pk_add_f32 for the subtraction: both want ALIGNED REGISTER PAIRS, so every pair cost moves (180 v_mov per three k-tiles, placed
// right behind the loads).  The four instructions below take any registers: same values bit for bit (v_cvt_pk_bf16_f32 rounds to
// nearest even, the difference is exact), 768 x 768 x 38400 0.174 -> 0.152 ms, 3072 x 768 0.70 -> 0.56 ms (0.96 PF/s).
__device__ __forceinline__ void ws_split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  float la, lb;
  asm("v_sub_f32 %0, %1, %2" : "=v"(la) : "v"(a), "v"(ha));
  asm("v_sub_f32 %0, %1, %2" : "=v"(lb) : "v"(b), "v"(hb));
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
}

template <int ABL>
__device__ __forceinline__ void ws_barrier() {
  if constexpr (!(ABL & 64)) __builtin_amdgcn_s_barrier();      // (64: probe builds without any barrier)
}

// LDS image of a plane: [row][32 k] bf16 = 64 bytes per row, four 16-byte chunks (chunk c = k 8c .. 8c + 7).  Two permutations
// make BOTH sides bank-conflict-free: the 4 x 4 transpose of a row's low four bits (a loader instruction writes rows 4 rq + j for
// 16 row-quads rq: physical rows with all four residues mod 4, i.e. all four 64-byte bank groups, where the plain order put the
// whole 1 KiB of the instruction into 16 of the 64 banks), and the chunk rotation by the physical row's bits 2..3 (a fragment read
// covers 16 consecutive rows at one chunk: 16 x 16 bytes over all 64 banks) plus its 16-row block index (lanes rq and rq + 4 of a
// loader instruction are 16 rows apart: the same bank group, now another chunk -- eight consecutive lanes never share a bank).
__device__ __forceinline__ int ws_lds_off(int row, int chunk) {
  const int pr = (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3);
  return pr * 64 + ((chunk ^ ((4 - ((pr >> 2) & 3) + (pr >> 4)) & 3)) * 16);
}

// ABL (probe builds only, tools/wgrad_ws_probe.hip): 1 = no epilogue, 2 = loaders store unsplit bits, 4 = no MFMAs, 8 = no global loads, 16 = one LDS buffer re-read by the MFMA waves, 32 = MFMA waves read their fragments once, 64 = no barriers, 128 = loaders re-read k-tile 0
template <int NL, int WM, int WN, int TM, int TN, class AOp, class BOp, class Epi, bool FULLK, int ABL = 0>
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
    // per task, ONCE: operand, first output row of the lane's quad, and the matrix as "column pointer + k * stride" (the accessors'
    // col / kstride / live / fill): a full k-tile's 16 bytes are then one 64-bit add away from a running pointer, and the edge
    // handling is one select per loaded vector in the tasks that HAVE edge rows (r01..r04 form: row clamp, k clamp, a 64-bit
    // multiply and two compares + four selects per 16 bytes -- the loaders, not the matrix pipe, set the k-tile time:
    // profiles/r05_wgrad_ws_probe.txt)
    bool t_isb[NT_L], t_on[NT_L], t_any[NT_L], t_live[NT_L], t_edge[NT_L];
    int64_t t_row[NT_L], t_stride[NT_L];
    const float* t_next[NT_L];
    float t_fill[NT_L];
#pragma unroll
    for (int q = 0; q < NT_L; ++q) {
      const int wt = wave + q * NL;
      t_on[q] = t_any[q] = wt < WT_A + WT_B;       // (t_any: wave-uniform -- a wave past the last task loads nothing)
      t_isb[q] = wt >= WT_A;
      const int task0 = (t_isb[q] ? wt - WT_A : wt) * 64;
      const int local = task0 + 4 * rq;
      t_row[q] = (t_isb[q] ? (int64_t)n0 : m0) + local;
      if (local >= (t_isb[q] ? BN : BM)) t_on[q] = false;       // tile rows % 64 != 0: the last wave-task is partial
      t_stride[q] = t_isb[q] ? B.kstride(K) : A.kstride(K);
      t_next[q] = (t_isb[q] ? B.col(t_row[q], K) : A.col(t_row[q], K)) + (kbeg + 8 * ko) * t_stride[q];
      t_live[q] = t_isb[q] ? B.live(t_row[q]) : A.live(t_row[q]);
      t_fill[q] = t_isb[q] ? B.fill(t_row[q]) : A.fill(t_row[q]);
      // (wave-uniform: does any row of this task lie past the operand's rows?)
      const int64_t task_end = (t_isb[q] ? (int64_t)n0 : m0) + task0 + 64;
      t_edge[q] = t_isb[q] ? !B.live(task_end - 4) : !A.live(task_end - 4);
    }
    // tiles are loaded in k order (0, 1, 2, then 3, 4, ..): the running pointers follow
    auto load_tiles = [&](int64_t k0, Stage& S) {
#pragma unroll
      for (int q = 0; q < NT_L; ++q) {
        if (!t_any[q]) continue;
        if constexpr (FULLK) {
          if (t_on[q]) {        // (the lanes past a partial task's rows fetch nothing: 160 columns cost 160, not 192, columns of traffic)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if constexpr (ABL & 8) S.r[q][e] = make_float4((float)k0, 1.f, 2.f, (float)e);
              else S.r[q][e] = *reinterpret_cast<const float4*>(t_next[q] + e * t_stride[q]);
            }
          }
          if constexpr (!(ABL & 128)) t_next[q] += BK * t_stride[q];      // (128: every k-tile re-reads the first one -- cache hits)
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t k = k0 + 8 * ko + e;
            S.r[q][e] = t_isb[q] ? B.load(k, t_row[q], K) : A.load(k, t_row[q], K);
          }
        }
      }
    };
    auto store_tiles = [&](int buf, int64_t k0, Stage& S) {
      unsigned char* base = smem + buf * BUF;
#pragma unroll
      for (int q = 0; q < NT_L; ++q) {
        if (!t_any[q]) continue;
        const int wt = wave + q * NL;
        const int lrow = (t_isb[q] ? wt - WT_A : wt) * 64 + 4 * rq;
        unsigned char* hi_plane = base + (t_isb[q] ? 2 * PLANE_A : 0);
        unsigned char* lo_plane = hi_plane + (t_isb[q] ? PLANE_B : PLANE_A);
        if constexpr (!FULLK) {       // a k extent that is not whole tiles: the accessors' own per-element edge rules
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t k = k0 + 8 * ko + e;
            if (t_isb[q]) B.finish(S.r[q][e], k, t_row[q], kend);
            else A.finish(S.r[q][e], k, t_row[q], kend);
          }
        }
        if (!FULLK || t_live[q]) {
          if (t_on[q]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = lrow + j;
              const float v[8] = {(&S.r[q][0].x)[j], (&S.r[q][1].x)[j], (&S.r[q][2].x)[j], (&S.r[q][3].x)[j],
                                  (&S.r[q][4].x)[j], (&S.r[q][5].x)[j], (&S.r[q][6].x)[j], (&S.r[q][7].x)[j]};
              uint32_t h[4], l[4];
#pragma unroll
              for (int p = 0; p < 4; ++p) {
                if constexpr (ABL & 2) { h[p] = __float_as_uint(v[2 * p]); l[p] = __float_as_uint(v[2 * p + 1]); }
                else if constexpr (ABL & 512) split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);       // (the r01..r05 form, for the probe)
                else ws_split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
              }
              const int off = ws_lds_off(row, ko);
              *reinterpret_cast<uint4*>(hi_plane + off) = make_uint4(h[0], h[1], h[2], h[3]);
              *reinterpret_cast<uint4*>(lo_plane + off) = make_uint4(l[0], l[1], l[2], l[3]);
            }
          }
        } else if (t_on[q]) {
          // rows past the operand (whole float4s: rows % 4 == 0): (fill, 0, 0, 0) at every k -- constants, no split
          const uint32_t hf = t_fill[q] != 0.f ? 0x3f803f80u : 0u;      // bf16(1.0) twice; its lo part is 0
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = lrow + j;
            const int off = ws_lds_off(row, ko);
            const uint32_t hv = j == 0 ? hf : 0u;
            *reinterpret_cast<uint4*>(hi_plane + off) = make_uint4(hv, hv, hv, hv);
            *reinterpret_cast<uint4*>(lo_plane + off) = make_uint4(0u, 0u, 0u, 0u);
          }
        }
      }
    };
    // Register ring of three k-tiles.  Barrier b (b = 0, 1, ..) separates "tile b is complete in buffer b & 1" from its
    // consumption; the loader writes tile b + 1 into the other buffer while the MFMA waves work on tile b.  Step tt: the MFMA
    // waves consume tile tt; this wave re-issues the set that held tile tt for tile tt + 3 and stages tile tt + 1.
    // The steady-state loop is three UNCONDITIONAL steps (every tile it touches exists), the last <= 5 steps are straight-line
    // code behind it: the three sets keep fixed roles, so none of them becomes a loop-carried array the compiler copies
    // (the r01..r04 form guarded every step inside the loop: 32 v_mov_b64 per iteration and spills once the loads got cheaper).
    // steady state of whole-tile extents: the refill's eight loads per task go out in PAIRS between the four row splits of the tile
    // being staged.  As one block they queue on the CU's one address unit (56 KiB-instructions x >= 16 clocks per k-tile, all four
    // loader waves in the same phase right behind the barrier) while the vector ALUs idle, and then the ALUs split while the
    // address unit idles.  (This form only pays once the split no longer builds register pairs, `ws_split_pair`: with them hipcc put
    // the pair moves right behind the loads and drained vmcnt(0) every k-tile -- 3.7 us per k-tile against 2.0.  Round-robin medians,
    // profiles/r05_wgrad_ws_probe.txt: 768 x 768 x 38400  0.163 (pairs) -> 0.148 (ws_split_pair) -> 0.143 ms (this form);
    // 3072 x 768  0.643 -> 0.545 -> 0.522 ms = 1.04 PF/s.  ABL & 256: the two-block form, for the probe.)
    auto fused_step = [&](Stage& SL, int buf, Stage& SS) {
      unsigned char* base = smem + buf * BUF;
#pragma unroll
      for (int q = 0; q < NT_L; ++q) {
        if (!t_any[q]) continue;
        const int wt = wave + q * NL;
        const int lrow = (t_isb[q] ? wt - WT_A : wt) * 64 + 4 * rq;
        unsigned char* hi_plane = base + (t_isb[q] ? 2 * PLANE_A : 0);
        unsigned char* lo_plane = hi_plane + (t_isb[q] ? PLANE_B : PLANE_A);
        {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 2 * j; e < 2 * j + 2; ++e) SL.r[q][e] = *reinterpret_cast<const float4*>(t_next[q] + e * t_stride[q]);
            const int row = lrow + j;
            const int off = ws_lds_off(row, ko);
            const float v[8] = {(&SS.r[q][0].x)[j], (&SS.r[q][1].x)[j], (&SS.r[q][2].x)[j], (&SS.r[q][3].x)[j],
                                (&SS.r[q][4].x)[j], (&SS.r[q][5].x)[j], (&SS.r[q][6].x)[j], (&SS.r[q][7].x)[j]};
            uint32_t h[4], l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) ws_split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
            if (!t_live[q]) {
#pragma unroll
              for (int p = 0; p < 4; ++p) { h[p] = (j == 0 && t_fill[q] != 0.f) ? 0x3f803f80u : 0u; l[p] = 0u; }
            }
            if (t_on[q]) {
              *reinterpret_cast<uint4*>(hi_plane + off) = make_uint4(h[0], h[1], h[2], h[3]);
              *reinterpret_cast<uint4*>(lo_plane + off) = make_uint4(l[0], l[1], l[2], l[3]);
            }
          }
        }
        t_next[q] += BK * t_stride[q];
      }
    };
    Stage S0, S1, S2;
    load_tiles(kk(0), S0);
    if (ntiles > 1) load_tiles(kk(1), S1);
    if (ntiles > 2) load_tiles(kk(2), S2);
    store_tiles(0, kk(0), S0);
    ws_barrier<ABL>();                       // barrier 0: tile 0 visible
    int tt = 0;
    if constexpr (FULLK && !(ABL & 256)) {
      for (; tt + 5 < ntiles; tt += 3) {
        fused_step(S0, (tt + 1) & 1, S1);
        ws_barrier<ABL>();
        fused_step(S1, (tt + 2) & 1, S2);
        ws_barrier<ABL>();
        fused_step(S2, (tt + 3) & 1, S0);
        ws_barrier<ABL>();
      }
    }
    for (; tt + 5 < ntiles; tt += 3) {
      load_tiles(kk(tt + 3), S0);
      store_tiles((tt + 1) & 1, kk(tt + 1), S1);
      ws_barrier<ABL>();
      load_tiles(kk(tt + 4), S1);
      store_tiles((tt + 2) & 1, kk(tt + 2), S2);
      ws_barrier<ABL>();
      load_tiles(kk(tt + 5), S2);
      store_tiles((tt + 3) & 1, kk(tt + 3), S0);
      ws_barrier<ABL>();
    }
    // (Also tried, profiles/r05_ab.txt: odd loader waves splitting first and fetching after -- the two orders meet at the back edge and
    //  the register sets get copied, 7 us per k-tile; an L2 prefetch by the MFMA waves, one LDS-DMA dword per 128-byte line 2..12
    //  k-tiles ahead -- no change: with the loaders re-reading ONE k-tile out of cache their time only fell from 1.9 to 1.27 us per
    //  k-tile, it was the issue of 56 KiB-instructions through the address unit plus the split, not the memory latency.)
    auto tail = [&](int t2, Stage& cur_next, Stage& refill) {
      if (t2 >= ntiles) return;
      if (t2 + 3 < ntiles) load_tiles(kk(t2 + 3), refill);
      if (t2 + 1 < ntiles) store_tiles((t2 + 1) & 1, kk(t2 + 1), cur_next);
      ws_barrier<ABL>();                     // barrier t2 + 1
    };
    tail(tt, S1, S0);
    tail(tt + 1, S2, S1);
    tail(tt + 2, S0, S2);
    tail(tt + 3, S1, S0);
    tail(tt + 4, S2, S1);
    return;
  }

  // ================================== MFMA waves ==========================================================
  const int mw = wave - NL;
  const int wm = mw / WN, wn = mw % WN;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // every block of the wave's tile is computed, also the ones past M / N (their rows arrive as zeros from the loaders and the
  // epilogue does not store them): the k-loop is straight-line code.  (Skipping the dead blocks, r01..r04, cost a scalar
  // compare + branch in front of every MFMA of EVERY tile to save a few MFMAs in the last one.)
  //
  // Fragment schedule (r05): ONE MFMA wave lives on a SIMD, so nothing but this wave's own earlier reads can hide an LDS
  // latency.  The column fragments of the k-tile (TN x (hi, lo)) stay in registers; the row fragments stream two row blocks at
  // a time, the next pair's four reads issued BEFORE the current pair's 6 TN MFMAs (>= 480 matrix cycles of cover), and the
  // 2 TN accumulators of a pass are independent.  The earlier form (all row fragments up front, column fragments on demand)
  // left the compiler to re-read column fragments per pass behind s_waitcnt lgkmcnt(0..2): matrix time + LDS time ADDED
  // (profiles/r05_wgrad_ws_probe.txt: 0.134 ms of MFMA-wave time for 0.054 ms of MFMAs at 768 x 768 x 38400).
  static_assert(TM % 2 == 0, "row blocks stream in pairs");
  auto compute = [&](int buf) {
    const unsigned char* base = smem + ((ABL & 16) ? 0 : buf) * BUF;     // (16: the fragment addresses do not move)
    auto afrag = [&](int i, int plane) {
      const int row = (wm * TM + i) * 16 + l15;
      return *reinterpret_cast<const bf16x8*>(base + plane * PLANE_A + ws_lds_off(row, g));
    };
    auto bfrag = [&](int j, int plane) {
      const int row = (wn * TN + j) * 16 + l15;
      return *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_A + plane * PLANE_B + ws_lds_off(row, g));
    };
    bf16x8 a0h = afrag(0, 0), a0l = afrag(0, 1), a1h = afrag(1, 0), a1l = afrag(1, 1);
    bf16x8 bh[TN], bl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bl[j] = bfrag(j, 1);
      bh[j] = bfrag(j, 0);
    }
#pragma unroll
    for (int ip = 0; ip < TM; ip += 2) {
      bf16x8 n0h = a0h, n0l = a0l, n1h = a1h, n1l = a1l;
      if (ip + 2 < TM) {
        n0h = afrag(ip + 2, 0);
        n0l = afrag(ip + 2, 1);
        n1h = afrag(ip + 3, 0);
        n1l = afrag(ip + 3, 1);
      }
      __builtin_amdgcn_sched_barrier(0);       // (the reads go out BEFORE the pair's MFMAs, not between them)
      if constexpr (ABL & 4) {
        asm volatile("" ::"v"(a0h), "v"(a0l), "v"(a1h), "v"(a1l));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
      } else {
        // (lo terms first, as ever: hi * lo, lo * hi, hi * hi per accumulator)
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[ip][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? a0l : a0h, pass == 0 ? bl[j] : bh[j], acc[ip][j], 0, 0, 0);
            acc[ip + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? a1l : a1h, pass == 0 ? bl[j] : bh[j], acc[ip + 1][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);       // (keep the pair structure: the scheduler would hoist every read to the top)
      a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
    }
  };

  ws_barrier<ABL>();                         // barrier 0
  if constexpr (ABL & 32) {       // (probe: matrix pace with the barriers, without LDS reads in the loop)
    const unsigned char* base = smem;
    bf16x8 f[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) f[c] = *reinterpret_cast<const bf16x8*>(base + c * 1024 + lane * 16);
    for (int tt = 0; tt < ntiles; ++tt) {
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[pass & 1], f[2 + (pass >> 1)], acc[i][j], 0, 0, 0);
      ws_barrier<ABL>();
    }
  } else
  for (int tt = 0; tt < ntiles; ++tt) {
    compute(tt & 1);
    ws_barrier<ABL>();                       // barrier tt + 1
  }
  if constexpr (ABL & 1) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) scratch[tid] = sum;
    return;
  }
  if (scratch != nullptr) {   // split-K in two steps (wgrad_reduce_kernel, nrl_gemm.h): partial tile -> scratch[split][tile]
    const EpiStore part{scratch + (split * tiles_total + t) * (BM * BN), BN};
    store_accumulators<TM, TN>(part, acc, 0, 0, wm, wn, l15, g, BM, BN);
  } else {
    store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
  }
}

template <int NL, int WM, int WN, int TM, int TN, int ABL = 0, class AOp, class BOp, class Epi>
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
  // (FULLK: every k-tile of every split is 32 whole rows -- the loaders then run on running pointers without k-edge rules)
  if (K % 32 == 0)
    hipLaunchKernelGGL((gemm_bf16x3_ws_kernel<NL, WM, WN, TM, TN, AOp, BOp, Epi, true, ABL>), dim3((unsigned)nblocks),
                       dim3((NL + WM * WN) * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps, splits, scratch);
  else
    hipLaunchKernelGGL((gemm_bf16x3_ws_kernel<NL, WM, WN, TM, TN, AOp, BOp, Epi, false, ABL>), dim3((unsigned)nblocks),
                       dim3((NL + WM * WN) * 64), 0, stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps, splits, scratch);
  NRL_LAUNCH_CHECK();
  if (scratch != nullptr) return launch_wgrad_reduce<BM, BN>(scratch, splits, tiles_n, tiles_total, M, N, epi, stream);
  return NRL_OK;
}

}  // namespace nrl
