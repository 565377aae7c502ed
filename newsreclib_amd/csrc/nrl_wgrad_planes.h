// Weight gradient of the news encoder's in-projection from PRE-SPLIT operands (gfx950):
//
//     dW'[i'][j] += sum over token rows m of  dqkv[m][i'] * x[m][j]        (embedding_dim 300: 960 x 301 outputs)
//
// The register-staged / wave-specialised kernels (nrl_gemm_bf16x3.h, nrl_gemm_ws.h) fetch both operands as fp32,
// split every element to (hi, lo) bf16 in VALU once per tile that uses it, transpose through registers into LDS
// and reach 25 % matrix-core occupancy (0.62 ms at B = 128: 4 x the MFMA time).  Here the PRODUCERS of the two
// operands -- the token-attention backward (dqkv) and the fused forward (x, which holds exactly these fragments in
// registers) -- write the split once, as "fragment-block planes":
//
//   block (mb = m / 16, cb = feature / 16), plane p (0 = hi, 1 = lo): 16 x 16 bf16, [m % 16][feature % 16] row-major,
//   512 bytes; a token-row block of x is [cb 0 .. ncb - 1][p][512], of dqkv (per head) [cb 0 .. 3][p][512]
//   (rows m are padded to 32 per news: two blocks per news, pad rows zero in dqkv).
//
// and this kernel is nothing but LDS-DMA (global_load_lds_dwordx4: a block pair is 1 KiB of contiguous global and
// LDS memory), `ds_read_b64_tr_b16` fragment reads (the 16 x 16 block is [k = token][i = feature]: the transpose
// read hands lane (i, g) its four tokens 4g .. 4g + 3, two blocks = the 8 k of a 16 x 16 x 32 MFMA operand -- the
// SAME token set on the A and the B side, which is all a reduction needs) and MFMAs.  No VALU in the loop.
//
// Workgroup = 4 wavefronts (2 x 2), tile 256 (4 heads x 64) x 160, k-tile = 32 token rows = one news; three 52 KiB
// LDS stages; split-K over news with atomic accumulation (EpiAtomicWBHeads remaps the 64-row head groups to the
// [Wq; Wk; Wv] rows).  Arithmetic: hi * lo + lo * hi + hi * hi into fp32, as everywhere else.
#pragma once
#include <atomic>
#include "nrl_gemm_bf16x3_dma.h"

namespace nrl {

// tile = 256 rows (4 heads x 64) x 32 * WP_TN columns (WP_TN = column blocks per wave; 10 = the whole width would fetch
// every dqkv byte once per launch, but its 320 accumulator registers per lane make hipcc shuttle them through
// v_accvgpr moves: 0.68 ms against 0.47; the same 256 x 320 tile on EIGHT waves (2 x 4, two waves per SIMD, 160
// accumulator registers each) fits registers but only two 72 KiB LDS stages: 0.40 ms against 0.36 -- the kernel lives
// on bytes in flight (~7-8 TB/s of L2 -> LDS traffic at two k-tiles ahead), and one k-tile ahead is not enough)
constexpr int WP_TN = 5;
constexpr int WP_A_STAGE = 32 * 1024, WP_B_STAGE = 2 * (2 * WP_TN) * 1024, WP_STAGE = WP_A_STAGE + WP_B_STAGE, WP_STAGES = 3;
constexpr int WP_PIECES = WP_STAGE / 1024;                 // 72 one-KiB pieces per k-tile, 18 per wave
static_assert(WP_PIECES % 4 == 0, "pieces are dealt to four waves");

typedef short wp_v4i16 __attribute__((ext_vector_type(4)));

struct WgradPlanesArgs {
  const unsigned char* a;   // dqkv planes: ((head * n_mb + mb) * 4 + cb) * 1024 + p * 512
  const unsigned char* b;   // x planes:    (mb * ncb_b + cb) * 1024 + p * 512
  int64_t n_mb;             // token-row blocks (even: 2 per news)
  int heads, ncb_b;         // ncb_b = 20: 320 feature columns (300 + the ones column + pad)
  int tiles_m, tiles_n;     // ceil(heads / 4), ncb_b / (2 * WP_TN)
  int nsplit;
  int64_t kt_per_split;     // k-tiles (news) per split
  int64_t M;                // heads * 64 logical output rows
  int N;                    // valid output columns (D + 1)
  float* scratch;           // (nsplit, tiles, BM x BN) partial tiles, or null: atomics straight into the gradient
};

// (split-K reduction in two steps: wgrad_reduce_kernel, nrl_gemm.h)
// ABL (tools/wp_probe.hip only): 1 = no DMA inside the loop, 2 = no MFMAs
// NW (round 6): wavefronts per workgroup over the SAME 256 x 160 tile and the same three LDS stages.  4 = (2 x 2), one wave per
// SIMD, 8 x 5 accumulator blocks each (rounds 2-5).  8 = (4 x 2), TWO waves per SIMD, 4 x 5 blocks each: a k-tile's fixed costs
// -- the counted wait, the barrier, the first fragment reads behind it, the scalar address arithmetic of the DMA issues -- are
// paid by one wave while its SIMD neighbour runs MFMAs (HISTORY.md section 8.0b priced them at 0.28 us per k-tile with one wave
// per SIMD); every wave issues 7 one-KiB pieces per k-tile (4 of A, 3 of B: the 20 B pieces over 8 waves leave four
// duplicates, which rewrite the same bytes).
template <class Epi, int ABL = 0, int NW = 4>
__global__ void __launch_bounds__(NW * 64, 1) wgrad_planes_kernel(const WgradPlanesArgs P, const Epi epi) {
  static_assert(NW == 4 || NW == 8, "4 waves (2 x 2) or 8 waves (4 x 2)");
  constexpr int RBW = 32 / NW;                             // row blocks per wave: 8 or 4
  constexpr int NA = 32 / NW, NB = (20 + NW - 1) / NW;     // DMA pieces per wave and k-tile: 8 + 5 or 4 + 3
  extern __shared__ __attribute__((aligned(1024))) unsigned char wp_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wp_smem;

  // workgroup -> (tile, split): all tiles of a split on ONE XCD (its operand slices are re-read from that L2)
  const int tiles_total = P.tiles_m * P.tiles_n;
  const int64_t bid = blockIdx.x;
  const int64_t xcd = bid % 8, local = bid / 8;
  const int t = (int)(local % tiles_total);
  const int64_t split = (local / tiles_total) * 8 + xcd;
  if (split >= P.nsplit) return;
  const int tm = t / P.tiles_n, tn = t % P.tiles_n;
  const int64_t kt_all = P.n_mb / 2;
  const int64_t kt0 = split * P.kt_per_split;
  const int64_t kt1 = kt0 + P.kt_per_split < kt_all ? kt0 + P.kt_per_split : kt_all;
  if (kt0 >= kt1) return;
  const int nkt = (int)(kt1 - kt0);

  // ---- DMA: one-KiB pieces (a 16 x 16 block, both planes).  LDS stage = A [hh 0..3][mbi 0..1][cb 0..3] | B [mbi 0..1][cb 0..9].
  //   A piece pa = q * NW + wave (q < NA): (hh * 2 + mbi) = pa >> 2, cb = pa & 3 -- LDS byte pa * 1024
  //   B piece pb = q * NW + wave (q < NB), clamped to 19: mbi = pb / 10, cb = pb % 10 -- LDS byte A_STAGE + pb * 1024
  // Wave-uniform base pointers of the split's first k-tile live in SGPRs; a k-tile advances them by a constant.
  static_assert(WP_TN == 5, "piece dealing below assumes 32 A + 20 B pieces");
  const unsigned char* base_a[NA];
  const unsigned char* base_b[NB];
  uint32_t lds_b[NB];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int pa = q * NW + wave;
    int head = 4 * tm + (pa >> 3);
    head = head < P.heads ? head : P.heads - 1;           // empty head slots of the last tile: rows dropped by the epilogue
    base_a[q] = P.a + (((int64_t)head * P.n_mb + 2 * kt0 + ((pa >> 2) & 1)) * 4 + (pa & 3)) * 1024;
  }
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    int pb = q * NW + wave;
    pb = pb < 20 ? pb : 19;
    const int mbi = pb / 10, cb = pb - mbi * 10;
    base_b[q] = P.b + ((2 * kt0 + mbi) * P.ncb_b + 10 * tn + cb) * 1024;
    lds_b[q] = (uint32_t)WP_A_STAGE + (uint32_t)pb * 1024u;
  }
  const int64_t step_a = 2 * 4 * 1024, step_b = 2 * (int64_t)P.ncb_b * 1024;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // pieces [c0, c1) of this wave's NA + NB (0 .. NA - 1: A, then B) of k-tile `rel` (relative to kt0) into `stage`
  auto issue = [&](int rel, int stage, int c0, int c1) {
    const uint32_t sbase = smem_base + (uint32_t)stage * WP_STAGE;
#pragma unroll
    for (int c = c0; c < c1; ++c) {
      if (c < NA) glds16_saddr(base_a[c] + rel * step_a, lane16, sbase + (uint32_t)(c * NW + wave) * 1024u);
      else glds16_saddr(base_b[c - NA] + rel * step_b, lane16, sbase + lds_b[c - NA]);
    }
  };

  f32x4 acc[RBW][WP_TN];
#pragma unroll
  for (int i = 0; i < RBW; ++i)
#pragma unroll
    for (int j = 0; j < WP_TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read address of this lane inside a 16 x 16 block: row 4g + (l15 >> 2), 4 bf16 at column 4 (l15 & 3)
  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  auto frag = [&](uint32_t blk0, uint32_t blk1) -> bf16x8 {
    typedef __attribute__((address_space(3))) wp_v4i16* lds_v4;
    const wp_v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(blk0 + lane_off));
    const wp_v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(blk1 + lane_off));
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    const v8i16 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  issue(0, 0, 0, NA + NB);
  issue(nkt > 1 ? 1 : 0, 1, 0, NA + NB);
  int stage = 0;
  for (int it = 0; it < nkt; ++it) {
    // tile `it` (issued two iterations ago) has landed for this wave: at most the pieces of tile it + 1 are in flight
    if constexpr (ABL & 1) wait_vmcnt<0>(); else wait_vmcnt<NA + NB>();
    __builtin_amdgcn_s_barrier();                          // ... and for every wave; stage (it + 2) % 3 is free
    const int nx = it + 2 < nkt ? it + 2 : nkt - 1;        // uniform control flow: the tail re-fetches the last tile
    const int st2 = stage == 0 ? 2 : stage - 1;            // (it + 2) % 3
    // (the DMA issues of tile it + 2 are spread over the row steps below: as one block in front of the MFMAs their scalar
    //  address arithmetic was ~1000 cycles per k-tile that nothing overlapped)
    auto issue_part = [&](int step) {
      if constexpr (!(ABL & 1)) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NW == 4) issue(nx, st2, step < 5 ? 2 * step : 5 + step, step < 5 ? 2 * step + 2 : 6 + step);   // 2 2 2 2 2 1 1 1
        else issue(nx, st2, step < 3 ? 2 * step : 6, step < 3 ? 2 * step + 2 : 7);                                   // 2 2 2 1
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const uint32_t sa = smem_base + (uint32_t)stage * WP_STAGE;
    const uint32_t sb = sa + WP_A_STAGE;
    // B fragments of this wave's column blocks
    bf16x8 bh[WP_TN], bl[WP_TN];
#pragma unroll
    for (int j = 0; j < WP_TN; ++j) {
      const uint32_t c0 = sb + (uint32_t)(WP_TN * wn + j) * 1024u, c1 = c0 + (uint32_t)(2 * WP_TN) * 1024u;   // mbi = 0, 1
      bh[j] = frag(c0, c1);
      bl[j] = frag(c0 + 512u, c1 + 512u);
    }
    // A fragments of row block i + 1 are fetched while the MFMAs of row block i run (two named sets; on its own hipcc
    // reads each fragment right before its first MFMA and waits for the LDS there)
    auto read_a = [&](int i, bf16x8& ah, bf16x8& al) {
      const int hh = NW == 4 ? 2 * wm + (i >> 2) : wm, cb = i & 3;
      const uint32_t a0 = sa + (uint32_t)((hh * 2 + 0) * 4 + cb) * 1024u, a1 = sa + (uint32_t)((hh * 2 + 1) * 4 + cb) * 1024u;
      ah = frag(a0, a1);
      al = frag(a0 + 512u, a1 + 512u);
    };
    auto mfma_row = [&](int i, const bf16x8& ah, const bf16x8& al) {
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int j = 0; j < WP_TN; ++j) {
          if constexpr (ABL & 2) asm volatile("" ::"v"(ah), "v"(al), "v"(bh[j]), "v"(bl[j]));
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al : ah, pass == 0 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
        }
    };
    bf16x8 ah0, al0, ah1, al1;
    read_a(0, ah0, al0);
#pragma unroll
    for (int i = 0; i < RBW; i += 2) {
      read_a(i + 1, ah1, al1);
      mfma_row(i, ah0, al0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * WP_TN, 0);
      issue_part(i);
      if (i + 2 < RBW) read_a(i + 2, ah0, al0);
      mfma_row(i + 1, ah1, al1);
      if (i + 2 < RBW) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * WP_TN, 0);
      issue_part(i + 1);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  wait_vmcnt<0>();                                          // the re-fetched tail tile must not outlive the LDS allocation

  if (P.scratch != nullptr) {
    const EpiStore part{P.scratch + ((int64_t)split * tiles_total + t) * (256 * 32 * WP_TN), 32 * WP_TN};
    store_accumulators<RBW, WP_TN>(part, acc, 0, 0, wm, wn, l15, g, 256, 32 * WP_TN);
  } else {
    store_accumulators<RBW, WP_TN>(epi, acc, (int64_t)tm * 256, tn * 32 * WP_TN, wm, wn, l15, g, P.M, P.N);
  }
}

static inline size_t wgrad_planes_scratch_floats(int heads, int ncb_b, int nsplit) {
  return (size_t)nsplit * ((heads + 3) / 4) * (ncb_b / (2 * WP_TN)) * 256 * 32 * WP_TN;
}
template <int ABL = 0, int NW = 4, class Epi>
static inline int launch_wgrad_planes(const void* a_planes, const void* b_planes, int64_t n_news, int heads, int ncb_b,
                                      int n_valid, const Epi& epi, int nsplit, hipStream_t st, float* scratch = nullptr) {
  if (n_news <= 0) return NRL_OK;
  NRL_REQUIRE(a_planes && b_planes && heads > 0 && ncb_b > 0 && ncb_b % (2 * WP_TN) == 0, "wgrad_planes: bad arguments");
  WgradPlanesArgs P;
  P.a = (const unsigned char*)a_planes; P.b = (const unsigned char*)b_planes;
  P.n_mb = 2 * n_news; P.heads = heads; P.ncb_b = ncb_b;
  P.tiles_m = (heads + 3) / 4; P.tiles_n = ncb_b / (2 * WP_TN);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > n_news) nsplit = (int)n_news;
  P.kt_per_split = ceil_div(n_news, nsplit);
  P.nsplit = (int)ceil_div(n_news, P.kt_per_split);
  P.M = (int64_t)heads * 64; P.N = n_valid;
  const int64_t blocks = (int64_t)P.tiles_m * P.tiles_n * ceil_div(P.nsplit, 8) * 8;
  NRL_REQUIRE(blocks < (1LL << 31), "wgrad_planes: grid too large");
  static std::atomic<uint64_t> attr_done{0};                // 156 KiB of dynamic LDS: opt in once per kernel instance AND device
  int dev = 0;
  NRL_HIP(hipGetDevice(&dev));
  if (!((attr_done.load(std::memory_order_relaxed) >> (dev & 63)) & 1u)) {
    NRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_kernel<Epi, ABL, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, WP_STAGES * WP_STAGE));
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
  }
  P.scratch = P.nsplit > 1 ? scratch : nullptr;
  hipLaunchKernelGGL((wgrad_planes_kernel<Epi, ABL, NW>), dim3((unsigned)blocks), dim3(NW * 64), WP_STAGES * WP_STAGE, st, P, epi);
  NRL_LAUNCH_CHECK();
  if (P.scratch != nullptr) {
    const int tiles = P.tiles_m * P.tiles_n;
    const int64_t threads = (int64_t)tiles * 256 * 32 * WP_TN / 4;
    hipLaunchKernelGGL((wgrad_reduce_kernel<256, 32 * WP_TN, Epi>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st,
                       P.scratch, P.nsplit, P.tiles_n, tiles, P.M, P.N, epi);
    NRL_LAUNCH_CHECK();
  }
  return NRL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Generic form: both operands plain fragment-block planes over the SAME row blocks,
//   A: (mb * ncb_a + cb) * 1024 + p * 512,  B: (mb * ncb_b + cb) * 1024 + p * 512        (mb = row / 16)
// C[i][j] += sum over rows A[m][i] B[m][j]; tile (2 TM x 16) x (2 TN x 16), 4 waves (2 x 2), three LDS stages.  Used for the
// out-projection weight gradient of the fused news path (dy planes from the row-panel epilogue, o planes from the fused
// forward): 160 x 160 tiles (TM = TN = 5) cover its 19 x 19 blocks with two tiles a side.
struct WgradPlanesGArgs {
  const unsigned char* a;
  const unsigned char* b;
  int64_t n_mb;             // row blocks (even)
  int ncb_a, ncb_b;
  int tiles_m, tiles_n;
  int nsplit;
  int64_t kt_per_split;     // k-tiles (32 rows) per split
  int64_t M;                // valid output rows
  int N;                    // valid output columns
  float* scratch;           // (nsplit, tiles, BM x BN) partial tiles, or null
};

// NW (round 6): 4 = (2 x 2) waves, TM x TN accumulator blocks each, one wave per SIMD (rounds 2-5).  8 = (4 x 2) waves over the SAME
// tile and LDS stages, two waves per SIMD: the 2 TM row blocks are dealt to the four wave rows as evenly as they go (TM = 5:
// 3, 3, 2, 2; TM = 7: 4, 4, 3, 3) -- waves w and w + 4 share a SIMD, so every SIMD carries TM row blocks either way.
template <int TM, int TN, class Epi, int NW = 4>
__global__ void __launch_bounds__(NW * 64, 1) wgrad_planes_g_kernel(const WgradPlanesGArgs P, const Epi epi) {
  static_assert(NW == 4 || NW == 8, "4 waves (2 x 2) or 8 waves (4 x 2)");
  constexpr int A_ST = 4 * TM * 1024, B_ST = 4 * TN * 1024, STAGE = A_ST + B_ST;
  constexpr int NPA_ALL = 4 * TM, NP_ALL = 4 * (TM + TN);  // one-KiB pieces of a k-tile (A first)
  static_assert(NP_ALL % NW == 0, "pieces are dealt evenly to the waves");
  constexpr int NP = NP_ALL / NW;                          // pieces per wave and k-tile
  constexpr int RQ = (2 * TM) / (NW / 2), RR = (2 * TM) % (NW / 2);
  constexpr int RB = NW == 4 ? TM : RQ + (RR > 0 ? 1 : 0);  // accumulator row blocks per wave (the most any wave row has)
  constexpr int PER_STEP = (NP + RB - 1) / RB;
  extern __shared__ __attribute__((aligned(1024))) unsigned char wp_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // this wave's row blocks of the tile: [r0, r0 + nrows)
  const int r0 = NW == 4 ? wm * TM : (wm < RR ? wm * (RQ + 1) : RR * (RQ + 1) + (wm - RR) * RQ);
  const int nrows = NW == 4 ? TM : RQ + (wm < RR ? 1 : 0);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wp_smem;

  const int tiles_total = P.tiles_m * P.tiles_n;
  const int64_t bid = blockIdx.x;
  const int64_t xcd = bid % 8, local = bid / 8;
  const int t = (int)(local % tiles_total);
  const int64_t split = (local / tiles_total) * 8 + xcd;
  if (split >= P.nsplit) return;
  const int tm = t / P.tiles_n, tn = t % P.tiles_n;
  const int64_t kt_all = P.n_mb / 2;
  const int64_t kt0 = split * P.kt_per_split;
  const int64_t kt1 = kt0 + P.kt_per_split < kt_all ? kt0 + P.kt_per_split : kt_all;
  if (kt0 >= kt1) return;
  const int nkt = (int)(kt1 - kt0);

  // piece p = NW q + wave of the k-tile: p < 4 TM: operand A, else operand B (p - 4 TM); inside an operand piece pp = (block column
  // cbi = pp >> 1, row block mbi = pp & 1): LDS [cbi][mbi][p][512] = byte pp * 1024 of the operand's part of the stage
  const unsigned char* base[NP];
  int64_t step[NP];
  uint32_t lds_off[NP];
  const int64_t step_a = 2 * (int64_t)P.ncb_a * 1024, step_b = 2 * (int64_t)P.ncb_b * 1024;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int pc = NW * q + wave;
    if (pc < NPA_ALL) {
      int cb = 2 * TM * tm + (pc >> 1);
      cb = cb < P.ncb_a ? cb : P.ncb_a - 1;                // block columns past the matrix: rows dropped by the epilogue
      base[q] = P.a + ((2 * kt0 + (pc & 1)) * P.ncb_a + cb) * 1024;
      step[q] = step_a;
      lds_off[q] = (uint32_t)pc * 1024u;
    } else {
      const int pb = pc - NPA_ALL;
      int cb = 2 * TN * tn + (pb >> 1);
      cb = cb < P.ncb_b ? cb : P.ncb_b - 1;
      base[q] = P.b + ((2 * kt0 + (pb & 1)) * P.ncb_b + cb) * 1024;
      step[q] = step_b;
      lds_off[q] = (uint32_t)A_ST + (uint32_t)pb * 1024u;
    }
  }
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto issue = [&](int rel, int stage, int c0, int c1) {
    const uint32_t sbase = smem_base + (uint32_t)stage * STAGE;
#pragma unroll
    for (int c = c0; c < c1; ++c)
      if (c < NP) glds16_saddr(base[c] + rel * step[c], lane16, sbase + lds_off[c]);
  };

  f32x4 acc[RB][TN];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  auto frag = [&](uint32_t blk0) -> bf16x8 {               // row blocks mbi = 0, 1 of one block column are 1 KiB apart
    typedef __attribute__((address_space(3))) wp_v4i16* lds_v4;
    const wp_v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(blk0 + lane_off));
    const wp_v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(blk0 + 1024u + lane_off));
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, (v8i16)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  issue(0, 0, 0, NP);
  issue(nkt > 1 ? 1 : 0, 1, 0, NP);
  int stage = 0;
  for (int it = 0; it < nkt; ++it) {
    wait_vmcnt<NP>();                                      // tile `it` landed for this wave (tile it + 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    const int nx = it + 2 < nkt ? it + 2 : nkt - 1;
    const int st2 = stage == 0 ? 2 : stage - 1;
    const uint32_t sa = smem_base + (uint32_t)stage * STAGE;
    const uint32_t sb = sa + A_ST;
    bf16x8 bh[TN], bl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const uint32_t c0 = sb + (uint32_t)(wn * TN + j) * 2048u;
      bh[j] = frag(c0);
      bl[j] = frag(c0 + 512u);
    }
    bf16x8 ah[2], al[2];
    {
      const uint32_t a0 = sa + (uint32_t)r0 * 2048u;
      ah[0] = frag(a0);
      al[0] = frag(a0 + 512u);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      // (a wave row with one row block fewer runs its last step without MFMAs -- wave-uniform -- but still issues its share of
      //  the DMA; the fragment it reads ahead then belongs to the next wave row: a valid address, never used)
      if (i + 1 < RB) {
        const uint32_t a0 = sa + (uint32_t)(r0 + i + 1 < 2 * TM ? r0 + i + 1 : 2 * TM - 1) * 2048u;
        ah[(i + 1) & 1] = frag(a0);
        al[(i + 1) & 1] = frag(a0 + 512u);
      }
      if (NW == 4 || i < nrows) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i & 1] : ah[i & 1], pass == 0 ? bl[j] : bh[j],
                                                               acc[i][j], 0, 0, 0);
      }
      if constexpr (NW == 4) {
        if (i + 1 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * TN, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      issue(nx, st2, i * PER_STEP, (i + 1) * PER_STEP);    // this k-tile's DMA issues spread over the row steps
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  wait_vmcnt<0>();

  // (a wave stores its own row blocks only: the row bound clips the accumulator rows past them)
  if (P.scratch != nullptr) {
    const EpiStore part{P.scratch + ((int64_t)split * tiles_total + t) * (32 * TM * 32 * TN), 32 * TN};
    store_accumulators<RB, TN>(part, acc, (int64_t)r0 * 16, 0, 0, wn, l15, g, (int64_t)(r0 + nrows) * 16, 32 * TN);
  } else {
    const int64_t m_tile = (int64_t)tm * (32 * TM);
    const int64_t m_end = m_tile + (int64_t)(r0 + nrows) * 16;
    store_accumulators<RB, TN>(epi, acc, m_tile + (int64_t)r0 * 16, tn * (32 * TN), 0, wn, l15, g, m_end < P.M ? m_end : P.M, P.N);
  }
}

static inline size_t wgrad_planes_g_scratch_floats(int TM, int TN, int ncb_a, int ncb_b, int nsplit) {
  return (size_t)nsplit * ((ncb_a + 2 * TM - 1) / (2 * TM)) * ((ncb_b + 2 * TN - 1) / (2 * TN)) * (32 * TM) * (32 * TN);
}
template <int TM, int TN, int NW = 4, class Epi>
static inline int launch_wgrad_planes_g(const void* a_planes, int ncb_a, const void* b_planes, int ncb_b, int64_t rows,
                                        int64_t m_valid, int n_valid, const Epi& epi, int nsplit, hipStream_t st,
                                        float* scratch = nullptr) {
  if (rows <= 0) return NRL_OK;
  NRL_REQUIRE(a_planes && b_planes && ncb_a > 0 && ncb_b > 0 && rows % 32 == 0, "wgrad_planes_g: bad arguments (rows % 32 == 0)");
  constexpr int LDS = 3 * 4 * (TM + TN) * 1024;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  WgradPlanesGArgs P;
  P.a = (const unsigned char*)a_planes; P.b = (const unsigned char*)b_planes;
  P.n_mb = rows / 16; P.ncb_a = ncb_a; P.ncb_b = ncb_b;
  P.tiles_m = (ncb_a + 2 * TM - 1) / (2 * TM); P.tiles_n = (ncb_b + 2 * TN - 1) / (2 * TN);
  const int64_t kt = rows / 32;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > kt) nsplit = (int)kt;
  P.kt_per_split = ceil_div(kt, nsplit);
  P.nsplit = (int)ceil_div(kt, P.kt_per_split);
  P.M = m_valid; P.N = n_valid;
  const int64_t blocks = (int64_t)P.tiles_m * P.tiles_n * ceil_div(P.nsplit, 8) * 8;
  NRL_REQUIRE(blocks < (1LL << 31), "wgrad_planes_g: grid too large");
  static std::atomic<uint64_t> attr_done{0};                // (per kernel instance and device; a racing second call is harmless)
  int dev = 0;
  NRL_HIP(hipGetDevice(&dev));
  if (!((attr_done.load(std::memory_order_relaxed) >> (dev & 63)) & 1u)) {
    NRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_g_kernel<TM, TN, Epi, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
  }
  P.scratch = P.nsplit > 1 ? scratch : nullptr;
  hipLaunchKernelGGL((wgrad_planes_g_kernel<TM, TN, Epi, NW>), dim3((unsigned)blocks), dim3(NW * 64), LDS, st, P, epi);
  NRL_LAUNCH_CHECK();
  if (P.scratch != nullptr) {
    const int tiles = P.tiles_m * P.tiles_n;
    const int64_t threads = (int64_t)tiles * (32 * TM) * (32 * TN) / 4;
    hipLaunchKernelGGL((wgrad_reduce_kernel<32 * TM, 32 * TN, Epi>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st,
                       P.scratch, P.nsplit, P.tiles_n, tiles, P.M, P.N, epi);
    NRL_LAUNCH_CHECK();
  }
  return NRL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Convolution weight gradient (Conv1d over the tokens of a news, window 3; CNNAddAtt text.py:169, LSTUR / NAML):
//
//     dWc[f][w * D + d] += sum over token rows m of  dc[m][f] * x[m + w - 1][d]       (x zero outside its news)
//     db_c[f]           += sum over m of dc[m][f]
//
// Both operands as fragment-block planes over PADDED rows: one news = 32 rows = one k-tile, rows L .. 31 zero on both
// sides (planes_from_rows_kernel below).  Then the three taps are three ROTATIONS of the k-tile's rows on the x side:
// the lane that fetches token row r for the A operand fetches row (r + w - 1) & 31 for the B operand -- row -1 is row
// 31 (zero), row L is zero, and whatever a zero dc row meets is finite.  `ds_read_b64_tr_b16` takes an address per
// lane, so a rotation is three precomputed lane offsets and nothing else: no halo, no masks, no VALU in the loop.
// One staged x block column feeds three output column groups (the taps): tile = (2 TM x 16) filters x (2 TN x 16)
// features x 3 taps; 4 waves (2 x 2), three LDS stages, split-K over news with the two-step reduction.
// The fp32-fed kernel it replaces (gemm_bf16x3_dma_tn<RCPlain, RCWindow>) re-split every operand element once per tile
// that used it and ran at 0.07 of the dense bf16 peak (2 x 0.97 ms per LSTUR step, profiles/r04_lstur_kernel_stats.txt).
struct EpiConvWB {
  float* dw;       // (F, W * D) row-major, tap-major columns (the C ABI's conv_weight layout)
  int64_t ldc;
  float* db;       // may be null
  int D;
  int tn16;        // 16 TN: feature columns of one wave and tap
  int center;      // tap whose ones column is the bias gradient
  struct Row {
    float* out;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{dw + m * ldc}; }
  // tile column c = (wave column wn) * 3 tn16 + tap * tn16 + feature-in-wave
  __device__ __forceinline__ void operator()(const Row& r, int64_t m, int n, float v) const {
    const int tw = 6 * tn16;
    const int tile = n / tw, c = n - tile * tw;
    const int wn = c / (3 * tn16), rem = c - wn * 3 * tn16;
    const int tap = rem / tn16;
    const int col = tile * 2 * tn16 + wn * tn16 + (rem - tap * tn16);
    if (col < D)
      atomicAdd(r.out + tap * D + col, v);
    else if (col == D && tap == center && db != nullptr)
      atomicAdd(db + m, v);
  }
};

// KT = k-tiles of 32 rows per news (1: L <= 31, three LDS stages of one news each; 2: L <= 63, two stages -- the same bytes
// in flight); the rotation is modulo 32 KT.
// NW (round 6): 4 = (2 x 2) waves, one per SIMD; 8 = (4 x 2) waves over the same tile and LDS stages, two per SIMD, the 2 TM row
// blocks dealt 3, 3, 2, 2 to the wave rows (waves w and w + 4 share a SIMD: TM row blocks per SIMD either way) -- as
// wgrad_planes_g_kernel above.
template <int TM, int TN, int KT, class Epi, int NW = 4>
__global__ void __launch_bounds__(NW * 64, 1) wgrad_planes_conv_kernel(const WgradPlanesGArgs P, const Epi epi) {
  static_assert(NW == 4 || NW == 8, "4 waves (2 x 2) or 8 waves (4 x 2)");
  constexpr int NS = 3, TNS = NS * TN;
  constexpr int A_ST = 4 * TM * KT * 1024, B_ST = 4 * TN * KT * 1024, STAGE = A_ST + B_ST;
  constexpr int STAGES = KT == 1 ? 3 : 2, PD = STAGES - 1;     // news in flight ahead of the one being multiplied
  constexpr int NPA_ALL = 4 * TM * KT, NP_ALL = 4 * (TM + TN) * KT;   // one-KiB pieces of a news (A first)
  constexpr int NP = (NP_ALL + NW - 1) / NW;                   // pieces per wave and news (a remainder is dealt as duplicates)
  constexpr int RQ = (2 * TM) / (NW / 2), RR = (2 * TM) % (NW / 2);
  constexpr int RB = NW == 4 ? TM : RQ + (RR > 0 ? 1 : 0);     // accumulator row blocks per wave
  constexpr int PER_STEP = (NP + RB * KT - 1) / (RB * KT);
  constexpr int ROWS = 32 * KT;
  extern __shared__ __attribute__((aligned(1024))) unsigned char wp_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int rb0 = NW == 4 ? wm * TM : (wm < RR ? wm * (RQ + 1) : RR * (RQ + 1) + (wm - RR) * RQ);
  const int nrows = NW == 4 ? TM : RQ + (wm < RR ? 1 : 0);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)wp_smem;

  const int tiles_total = P.tiles_m * P.tiles_n;
  const int64_t bid = blockIdx.x;
  const int64_t xcd = bid % 8, local = bid / 8;
  const int t = (int)(local % tiles_total);
  const int64_t split = (local / tiles_total) * 8 + xcd;
  if (split >= P.nsplit) return;
  const int tm = t / P.tiles_n, tn = t % P.tiles_n;
  const int64_t kt_all = P.n_mb / (2 * KT);                  // news
  const int64_t kt0 = split * P.kt_per_split;
  const int64_t kt1 = kt0 + P.kt_per_split < kt_all ? kt0 + P.kt_per_split : kt_all;
  if (kt0 >= kt1) return;
  const int nkt = (int)(kt1 - kt0);

  // piece pc = NW q + wave of the news (clamped: a duplicate rewrites the same bytes): pc < 4 TM KT: operand A, else operand B
  // (pc - 4 TM KT); inside an operand piece pp = (block column cbi = pp / (2 KT), row block mbi = pp % (2 KT)): LDS [cbi][mbi][p][512]
  const unsigned char* base[NP];
  int64_t step[NP];
  uint32_t lds_off[NP];
  const int64_t step_a = 2 * KT * (int64_t)P.ncb_a * 1024, step_b = 2 * KT * (int64_t)P.ncb_b * 1024;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    int pc = NW * q + wave;
    pc = pc < NP_ALL ? pc : NP_ALL - 1;
    if (pc < NPA_ALL) {
      int cb = 2 * TM * tm + pc / (2 * KT);
      cb = cb < P.ncb_a ? cb : P.ncb_a - 1;
      base[q] = P.a + ((2 * KT * kt0 + pc % (2 * KT)) * P.ncb_a + cb) * 1024;
      step[q] = step_a;
      lds_off[q] = (uint32_t)pc * 1024u;
    } else {
      const int pb = pc - NPA_ALL;
      int cb = 2 * TN * tn + pb / (2 * KT);
      cb = cb < P.ncb_b ? cb : P.ncb_b - 1;                  // block columns past the matrix: dropped by the epilogue (col > D)
      base[q] = P.b + ((2 * KT * kt0 + pb % (2 * KT)) * P.ncb_b + cb) * 1024;
      step[q] = step_b;
      lds_off[q] = (uint32_t)A_ST + (uint32_t)pb * 1024u;
    }
  }
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto issue = [&](int rel, int stage, int c0, int c1) {
    const uint32_t sbase = smem_base + (uint32_t)stage * STAGE;
#pragma unroll
    for (int c = c0; c < c1; ++c)
      if (c < NP) glds16_saddr(base[c] + rel * step[c], lane16, sbase + lds_off[c]);
  };

  f32x4 acc[RB][TNS];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < TNS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // token row r of a news sits at (r >> 4) * 1024 + (r & 15) * 32 of a (block column, plane); this lane's row of the first
  // half of k-tile u is 32 u + 4 g + (l15 >> 2), of the second half 16 more; tap w reads the row w - 1 further on, modulo ROWS
  const int r0 = 4 * g + (l15 >> 2);
  const uint32_t colb = (uint32_t)(l15 & 3) * 8u;
  auto row_addr = [&](int r) -> uint32_t { r &= ROWS - 1; return (uint32_t)((r >> 4) * 1024 + (r & 15) * 32) + colb; };
  uint32_t off[KT][NS][2];
#pragma unroll
  for (int u = 0; u < KT; ++u)
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      off[u][w][0] = row_addr(32 * u + r0 + w - 1);
      off[u][w][1] = row_addr(32 * u + r0 + 16 + w - 1);
    }
  auto frag2 = [&](uint32_t a0, uint32_t a1) -> bf16x8 {
    typedef __attribute__((address_space(3))) wp_v4i16* lds_v4;
    const wp_v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)a0);
    const wp_v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)a1);
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, (v8i16)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  constexpr uint32_t CBS = 2048u * KT;                       // one block column of a stage: 2 KT row blocks x (hi, lo)

#pragma unroll
  for (int p = 0; p < PD; ++p) issue(p < nkt ? p : nkt - 1, p, 0, NP);
  int stage = 0;
  for (int it = 0; it < nkt; ++it) {
    wait_vmcnt<(PD - 1) * NP>();                           // news `it` landed for this wave (PD - 1 later ones may be in flight)
    __builtin_amdgcn_s_barrier();
    const int nx = it + PD < nkt ? it + PD : nkt - 1;
    const int st2 = stage + PD >= STAGES ? stage + PD - STAGES : stage + PD;
    const uint32_t sa = smem_base + (uint32_t)stage * STAGE;
    const uint32_t sb = sa + A_ST;
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      bf16x8 bh[TNS], bl[TNS];
#pragma unroll
      for (int w = 0; w < NS; ++w)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const uint32_t c0 = sb + (uint32_t)(wn * TN + j) * CBS;
          bh[w * TN + j] = frag2(c0 + off[u][w][0], c0 + off[u][w][1]);
          bl[w * TN + j] = frag2(c0 + 512u + off[u][w][0], c0 + 512u + off[u][w][1]);
        }
      bf16x8 ah[2], al[2];
      {
        const uint32_t a0 = sa + (uint32_t)rb0 * CBS;
        ah[0] = frag2(a0 + off[u][1][0], a0 + off[u][1][1]);
        al[0] = frag2(a0 + 512u + off[u][1][0], a0 + 512u + off[u][1][1]);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (i + 1 < RB) {
          // (a wave row with one row block fewer reads ahead into the next wave row's block: a valid address, never multiplied)
          const uint32_t a0 = sa + (uint32_t)(rb0 + i + 1 < 2 * TM ? rb0 + i + 1 : 2 * TM - 1) * CBS;
          ah[(i + 1) & 1] = frag2(a0 + off[u][1][0], a0 + off[u][1][1]);
          al[(i + 1) & 1] = frag2(a0 + 512u + off[u][1][0], a0 + 512u + off[u][1][1]);
        }
        if (NW == 4 || i < nrows) {
#pragma unroll
          for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int j = 0; j < TNS; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i & 1] : ah[i & 1], pass == 0 ? bl[j] : bh[j],
                                                                 acc[i][j], 0, 0, 0);
        }
        if constexpr (NW == 4) {
          if (i + 1 < RB) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * TNS, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(nx, st2, (u * RB + i) * PER_STEP, (u * RB + i + 1) * PER_STEP);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stage = stage + 1 == STAGES ? 0 : stage + 1;
  }
  wait_vmcnt<0>();

  // (a wave stores its own row blocks only: the row bound clips the accumulator rows past them)
  if (P.scratch != nullptr) {
    const EpiStore part{P.scratch + ((int64_t)split * tiles_total + t) * (32 * TM * 32 * TNS), 32 * TNS};
    store_accumulators<RB, TNS>(part, acc, (int64_t)rb0 * 16, 0, 0, wn, l15, g, (int64_t)(rb0 + nrows) * 16, 32 * TNS);
  } else {
    const int64_t m_tile = (int64_t)tm * (32 * TM);
    const int64_t m_end = m_tile + (int64_t)(rb0 + nrows) * 16;
    store_accumulators<RB, TNS>(epi, acc, m_tile + (int64_t)rb0 * 16, tn * (32 * TNS), 0, wn, l15, g, m_end < P.M ? m_end : P.M, P.N);
  }
}

static inline size_t wgrad_planes_conv_scratch_floats(int TM, int TN, int ncb_a, int ncb_b, int nsplit) {
  return (size_t)nsplit * ((ncb_a + 2 * TM - 1) / (2 * TM)) * ((ncb_b + 2 * TN - 1) / (2 * TN)) * (32 * TM) * (96 * TN);
}
// a: dc planes (ncb_a block columns, F = m_valid filters), b: x planes (ncb_b block columns holding D features + the ones
// column); rows = padded rows (32 per news).  dw (F, 3 D), db (F).
template <int TM, int TN, int KT, int NW = 4>
static inline int launch_wgrad_planes_conv(const void* a_planes, int ncb_a, const void* b_planes, int ncb_b, int64_t rows,
                                           int64_t m_valid, int D, float* dw, float* db, int nsplit, hipStream_t st,
                                           float* scratch = nullptr) {
  if (rows <= 0) return NRL_OK;
  NRL_REQUIRE(a_planes && b_planes && ncb_a > 0 && ncb_b > 0 && rows % (32 * KT) == 0 && ncb_b * 16 > D,
              "wgrad_planes_conv: bad arguments (whole news of 32 KT rows, a ones column after the features)");
  typedef EpiConvWB Epi;
  constexpr int LDS = (KT == 1 ? 3 : 2) * 4 * (TM + TN) * KT * 1024;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  WgradPlanesGArgs P;
  P.a = (const unsigned char*)a_planes; P.b = (const unsigned char*)b_planes;
  P.n_mb = rows / 16; P.ncb_a = ncb_a; P.ncb_b = ncb_b;
  P.tiles_m = (ncb_a + 2 * TM - 1) / (2 * TM); P.tiles_n = (ncb_b + 2 * TN - 1) / (2 * TN);
  const int64_t kt = rows / (32 * KT);                       // news
  if (nsplit < 1) nsplit = 1;
  if (nsplit > kt) nsplit = (int)kt;
  P.kt_per_split = ceil_div(kt, nsplit);
  P.nsplit = (int)ceil_div(kt, P.kt_per_split);
  P.M = m_valid; P.N = P.tiles_n * 96 * TN;
  const Epi epi{dw, (int64_t)3 * D, db, D, 16 * TN, 1};
  const int64_t blocks = (int64_t)P.tiles_m * P.tiles_n * ceil_div(P.nsplit, 8) * 8;
  NRL_REQUIRE(blocks < (1LL << 31), "wgrad_planes_conv: grid too large");
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  NRL_HIP(hipGetDevice(&dev));
  if (!((attr_done.load(std::memory_order_relaxed) >> (dev & 63)) & 1u)) {
    NRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_conv_kernel<TM, TN, KT, Epi, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
  }
  P.scratch = P.nsplit > 1 ? scratch : nullptr;
  hipLaunchKernelGGL((wgrad_planes_conv_kernel<TM, TN, KT, Epi, NW>), dim3((unsigned)blocks), dim3(NW * 64), LDS, st, P, epi);
  NRL_LAUNCH_CHECK();
  if (P.scratch != nullptr) {
    const int tiles = P.tiles_m * P.tiles_n;
    const int64_t threads = (int64_t)tiles * (32 * TM) * (96 * TN) / 4;
    hipLaunchKernelGGL((wgrad_reduce_kernel<32 * TM, 96 * TN, Epi>), dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st,
                       P.scratch, P.nsplit, P.tiles_n, tiles, P.M, P.N, epi);
    NRL_LAUNCH_CHECK();
  }
  return NRL_OK;
}

// fp32 rows (n_news * L of them, `ld` floats apart, `ncols` valid columns, ncols % 4 == 0) -> (hi, lo) fragment-block planes
// over 16 nrb padded rows per news: block (mb = nrb * news + row / 16, cb) at (mb * ncb + cb) * 1024, hi plane first; rows >= L and
// columns >= ncols zero, except column `ncols` of the real rows when `ones` (the bias gradient's ones column).
struct PlanesFromRowsArgs {
  const float* src;
  int64_t ld, n_news;
  int L, ncols, ncb, ones;
  int nrb;                   // row blocks per news (2: 32 padded rows, 4: 64)
  unsigned char* dst;
};
static __global__ void __launch_bounds__(256) planes_from_rows_kernel(const PlanesFromRowsArgs P) {
  const int64_t news = blockIdx.x;
  const int items = P.nrb * P.ncb * 32;                      // (row block, block column, row, half row of 8 features)
  for (int it = threadIdx.x; it < items; it += 256) {
    const int half = it & 1, r16 = (it >> 1) & 15, rest = it >> 5;
    const int mbi = rest / P.ncb, cb = rest - mbi * P.ncb;
    const int r = 16 * mbi + r16, col0 = 16 * cb + 8 * half;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (r < P.L) {
      const float* row = P.src + (news * P.L + r) * P.ld + col0;
      if (col0 < P.ncols) v0 = *reinterpret_cast<const float4*>(row);
      if (col0 + 4 < P.ncols) v1 = *reinterpret_cast<const float4*>(row + 4);
      if (P.ones) {
        if (col0 == P.ncols) v0.x = 1.0f;
        if (col0 + 4 == P.ncols) v1.x = 1.0f;
      }
    }
    uint32_t h[4], l[4];
    split_pair(v0.x, v0.y, h[0], l[0]);
    split_pair(v0.z, v0.w, h[1], l[1]);
    split_pair(v1.x, v1.y, h[2], l[2]);
    split_pair(v1.z, v1.w, h[3], l[3]);
    unsigned char* dst = P.dst + ((P.nrb * news + mbi) * P.ncb + cb) * 1024 + r16 * 32 + half * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(dst + 512) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}
// x = dropout(E[ids]) written ONLY as those planes (round 5): the embedding lookup of the CNN text encoders whose convolution
// forward reads planes too (KCWindowPlanes, nrl_conv.h) -- no fp32 copy of x, no conversion launch in the backward.  Element (row m,
// column d) gets the keep multiplier of flat index m * D + d (the lookup's stream), the ones column sits at column D of the real rows.
struct EmbeddingPlanesArgs {
  const float* table;
  const int64_t* ids;
  int64_t n_news;
  int L, D, ncb, nrb;
  Dropout drop;
  unsigned char* dst;
};
static __global__ void __launch_bounds__(256) embedding_rows_planes_kernel(const EmbeddingPlanesArgs P) {
  const int64_t news = blockIdx.x;
  __shared__ int64_t s_id[64];
  if (threadIdx.x < 64) s_id[threadIdx.x] = (int)threadIdx.x < P.L ? P.ids[news * P.L + threadIdx.x] : 0;
  __syncthreads();
  const int items = P.nrb * P.ncb * 32;                      // (row block, block column, row, half row of 8 features)
  for (int it = threadIdx.x; it < items; it += 256) {
    const int half = it & 1, r16 = (it >> 1) & 15, rest = it >> 5;
    const int mbi = rest / P.ncb, cb = rest - mbi * P.ncb;
    const int r = 16 * mbi + r16, col0 = 16 * cb + 8 * half;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (r < P.L) {
      const float* row = P.table + s_id[r] * (int64_t)P.D + col0;
      const uint32_t idx = (uint32_t)(news * P.L + r) * (uint32_t)P.D + (uint32_t)col0;
      if (col0 < P.D) {
        v0 = *reinterpret_cast<const float4*>(row);
        if (P.drop.thresh != 0u) {
          v0.x *= P.drop.mult(idx);     v0.y *= P.drop.mult(idx + 1);
          v0.z *= P.drop.mult(idx + 2); v0.w *= P.drop.mult(idx + 3);
        }
      }
      if (col0 + 4 < P.D) {
        v1 = *reinterpret_cast<const float4*>(row + 4);
        if (P.drop.thresh != 0u) {
          v1.x *= P.drop.mult(idx + 4); v1.y *= P.drop.mult(idx + 5);
          v1.z *= P.drop.mult(idx + 6); v1.w *= P.drop.mult(idx + 7);
        }
      }
      if (col0 == P.D) v0.x = 1.0f;
      if (col0 + 4 == P.D) v1.x = 1.0f;
    }
    uint32_t h[4], l[4];
    split_pair(v0.x, v0.y, h[0], l[0]);
    split_pair(v0.z, v0.w, h[1], l[1]);
    split_pair(v1.x, v1.y, h[2], l[2]);
    split_pair(v1.z, v1.w, h[3], l[3]);
    unsigned char* dst = P.dst + ((P.nrb * news + mbi) * P.ncb + cb) * 1024 + r16 * 32 + half * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(dst + 512) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}
static inline int launch_embedding_rows_planes(const float* table, const int64_t* ids, int64_t n_news, int L, int D, int ncb, int nrb,
                                               const Dropout& drop, void* dst, hipStream_t st) {
  if (n_news <= 0) return NRL_OK;
  NRL_REQUIRE(table && ids && dst && L > 0 && L <= 16 * nrb && L <= 64 && D % 4 == 0 && ncb * 16 >= D + 1 &&
                  ((uintptr_t)table & 15) == 0 && n_news < (1LL << 31), "embedding_rows_planes: bad arguments");
  const EmbeddingPlanesArgs P{table, ids, n_news, L, D, ncb, nrb, drop, (unsigned char*)dst};
  hipLaunchKernelGGL(embedding_rows_planes_kernel, dim3((unsigned)n_news), dim3(256), 0, st, P);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// zero the pad rows (L <= r < 16 nrb) of every news of such planes: for a producer that writes only real rows (EpiPoolBwdNewsPlanes)
static __global__ void __launch_bounds__(256) planes_zero_pad_rows_kernel(unsigned char* dst, int L, int ncb, int nrb) {
  const int64_t news = blockIdx.x;
  const int pad = 16 * nrb - L, items = pad * ncb * 2 * 2;          // (pad row, block column, plane, 16-byte half)
  for (int it = threadIdx.x; it < items; it += 256) {
    const int half = it & 1, plane = (it >> 1) & 1, rest = it >> 2;
    const int cb = rest % ncb, r = L + rest / ncb;
    *reinterpret_cast<uint4*>(dst + ((nrb * news + (r >> 4)) * ncb + cb) * 1024 + plane * 512 + (r & 15) * 32 + half * 16) =
        make_uint4(0u, 0u, 0u, 0u);
  }
}
static inline int launch_planes_zero_pad_rows(void* dst, int64_t n_news, int L, int ncb, int nrb, hipStream_t st) {
  if (n_news <= 0 || L >= 16 * nrb) return NRL_OK;
  NRL_REQUIRE(dst && L > 0 && n_news < (1LL << 31), "planes_zero_pad_rows: bad arguments");
  hipLaunchKernelGGL(planes_zero_pad_rows_kernel, dim3((unsigned)n_news), dim3(256), 0, st, (unsigned char*)dst, L, ncb, nrb);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

static inline size_t planes_from_rows_bytes(int64_t n_news, int ncb, int nrb) { return (size_t)n_news * nrb * ncb * 1024; }
static inline int launch_planes_from_rows(const float* src, int64_t ld, int64_t n_news, int L, int ncols, int ncb, int nrb, bool ones,
                                          void* dst, hipStream_t st) {
  if (n_news <= 0) return NRL_OK;
  NRL_REQUIRE(src && dst && L > 0 && L <= 16 * nrb && ncols % 4 == 0 && ncb * 16 >= ncols + (ones ? 1 : 0) && (ld & 3) == 0 &&
                  ((uintptr_t)src & 15) == 0 && n_news < (1LL << 31),
              "planes_from_rows: bad arguments");
  const PlanesFromRowsArgs P{src, ld, n_news, L, ncols, ncb, ones ? 1 : 0, nrb, (unsigned char*)dst};
  hipLaunchKernelGGL(planes_from_rows_kernel, dim3((unsigned)n_news), dim3(256), 0, st, P);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
