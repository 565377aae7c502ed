// Weight gradient C (I x J) += A^T B from fp32 k-major operands, second form (gfx950):
//
// gemm_bf16x3_dma_tn_kernel (nrl_gemm_bf16x3_dma.h) lets every wave read its fragments k-strided out of the fp32 LDS
// tile (eight ds_read_b32 each) and split them itself: a fragment is split by every wave that uses it, and the four
// weight gradients of an NRMS step that run on it sit at 0.19 matrix-core occupancy.  Here the workgroup converts each
// fp32 k-tile ONCE, cooperatively, into (hi, lo) bf16 planes in LDS (row-major [k][rows], row stride an odd multiple
// of 32 bytes), and the MFMA operands come out of those planes with `ds_read_b64_tr_b16` -- the transposition is the
// read's, no VALU in the product phase (the scheme of nrl_wgrad_planes.h with the producer's split done in LDS):
//   DMA(t + 2)  ->  fp32 stage [t % 2]           global_load_lds_dwordx4, 1 KiB per wave-instruction, linear chunks
//   conv(t + 1):    fp32 stage -> planes [(t + 1) % 2]   9 float4 per thread at 128 x 160: read, split, two 8-byte writes
//   mma(t):         planes [t % 2]                       tr-read fragments + MFMA
// conv(t + 1) and mma(t) share an iteration (independent: VALU / LDS work under the MFMAs), one barrier per k-tile.
// Out-of-range rows / k tails / the ones column arrive as zeros / ones through the accessors' `src()` (as in the tn kernel).
#pragma once
#include "nrl_gemm_bf16x3_dma.h"

namespace nrl {

template <int ROWS>
struct Tn2Stride {   // bytes per k-row of a bf16 plane: ROWS * 2, padded so that (stride / 32) is odd
  static constexpr int value = ((ROWS * 2 / 32) % 2 == 1) ? ROWS * 2 : ROWS * 2 + 32;
};

template <int TM, int TN, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__(256, 1)
    gemm_bf16x3_tn2_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N, const int64_t K,
                           const int tiles_n, const int64_t tiles_total, const int64_t k_per_split, const int nsplit) {
  constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16, BK = 32;
  constexpr int CA = BM / 4, CB = BN / 4;                 // float4 chunks per k-row
  constexpr int FA = BK * BM * 4, FB = BK * BN * 4, FSTAGE = FA + FB;
  constexpr int SA = Tn2Stride<BM>::value, SB = Tn2Stride<BN>::value;
  constexpr int PA = BK * SA, PB = BK * SB, PSTAGE = 2 * PA + 2 * PB;   // [A hi][A lo][B hi][B lo]
  constexpr int PA_TOT = BK * CA / 64, PB_TOT = BK * CB / 64;
  static_assert((BK * CA) % 64 == 0 && (BK * CB) % 64 == 0, "whole 1-KiB DMA pieces");
  constexpr int GA = (PA_TOT + 3) / 4, GB = (PB_TOT + 3) / 4;
  static_assert(AOp::kLayout == SRC_RC && BOp::kLayout == SRC_RC, "both operands k-major fp32");
  static_assert(2 * (FSTAGE + PSTAGE) <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(1024))) unsigned char tn2_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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

  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tn2_smem;
  unsigned char* const fstage = tn2_smem;                       // 2 x FSTAGE
  unsigned char* const pstage = tn2_smem + 2 * FSTAGE;          // 2 x PSTAGE

  // ---- DMA: piece p = chunks 64 p .. 64 p + 63 of the row-major [k][chunk] tile -------------------------------
  int kra[GA], ca[GA], krb[GB], cb[GB];
  uint32_t da[GA], db[GB];
#pragma unroll
  for (int c = 0; c < GA; ++c) {
    int piece = wave + 4 * c;
    piece = piece < PA_TOT ? piece : PA_TOT - 1;
    const int ch = piece * 64 + lane;
    kra[c] = ch / CA;
    ca[c] = 4 * (ch % CA);
    da[c] = (uint32_t)piece * 1024u;
  }
#pragma unroll
  for (int c = 0; c < GB; ++c) {
    int piece = wave + 4 * c;
    piece = piece < PB_TOT ? piece : PB_TOT - 1;
    const int ch = piece * 64 + lane;
    krb[c] = ch / CB;
    cb[c] = 4 * (ch % CB);
    db[c] = (uint32_t)FA + (uint32_t)piece * 1024u;
  }
  auto issue = [&](int tile, int buf) {
    const int64_t k0 = kbeg + (int64_t)tile * BK;
    const uint32_t base = smem_base + (uint32_t)buf * (uint32_t)FSTAGE;
#pragma unroll
    for (int c = 0; c < GA; ++c) glds16_asm(A.src(k0 + kra[c], m0 + ca[c], kend), base + da[c]);
#pragma unroll
    for (int c = 0; c < GB; ++c) glds16_asm(B.src(k0 + krb[c], (int64_t)n0 + cb[c], kend), base + db[c]);
  };

  // ---- conversion: float4 f of the stage -> 8 bytes of the hi plane + 8 of the lo plane -------------------------
  auto convert = [&](int fbuf, int pbuf) {
    const unsigned char* fs = fstage + fbuf * FSTAGE;
    unsigned char* ps = pstage + pbuf * PSTAGE;
    constexpr int NA = BK * CA, NB = BK * CB;
#pragma unroll
    for (int q = 0; q < (NA + 255) / 256; ++q) {
      const int f = tid + 256 * q;
      if (NA % 256 == 0 || f < NA) {
        const float4 v = *reinterpret_cast<const float4*>(fs + f * 16);
        const int kr = f / CA, c4 = f % CA;
        uint32_t h0, l0, h1, l1;
        split_pair(v.x, v.y, h0, l0);
        split_pair(v.z, v.w, h1, l1);
        *reinterpret_cast<uint2*>(ps + kr * SA + c4 * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(ps + PA + kr * SA + c4 * 8) = make_uint2(l0, l1);
      }
    }
#pragma unroll
    for (int q = 0; q < (NB + 255) / 256; ++q) {
      const int f = tid + 256 * q;
      if (NB % 256 == 0 || f < NB) {
        const float4 v = *reinterpret_cast<const float4*>(fs + FA + f * 16);
        const int kr = f / CB, c4 = f % CB;
        uint32_t h0, l0, h1, l1;
        split_pair(v.x, v.y, h0, l0);
        split_pair(v.z, v.w, h1, l1);
        *reinterpret_cast<uint2*>(ps + 2 * PA + kr * SB + c4 * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(ps + 2 * PA + PB + kr * SB + c4 * 8) = make_uint2(l0, l1);
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read: lane (l15, g) addresses row 4g + (l15 >> 2) (+ 16 for the second half of the k-tile), 4 bf16 at column
  // 4 (l15 & 3) of the 16-column block, and receives rows 4g .. 4g + 3 of column l15
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef short v8i16 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) v4i16* lds_v4;
  const uint32_t rowa = (uint32_t)((4 * g + (l15 >> 2)) * SA + (l15 & 3) * 8);
  const uint32_t rowb = (uint32_t)((4 * g + (l15 >> 2)) * SB + (l15 & 3) * 8);
  auto frag = [&](uint32_t addr, uint32_t half_stride) -> bf16x8 {
    const v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)addr);
    const v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(addr + half_stride));
    return __builtin_bit_cast(bf16x8, (v8i16)__builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto mma = [&](int pbuf) {
    const uint32_t ps = smem_base + 2u * FSTAGE + (uint32_t)pbuf * PSTAGE;
    bf16x8 bh[TN], bl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const uint32_t a = ps + 2u * PA + rowb + (uint32_t)((wn * TN + j) * 32);
      bh[j] = frag(a, 16u * SB);
      bl[j] = frag(a + PB, 16u * SB);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const uint32_t a = ps + rowa + (uint32_t)((wm * TM + i) * 32);
      const bf16x8 ah = frag(a, 16u * SA), al = frag(a + PA, 16u * SA);
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al : ah, pass == 0 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
    }
  };

  issue(0, 0);
  if (ntiles > 1) issue(1, 1);
  if (ntiles > 1) wait_vmcnt<GA + GB>(); else wait_vmcnt<0>();
  __syncthreads();
  convert(0, 0);
  for (int tt = 0; tt < ntiles; ++tt) {
    wait_vmcnt<0>();                      // DMA(tt + 1), issued one iteration ago, has landed for this wave
    __syncthreads();                      // ... for all; planes[(tt + 1) & 1] and stage[tt & 1] are free, planes[tt & 1] complete
    if (tt + 2 < ntiles) issue(tt + 2, tt & 1);
    if (tt + 1 < ntiles) convert((tt + 1) & 1, (tt + 1) & 1);
    mma(tt & 1);
  }
  wait_vmcnt<0>();

  store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
}

template <int TM, int TN, class AOp, class BOp, class Epi>
int launch_gemm_bf16x3_tn2(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
                           hipStream_t stream) {
  constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
  constexpr int FSTAGE = 32 * (BM + BN) * 4, PSTAGE = 2 * 32 * (Tn2Stride<BM>::value + Tn2Stride<BN>::value);
  constexpr int LDS = 2 * (FSTAGE + PSTAGE);
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  int64_t kps = ceil_div(ceil_div(K, splits), 32) * 32;
  splits = (int)ceil_div(K, kps);
  const int64_t nblocks = splits > 1 ? ceil_div(splits, 8) * 8 * tiles_total : tiles_total;
  NRL_REQUIRE(nblocks < (1LL << 31), "gemm grid too large");
  static bool attr_done = false;
  if (!attr_done) {
    NRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_tn2_kernel<TM, TN, AOp, BOp, Epi>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_tn2_kernel<TM, TN, AOp, BOp, Epi>), dim3((unsigned)nblocks), dim3(256), LDS, stream, A, B, epi,
                     M, N, K, tiles_n, tiles_total, kps, splits);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
