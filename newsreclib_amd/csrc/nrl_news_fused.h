// Fused front half of the NRMS news encoder (gfx950):  token ids -> attention output `o`
//
//   x = dropout(E[ids])                         text.py:224-225   (bit-exact gather, counter-based mask)
//   q|k|v = x W_in^T + b_in                     text.py:229       (bf16x3 on the matrix cores, K = D (+ bias column))
//   o[:, head] = softmax(q k^T / sqrt(dh)) v    nn.MultiheadAttention, per (news, head), no mask
//
// One wavefront owns ONE news (L <= 32 tokens = 2 MFMA row blocks).  Its post-dropout embedding rows are
// gathered ONCE, split to (hi, lo) bf16 ONCE and kept in registers as MFMA A fragments (160 VGPRs at
// D <= 316) for all heads -- the tiled in-projection GEMM re-read and re-split every row for each of its
// six column tiles, wrote q|k|v (3.6 KB per token) to HBM and the attention kernel read it back.  Per head the
// 64 image columns [q 20 | k 20 | v 20 | 0] of W_in stream through a 5-slot LDS ring as fragment-ordered 16 KB
// chunks (rp_weight_image_kernel with the head remap; the bias rides in the image as column K = D against a
// ones column on the A side), the 32 x 64 result goes accumulators -> a private LDS image, and the 32 x 32
// attention of that head runs on the matrix cores straight from it:
//   S^T = K Q^T   (bf16x3; the accumulator layout of S^T -- lane (query, g) holds keys {4g.., 16+4g..} -- IS the
//                  A-operand layout of P V under the key permutation kappa(g, e), so P never leaves registers)
//   softmax over the keys of a query = 8 in-register values + two cross-lane steps (xor 16, 32)
//   O = P V       (bf16x3, V fragments read k-strided from the image)
// Only `o` (and, for training, q|k|v + log-sum-exp + x for the existing backward kernels) leaves the chip.
//
// Workgroup = 8 wavefronts = 8 news, two waves per SIMD (<= 256 VGPRs); LDS = 80 KB weight ring + 8 x 8.5 KB
// images.  Sync: one barrier per 16 KB chunk (slot reuse) + one `vmcnt(0)` per head; the DMA of head h + 1's
// chunk c is issued right after chunk c of head h has been consumed, so it has the rest of the head to land.
#pragma once
#include <type_traits>

#include "nrl_rowpanel.h"

namespace nrl {

constexpr int NF_WAVES = 8;        // news per workgroup
constexpr int NF_KB = 10;          // k-blocks of 32: D + 1 (bias column) <= 320
constexpr int NF_IMG_LD = 68;      // floats per row of the private q|k|v image (64 + pad: conflict-free C-layout stores)
constexpr int NF_IMG_FLOATS = 32 * NF_IMG_LD;
constexpr int NF_RING = 5 * 16384;  // one head = 5 chunks of (2 k-blocks x 4 column blocks x hi|lo) KiB

struct NewsFusedArgs {
  const float* table;       // (V, D)
  const int64_t* ids;       // (n_news, L)
  const uint16_t* img;      // rp image: nblk = heads * 4, kblocks = NF_KB, bias at k = D
  int64_t n_news;
  int L, D, heads, dh;      // dh == 20
  float scale;              // 1 / sqrt(dh)
  Dropout drop1;
  float* o;                 // (n_news * L, D)
  unsigned char* o_planes;  // or (instead of o): (hi, lo) bf16 fragment-block planes over the REAL rows m = news * L + t,
                            // [m / 16][cb 0 .. 18][p][16 x 16], feature order permuted so that a head writes whole block
                            // rows: block cb < heads = features 0 .. 15 of head cb, block heads + s = features 16 .. 19 of
                            // heads 4s .. 4s + 3 (4 columns each); the first free column (block 18, column 12 at 15 heads)
                            // carries the ones of the bias gradient (KCPlanesG / rp_jobs_add_kperm / EpiAtomicWBPerm)
  float* x_save;            // (n_news * L, D) post-dropout rows, or null
  unsigned char* x_planes;  // or (instead of x_save): the rows' (hi, lo) bf16 fragments as fragment-block planes
                            // [news * 2 + rb][cb 0 .. 19][p][16 x 16] (nrl_wgrad_planes.h) -- exactly the registers
                            // this kernel holds, so the in-projection weight gradient never splits or transposes x
  float* qkv_save;          // q|k|v (unscaled q) for the backward, or null (news_fused_bwd_kernel recomputes them):
                            //   qkv_head_major == 0: (n_news * L, 3D) packed rows (attn_bwd_small's layout)
                            //   qkv_head_major == 1: (n_news, heads, L, 64) = the private image rows [q | k | v | 0 4],
                            //                        one contiguous L x 256 B slab per (news, head) (news_attn_bwd_kernel)
  int qkv_head_major;
  float* lse;               // (n_news * heads, L) or null
  int full_wgs;             // workgroups [0, full_wgs) own 8 news each, the rest 4 (set by the launcher: tail balancing)
  // Evaluation only (no saves, no dropout): pad-row sharing.  `perm` lists the news with the SHORT ones first, `n_short` (device
  // scalar) says how many: a short news has token 15 and every later token equal to the padding id 0 (news_classify_kernel), so
  // without dropout its token rows 15 .. L - 1 are ONE row: identical embedding row -> identical q|k|v -> identical attention
  // output.  A wave that owns a short news computes row block 0 only and fills the keys / values of block 1 with copies of
  // row 15; `o` is written for rows 0 .. 15 only (the tail kernel, given the same lists, never reads the rest).  Both null: off.
  const int32_t* perm = nullptr;
  const int32_t* n_short = nullptr;
  // TBL (token q|k|v table, evaluation: nrl_token_table_build): `ids` is unused -- "news" n is the run of 32 vocabulary ids
  // 32 n .. 32 n + 31 (ids >= vocab: zero rows) and only the head-major q|k|v slabs are written (qkv_save = the table)
  int64_t vocab = 0;
};

// One launch that partitions the news of an evaluation call into short (token 15 and everything after it is the padding id;
// L >= 17) and long ones: perm[0 .. n_short) = the short news, perm[n_short .. n_news) = the long ones (filled from the back),
// hdr[0] = n_short, hdr[1] = n_long (both zeroed by the caller).  The order inside a class depends on the atomics' arrival
// order -- irrelevant: a news vector does not depend on which wave computes it.  One returning atomic per workgroup and class.
static __global__ void __launch_bounds__(256) news_classify_kernel(const int64_t* __restrict__ ids, int64_t n_news, int L,
                                                                    int32_t* __restrict__ hdr, int32_t* __restrict__ perm) {
  __shared__ int cnt[4][2];
  __shared__ int base[2];
  __shared__ int longf[256];            // news of this workgroup with a real token at position >= 15
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * 256;
  longf[tid] = 0;
  __syncthreads();
  // the workgroup's 256 x L ids are one contiguous run: coalesced loads, flag = any non-zero id at a token position >= 15
  const int64_t rows = n_news - n0 < 256 ? n_news - n0 : 256;
  const int total = (int)rows * L;                        // <= 256 * 32
  const int64_t* src = ids + n0 * L;
  // 32-bit indices and a multiply-shift division by L (exact for L <= 32, e < 2^13 with the 2^18 reciprocal rounded up: checked
  // exhaustively; a 64-bit e / L is a ~100-instruction software division per element)
  const unsigned rcp = ((1u << 18) + (unsigned)L - 1u) / (unsigned)L;
  for (int e = tid; e < total; e += 256) {
    const int nl = (int)(((unsigned)e * rcp) >> 18), t = e - nl * L;
    if (t >= 15 && src[e] != 0) longf[nl] = 1;          // (racing writers agree)
  }
  __syncthreads();
  const int64_t n = n0 + tid;
  const bool valid = n < n_news;
  const bool is_short = valid && L >= 17 && longf[tid] == 0;
  const unsigned long long bs = __ballot(valid && is_short), bl = __ballot(valid && !is_short);
  if (lane == 0) { cnt[wave][0] = __popcll(bs); cnt[wave][1] = __popcll(bl); }
  __syncthreads();
  if (tid < 2) {
    int tot = 0;
    for (int w = 0; w < 4; ++w) { const int c = cnt[w][tid]; cnt[w][tid] = tot; tot += c; }
    base[tid] = tot > 0 ? atomicAdd(hdr + tid, tot) : 0;
  }
  __syncthreads();
  if (valid) {
    const unsigned long long below = (1ull << lane) - 1ull;
    if (is_short) perm[base[0] + cnt[wave][0] + __popcll(bs & below)] = (int32_t)n;
    else perm[n_news - 1 - (base[1] + cnt[wave][1] + __popcll(bl & below))] = (int32_t)n;
  }
}
static inline int launch_news_classify(const int64_t* ids, int64_t n_news, int L, int32_t* hdr, int32_t* perm, hipStream_t st) {
  NRL_REQUIRE(n_news < (1LL << 31), "news_classify: too many news");
  NRL_HIP(hipMemsetAsync(hdr, 0, 2 * sizeof(int32_t), st));
  hipLaunchKernelGGL(news_classify_kernel, dim3((unsigned)ceil_div(n_news, 256)), dim3(256), 0, st, ids, n_news, L, hdr, perm);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__device__ __forceinline__ float nf_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
// lane ^ 16 / lane ^ 32 exchanges on the gfx950 row / half swaps: one VALU op + a select instead of a ds_bpermute
// round trip through the LDS crossbar (tools/nf_probe.hip: the softmax's four dependent exchanges per query block)
__device__ __forceinline__ float nf_xor16(float v, int lane) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (lane & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float nf_xor32(float v, int lane) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
}

// Branch-free fragment readers of the per-wave images (dh = 20).  Lanes whose features fall outside the head read a
// valid neighbouring address and are zeroed by their multiplier (hipcc turns `cond ? load : 0` into exec-masked
// branches: 47 of them in the attention backward, each with its own waitcnt drain).
//   nf_frag8: features 8g .. 8g + 7 of one row (rowp = row base + first column of the part), times mul
__device__ __forceinline__ void nf_frag8(const float* rowp, int g, float mul, bf16x8& hi, bf16x8& lo) {
  const int c = g < 3 ? 8 * g : 0;
  float4 v0 = *reinterpret_cast<const float4*>(rowp + c);
  float4 v1 = *reinterpret_cast<const float4*>(rowp + c + 4);
  const float m0 = g < 3 ? mul : 0.f, m1 = g < 2 ? mul : 0.f;
  v0.x *= m0; v0.y *= m0; v0.z *= m0; v0.w *= m0;
  v1.x *= m1; v1.y *= m1; v1.z *= m1; v1.w *= m1;
  rp_split8(v0, v1, hi, lo);
}
//   nf_kfrag: B operand of a product over rows -- lane (d, g) <- rows kappa(g, e), e = 0..7, of column d
//   (colp = image + first column of the part + d; mk = d < dh ? mul : 0)
__device__ __forceinline__ void nf_kfrag(const float* colp, int ld, int g, float mk, bf16x8& hi, bf16x8& lo) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int rowk = (q < 4) ? 4 * g + q : 16 + 4 * g + (q - 4);
    v[q] = colp[rowk * ld] * mk;
  }
  rp_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
}

// ---- (hi, lo) bf16 planes of one (news, head) in LDS ("nap"), news_attn_bwd_p_kernel ------------------------------------
//   [part q|k|v][row block 0|1][plane hi|lo] = 12 planes of 640 bytes; a plane holds 16 token rows x 20 features CHUNK-major:
//     [features 0..7: 16 rows x 16 B][features 8..15: 16 rows x 16 B][features 16..19: 16 rows x 8 B]
//   = what the lanes g = 0, 1, 2 of a row-form fragment hold, each piece 16- (8-) byte aligned.
constexpr int NAP_PLANE = 640, NAP_SLAB = 12 * NAP_PLANE;
__device__ __forceinline__ constexpr int nap_off(int part, int rb, int plane) { return ((part * 2 + rb) * 2 + plane) * NAP_PLANE; }
// byte offset inside a plane of features 4q .. 4q + 3 (q = 0 .. 4) of token row r
__device__ __forceinline__ int nap_quad(int r, int q) { return q < 4 ? (q >> 1) * 256 + r * 16 + (q & 1) * 8 : 512 + r * 8; }
typedef uint2 __attribute__((may_alias)) nap_u2;           // the LDS planes are written as pairs, read as pairs / quads, reused as floats
typedef uint4 __attribute__((may_alias)) nap_u4a;

// ABL (tools/nf_probe.hip only; product code uses 0): 1 = no attention phase, 2 = no in-projection MFMAs,
// 4 = no weight DMA, 8 = no o / q|k|v / lse stores, 16 = streaming saves, 32 = streaming `o` stores,
// 64 = counted head-top wait (vmcnt(8): the q|k|v slab stores stay in flight), 128 = every slab store of a wave lands on
// the same 7.5 KB (cache-resident: the stores are issued, nothing drains to HBM), 256 = slab stores in front of the attention phase instead of after the score MFMAs,
// 512 = half of them there, half after the P V MFMAs
// WV (round 5): waves = news per workgroup.  8: one workgroup per CU, the whole head's image (5 chunks, 80 KB) resident one head
// ahead.  4: TWO workgroups per CU (2 x 16 KB ring + 4 images = 66 KB each), chunks one ahead -- the two waves of a SIMD then
// belong to different workgroups and share no barrier, so one's softmax / store phase runs under the other's MFMAs (what took the
// tail kernels from 0.25 to 0.20 ms, nrl_news_tail.h); every weight byte is streamed L2 -> LDS once per 4 news instead of once per 8.
// TBL (round 6): the in-projection of the VOCABULARY instead of a batch -- no dropout, no attention, nothing written but the q|k|v
// slabs.  The accumulators of a token row are produced by the same MFMAs in the same order as in every other shape of this
// kernel (an MFMA's output row depends on its own A row only), so a table row is bit for bit what an evaluation forward computes
// for every position that holds this id; news_tab_attn_fwd_kernel below gathers them.
template <int DH, bool SAVE, int ABL = 0, bool SHARE = false, int WV = NF_WAVES, bool TBL = false>
__global__ void __launch_bounds__(WV * 64, 2) news_fused_fwd_kernel(const NewsFusedArgs P) {
  static_assert(!TBL || (SAVE && !SHARE && ABL == 0), "table build: the slab-saving shape, no sharing, no probe switches");
  static_assert(DH == 20, "image packing below assumes 3 * dh <= 64 with dh = 20");
  static_assert(WV == 8 || WV == 4, "8 waves (whole-head ring) or 4 waves (two-slot ring)");
  constexpr int RING = WV == 8 ? NF_RING : 2 * 16384;
  constexpr int PPW = 16 / WV;                          // 1 KiB DMA pieces per wave and chunk
  static_assert(!(SHARE && SAVE), "pad-row sharing is for evaluation forwards (no dropout, nothing saved)");
  // plain (write-allocate) stores: this kernel is not store-bound and the L2 merges the 80-byte `o` pieces; streaming
  // saves measured slower (train 0.62-0.65 vs 0.60 ms, tools/nf_probe.hip).  ABL 16 / 32 select the streaming forms.
  constexpr int NT = (ABL & 16) ? 1 : 0;
  constexpr int NT_O = (ABL & 32) ? 1 : 0;
  // (+ 16 bytes: the masked lanes of the last wave's V-fragment reads run 3 floats past its image)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[RING + WV * NF_IMG_FLOATS * 4 + 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* const image = reinterpret_cast<float*>(smem + RING) + wave * NF_IMG_FLOATS;

  const int L = P.L, D = P.D, heads = P.heads;
  const int nblk = heads * 4;
  // Tail balancing: a workgroup costs the same with 1 or 8 news (every wave streams all the weights), so the last,
  // partly filled round of 8-news workgroups (7040 news = 3.44 rounds of 256) is replaced by 4-news workgroups whose
  // waves have a SIMD to themselves; their other four waves only keep the barrier / DMA protocol going.
  const bool half_wg = WV == 8 && (int)blockIdx.x >= P.full_wgs;
  const int64_t news_v = half_wg ? (int64_t)P.full_wgs * NF_WAVES + (int64_t)((int)blockIdx.x - P.full_wgs) * 4 + wave
                                 : (int64_t)blockIdx.x * WV + wave;
  const bool wave_active = !half_wg || wave < 4;
  const bool news_ok = wave_active && news_v < P.n_news;
  // SHARE: the wave's news comes from the short-first list; position < n_short <=> a short news (wave-uniform)
  int64_t news = news_v;
  bool short_w = false;
  if constexpr (SHARE) {
    if (news_ok) {
      news = __builtin_amdgcn_readfirstlane(P.perm[news_v]);
      short_w = news_v < (int64_t)__builtin_amdgcn_readfirstlane(*P.n_short);
    }
  }
  const int Lw = short_w ? 16 : L;                      // token rows this wave computes and writes
  const int64_t row0 = (news_ok ? news : 0) * L;        // first token row of this wave's news

  // ---- weight DMA: chunk c of head h = k-blocks 2c, 2c + 1 x column blocks 4h .. 4h + 3 x (hi, lo): 16 pieces of
  // 1 KiB, two per wave.  LDS slot c holds [kbi][nb][plane][lane * 16].
  // (WV == 8: chunk c lives in slot c; WV == 4: chunk c of head h lives in slot (h + c) & 1 -- five chunks per head)
  auto issue_chunk = [&](int h, int c, int slot_i = -1) {
    const uint32_t slot = slot_i >= 0 ? (uint32_t)slot_i : (WV == 8 ? (uint32_t)c : (uint32_t)((h + c) & 1));
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave * PPW + q;                  // 0..15 = (kbi, nb, plane)
      const int kbi = piece >> 3, nb = (piece >> 1) & 3, plane = piece & 1;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img) +
                                 ((size_t)(((2 * c + kbi) * nblk + h * 4 + nb) * 2 + plane)) * 1024 + lane * 16;
      glds16_asm(src, smem_base + slot * 16384u + (uint32_t)piece * 1024u);
    }
  };
  if constexpr (!(ABL & 4)) {
#pragma unroll
    for (int c = 0; c < (WV == 8 ? 5 : 2); ++c) issue_chunk(0, c);
  }
  if (WV == 8 && !wave_active) {
    // idle wave of a 4-news workgroup: same barriers, same share of the weight DMA, nothing else
    for (int h = 0; h < heads; ++h) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const int hn = h + 1 < heads ? h + 1 : h;
      for (int c = 0; c < 5; ++c) {
        __builtin_amdgcn_s_barrier();
        if constexpr (!(ABL & 4)) issue_chunk(hn, c);
      }
    }
    wait_vmcnt<0>();
    return;
  }

  // ---- gather + dropout + split: the A fragments of this news, resident for all heads ------------------
  // All 40 row loads of a lane are issued before the first is consumed (branch-free: counted waits, one memory
  // latency per news instead of one per k-block); the raw registers turn into the fragments in place.
  bf16x8 ah[2][NF_KB], al[2][NF_KB];
  {
    float4 raw[2][NF_KB][2];
    bool okr[2];
    int64_t growr[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int t = rb * 16 + l15;
      okr[rb] = news_ok && t < L;
      if constexpr (TBL) okr[rb] = okr[rb] && row0 + t < P.vocab;   // (L == 32: row0 + t is the vocabulary id itself)
      growr[rb] = row0 + (okr[rb] ? t : 0);
      if (SHARE && rb == 1 && short_w) continue;          // (row block 1 of a short news is never computed)
      const float* rowp = P.table + (TBL ? (okr[rb] ? growr[rb] : 0) : P.ids[growr[rb]]) * (int64_t)D;
#pragma unroll
      for (int kb = 0; kb < NF_KB; ++kb) {
        const int k = kb * 32 + 8 * g;
        // k-blocks 0 .. NF_KB - 2 lie inside every supported D (>= 288); only the last one needs clamping
        const int k0 = (kb < NF_KB - 1 || k < D) ? k : D - 4;
        const int k1 = (kb < NF_KB - 1 || k + 4 < D) ? k + 4 : D - 4;
        raw[rb][kb][0] = *reinterpret_cast<const float4*>(rowp + k0);
        raw[rb][kb][1] = *reinterpret_cast<const float4*>(rowp + k1);
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if (SHARE && rb == 1 && short_w) {
#pragma unroll
        for (int kb = 0; kb < NF_KB; ++kb) ah[1][kb] = al[1][kb] = bf16x8{};
        continue;
      }
      const uint32_t idx0 = (uint32_t)growr[rb] * (uint32_t)D;
      const float live = okr[rb] ? 1.0f : 0.0f;
#pragma unroll
      for (int kb = 0; kb < NF_KB; ++kb) {
        const int k = kb * 32 + 8 * g;
        float4 v0 = raw[rb][kb][0], v1 = raw[rb][kb][1];
        const bool in0 = kb < NF_KB - 1 || k < D, in1 = kb < NF_KB - 1 || k + 4 < D;
        const float m0 = in0 ? live : 0.0f, m1 = in1 ? live : 0.0f;
        const uint32_t idx = idx0 + (uint32_t)k;
        // (p == 0: thresh 0, scale 1 -> every multiplier is 1; no branch on it)
        v0.x *= m0 * P.drop1.mult(idx);     v0.y *= m0 * P.drop1.mult(idx + 1);
        v0.z *= m0 * P.drop1.mult(idx + 2); v0.w *= m0 * P.drop1.mult(idx + 3);
        v1.x *= m1 * P.drop1.mult(idx + 4); v1.y *= m1 * P.drop1.mult(idx + 5);
        v1.z *= m1 * P.drop1.mult(idx + 6); v1.w *= m1 * P.drop1.mult(idx + 7);
        if constexpr (SAVE && !TBL) {
          if (P.x_planes == nullptr && okr[rb]) {
            if (in0) store4(P.x_save + growr[rb] * D + k, v0, NT);
            if (in1) store4(P.x_save + growr[rb] * D + k + 4, v1, NT);
          }
        }
        // ones column at k == D: the bias row of the image is added by the matrix cores (D % 4 == 0)
        if (kb == NF_KB - 1) {
          if (k == D) v0.x = 1.0f;
          if (k + 4 == D) v1.x = 1.0f;
        }
        rp_split8(v0, v1, ah[rb][kb], al[rb][kb]);
        if constexpr (SAVE && !TBL) {
          if (P.x_planes != nullptr && news_ok) {
            // block (mb = 2 news + rb, cb = 2 kb + (g >> 1)): row l15, columns 8 (g & 1) .. + 7 (pad rows: zeros + the ones column)
            unsigned char* dst = P.x_planes + (((news * 2 + rb) * 20 + 2 * kb + (g >> 1)) * 2) * 512 + l15 * 32 + (g & 1) * 16;
            *reinterpret_cast<bf16x8*>(dst) = ah[rb][kb];
            *reinterpret_cast<bf16x8*>(dst + 512) = al[rb][kb];
          }
        }
      }
    }
  }

  // image (q columns = O, column 60 = log-sum-exp of the query) of head `hp` -> global, 16-byte row stores
  auto flush_o = [&](int hp) {
    if (TBL || !news_ok || (ABL & 8)) return;
    // (the lane id is made opaque here: hipcc otherwise hoists every pass's address arithmetic out of the head
    //  loop and spills it -- the MFMA phase has no registers to spare)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if (P.o_planes != nullptr) {
      // main block of the head: 32 rows x 2 halves of 8 features = 64 lanes, one 16-byte chunk per plane each
      {
        const int row = ln >> 1, half = ln & 1;
        const int64_t m = row0 + row;
        bf16x8 hi, lo;
        rp_split8(*reinterpret_cast<const float4*>(image + row * NF_IMG_LD + half * 8),
                  *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + half * 8 + 4), hi, lo);
        unsigned char* dst = P.o_planes + (((m >> 4) * 19 + hp) * 2) * 512 + (m & 15) * 32 + half * 16;
        if (row < Lw) {
          *reinterpret_cast<bf16x8*>(dst) = hi;
          *reinterpret_cast<bf16x8*>(dst + 512) = lo;
        }
      }
      // features 16 .. 19 -> 4 columns of the block shared by four heads (+ the ones column after the last head)
      {
        const int row = ln & 31;
        const int64_t m = row0 + row;
        const float4 v = *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + 16);
        uint32_t h0, l0, h1, l1;
        split_pair(v.x, v.y, h0, l0);
        split_pair(v.z, v.w, h1, l1);
        unsigned char* blk = P.o_planes + (((m >> 4) * 19 + heads + (hp >> 2)) * 2) * 512 + (m & 15) * 32;
        if (row < Lw) {
          if (ln < 32) {
            *reinterpret_cast<uint2*>(blk + (hp & 3) * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(blk + 512 + (hp & 3) * 8) = make_uint2(l0, l1);
          } else if (hp == heads - 1 && (heads & 3) != 0) {
            // free columns of the last shared block: 1.0 (bf16 0x3F80) in the first, zeros after it
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const int slot = (heads & 3) + q;
              if (slot < 4) {
                *reinterpret_cast<uint2*>(blk + slot * 8) = make_uint2(q == 0 ? 0x3F80u : 0u, 0u);
                *reinterpret_cast<uint2*>(blk + 512 + slot * 8) = make_uint2(0u, 0u);
              }
            }
          }
        }
      }
    } else {
      float* o_out = P.o + row0 * (int64_t)D;             // wave-uniform base + 32-bit lane offsets
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {               // 32 rows x 5 float4 = 160 slots
        const int slot_i = pass * 64 + ln;
        const int row = slot_i / 5, c4 = slot_i - row * 5;
        if (slot_i < 160 && row < Lw)
          store4(o_out + (row * D + hp * DH + 4 * c4), *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + 4 * c4), NT_O);
      }
    }
    if (SAVE && ln < L) P.lse[(news * heads + hp) * L + ln] = image[ln * NF_IMG_LD + 60];
  };

  // (Probe only, ABL 64: a COUNTED head-top wait.  The vector-memory counter retires in issue order and after the last weight
  //  DMA of the next head (chunk 4, step 18) a wave issues exactly the eight 16-byte stores of its q|k|v slab (L >= 29: every
  //  pass has live lanes), so vmcnt(8) is enough for the weights and lets the slab stores fly through the next head.  Measured:
  //  0.643 vs 0.645 ms, bit-identical output -- the stores' latency is not what the kernel waits for; its 1.49 GB of output
  //  drain at the ~3 TB/s a larger-than-cache write stream gets (profiles/r01_store_probe.txt).  Product code waits vmcnt(0).)
  const bool slab_tail = (ABL & 64) && SAVE && !(ABL & 8) && P.qkv_save != nullptr && P.qkv_head_major && news_ok && L >= 29;
  // The head loop exists in two compile-time shapes: NI = 2 row blocks (every training wave, the long news of an evaluation
  // call) and NI = 1 (SHARE: a short news -- half the in-projection MFMAs, half the score / P V MFMAs, one softmax block).
  // Both execute the same barriers, so waves of either shape can share a workgroup (the short-first list makes that rare).
  auto run_heads = [&](auto sh_c) {
    constexpr bool SH = decltype(sh_c)::value;
    constexpr int NI = SH ? 1 : 2;
    for (int h = 0; h < heads; ++h) {
      // every chunk of head h was issued while head h - 1 ran: landed for this wave, then (barrier) for all
      // (WV == 4: chunk 0 of this head was waited for at the last chunk boundary of the previous head; only head 0 waits here)
      if (WV == 8 || h == 0) {
        if (h > 0 && slab_tail) wait_vmcnt<8>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
      const unsigned char* const slot_even = smem + (WV == 8 ? 0 : (h & 1) * 16384);        // WV == 4: chunks 0, 2, 4 of this head
      const unsigned char* const slot_odd = smem + (WV == 8 ? 0 : ((h & 1) ^ 1) * 16384);   //          chunks 1, 3
      f32x4 acc[2][4];
  #pragma unroll
      for (int i = 0; i < 2; ++i)
  #pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int hn = h + 1 < heads ? h + 1 : h;            // last head: re-issue its own chunks (uniform control flow)
      // 20 steps per head = (chunk c, k-block kbi, column-block pair np); the B fragments of step t + 1 are fetched
      // before the 12 MFMAs of step t (two named fragment sets -- hipcc alone fetches each pair right before its
      // first MFMA and waits lgkmcnt(0) there, exposing the LDS latency every 2-4 MFMAs).  Reading ahead across a
      // chunk boundary is safe: the barrier there only guards the REFILL of the slot just finished.
      auto read_step = [&](int t, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
        const int c = t >> 2, kbi = (t >> 1) & 1, np = t & 1;
        const unsigned char* slot = WV == 8 ? smem + c * 16384 + lane * 16 : ((c & 1) ? slot_odd : slot_even) + lane * 16;
  #pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          bh[jj] = *reinterpret_cast<const bf16x8*>(slot + ((kbi * 4 + 2 * np + jj) * 2) * 1024);
          bl[jj] = *reinterpret_cast<const bf16x8*>(slot + ((kbi * 4 + 2 * np + jj) * 2 + 1) * 1024);
        }
      };
      auto mfma_step = [&](int t, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
        const int kb = t >> 1, np = t & 1;
        if constexpr (ABL & 2) {
  #pragma unroll
          for (int jj = 0; jj < 2; ++jj) asm volatile("" ::"v"(bh[jj]), "v"(bl[jj]));
          return;
        }
  #pragma unroll
        for (int pass = 0; pass < 3; ++pass)
  #pragma unroll
          for (int jj = 0; jj < 2; ++jj)
  #pragma unroll
            for (int i = 0; i < NI; ++i)
              acc[i][2 * np + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  pass == 1 ? al[i][kb] : ah[i][kb], pass == 0 ? bl[jj] : bh[jj], acc[i][2 * np + jj], 0, 0, 0);
      };
      bf16x8 bh0[2], bl0[2], bh1[2], bl1[2];
      read_step(0, bh0, bl0);
  #pragma unroll
      for (int t = 0; t < 20; t += 2) {
        read_step(t + 1, bh1, bl1);
        mfma_step(t, bh0, bl0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NI, 0);
        // (WV == 4: the next chunk may still be in flight at a chunk boundary -- its fragments are read after the boundary's wait)
        const bool ahead = t + 2 < 20 && (WV == 8 || (t & 3) != 2);
        if (ahead) read_step(t + 2, bh0, bl0);
        mfma_step(t + 1, bh1, bl1);
        if (ahead) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NI, 0);
        if ((t & 3) == 2) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (WV == 8) {
            // steps 4c .. 4c + 3 done: all waves are finished with slot c -> refill it with chunk c of the next head
            __builtin_amdgcn_s_barrier();
            if constexpr (!(ABL & 4)) issue_chunk(hn, t >> 2);
          } else {
            // two-slot ring: chunk c + 1 (chunk 0 of the next head after c = 4) was issued one chunk ago into the other slot: this
            // wave's pieces have landed (vmcnt), then -- barrier -- everybody's, and everybody is done with chunk c's slot, which
            // takes chunk c + 2 (of the next head past the end of this one; last head: its own chunks again, uniform control flow)
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const int c2 = (t >> 2) + 2;
            if constexpr (!(ABL & 4)) {
              // (the slot is the one chunk c just left, whichever head the refill belongs to: (h + c) & 1)
              if (c2 < 5) issue_chunk(h, c2, (h + c2) & 1);
              else issue_chunk(hn, c2 - 5, (h + c2) & 1);
            }
            if (t + 2 < 20) read_step(t + 2, bh0, bl0);
          }
          if (t == 2 && h > 0) flush_o(h - 1);              // the previous head's O, under this head's MFMAs
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // ---- accumulators -> private image [token][q 20 | k 20 | v 20 | 0 4] -----------------------------
  #pragma unroll
      for (int i = 0; i < NI; ++i)
  #pragma unroll
        for (int nb = 0; nb < 4; ++nb)
  #pragma unroll
          for (int r = 0; r < 4; ++r) image[(i * 16 + 4 * g + r) * NF_IMG_LD + nb * 16 + l15] = acc[i][nb][r];
      if constexpr (SH) {
        // short news: token rows 16 .. L - 1 are the row 15 (all of them the padding id, no dropout): its q|k|v -- held by the
        // lanes g = 3 as element 3 -- are copied into the image rows of block 1, where the keys / values of the attention read them
  #pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const float v15 = __shfl(acc[0][nb][3], 48 + l15, 64);
  #pragma unroll
          for (int r = 0; r < 4; ++r) image[(16 + 4 * g + r) * NF_IMG_LD + nb * 16 + l15] = v15;
        }
      }

      auto save_qkv = [&](int pass_lo, int pass_hi) {
      // ---- save q|k|v for the backward kernels: (row, 3D) layout, q at head*dh, k at D + .., v at 2D + .. ---
        // (the output base pointers are made opaque per head: hipcc otherwise hoists the per-pass store addresses out
        //  of the head loop and spills them)
        // wave-uniform bases + 32-bit lane offsets (scalar-base addressing)
        float* qkv_out = P.qkv_save + row0 * (int64_t)(3 * D);
        if (P.qkv_head_major) qkv_out = P.qkv_save + ((news_ok ? news : 0) * heads + h) * (int64_t)L * 64;
        if constexpr (ABL & 128) qkv_out = P.qkv_save + ((int64_t)(blockIdx.x & 255) * NF_WAVES + wave) * (int64_t)L * 64;
        asm volatile("" : "+s"(qkv_out));
        if (SAVE && P.qkv_save != nullptr && news_ok && !(ABL & 8)) {
          int ln = lane;
          asm volatile("" : "+v"(ln));
          if (P.qkv_head_major) {
            // whole image rows: 1 KiB per pass, every 128-byte line written in full (the packed-row form below writes
            // 80-byte pieces at a 3.6 KB stride)
  #pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
              if (pass < pass_lo || pass >= pass_hi) continue;
              const int slot_i = pass * 64 + ln;
              const int row = slot_i >> 4, ch = slot_i & 15;
              if (row < L)
                store4(qkv_out + 4 * slot_i, *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + 4 * ch), NT);
            }
          } else {
  #pragma unroll
            for (int pass = 0; pass < 8; ++pass) {            // 32 rows x 15 float4 (3 parts x 5) -> 480 of 512 slots
              if (pass < pass_lo || pass >= pass_hi) continue;
              const int slot_i = pass * 64 + ln;
              const int row = slot_i >> 4, ch = slot_i & 15;
              if (ch < 15 && row < L) {
                const int part = ch / 5, c4 = ch - part * 5;
                const float4 v = *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + part * DH + 4 * c4);
                *reinterpret_cast<float4*>(qkv_out + (row * 3 * D + part * D + h * DH + 4 * c4)) = v;
              }
            }
          }
        }
      };
      if constexpr (ABL & 256) save_qkv(0, 8);                 // (probe: the stores in front of the attention phase)
      if constexpr (TBL) {                                     // table build: the slab is the result
        save_qkv(0, 8);
        continue;
      }

      if constexpr (ABL & 1) continue;
      // ---- S^T = K Q^T on the matrix cores ---------------------------------------------------------------
      bf16x8 kh[2], kl[2], qh[2], ql[2];
  #pragma unroll
      for (int b = 0; b < 2; ++b) {
        nf_frag8(image + (b * 16 + l15) * NF_IMG_LD + DH, g, 1.0f, kh[b], kl[b]);
        if (b < NI) nf_frag8(image + (b * 16 + l15) * NF_IMG_LD, g, P.scale, qh[b], ql[b]);   // (short: the queries of block 0 only)
      }
      f32x4 s[2][2];                                         // [key block][query block]
  #pragma unroll
      for (int jb = 0; jb < 2; ++jb)
  #pragma unroll
        for (int ib = 0; ib < 2; ++ib) s[jb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int pass = 0; pass < 3; ++pass)
  #pragma unroll
        for (int jb = 0; jb < 2; ++jb)
  #pragma unroll
          for (int ib = 0; ib < NI; ++ib)
            s[jb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? kl[jb] : kh[jb], pass == 0 ? ql[ib] : qh[ib],
                                                                s[jb][ib], 0, 0, 0);
      // the slab stores go out under the latency of the score MFMAs (in front of the attention phase they cost 8-15 us more per
      // launch at B = 128, tools/nf_probe.hip; the q columns are overwritten by O only at the end of the phase)
      if constexpr (!(ABL & (256 | 512))) save_qkv(0, 8);
      if constexpr (ABL & 512) save_qkv(0, 4);                // (probe: half here, half under the P V MFMAs)
      // lane (query = ib * 16 + l15, g) holds keys jb * 16 + 4g + r: softmax over all L keys of the query
      bf16x8 ph[2], pl[2];
  #pragma unroll
      for (int ib = 0; ib < NI; ++ib) {
        float e[8];
        float m = -INFINITY;
  #pragma unroll
        for (int jb = 0; jb < 2; ++jb)
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = jb * 16 + 4 * g + r;
            e[jb * 4 + r] = key < L ? s[jb][ib][r] : -INFINITY;
            m = fmaxf(m, e[jb * 4 + r]);
          }
        m = fmaxf(m, nf_xor16(m, lane));
        m = fmaxf(m, nf_xor32(m, lane));
        // exp(s - m) as one fma + v_exp_f32 (2^x, 1 ulp): |s - m| stays far below the 2^-24 |x| the pre-scaling adds
        constexpr float LOG2E = 1.4426950408889634f;
        const float m2 = m * LOG2E;
        float sum = 0.f;
  #pragma unroll
        for (int q = 0; q < 8; ++q) {
          e[q] = __builtin_amdgcn_exp2f(fmaf(e[q], LOG2E, -m2));   // masked keys: 2^(-inf) = 0
          sum += e[q];
        }
        sum += nf_xor16(sum, lane);
        sum += nf_xor32(sum, lane);
        const float inv = __builtin_amdgcn_rcpf(sum);
        if (SAVE && g == 0) image[(ib * 16 + l15) * NF_IMG_LD + 60] = m + logf(sum);   // -> flush_o of the next head
        // A fragment of P V: slot e of lane group g <-> key kappa(g, e) = (e < 4 ? 4g + e : 16 + 4g + e - 4)
        rp_split8(make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv),
                  make_float4(e[4] * inv, e[5] * inv, e[6] * inv, e[7] * inv), ph[ib], pl[ib]);
      }
      // B fragments of V under the same key permutation: lane (d = db * 16 + l15, g) <- V[kappa(g, e)][d]
      bf16x8 vh[2], vl[2];
  #pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int d = db * 16 + l15;
        nf_kfrag(image + 2 * DH + d, NF_IMG_LD, g, d < DH ? 1.0f : 0.f, vh[db], vl[db]);
      }
      f32x4 oacc[2][2];                                      // [query block][feature block]
  #pragma unroll
      for (int ib = 0; ib < 2; ++ib)
  #pragma unroll
        for (int db = 0; db < 2; ++db) oacc[ib][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int pass = 0; pass < 3; ++pass)
  #pragma unroll
        for (int ib = 0; ib < NI; ++ib)
  #pragma unroll
          for (int db = 0; db < 2; ++db)
            oacc[ib][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? pl[ib] : ph[ib], pass == 0 ? vl[db] : vh[db],
                                                                   oacc[ib][db], 0, 0, 0);
      if constexpr (ABL & 512) save_qkv(4, 8);
      // O -> image (over the q columns, dead by now) -> 16-byte row stores into o[:, head * dh ..]
  #pragma unroll
      for (int ib = 0; ib < NI; ++ib)
  #pragma unroll
        for (int db = 0; db < 2; ++db)
  #pragma unroll
          for (int r = 0; r < 4; ++r)
            if (db * 16 + l15 < DH) image[(ib * 16 + 4 * g + r) * NF_IMG_LD + db * 16 + l15] = oacc[ib][db][r];
      // (the global stores of O / lse are issued by `flush_o` during the NEXT head's MFMA phase: the head-top
      //  vmcnt(0) that the weight DMA needs would otherwise also wait out these stores' full latency -- 0.15 ms of
      //  0.64 at B = 128, tools/nf_probe.hip)
    }
  };
  if constexpr (SHARE) {
    if (short_w) run_heads(std::true_type{});
    else run_heads(std::false_type{});
  } else {
    run_heads(std::false_type{});
  }
  flush_o(heads - 1);
  // outstanding re-issued DMA of the last head must not outlive the workgroup's LDS allocation
  wait_vmcnt<0>();
}

static inline bool news_fused_ok(int L, int D, int heads) {
  // (the k-loop is unrolled for exactly NF_KB k-blocks: 288 <= D <= 316; the reference configuration is D = 300)
  return L >= 1 && L <= 32 && D % 4 == 0 && rp_kblocks(D, true) == NF_KB && heads > 0 && D == heads * 20;
}

// Workgroup shape: 8 waves / one workgroup per CU.  The 4-wave / two-per-CU shape (WV = 4: two-slot weight ring, chunks one ahead,
// bit-identical output) is instantiated for probes only (ALL = true, tools/nf4_probe.hip): measured in round 5 it loses 3 % on
// training forwards (every chunk boundary is a vmcnt(0) that also waits out the saves' stores) and, although the bare evaluation
// kernel gains 2-5 % in the probe, the evaluation forward of the step with pad-row sharing is slower with it (0.554 vs 0.547 ms;
// profiles/r05_ab.txt).
template <int ABL = 0, bool ALL = false>
static inline int launch_news_fused_fwd(const NewsFusedArgs& a_in, hipStream_t st, int wv = 0) {
  if (a_in.n_news <= 0) return NRL_OK;
  NewsFusedArgs a = a_in;
  const bool training = (a.x_save != nullptr || a.x_planes != nullptr) && a.lse != nullptr;
  if (!ALL || wv != 4) wv = 8;
  // full rounds of 8-news workgroups over the 256 CUs; what is left goes to 4-news workgroups if that fits one round
  constexpr int64_t CUS = 256;
  const int64_t full = (a.n_news / NF_WAVES) / CUS * CUS;
  const int64_t rem = a.n_news - full * NF_WAVES;
  int64_t blocks;
  if (wv == 4) {
    blocks = ceil_div(a.n_news, 4);
    a.full_wgs = (int)blocks;
  } else if (rem > 0 && rem <= 4 * CUS) {
    a.full_wgs = (int)full;
    blocks = full + ceil_div(rem, 4);
  } else {
    blocks = ceil_div(a.n_news, NF_WAVES);
    a.full_wgs = (int)blocks;
  }
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  NRL_REQUIRE(a.x_planes == nullptr || NF_KB == 10, "x planes assume 20 column blocks");
  const dim3 grid((unsigned)blocks), blk((unsigned)(wv * 64));
  if (training) {   // x + lse (+ q|k|v unless recomputed)
    NRL_REQUIRE(a.perm == nullptr, "fused news encoder: pad-row sharing is for evaluation forwards");
    if constexpr (ALL) {
      if (wv == 4) hipLaunchKernelGGL((news_fused_fwd_kernel<20, true, ABL, false, 4>), grid, blk, 0, st, a);
      else hipLaunchKernelGGL((news_fused_fwd_kernel<20, true, ABL>), grid, blk, 0, st, a);
    } else {
      hipLaunchKernelGGL((news_fused_fwd_kernel<20, true, ABL>), grid, blk, 0, st, a);
    }
  } else {
    NRL_REQUIRE(a.x_save == nullptr && a.x_planes == nullptr && a.qkv_save == nullptr && a.lse == nullptr,
                "fused news encoder: save x and lse (and optionally q|k|v), or nothing");
    if (a.perm != nullptr) {
      NRL_REQUIRE(a.n_short != nullptr && a.drop1.thresh == 0u, "fused news encoder: pad-row sharing needs the short-first list and no dropout");
      if constexpr (ALL) {
        if (wv == 4) { hipLaunchKernelGGL((news_fused_fwd_kernel<20, false, ABL, true, 4>), grid, blk, 0, st, a); NRL_LAUNCH_CHECK(); return NRL_OK; }
      }
      hipLaunchKernelGGL((news_fused_fwd_kernel<20, false, ABL, true>), grid, blk, 0, st, a);
    } else {
      if constexpr (ALL) {
        if (wv == 4) { hipLaunchKernelGGL((news_fused_fwd_kernel<20, false, ABL, false, 4>), grid, blk, 0, st, a); NRL_LAUNCH_CHECK(); return NRL_OK; }
      }
      hipLaunchKernelGGL((news_fused_fwd_kernel<20, false, ABL>), grid, blk, 0, st, a);
    }
  }
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}


// =====================================================================================================
// Evaluation forwards from a per-token q|k|v table (round 6; SURVEY.md section 8(f) row 3, reference rec_dataset.py:98-121 /
// nrms_module.py:398-535: validation and test re-encode every news of every impression with FROZEN weights).
//
// Without dropout the q|k|v rows of a token position are a function of its token id alone (text.py:224,229: x = E[id], then
// x W_in^T + b_in), and a MIND-shaped batch holds 211,200 positions over <= 14 k distinct ids (62 % of them the padding id).
// `launch_news_qkv_table` runs the in-projection ONCE per vocabulary id (news_fused_fwd_kernel<.., TBL>: 70 k rows instead of
// 211 k per forward, once per weight version) into
//     table[(id >> 5)][head][id & 31][64 floats = q 20 | k 20 | v 20 | 0 4]        (256 contiguous bytes per (id, head))
// and `news_tab_attn_fwd_kernel` below is the fused forward without its projection phase: one wavefront = one news, per head
// the 32 x 64 image is GATHERED from the table (eight 16-byte loads per lane, the next head's in flight under this head's
// arithmetic), then the same matrix-core attention, the same `o` planes.  No weight ring, no barrier, 35 KB of LDS per 4-wave
// workgroup.  The attention code below is a copy of the phase in news_fused_fwd_kernel on purpose (that kernel's register
// allocation is tuned and stays untouched); `test_token_table_forward_is_bit_identical` keeps the two from drifting apart:
// a table row carries the bits the projection phase would produce, so `o` -- and every news vector -- must be EQUAL.
struct NewsTabArgs {
  const float* tab;         // token q|k|v table
  const int64_t* ids;       // (n_news, L)
  int64_t n_news, vocab;
  int L, heads;
  float scale;              // 1 / sqrt(dh)
  unsigned char* o_planes;  // as NewsFusedArgs::o_planes
  const int32_t* perm = nullptr;      // pad-row sharing lists (news_classify_kernel), or both null
  const int32_t* n_short = nullptr;
};

constexpr int NTA_WAVES = 4;

// O of head `hp` (image columns 0 .. 19) -> the head-permuted (hi, lo) planes (the flush of news_fused_fwd_kernel)
__device__ __forceinline__ void nta_flush_o(const float* image, unsigned char* o_planes, int64_t row0, int hp, int heads, int Lw,
                                            int lane) {
  int ln = lane;
  asm volatile("" : "+v"(ln));
  {
    const int row = ln >> 1, half = ln & 1;
    const int64_t m = row0 + row;
    bf16x8 hi, lo;
    rp_split8(*reinterpret_cast<const float4*>(image + row * NF_IMG_LD + half * 8),
              *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + half * 8 + 4), hi, lo);
    unsigned char* dst = o_planes + (((m >> 4) * 19 + hp) * 2) * 512 + (m & 15) * 32 + half * 16;
    if (row < Lw) {
      *reinterpret_cast<bf16x8*>(dst) = hi;
      *reinterpret_cast<bf16x8*>(dst + 512) = lo;
    }
  }
  {
    const int row = ln & 31;
    const int64_t m = row0 + row;
    const float4 v = *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + 16);
    uint32_t h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    unsigned char* blk = o_planes + (((m >> 4) * 19 + heads + (hp >> 2)) * 2) * 512 + (m & 15) * 32;
    if (row < Lw) {
      if (ln < 32) {
        *reinterpret_cast<uint2*>(blk + (hp & 3) * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(blk + 512 + (hp & 3) * 8) = make_uint2(l0, l1);
      } else if (hp == heads - 1 && (heads & 3) != 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int slot = (heads & 3) + q;
          if (slot < 4) {
            *reinterpret_cast<uint2*>(blk + slot * 8) = make_uint2(q == 0 ? 0x3F80u : 0u, 0u);
            *reinterpret_cast<uint2*>(blk + 512 + slot * 8) = make_uint2(0u, 0u);
          }
        }
      }
    }
  }
}

// the head loop of one news in two compile-time shapes: NI = 2 query blocks, and (a short news under pad-row sharing) NI = 1
template <int DH, int NI>
__device__ __forceinline__ void nta_heads(const NewsTabArgs& P, float* image, int64_t row0, int Lw, int lane) {
  const int l15 = lane & 15, g = lane >> 4;
  const int L = P.L, heads = P.heads;
  // slot = pass * 64 + lane of the 32 x 16 float4 image: row = pass * 4 + g, 16-byte column l15.  Rows past the news (L < 32)
  // read the last token's row: their keys are masked, their values meet P = 0, their query rows are never stored.
  // (eight NAMED registers sets, not an array: a float4 array carried around the head loop stays in scratch memory -- hipcc
  //  stores every load to the stack at the loop's end, behind a vmcnt(0), and reloads it at the top: no prefetch at all)
  const float *s0, *s1, *s2, *s3, *s4, *s5, *s6, *s7;
  float4 n0, n1, n2, n3, n4, n5, n6, n7;
#define NTA_SRC(i)                                                                                        \
  {                                                                                                       \
    const int row = (i) * 4 + g;                                                                          \
    const int64_t id = P.ids[row0 + (row < L ? row : L - 1)];                                             \
    s##i = P.tab + ((size_t)((id >> 5) * heads) * 2048u + (size_t)(id & 31) * 64u + (size_t)l15 * 4u);    \
  }
  NTA_SRC(0) NTA_SRC(1) NTA_SRC(2) NTA_SRC(3) NTA_SRC(4) NTA_SRC(5) NTA_SRC(6) NTA_SRC(7)
#undef NTA_SRC
#define NTA_LOAD(i) n##i = *reinterpret_cast<const float4*>(s##i);
  NTA_LOAD(0) NTA_LOAD(1) NTA_LOAD(2) NTA_LOAD(3) NTA_LOAD(4) NTA_LOAD(5) NTA_LOAD(6) NTA_LOAD(7)
  for (int h = 0; h < heads; ++h) {
    // this head's rows -> the private image; the next head's leave for the registers they came from
#define NTA_PUT(i) *reinterpret_cast<float4*>(image + ((i) * 4 + g) * NF_IMG_LD + 4 * l15) = n##i;
    NTA_PUT(0) NTA_PUT(1) NTA_PUT(2) NTA_PUT(3) NTA_PUT(4) NTA_PUT(5) NTA_PUT(6) NTA_PUT(7)
#undef NTA_PUT
    {
      const int step = h + 1 < heads ? 2048 : 0;          // (last head: its own rows again, uniform control flow)
      s0 += step; s1 += step; s2 += step; s3 += step; s4 += step; s5 += step; s6 += step; s7 += step;
      NTA_LOAD(0) NTA_LOAD(1) NTA_LOAD(2) NTA_LOAD(3) NTA_LOAD(4) NTA_LOAD(5) NTA_LOAD(6) NTA_LOAD(7)
      // (left alone, the scheduler sinks these loads below the attention phase -- to the end of the head, a few dozen
      //  instructions in front of the wait for them)
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- S^T = K Q^T on the matrix cores (as news_fused_fwd_kernel) --------------------------------------
    bf16x8 kh[2], kl[2], qh[2], ql[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      nf_frag8(image + (b * 16 + l15) * NF_IMG_LD + DH, g, 1.0f, kh[b], kl[b]);
      if (b < NI) nf_frag8(image + (b * 16 + l15) * NF_IMG_LD, g, P.scale, qh[b], ql[b]);
    }
    f32x4 s[2][2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) s[jb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int ib = 0; ib < NI; ++ib)
          s[jb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? kl[jb] : kh[jb], pass == 0 ? ql[ib] : qh[ib],
                                                              s[jb][ib], 0, 0, 0);
    bf16x8 ph[2], pl[2];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      float e[8];
      float m = -INFINITY;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = jb * 16 + 4 * g + r;
          e[jb * 4 + r] = key < L ? s[jb][ib][r] : -INFINITY;
          m = fmaxf(m, e[jb * 4 + r]);
        }
      m = fmaxf(m, nf_xor16(m, lane));
      m = fmaxf(m, nf_xor32(m, lane));
      constexpr float LOG2E = 1.4426950408889634f;
      const float m2 = m * LOG2E;
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        e[q] = __builtin_amdgcn_exp2f(fmaf(e[q], LOG2E, -m2));
        sum += e[q];
      }
      sum += nf_xor16(sum, lane);
      sum += nf_xor32(sum, lane);
      const float inv = __builtin_amdgcn_rcpf(sum);
      rp_split8(make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv),
                make_float4(e[4] * inv, e[5] * inv, e[6] * inv, e[7] * inv), ph[ib], pl[ib]);
    }
    bf16x8 vh[2], vl[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int d = db * 16 + l15;
      nf_kfrag(image + 2 * DH + d, NF_IMG_LD, g, d < DH ? 1.0f : 0.f, vh[db], vl[db]);
    }
    f32x4 oacc[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[ib][db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int ib = 0; ib < NI; ++ib)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          oacc[ib][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? pl[ib] : ph[ib], pass == 0 ? vl[db] : vh[db],
                                                                 oacc[ib][db], 0, 0, 0);
#pragma unroll
    for (int ib = 0; ib < NI; ++ib)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (db * 16 + l15 < DH) image[(ib * 16 + 4 * g + r) * NF_IMG_LD + db * 16 + l15] = oacc[ib][db][r];
    nta_flush_o(image, P.o_planes, row0, h, heads, Lw, lane);
  }
#undef NTA_LOAD
}

template <int DH, bool SHARE>
__global__ void __launch_bounds__(NTA_WAVES * 64, 3) news_tab_attn_fwd_kernel(const NewsTabArgs P) {
  static_assert(DH == 20, "image packing assumes 3 * dh <= 64 with dh = 20");
  __shared__ __attribute__((aligned(16))) float smem[NTA_WAVES * NF_IMG_FLOATS + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const image = smem + wave * NF_IMG_FLOATS;
  const int64_t news_v = (int64_t)blockIdx.x * NTA_WAVES + wave;
  if (news_v >= P.n_news) return;                          // (no workgroup-level synchronisation anywhere below)
  int64_t news = news_v;
  bool short_w = false;
  if constexpr (SHARE) {
    news = __builtin_amdgcn_readfirstlane(P.perm[news_v]);
    short_w = news_v < (int64_t)__builtin_amdgcn_readfirstlane(*P.n_short);
  }
  const int64_t row0 = news * P.L;
  if (SHARE && short_w) nta_heads<DH, 1>(P, image, row0, 16, lane);
  else nta_heads<DH, 2>(P, image, row0, P.L, lane);
}

static inline size_t news_qkv_table_floats(int64_t vocab, int heads) { return (size_t)((vocab + 31) / 32) * heads * 32 * 64; }

// the table: the in-projection of every vocabulary id (news_fused_fwd_kernel in its TBL shape)
static inline int launch_news_qkv_table(const float* emb_table, int64_t vocab, int D, int heads, const uint16_t* img_heads,
                                        float* table, hipStream_t st) {
  NRL_REQUIRE(news_fused_ok(32, D, heads) && vocab > 0, "token table: unsupported geometry");
  NRL_REQUIRE(news_qkv_table_floats(vocab, heads) < (1ull << 32), "token table: more than 2^32 floats");
  NewsFusedArgs a;
  a.table = emb_table; a.ids = nullptr; a.img = img_heads; a.n_news = (vocab + 31) / 32; a.L = 32; a.D = D; a.heads = heads;
  a.dh = 20; a.scale = 0.f; a.drop1 = make_dropout(0.0, 0, 0); a.o = nullptr; a.o_planes = nullptr; a.x_save = nullptr;
  a.x_planes = nullptr; a.qkv_save = table; a.qkv_head_major = 1; a.lse = nullptr; a.vocab = vocab;
  const int64_t blocks = ceil_div(a.n_news, NF_WAVES);
  a.full_wgs = (int)blocks;
  hipLaunchKernelGGL((news_fused_fwd_kernel<20, true, 0, false, NF_WAVES, true>), dim3((unsigned)blocks), dim3(NF_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

static inline int launch_news_tab_attn_fwd(const NewsTabArgs& a, hipStream_t st) {
  if (a.n_news <= 0) return NRL_OK;
  NRL_REQUIRE(a.L >= 1 && a.L <= 32 && a.heads > 0 && a.tab != nullptr && a.o_planes != nullptr, "token-table forward: bad arguments");
  NRL_REQUIRE(news_qkv_table_floats(a.vocab, a.heads) < (1ull << 32), "token table: more than 2^32 floats");
  const int64_t blocks = ceil_div(a.n_news, NTA_WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  if (a.perm != nullptr) {
    NRL_REQUIRE(a.n_short != nullptr, "token-table forward: pad-row sharing needs both lists");
    hipLaunchKernelGGL((news_tab_attn_fwd_kernel<20, true>), dim3((unsigned)blocks), dim3(NTA_WAVES * 64), 0, st, a);
  } else {
    hipLaunchKernelGGL((news_tab_attn_fwd_kernel<20, false>), dim3((unsigned)blocks), dim3(NTA_WAVES * 64), 0, st, a);
  }
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}


// =====================================================================================================
// Token-attention backward on the matrix cores from the forward's head-major q|k|v slabs:
//   (q|k|v slab of (news, head): L x 256 B contiguous, d_o, lse) -> dq|dk|dv slab of the head's plane of `dqkv`
//
// Replaces attn_bwd_small for the fused news path (fp32 VALU, 80-byte strided q|k|v reads: 2.4 GB of HBM traffic for
// 1.77 GB of operands, 0.58 ms at B = 128 -- profiles/r02_x3_kernel_stats.csv).  One wavefront owns `hpw` consecutive
// heads of one news; per head the 32 x 64 slab goes registers -> a private LDS image (rows padded to 68 floats) and the
// whole 32 x 32 backward runs as in news_fused_bwd_kernel below:
//   S = Q K^T, S^T = K Q^T, dP = dO V^T, dP^T = V dO^T     (both orientations: an accumulator block holds 4 rows x 1
//   P = exp(S - lse), delta = rowsum(P o dP), dS = P o (dP - delta)   column per lane = the A-operand layout of the
//   dV = P^T dO,  dK = dS^T (scale Q),  dQ = scale dS K             next product under the slot permutation kappa)
// The next head's slab / d_o slice / lse are in flight (registers) during the current head's arithmetic.  No
// workgroup-level synchronisation: every LDS byte is private to its wave.
struct NewsAttnBwdArgs {
  const float* qkv_hm;  // (n_news, heads, L, 64)
  const float* d_o;     // (n_news * L, D)
  const float* lse;     // (n_news * heads, L)
  float* dqkv;          // head planes (heads, n_news * L, 64) = [dq | dk | dv | 0 4] rows (KCSlab / RCSlab, nrl_gemm.h):
                        // the L rows of a (news, head) pair are one contiguous slab, written in full 128-byte lines
  int64_t n_news;
  int L, D, heads;
  float scale;
  int hpw;              // heads per wave (divides heads)
  int planes;           // 1: dqkv as (hi, lo) bf16 fragment-block planes [head][news * 2 + mb][cb 0 .. 3][p][16 x 16]
                        // (nrl_wgrad_planes.h; token rows padded to 32 per news) instead of fp32 head planes
};

constexpr int NAB_WAVES = 4;
constexpr int NAB_DO_FLOATS = 32 * 20;
constexpr int NAB_VEC_FLOATS = 64;                         // [0, 32): lse of the query, [32, 64): delta of the query
constexpr int NAB_WAVE_FLOATS = NF_IMG_FLOATS + NAB_DO_FLOATS + NAB_VEC_FLOATS;

// ABL (tools/nf_probe.hip only): 1 = no dqkv stores, 2 = no arithmetic (slab -> image -> stores), 4 = no prefetch loads,
// 8 = plain instead of streaming stores
template <int DH, int OCC, int ABL = 0>
__global__ void __launch_bounds__(NAB_WAVES * 64, OCC) news_attn_bwd_kernel(const NewsAttnBwdArgs P) {
  static_assert(DH == 20, "image packing assumes dh = 20");
  __shared__ __attribute__((aligned(16))) float smem[NAB_WAVES * NAB_WAVE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  float* const image = smem + wave * NAB_WAVE_FLOATS;
  float* const dO_s = image + NF_IMG_FLOATS;              // [32][20]
  float* const vec = dO_s + NAB_DO_FLOATS;                // lse | delta

  const int L = P.L, D = P.D, heads = P.heads, hpw = P.hpw;
  const int groups = heads / hpw;
  const int64_t unit = (int64_t)blockIdx.x * NAB_WAVES + wave;
  const int64_t news = unit / groups;
  if (news >= P.n_news) return;
  const int h0 = (int)(unit - news * groups) * hpw;
  const int64_t row0 = news * L;
  const float* const do_base = P.d_o + row0 * (int64_t)D;

  // slab (8 float4 per lane), d_o slice (32 x 20, rows >= L zero) and lse of head `hd`: global -> registers
  auto get_head_inputs = [&](int hd, float4 (&qv)[8], float4 (&dv)[3], float& ls) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const float* slab = P.qkv_hm + (news * heads + hd) * (int64_t)L * 64;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int slot_i = pass * 64 + ln;
      const bool ok = (slot_i >> 4) < L;
      qv[pass] = *reinterpret_cast<const float4*>(slab + (ok ? 4 * slot_i : 0));
      if (!ok) qv[pass] = f4zero();
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + ln;
      const int row = slot_i / 5, c4 = slot_i - row * 5;
      const bool ok = slot_i < 160 && row < L;
      dv[pass] = *reinterpret_cast<const float4*>(do_base + (ok ? row * D + hd * DH + 4 * c4 : 0));
      if (!ok) dv[pass] = f4zero();
    }
    const bool lok = ln < L;
    ls = P.lse[(news * heads + hd) * L + (lok ? ln : 0)];
    if (!lok) ls = 1e30f;                                  // pad queries: P = exp(s - 1e30) = 0
  };
  auto put_head_inputs = [&](const float4 (&qv)[8], const float4 (&dv)[3], float ls) {
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int slot_i = pass * 64 + lane;
      *reinterpret_cast<float4*>(image + (slot_i >> 4) * NF_IMG_LD + 4 * (slot_i & 15)) = qv[pass];
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + lane;
      if (slot_i < 160) *reinterpret_cast<float4*>(dO_s + 4 * slot_i) = dv[pass];
    }
    if (lane < 32) vec[lane] = ls;
  };
  {
    float4 qv[8], dv[3];
    float ls;
    get_head_inputs(h0, qv, dv, ls);
    put_head_inputs(qv, dv, ls);
  }

  for (int hh = 0; hh < hpw; ++hh) {
    const int h = h0 + hh;
    const int hn = hh + 1 < hpw ? h + 1 : h;
    float4 nx_qv[8], nx_dv[3];
    float nx_ls;
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) nx_qv[i] = f4zero();
#pragma unroll
      for (int i = 0; i < 3; ++i) nx_dv[i] = f4zero();
      nx_ls = 0.f;
    } else {
      get_head_inputs(hn, nx_qv, nx_dv, nx_ls);
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (!(ABL & 2)) {

    auto frag8 = [&](const float* src, int ld, int row, int col0, float mul, bf16x8& hi, bf16x8& lo) {
      nf_frag8(src + row * ld + col0, g, mul, hi, lo);
    };
    auto kfrag = [&](const float* src, int ld, int col0, int db, float mul, bf16x8& hi, bf16x8& lo) {
      const int d = db * 16 + l15;
      nf_kfrag(src + col0 + d, ld, g, d < DH ? mul : 0.f, hi, lo);
    };
    auto mm3 = [&](f32x4& c, const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, c, 0, 0, 0);
    };
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;

    // ---- scores in both orientations: s[ib][jb] = (queries x keys), sT[jb][ib] = (keys x queries) --------
    f32x4 s[2][2], sT[2][2];
    {
      bf16x8 qh[2], ql[2], kh[2], kl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        frag8(image, NF_IMG_LD, b * 16 + l15, 0, P.scale, qh[b], ql[b]);
        frag8(image, NF_IMG_LD, b * 16 + l15, DH, 1.0f, kh[b], kl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          s[a][b] = z4; sT[a][b] = z4;
          mm3(s[a][b], qh[a], ql[a], kh[b], kl[b]);
          mm3(sT[a][b], kh[a], kl[a], qh[b], ql[b]);
        }
    }
    // ---- dP = dO V^T (queries x keys), dP^T = V dO^T --------------------------------------------------------
    f32x4 dp[2][2], dpT[2][2];
    {
      bf16x8 oh[2], ol[2], vh[2], vl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        frag8(dO_s, DH, b * 16 + l15, 0, 1.0f, oh[b], ol[b]);
        frag8(image, NF_IMG_LD, b * 16 + l15, 2 * DH, 1.0f, vh[b], vl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          dp[a][b] = z4; dpT[a][b] = z4;
          mm3(dp[a][b], oh[a], ol[a], vh[b], vl[b]);
          mm3(dpT[a][b], vh[a], vl[a], oh[b], ol[b]);
        }
    }
    // ---- P^T, delta (per query = per column of the transposed orientation), dS^T ---------------------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float lse2 = vec[ib * 16 + l15] * LOG2E;
      float dl = 0.f;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = jb * 16 + 4 * g + r;
          const float p = key < L ? __builtin_amdgcn_exp2f(fmaf(sT[jb][ib][r], LOG2E, -lse2)) : 0.f;
          dl += p * dpT[jb][ib][r];
          sT[jb][ib][r] = p;
        }
      dl += nf_xor16(dl, lane);
      dl += nf_xor32(dl, lane);
      if (g == 0) vec[32 + ib * 16 + l15] = dl;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[jb][ib][r] *= dpT[jb][ib][r] - dl;          // sT now holds dS^T
    }
    __builtin_amdgcn_wave_barrier();
    // ---- P, dS in the (queries x keys) orientation: rows 4g + r need lse / delta of THEIR query ------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float4 lse_r = *reinterpret_cast<const float4*>(vec + ib * 16 + 4 * g);
      const float4 dl_r = *reinterpret_cast<const float4*>(vec + 32 + ib * 16 + 4 * g);
      const float ls4[4] = {lse_r.x * LOG2E, lse_r.y * LOG2E, lse_r.z * LOG2E, lse_r.w * LOG2E};
      const float dl4[4] = {dl_r.x, dl_r.y, dl_r.z, dl_r.w};
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const bool kok = jb * 16 + l15 < L;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = kok ? __builtin_amdgcn_exp2f(fmaf(s[ib][jb][r], LOG2E, -ls4[r])) : 0.f;
          s[ib][jb][r] = p;                                                          // s now holds P
          dp[ib][jb][r] = p * (dp[ib][jb][r] - dl4[r]);                              // dp now holds dS
        }
      }
    }
    // ---- dV = P^T dO: A = P with the query slots kappa-permuted (exactly what a key-column lane holds) --------
    f32x4 dv_[2][2], dk_[2][2], dq_[2][2];
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(dO_s, DH, 0, db, 1.0f, bh[db], bl[db]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(s[0][jb][0], s[0][jb][1], s[0][jb][2], s[0][jb][3]),
                  make_float4(s[1][jb][0], s[1][jb][1], s[1][jb][2], s[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv_[jb][db] = z4;
          mm3(dv_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    // ---- dK = dS^T (scale Q): A = dS in the same key-column form; dQ = scale dS K: A = dS^T (query-column form) ---
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(image, NF_IMG_LD, 0, db, P.scale, bh[db], bl[db]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(dp[0][jb][0], dp[0][jb][1], dp[0][jb][2], dp[0][jb][3]),
                  make_float4(dp[1][jb][0], dp[1][jb][1], dp[1][jb][2], dp[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dk_[jb][db] = z4;
          mm3(dk_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(image, NF_IMG_LD, DH, db, 1.0f, bh[db], bl[db]);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(sT[0][ib][0], sT[0][ib][1], sT[0][ib][2], sT[0][ib][3]),
                  make_float4(sT[1][ib][0], sT[1][ib][1], sT[1][ib][2], sT[1][ib][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dq_[ib][db] = z4;
          mm3(dq_[ib][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // every strided read of Q, K, V is done: dQ (x scale) -> Q columns, dK -> K columns, dV -> V columns
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (db * 16 + l15 < DH) {
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + db * 16 + l15] = dq_[b][db][r] * P.scale;
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + DH + db * 16 + l15] = dk_[b][db][r];
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + 2 * DH + db * 16 + l15] = dv_[b][db][r];
          }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- image [token][dq | dk | dv | 0] -> registers -> this head's plane of dqkv -------------------------------
    // The next head's inputs are parked in the image BETWEEN the LDS reads and the global stores of this head's
    // result: waiting for their loads (vmcnt) then never waits on this head's stores, which get the whole next
    // head to retire.
    float4 stv[8];
    if (P.planes) {
      // 16-byte chunks of 8 features, split once here: chunk ch = (block blk = ch >> 5, row r16, half) -> the block's hi
      // plane at (ch & 31) * 16, lo plane 512 bytes on: every 512-byte block plane is written by 32 consecutive lanes
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int ch = pass * 64 + ln;
        const int blk = ch >> 5, r16 = (ch >> 1) & 15, half = ch & 1;
        const float* src = image + ((blk >> 2) * 16 + r16) * NF_IMG_LD + ((blk & 3) * 2 + half) * 8;
        bf16x8 hi, lo;
        rp_split8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), hi, lo);
        stv[2 * pass] = __builtin_bit_cast(float4, hi);
        stv[2 * pass + 1] = __builtin_bit_cast(float4, lo);
      }
    } else {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int slot_i = pass * 64 + ln;
        stv[pass] = *reinterpret_cast<const float4*>(image + (slot_i >> 4) * NF_IMG_LD + 4 * (slot_i & 15));
      }
    }
    __builtin_amdgcn_wave_barrier();
    put_head_inputs(nx_qv, nx_dv, nx_ls);
    if constexpr (!(ABL & 1)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (P.planes) {
        float* out = P.dqkv + (((int64_t)h * P.n_news + news) * 2) * 4 * 256;   // 8 KiB per (head, news)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int ch = pass * 64 + ln;
          float* dst = out + (ch >> 5) * 256 + (ch & 31) * 4;
          store4(dst, stv[2 * pass], !(ABL & 8));
          store4(dst + 128, stv[2 * pass + 1], !(ABL & 8));
        }
      } else {
        float* out = P.dqkv + ((int64_t)h * P.n_news * L + row0) * 64;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const int slot_i = pass * 64 + ln;
          if ((slot_i >> 4) < L) store4(out + 4 * slot_i, stv[pass], !(ABL & 8));
        }
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass)
        asm volatile("" ::"v"(stv[pass].x), "v"(stv[pass].y), "v"(stv[pass].z), "v"(stv[pass].w));
    }
  }
}

template <int OCC = 2, int ABL = 0>
static inline int launch_news_attn_bwd(const NewsAttnBwdArgs& a_in, hipStream_t st) {
  if (a_in.n_news <= 0) return NRL_OK;
  NewsAttnBwdArgs a = a_in;
  a.hpw = a.heads % 5 == 0 ? 5 : (a.heads % 3 == 0 ? 3 : 1);   // ~7 rounds of workgroups at B = 128 instead of 2.3
  const int64_t units = a.n_news * (a.heads / a.hpw);
  const int64_t blocks = ceil_div(units, NAB_WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  hipLaunchKernelGGL((news_attn_bwd_kernel<20, OCC, ABL>), dim3((unsigned)blocks), dim3(NAB_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}


// =====================================================================================================
// The same backward with every operand split ONCE, into (hi, lo) bf16 planes in LDS.
//
// news_attn_bwd_kernel builds fourteen operand fragments of q, k, v and d_o per head from its fp32 image -- each value up to
// twice, once in row form (reduction over features) and once token-strided (reduction over tokens), 112 elements per lane
// turned into (hi, lo) -- against 84 MFMAs.  Here the slab's 32 and the d_o slice's 12 elements per lane are split on their
// way INTO LDS (nap_* planes above; q times 1/sqrt(dh) first, as the fragment builders did), and both fragment forms are
// plain reads of bf16: 16- / 8-byte pieces for the row form, `ds_read_b64_tr_b16` for the token-strided form (a 16-lane group
// transposes 4 token rows x 16 features: lane (feature, g) receives tokens 4g .. 4g + 3; two row blocks = the eight slots of
// kappa).  Left to split besides: P, dS, dS^T (accumulators).  Same products in the same order on the same operand bits as
// news_attn_bwd_kernel: the output is bit-identical (tools/nf_probe.hip compares the 1 GB of dqkv planes).
// (Measured on the way: with the FORWARD saving q|k|v pre-split -- its score fragments are exactly scale * q and k in row form --
//  this kernel needs no q|k|v split at all and ran 0.418 vs 0.456 ms, but the forward's 24 narrow stores per head from
//  fragment registers, instead of 8 whole-row stores from its image, cost it 0.04-0.07 ms: net loss, forward left alone.)
constexpr int NAP_DO = 4 * NAP_PLANE;                       // d_o planes: [hi | lo][row block 0 | 1], chunk-major like the slab's
constexpr int NAP_IN = NAP_SLAB + NAP_DO;                   // 10240 bytes; reused as the fp32 output image
constexpr int NAP_WAVE_BYTES = NAP_IN + NAB_VEC_FLOATS * 4;
static_assert(NF_IMG_FLOATS * 4 <= NAP_IN, "the output image lives in the input planes' space");

template <int DH, int OCC, int ABL = 0>
__global__ void __launch_bounds__(NAB_WAVES * 64, OCC) news_attn_bwd_p_kernel(const NewsAttnBwdArgs P) {
  static_assert(DH == 20, "plane rows hold 20 features");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NAB_WAVES * NAP_WAVE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  unsigned char* const in = smem + wave * NAP_WAVE_BYTES;
  float* const image = reinterpret_cast<float*>(in);       // [32][NF_IMG_LD] output staging, once every input fragment is read
  float* const vec = reinterpret_cast<float*>(in + NAP_IN);   // lse | delta
  const uint32_t in_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)wave * NAP_WAVE_BYTES;

  const int L = P.L, D = P.D, heads = P.heads, hpw = P.hpw;
  const int groups = heads / hpw;
  const int64_t unit = (int64_t)blockIdx.x * NAB_WAVES + wave;
  const int64_t news = unit / groups;
  if (news >= P.n_news) return;
  const int h0 = (int)(unit - news * groups) * hpw;
  const int64_t row0 = news * L;
  const float* const do_base = P.d_o + row0 * (int64_t)D;

  // slab (8 float4 per lane), d_o slice (32 x 20, rows >= L zero) and lse of head `hd`: global -> registers
  auto get_head_inputs = [&](int hd, f32x4 (&qv)[8], float4 (&dv)[3], float& ls) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const float* slab = P.qkv_hm + (news * heads + hd) * (int64_t)L * 64;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int slot_i = pass * 64 + ln;
      const bool ok = (slot_i >> 4) < L;
      qv[pass] = *reinterpret_cast<const f32x4*>(slab + (ok ? 4 * slot_i : 0));
      if (!ok) qv[pass] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + ln;
      const int row = slot_i / 5, c4 = slot_i - row * 5;
      const bool ok = slot_i < 160 && row < L;
      dv[pass] = *reinterpret_cast<const float4*>(do_base + (ok ? row * D + hd * DH + 4 * c4 : 0));
      if (!ok) dv[pass] = f4zero();
    }
    const bool lok = ln < L;
    ls = P.lse[(news * heads + hd) * L + (lok ? ln : 0)];
    if (!lok) ls = 1e30f;                                  // pad queries: P = exp(s - 1e30) = 0
  };
  // registers -> (hi, lo) planes: a lane's float4 = features 4 c4 .. 4 c4 + 3 of one part (q | k | v) of one token row
  auto put_head_inputs = [&](const f32x4 (&qv)[8], const float4 (&dv)[3], float ls) __attribute__((always_inline)) {
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int slot_i = pass * 64 + lane;
      const int row = slot_i >> 4, ch = slot_i & 15;
      const int part = ch / 5, c4 = ch - part * 5;          // ch == 15: the slab's four pad columns
      const float mul = part == 0 ? P.scale : 1.0f;
      uint32_t h0_, l0_, h1_, l1_;
      split_pair(qv[pass][0] * mul, qv[pass][1] * mul, h0_, l0_);
      split_pair(qv[pass][2] * mul, qv[pass][3] * mul, h1_, l1_);
      if (ch < 15) {
        unsigned char* dst = in + nap_off(part, row >> 4, 0) + nap_quad(row & 15, c4);
        *reinterpret_cast<nap_u2*>(dst) = make_uint2(h0_, h1_);
        *reinterpret_cast<nap_u2*>(dst + NAP_PLANE) = make_uint2(l0_, l1_);
      }
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + lane;
      if (slot_i < 160) {
        const int row = slot_i / 5, c4 = slot_i - row * 5;
        uint32_t h0_, l0_, h1_, l1_;
        split_pair(dv[pass].x, dv[pass].y, h0_, l0_);
        split_pair(dv[pass].z, dv[pass].w, h1_, l1_);
        unsigned char* dst = in + NAP_SLAB + (row >> 4) * NAP_PLANE + nap_quad(row & 15, c4);
        *reinterpret_cast<nap_u2*>(dst) = make_uint2(h0_, h1_);
        *reinterpret_cast<nap_u2*>(dst + 2 * NAP_PLANE) = make_uint2(l0_, l1_);
      }
    }
    if (lane < 32) vec[lane] = ls;
  };
  {
    f32x4 qv[8];
    float4 dv[3];
    float ls;
    get_head_inputs(h0, qv, dv, ls);
    put_head_inputs(qv, dv, ls);
  }

  // row form: lane (row l15 of the block, g) <- features 8g .. 8g + 7 of its row (20 per row: g = 2 keeps four, g = 3 none)
  auto rowfrag = [&](int hi_off, int plane_stride, bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
    // every lane reads a 16-byte piece (chunk g & 1) and an 8-byte piece (features 16 .. 19) and keeps the one that is its own
    const unsigned char* p16 = in + hi_off + (g & 1) * 256 + l15 * 16;
    const unsigned char* p8 = in + hi_off + 512 + l15 * 8;
    const uint4 a16 = *reinterpret_cast<const nap_u4a*>(p16), b16 = *reinterpret_cast<const nap_u4a*>(p16 + plane_stride);
    const uint2 a8 = *reinterpret_cast<const nap_u2*>(p8), b8 = *reinterpret_cast<const nap_u2*>(p8 + plane_stride);
    const uint32_t m8 = g == 2 ? 0xFFFFFFFFu : 0u;
    hi = __builtin_bit_cast(bf16x8, g < 2 ? a16 : make_uint4(a8.x & m8, a8.y & m8, 0u, 0u));
    lo = __builtin_bit_cast(bf16x8, g < 2 ? b16 : make_uint4(b8.x & m8, b8.y & m8, 0u, 0u));
  };
  // token-strided form: lane (feature d = db * 16 + l15, g) <- tokens kappa(g, e), e = 0 .. 7, of feature d
  typedef short nap_v4i16 __attribute__((ext_vector_type(4)));
  typedef short nap_v8i16 __attribute__((ext_vector_type(8)));
  auto trfrag = [&](int rb0_off, int rb1_off, int db, bf16x8& out) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) nap_v4i16* lds_v4;
    const uint32_t off = (uint32_t)nap_quad(4 * g + (l15 >> 2), db == 0 ? (l15 & 3) : 4);
    const nap_v4i16 e03 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(in_lds + (uint32_t)rb0_off + off));
    const nap_v4i16 e47 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(in_lds + (uint32_t)rb1_off + off));
    uint4 v = __builtin_bit_cast(uint4, (nap_v8i16)__builtin_shufflevector(e03, e47, 0, 1, 2, 3, 4, 5, 6, 7));
    const uint32_t m = (db == 0 || l15 < DH - 16) ? 0xFFFFFFFFu : 0u;   // features past dh: zero
    out = __builtin_bit_cast(bf16x8, make_uint4(v.x & m, v.y & m, v.z & m, v.w & m));
  };

  for (int hh = 0; hh < hpw; ++hh) {
    const int h = h0 + hh;
    const int hn = hh + 1 < hpw ? h + 1 : h;
    f32x4 nx_qv[8];
    float4 nx_dv[3];
    float nx_ls;
    get_head_inputs(hn, nx_qv, nx_dv, nx_ls);
    __builtin_amdgcn_wave_barrier();

    auto mm3 = [&](f32x4& c, const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, c, 0, 0, 0);
    };
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;

    // ---- scores in both orientations: s[ib][jb] = (queries x keys), sT[jb][ib] = (keys x queries) --------
    f32x4 s[2][2], sT[2][2];
    {
      bf16x8 qh[2], ql[2], kh[2], kl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        rowfrag(nap_off(0, b, 0), NAP_PLANE, qh[b], ql[b]);          // (scale * q)
        rowfrag(nap_off(1, b, 0), NAP_PLANE, kh[b], kl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          s[a][b] = z4; sT[a][b] = z4;
          mm3(s[a][b], qh[a], ql[a], kh[b], kl[b]);
          mm3(sT[a][b], kh[a], kl[a], qh[b], ql[b]);
        }
    }
    // ---- dP = dO V^T (queries x keys), dP^T = V dO^T --------------------------------------------------------
    f32x4 dp[2][2], dpT[2][2];
    {
      bf16x8 oh[2], ol[2], vh[2], vl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        rowfrag(NAP_SLAB + b * NAP_PLANE, 2 * NAP_PLANE, oh[b], ol[b]);
        rowfrag(nap_off(2, b, 0), NAP_PLANE, vh[b], vl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          dp[a][b] = z4; dpT[a][b] = z4;
          mm3(dp[a][b], oh[a], ol[a], vh[b], vl[b]);
          mm3(dpT[a][b], vh[a], vl[a], oh[b], ol[b]);
        }
    }
    // ---- P^T, delta (per query = per column of the transposed orientation), dS^T ---------------------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float lse2 = vec[ib * 16 + l15] * LOG2E;
      float dl = 0.f;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = jb * 16 + 4 * g + r;
          const float p = key < L ? __builtin_amdgcn_exp2f(fmaf(sT[jb][ib][r], LOG2E, -lse2)) : 0.f;
          dl += p * dpT[jb][ib][r];
          sT[jb][ib][r] = p;
        }
      dl += nf_xor16(dl, lane);
      dl += nf_xor32(dl, lane);
      if (g == 0) vec[32 + ib * 16 + l15] = dl;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[jb][ib][r] *= dpT[jb][ib][r] - dl;          // sT now holds dS^T
    }
    __builtin_amdgcn_wave_barrier();
    // ---- P, dS in the (queries x keys) orientation: rows 4g + r need lse / delta of THEIR query ------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float4 lse_r = *reinterpret_cast<const float4*>(vec + ib * 16 + 4 * g);
      const float4 dl_r = *reinterpret_cast<const float4*>(vec + 32 + ib * 16 + 4 * g);
      const float ls4[4] = {lse_r.x * LOG2E, lse_r.y * LOG2E, lse_r.z * LOG2E, lse_r.w * LOG2E};
      const float dl4[4] = {dl_r.x, dl_r.y, dl_r.z, dl_r.w};
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const bool kok = jb * 16 + l15 < L;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = kok ? __builtin_amdgcn_exp2f(fmaf(s[ib][jb][r], LOG2E, -ls4[r])) : 0.f;
          s[ib][jb][r] = p;                                                          // s now holds P
          dp[ib][jb][r] = p * (dp[ib][jb][r] - dl4[r]);                              // dp now holds dS
        }
      }
    }
    // ---- dV = P^T dO: A = P with the query slots kappa-permuted (exactly what a key-column lane holds) --------
    f32x4 dv_[2][2], dk_[2][2], dq_[2][2];
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        trfrag(NAP_SLAB, NAP_SLAB + NAP_PLANE, db, bh[db]);
        trfrag(NAP_SLAB + 2 * NAP_PLANE, NAP_SLAB + 3 * NAP_PLANE, db, bl[db]);
      }
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(s[0][jb][0], s[0][jb][1], s[0][jb][2], s[0][jb][3]),
                  make_float4(s[1][jb][0], s[1][jb][1], s[1][jb][2], s[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv_[jb][db] = z4;
          mm3(dv_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    // ---- dK = dS^T (scale Q): A = dS in the same key-column form; dQ = scale dS K: A = dS^T (query-column form) ---
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        trfrag(nap_off(0, 0, 0), nap_off(0, 1, 0), db, bh[db]);
        trfrag(nap_off(0, 0, 1), nap_off(0, 1, 1), db, bl[db]);
      }
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(dp[0][jb][0], dp[0][jb][1], dp[0][jb][2], dp[0][jb][3]),
                  make_float4(dp[1][jb][0], dp[1][jb][1], dp[1][jb][2], dp[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dk_[jb][db] = z4;
          mm3(dk_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        trfrag(nap_off(1, 0, 0), nap_off(1, 1, 0), db, bh[db]);
        trfrag(nap_off(1, 0, 1), nap_off(1, 1, 1), db, bl[db]);
      }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(sT[0][ib][0], sT[0][ib][1], sT[0][ib][2], sT[0][ib][3]),
                  make_float4(sT[1][ib][0], sT[1][ib][1], sT[1][ib][2], sT[1][ib][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dq_[ib][db] = z4;
          mm3(dq_[ib][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // every read of the input planes is done: dQ (x scale) | dK | dV -> the fp32 image in their place
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (db * 16 + l15 < DH) {
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + db * 16 + l15] = dq_[b][db][r] * P.scale;
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + DH + db * 16 + l15] = dk_[b][db][r];
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + 2 * DH + db * 16 + l15] = dv_[b][db][r];
          }
    __builtin_amdgcn_wave_barrier();
    // ---- image [token][dq | dk | dv | 0] -> registers -> this head's plane of dqkv (as news_attn_bwd_kernel) -----------
    float4 stv[8];
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int ch = pass * 64 + ln;
        const int blk = ch >> 5, r16 = (ch >> 1) & 15, half = ch & 1;
        const float* src = image + ((blk >> 2) * 16 + r16) * NF_IMG_LD + ((blk & 3) * 2 + half) * 8;
        float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        if ((blk & 3) == 3 && half == 1) v1 = f4zero();    // columns 60 .. 63: nobody wrote them
        bf16x8 hi, lo;
        rp_split8(v0, v1, hi, lo);
        stv[2 * pass] = __builtin_bit_cast(float4, hi);
        stv[2 * pass + 1] = __builtin_bit_cast(float4, lo);
      }
    }
    __builtin_amdgcn_wave_barrier();
    put_head_inputs(nx_qv, nx_dv, nx_ls);
    if constexpr (!(ABL & 1)) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      float* out = P.dqkv + (((int64_t)h * P.n_news + news) * 2) * 4 * 256;   // 8 KiB per (head, news)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int ch = pass * 64 + ln;
        float* dst = out + (ch >> 5) * 256 + (ch & 31) * 4;
        store4(dst, stv[2 * pass], !(ABL & 8));
        store4(dst + 128, stv[2 * pass + 1], !(ABL & 8));
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass)
        asm volatile("" ::"v"(stv[pass].x), "v"(stv[pass].y), "v"(stv[pass].z), "v"(stv[pass].w));
    }
  }
}

template <int OCC = 2, int ABL = 0>
static inline int launch_news_attn_bwd_p(const NewsAttnBwdArgs& a_in, hipStream_t st) {
  if (a_in.n_news <= 0) return NRL_OK;
  NRL_REQUIRE(a_in.planes == 1, "news_attn_bwd_p: dqkv goes out as fragment-block planes");
  NewsAttnBwdArgs a = a_in;
  a.hpw = a.heads % 5 == 0 ? 5 : (a.heads % 3 == 0 ? 3 : 1);
  const int64_t units = a.n_news * (a.heads / a.hpw);
  const int64_t blocks = ceil_div(units, NAB_WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  hipLaunchKernelGGL((news_attn_bwd_p_kernel<20, OCC, ABL>), dim3((unsigned)blocks), dim3(NAB_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}


// (news_fused_bwd_kernel -- q|k|v recomputed inside the matrix-core attention backward, round 2: 160 fragment VGPRs held through the
//  attention backward, 78 spilled registers, 1.69 ms against 0.60 + 0.22 ms, profiles/r02_fused_bwd_ab.txt -- lives in
//  tools/experimental/nrl_news_fused_bwd.h since round 5: it lost its A/B and no product path runs it.)

}  // namespace nrl
