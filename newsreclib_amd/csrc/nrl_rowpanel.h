// "Row-panel" bf16x3 GEMM for the narrow projections of the NRMS block (gfx950):
//
//     C (M x N) = epi( A (M x K) * B ),   N <= 320,  any K
//
// One wavefront owns 32 rows x ALL N columns of the output (accumulators: 2 x N/16 MFMA blocks in
// registers).  That turns the three costs that bound the tiled kernels (nrl_gemm_bf16x3*.h,
// 15-27 % matrix-core occupancy) into rounding errors:
//   * the fp32 -> (hi, lo) bf16 split of an activation fragment happens ONCE per (row block, k block)
//     and feeds 3 * N/16 MFMAs (57 at N = 300); in the 128 x 160 tiles the same fragment was split by
//     every column tile and every wave column (4-12 x redundant VALU work, 15 MFMAs per split);
//   * A never passes through LDS: a lane loads the 8 consecutive k of its row straight from global
//     (two 16-B loads per fragment, one k-block ahead), so LDS carries only the weights;
//   * the weights are laid out ONCE per step by `rp_weight_images` in MFMA-fragment order
//     ([k block][n block][hi|lo][lane][8 bf16]): a k-block is one contiguous chunk that travels
//     global -> LDS by `global_load_lds_dwordx4` with no address arithmetic, and a fragment read is one
//     conflict-free ds_read_b128 at lane * 16.
// Per k-block a wave issues 2 * N/16 ds_read_b128 and 6 * N/16 MFMAs (16 cycles each): LDS and VALU
// sit far below the matrix pipe.  Two 4-wave workgroups share a CU (2-slot LDS ring of <= 40 KB chunks
// each), so one workgroup's epilogue / barrier waits hide under the other's MFMAs.
//
// Arithmetic: identical to nrl_gemm_bf16x3.h (a = hi + lo, products hi*hi + hi*lo + lo*hi, fp32
// accumulate, lo terms first); outputs differ from the tiled kernel only by the accumulation order over k
// (the same k order, in fact: k-blocks ascending), i.e. they are bit-identical in practice.
#pragma once
#include "nrl_gemm_bf16x3_dma.h"

namespace nrl {

// ---- weight images -----------------------------------------------------------------------------------
// element (n, k) of the logical B^T (n = output column, k = reduction index) is src[n * sn + k * sk];
// k == K reads bias[n] when bias != nullptr (the "ones column": the A side then supplies 1.0 at k == K, so
// the bias is added by the matrix cores); everything outside (N, K [+ 1]) is zero.
struct RpImageJob {
  const float* src;
  const float* bias;
  uint16_t* img;
  int64_t sn, sk;
  int N, K, nblk, kblocks;
  // optional row remap for the fused news encoder (nrl_news_fused.h): logical column n of the image is
  // row `remap_base[n / remap_w] * ... ` -- see rp_image_row()
  int heads, dh;  // heads > 0: per-head packed q|k|v image (n = head * 64 + {q: 0..dh-1, k: dh.., v: 2dh..})
  // conv_w > 0: the tap-reversed conv weight of the CNN dgrad (nrl_conv.h): element (n = d, k = t' * F + f) is
  // src[(f * W + (W - 1 - t')) * N + d], src = Wc (F, W * N)
  int conv_f, conv_w;
  // kheads > 0: the reduction index of the image is a head-plane index (KCSlab, nrl_gemm.h): image k' = head * 64 + c
  // is logical k = part * (kheads * kdh) + head * kdh + d (c = part * kdh + d < 3 kdh), zero for the pad c
  int kheads, kdh;
  // kperm_heads > 0 (dh = 20): the reduction index follows the feature order of the fused forward's `o` planes -- block
  // cb < heads = features 0 .. 15 of head cb, block heads + s = features 16 .. 19 of heads 4s .. 4s + 3; every other slot
  // (the ones column, the tail) is a zero row
  int kperm_heads;
  // kappa != 0: inside every k-block of 32 the reduction index follows the accumulator-to-fragment permutation of the fused
  // news tail (nrl_news_tail.h): slot 8g + e holds logical k = 32 kb + (e < 4 ? 4g + e : 16 + 4g + e - 4)
  int kappa;
  // kpad_dp > 0 (round 5): the reduction index is TAP-PADDED -- image k' = tap * kpad_dp + d is logical k = tap * kpad_d + d for
  // d < kpad_d, a zero row for kpad_d <= d < kpad_dp.  The convolution forward over x as fragment-block planes (KCWindowPlanes,
  // nrl_conv.h): a tap is a whole number of 32-wide k-blocks of the planes' padded feature width (320 at D = 300)
  int kpad_d, kpad_dp;
};
constexpr int RP_MAX_JOBS = 12;
struct RpImageJobs {
  RpImageJob job[RP_MAX_JOBS];
  int count;
  int overflow;                           // a job past RP_MAX_JOBS was added: rp_jobs_launch fails instead of overrunning
  int64_t first_thread[RP_MAX_JOBS + 1];  // prefix sums of kblocks * nblk * 64
};

static inline int rp_kblocks(int K, bool bias) { return (K + (bias ? 1 : 0) + 31) / 32; }
static inline size_t rp_image_elems(int nblk, int kblocks) { return (size_t)kblocks * nblk * 1024; }  // uint16

// logical source row of image column n (or -1: zero column)
__device__ __forceinline__ int rp_image_row(const RpImageJob& J, int n) {
  if (J.heads <= 0) return n < J.N ? n : -1;
  const int head = n >> 6, c = n & 63;       // 64 image columns per head: q | k | v | pad
  if (head >= J.heads || c >= 3 * J.dh) return -1;
  const int part = c / J.dh, d = c - part * J.dh;
  return part * (J.heads * J.dh) + head * J.dh + d;   // rows [Wq; Wk; Wv] of in_proj_weight
}

static __global__ void __launch_bounds__(256) rp_weight_image_kernel(const RpImageJobs jobs) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= jobs.first_thread[jobs.count]) return;
  int j = 0;
#pragma unroll
  for (int q = 1; q < RP_MAX_JOBS; ++q)
    if (q < jobs.count && tid >= jobs.first_thread[q]) j = q;
  const RpImageJob& J = jobs.job[j];
  const int64_t t = tid - jobs.first_thread[j];
  const int lane = (int)(t & 63);
  const int nb = (int)((t >> 6) % J.nblk), kb = (int)((t >> 6) / J.nblk);
  const int n = nb * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
  const int row = rp_image_row(J, n);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int k = k0 + e;
    if (J.kappa != 0) {
      const int gg = lane >> 4;
      k = kb * 32 + (e < 4 ? 4 * gg + e : 16 + 4 * gg + (e - 4));
    }
    float x = 0.f;
    bool live = row >= 0;
    if (J.kperm_heads > 0) {
      const int cb = k >> 4, c = k & 15;
      int head, d;
      if (cb < J.kperm_heads) { head = cb; d = c; }
      else { head = 4 * (cb - J.kperm_heads) + (c >> 2); d = 16 + (c & 3); }
      // (slot K = 20 heads is the ones column of the `o` planes when heads % 4 != 0: the bias row, if the job has one)
      if (head < J.kperm_heads) k = head * 20 + d;
      else if (k != J.K || J.bias == nullptr) live = false;
    }
    if (J.kpad_dp > 0) {
      const int tap = k / J.kpad_dp, d = k - tap * J.kpad_dp;
      live = live && d < J.kpad_d;
      k = tap * J.kpad_d + d;
    }
    if (J.kheads > 0) {
      const int head = k >> 6, c = k & 63;
      live = live && head < J.kheads && c < 3 * J.kdh;
      const int part = c / J.kdh, d = c - part * J.kdh;
      k = part * (J.kheads * J.kdh) + head * J.kdh + d;
    }
    if (live) {
      if (k < J.K) {
        if (J.conv_w > 0) {
          const int tr = k / J.conv_f, f = k - tr * J.conv_f;
          x = J.src[((int64_t)f * J.conv_w + (J.conv_w - 1 - tr)) * J.N + row];
        } else {
          x = J.src[row * J.sn + k * J.sk];
        }
      }
      else if (k == J.K && J.bias != nullptr)
        x = J.bias[row];
    }
    v[e] = x;
  }
  uint32_t h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
  uint4* dst = reinterpret_cast<uint4*>(J.img) + ((int64_t)(kb * J.nblk + nb) * 2) * 64 + lane;
  dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
  dst[64] = make_uint4(l[0], l[1], l[2], l[3]);
}

static inline void rp_jobs_init(RpImageJobs* js) { js->count = 0; js->overflow = 0; js->first_thread[0] = 0; }
static inline RpImageJob* rp_jobs_add(RpImageJobs* js, const float* src, int64_t sn, int64_t sk, int N, int K,
                                      const float* bias, uint16_t* img, int nblk) {
  if (js->count >= RP_MAX_JOBS) {          // (reported by rp_jobs_launch; the last slot is overwritten, nothing overruns)
    js->overflow = 1;
    js->count = RP_MAX_JOBS - 1;
  }
  RpImageJob& J = js->job[js->count];
  J.src = src; J.bias = bias; J.img = img; J.sn = sn; J.sk = sk; J.N = N; J.K = K; J.nblk = nblk;
  J.kblocks = rp_kblocks(K, bias != nullptr); J.heads = 0; J.dh = 0; J.conv_f = 0; J.conv_w = 0; J.kheads = 0; J.kdh = 0; J.kperm_heads = 0;
  J.kappa = 0; J.kpad_d = 0; J.kpad_dp = 0;
  js->first_thread[js->count + 1] = js->first_thread[js->count] + (int64_t)J.kblocks * nblk * 64;
  js->count += 1;
  return &J;
}
// per-head packed in-projection image for the fused news encoder: 64 image columns per head = [q | k | v | 0] rows of
// in_proj_weight (3D, D), the bias as column K = D
static inline RpImageJob* rp_jobs_add_qkv_heads(RpImageJobs* js, const float* w_in, int D, const float* b_in,
                                                uint16_t* img, int heads, int dh) {
  RpImageJob* J = rp_jobs_add(js, w_in, D, 1, heads * 64, D, b_in, img, heads * 4);
  J->heads = heads;
  J->dh = dh;
  return J;
}
// image whose reduction index is tap-padded (see RpImageJob::kpad_dp): K' = taps * dp image rows, logical K = taps * d
static inline RpImageJob* rp_jobs_add_kpad(RpImageJobs* js, const float* src, int64_t sn, int64_t sk, int N, int taps, int d,
                                           int dp, uint16_t* img, int nblk) {
  RpImageJob* J = rp_jobs_add(js, src, sn, sk, N, taps * d, nullptr, img, nblk);
  J->kpad_d = d;
  J->kpad_dp = dp;
  J->kblocks = rp_kblocks(taps * dp, false);
  js->first_thread[js->count] = js->first_thread[js->count - 1] + (int64_t)J->kblocks * nblk * 64;
  return J;
}
// image whose reduction index runs over head planes (the in-projection dgrad of the fused news path: dx = dqkv W_in
// with dqkv in the KCSlab layout): K' = heads * 64 image rows, logical K = 3 * heads * dh
static inline RpImageJob* rp_jobs_add_kheads(RpImageJobs* js, const float* src, int64_t sn, int64_t sk, int N, int heads,
                                             int dh, uint16_t* img, int nblk) {
  RpImageJob* J = rp_jobs_add(js, src, sn, sk, N, 3 * heads * dh, nullptr, img, nblk);
  J->kheads = heads;
  J->kdh = dh;
  J->kblocks = heads * 2;
  js->first_thread[js->count] = js->first_thread[js->count - 1] + (int64_t)J->kblocks * nblk * 64;
  return J;
}
// image whose reduction index is the head-permuted feature order of the `o` planes (out-projection forward of the fused
// news path): K' = 16 * (heads + ceil(heads / 4)) slots
static inline RpImageJob* rp_jobs_add_kperm(RpImageJobs* js, const float* src, int64_t sn, int64_t sk, int N, int heads,
                                            uint16_t* img, int nblk, const float* bias = nullptr) {
  RpImageJob* J = rp_jobs_add(js, src, sn, sk, N, heads * 20, bias, img, nblk);
  J->kperm_heads = heads;
  J->kblocks = (16 * (heads + (heads + 3) / 4) + 31) / 32;
  js->first_thread[js->count] = js->first_thread[js->count - 1] + (int64_t)J->kblocks * nblk * 64;
  return J;
}
// image in kappa order (second projection of the fused news tail): bias at k = K against the ones feature of y
// `kblocks` > 0: the number of k-blocks the CONSUMER streams (the fused tail kernels walk a fixed count whatever K is): the
// blocks past K are written too, as zeros -- an image that stops at rp_kblocks(K) leaves them uninitialised, and a NaN bit
// pattern there times a zero operand is still NaN (found in round 4 with a NaN-poisoned workspace at Q = 64: the W_a^T
// image of the tail backward had 2 of its 7 k-blocks written)
static inline RpImageJob* rp_jobs_add_kappa(RpImageJobs* js, const float* src, int64_t sn, int64_t sk, int N, int K,
                                            const float* bias, uint16_t* img, int nblk, int kblocks = 0) {
  RpImageJob* J = rp_jobs_add(js, src, sn, sk, N, K, bias, img, nblk);
  J->kappa = 1;
  if (kblocks > J->kblocks) {
    J->kblocks = kblocks;
    js->first_thread[js->count] = js->first_thread[js->count - 1] + (int64_t)J->kblocks * nblk * 64;
  }
  return J;
}
static inline int rp_jobs_launch(const RpImageJobs& js, hipStream_t st) {
  NRL_REQUIRE(js.overflow == 0, "row-panel image jobs: more than %d images in one launch", RP_MAX_JOBS);
  if (js.count == 0) return NRL_OK;
  const int64_t total = js.first_thread[js.count];
  hipLaunchKernelGGL(rp_weight_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, js);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// handle of a built image
struct RpImage {
  const uint16_t* img = nullptr;
  int nblk = 0, kblocks = 0;
};

// one fragment (8 consecutive k of one row) fp32 -> packed bf16 (hi, lo)
__device__ __forceinline__ void rp_split8(const float4& v0, const float4& v1, bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
  split_pair(v0.x, v0.y, h[0], l[0]);
  split_pair(v0.z, v0.w, h[1], l[1]);
  split_pair(v1.x, v1.y, h[2], l[2]);
  split_pair(v1.z, v1.w, h[3], l[3]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// EpiPoolBwd writing dy as (hi, lo) bf16 fragment-block planes over the real rows (KCPlanesG) instead of fp32: dy is read
// only by the out-projection's dgrad and wgrad GEMMs, so the split is done once, here.  A lane of the 16-byte output stage
// holds 4 consecutive columns of one row = 8 bytes of each plane; the wave's stores of one accumulator block fill one
// 512-byte block plane.
struct EpiPoolBwdPlanes {
  EpiPoolBwd inner;          // `c` unused; ldc = row length of d_out / relu_src (= N)
  unsigned char* planes;
  int ncb;
  struct Row {
    EpiPoolBwd::Row in;
    unsigned char* blk;      // block row of this output row in block column 0, hi plane
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    return Row{inner.row(m), planes + (m >> 4) * ncb * 1024 + (m & 15) * 32};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    float x = v + r.in.wm * r.in.g[n];
    if (inner.drop.thresh != 0u) x *= inner.drop.mult(r.in.idx0 + (uint32_t)n);
    if (r.in.src != nullptr && !(r.in.src[n] > 0.0f)) x = 0.0f;
    uint32_t h, l;
    split_pair(x, 0.0f, h, l);                       // low halves = this element
    unsigned char* dst = r.blk + (n >> 4) * 1024 + (n & 15) * 2;
    *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(h & 0xFFFFu);
    *reinterpret_cast<uint16_t*>(dst + 512) = (uint16_t)(l & 0xFFFFu);
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return inner.vec_ok(); }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    v = inner.apply4(r.in, n, v);
    uint32_t h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    unsigned char* dst = r.blk + (n >> 4) * 1024 + (n & 15) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 512) = make_uint2(l0, l1);
  }
};

// EpiLinear that ALSO writes its output as (hi, lo) bf16 fragment-block planes over the real rows (KCPlanesG): y of the fused
// news path is read as fp32 by the pooling kernels and as a GEMM operand by the additive-attention forward and weight
// gradient.  The lane that owns the last four columns appends the ones column of the bias gradient (column n_cols) and
// zeros to the end of the block (n_cols % 16 == 12: the reference width 300).
struct EpiLinearPlanes {
  EpiLinear inner;
  unsigned char* planes;
  int ncb;
  struct Row {
    EpiLinear::Row in;
    unsigned char* blk;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    return Row{inner.row(m), planes + (m >> 4) * ncb * 1024 + (m & 15) * 32};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t m, int n, float v) const { inner(r.in, m, n, v); }   // (not used: vec_ok)
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return inner.vec_ok(); }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    v = inner.apply4(r.in, n, v);
    store4(r.in.out + n, v, inner.stream);
    uint32_t h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    unsigned char* dst = r.blk + (n >> 4) * 1024 + (n & 15) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 512) = make_uint2(l0, l1);
    if (n + 4 == inner.n_cols && (inner.n_cols & 15) == 12) {
      *reinterpret_cast<uint2*>(dst + 8) = make_uint2(0x3F80u, 0u);      // 1.0, 0, 0, 0
      *reinterpret_cast<uint2*>(dst + 512 + 8) = make_uint2(0u, 0u);
    }
  }
};

// A operands that deliver their fragments already split (KCPlanes, nrl_gemm.h)
template <class T, class = void>
struct RpPreSplit : std::false_type {};
template <class T>
struct RpPreSplit<T, std::enable_if_t<T::kPreSplit>> : std::true_type {};

// ---- kernel ------------------------------------------------------------------------------------------
// WAVES wavefronts x 16 RB rows each (RB = 2 row blocks; RB = 1 halves the MFMA work of a k-block per wave and doubles the
// workgroup count: for the few-row launches of the user encoder, where a panel's k-loop latency is the whole kernel);
// NBLK = ceil(N / 16) column blocks per wave; LDS ring of 2 k-block chunks.
template <int NBLK, int WAVES, class AOp, class Epi, int DEEP = 0, int RB = 2>
__global__ void __launch_bounds__(WAVES * 64, 2)
    rp_gemm_kernel(const AOp A, const uint16_t* __restrict__ img_base, const Epi epi, const int64_t M, const int N,
                   const int K, const int kblocks, const int64_t panel_elems, const int panels, const int panel_group,
                   const int64_t row_blocks) {
  // blockIdx.y = column panel of a WIDE output (N > 16 NBLK: nn.Linear layers of a transformer body, nrl_linear_fwd): panel p
  // owns columns [p * 16 NBLK, ...) and its own fragment-ordered image, `panel_elems` uint16 further on.  Row blocks are the
  // fast grid axis, so the workgroups in flight share one panel image (L2-resident) and stream the activation rows.
  //
  // Round 5, wide outputs: with the row blocks as the fast axis every panel re-streams the WHOLE activation (12 times at N = 3072:
  // 1.6 of the 2.8 GB a feed-forward GEMM of config 4 moved were those re-reads, profiles/r05_plm_lstur_pmc.txt, and its matrix pipe sat
  // at 0.40 against 0.52 for the 768-wide layers).  panel_group > 0: a 1-D grid, workgroup id -> (XCD = id % 8, place on that XCD);
  // an XCD owns the row blocks rb % 8 == xcd and walks them panel GROUP by panel group, the panel_group panels of one row block on
  // consecutive places -- they run on the same XCD at the same time, so the row block crosses HBM -> L2 once per group while the
  // group's images (panel_group x 0.79 MB at K = 768) stay L2-resident.
  int64_t rb_idx = blockIdx.x;
  int panel = (int)blockIdx.y;
  if (panel_group > 0) {
    const int64_t id = blockIdx.x;
    const int xcd = (int)(id % 8);
    const int64_t local = id / 8;
    const int64_t rx = (row_blocks + 7) / 8;                  // row blocks per XCD (the last ones of some XCDs do not exist)
    const int64_t per_group = rx * panel_group;
    const int64_t grp = local / per_group, rem = local % per_group;
    rb_idx = (rem / panel_group) * 8 + xcd;
    panel = (int)(grp * panel_group + rem % panel_group);
    if (rb_idx >= row_blocks || panel >= panels) return;      // (before any barrier: the whole workgroup leaves)
  }
  const uint16_t* __restrict__ img = img_base + (int64_t)panel * panel_elems;
  const int n_panel0 = panel * (NBLK * 16);
  constexpr int CHUNK = NBLK * 2048;          // bytes of one k-block of the image
  constexpr int PIECES = 2 * NBLK;            // 1-KiB pieces per chunk
  constexpr int G = (PIECES + WAVES - 1) / WAVES;
  static_assert(AOp::kLayout == SRC_KC, "A: fp32 k-contiguous rows");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CHUNK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t m0 = rb_idx * (WAVES * 16 * RB);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // this wave's share of a chunk: pieces wave, wave + WAVES, ... (clamped: a duplicate rewrites the same bytes)
  auto issue = [&](int kb, int slot) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(img) + (size_t)kb * CHUNK + lane * 16;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      int piece = wave + c * WAVES;
      piece = piece < PIECES ? piece : PIECES - 1;
      glds16_asm(src + piece * 1024, smem_base + (uint32_t)slot * CHUNK + (uint32_t)piece * 1024u);
    }
  };

  // (an operand over a compact list of live rows, KCPlanesLive: the grid is sized for the worst case; workgroups past the
  //  list's end leave together, before the first barrier, and the epilogue stores nothing past it)
  int64_t m_end = M;
  if constexpr (HasLiveRows<AOp>::value) {
    m_end = A.live_rows();
    if (m0 >= m_end) return;
  }
  typename AOp::State st[RB];
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
      if constexpr (RpPreSplit<AOp>::value) {
        ah[i] = __builtin_bit_cast(bf16x8, r[i][0]);       // the producer split once (hi | lo planes)
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

  // B fragments of column-block pair p are fetched while the 12 MFMAs of pair p - 1 run (two named fragment
  // sets; hipcc on its own fetches each fragment right before its first MFMA and waits lgkmcnt(0) there, which
  // exposes the LDS latency every 2-4 MFMAs).  Inside a pair the three split products go pass by pass over four
  // independent accumulators, so dependent MFMAs are 64 cycles apart.
  constexpr int NPAIR = (NBLK + 1) / 2;
  auto read_pair = [&](const unsigned char* base, int p, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * p + jj < NBLK ? 2 * p + jj : NBLK - 1;
      bh[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048);
      bl[jj] = *reinterpret_cast<const bf16x8*>(base + j * 2048 + 1024);
    }
  };
  auto mfma_pair = [&](int p, const bf16x8 (&ah)[RB], const bf16x8 (&al)[RB], const bf16x8 (&bh)[2],
                       const bf16x8 (&bl)[2]) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (2 * p + jj < NBLK)
#pragma unroll
          for (int i = 0; i < RB; ++i)
            acc[i][2 * p + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? al[i] : ah[i],
                                                                        pass == 0 ? bl[jj] : bh[jj], acc[i][2 * p + jj], 0, 0, 0);
  };
  auto mfma_chunk = [&](int slot, const bf16x8 (&ah)[RB], const bf16x8 (&al)[RB]) {
    const unsigned char* base = smem + slot * CHUNK + lane * 16;
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
    // pin the interleave: 4 fragment reads, then the 12 (last pair of an odd NBLK: 6) MFMAs they overlap with
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      if (p + 1 < NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      if (2 * p + 1 < NBLK) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * RB, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * RB, 0);
      }
    }
  };

  // k-loop.  The (hi, lo) fragments of k-block kb are loop-carried; the raw loads of k-block kb + 1 are issued at
  // the top of iteration kb and converted at its END, behind ~1800 cycles of MFMAs -- hipcc's own wait for them
  // (a vmcnt(0): it cannot see the asm DMA) then finds both the loads and the chunk DMA long landed.
  bf16x8 ah[RB], al[RB];
  if constexpr (DEEP) {
    // two k-blocks of raw A loads in flight (two NAMED register sets, loop unrolled by two): the rows of k-block
    // kb + 2 are requested at the top of iteration kb and converted at the end of iteration kb + 1.  The chunk DMA
    // of an iteration is issued BEFORE its four row loads, so `vmcnt(4)` at the top of the next iteration covers it.
    float4 ra[RB][2], rb[RB][2];
    auto clampk = [&](int k) { return k < kblocks ? k : kblocks - 1; };
    {
      float4 r0[RB][2];
      issue(0, 0);
      load_raw(0, r0);
      load_raw(clampk(1), ra);
      convert(0, r0, ah, al);
    }
    auto step = [&](int kb, float4 (&cur)[RB][2], float4 (&nxt)[RB][2]) {
      wait_vmcnt<4>();
      __builtin_amdgcn_s_barrier();
      issue(clampk(kb + 1), (kb + 1) & 1);
      load_raw(clampk(kb + 2), nxt);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(kb & 1, ah, al);
      __builtin_amdgcn_sched_barrier(0);
      convert(clampk(kb + 1), cur, ah, al);
    };
    for (int kb = 0; kb < kblocks; kb += 2) {
      step(kb, ra, rb);
      if (kb + 1 < kblocks) step(kb + 1, rb, ra);
    }
  } else {
    {
      float4 r0[RB][2];
      issue(0, 0);
      load_raw(0, r0);
      convert(0, r0, ah, al);
    }
    for (int kb = 0; kb < kblocks; ++kb) {
      // chunk kb has landed for THIS wave (it was issued one iteration ago) ...
      wait_vmcnt<0>();
      // ... and after the barrier for every wave; everyone has also finished reading the other slot
      __builtin_amdgcn_s_barrier();
      // the last iteration re-issues its own chunk into the idle slot and re-loads its own rows: uniform control
      // flow instead of a branch (one redundant chunk per workgroup)
      const int kn = kb + 1 < kblocks ? kb + 1 : kb;
      float4 r[RB][2];
      issue(kn, (kb + 1) & 1);
      load_raw(kn, r);
      __builtin_amdgcn_sched_barrier(0);
      mfma_chunk(kb & 1, ah, al);
      __builtin_amdgcn_sched_barrier(0);
      convert(kn, r, ah, al);
    }
  }

  store_accumulators<RB, NBLK>(epi, acc, m0, n_panel0, wave, 0, l15, g, m_end, N);
}

template <int NBLK, int WAVES = 4, int DEEP = 0, int RB = 2, class AOp, class Epi>
int launch_rp_gemm(const AOp& A, const RpImage& B, const Epi& epi, int64_t M, int N, int K, hipStream_t stream, int panels = 1) {
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  NRL_REQUIRE(B.img != nullptr && B.nblk == NBLK && N <= panels * NBLK * 16 && B.kblocks * 32 >= K && panels >= 1 && panels < 65536,
              "row-panel GEMM: image / shape mismatch");
  const int64_t blocks = ceil_div(M, WAVES * 16 * RB);
  NRL_REQUIRE(blocks < (1LL << 31), "gemm grid too large");
  // panel grouping of wide outputs (see the kernel): NRL_RP_PANEL_GROUP = 0 restores the row-blocks-fast 2-D grid (A/B runs)
  // (the group size must DIVIDE the panel count: a padded last group leaves holes in every XCD's walk -- 3 panels in groups of 2:
  //  0.214 ms against 0.165 at 38400 x 768 x 768)
  static const int group_env = [] { const char* e = getenv("NRL_RP_PANEL_GROUP"); return e ? atoi(e) : -1; }();
  int group = 0;
  if (panels > 1) {
    if (group_env >= 0) group = group_env;
    else group = panels % 3 == 0 ? 3 : (panels % 4 == 0 ? 4 : (panels % 2 == 0 ? 2 : 0));
    if (group > panels || (group > 0 && panels % group != 0)) group = 0;
  }
  if (group > 0) {
    const int64_t rx = ceil_div(blocks, 8), groups = ceil_div(panels, group);
    const int64_t grid = 8 * rx * group * groups;
    NRL_REQUIRE(grid < (1LL << 31), "gemm grid too large");
    hipLaunchKernelGGL((rp_gemm_kernel<NBLK, WAVES, AOp, Epi, DEEP, RB>), dim3((unsigned)grid), dim3(WAVES * 64), 0, stream, A, B.img,
                       epi, M, N, K, B.kblocks, (int64_t)rp_image_elems(NBLK, B.kblocks), panels, group, blocks);
  } else {
    hipLaunchKernelGGL((rp_gemm_kernel<NBLK, WAVES, AOp, Epi, DEEP, RB>), dim3((unsigned)blocks, (unsigned)panels), dim3(WAVES * 64), 0,
                       stream, A, B.img, epi, M, N, K, B.kblocks, (int64_t)rp_image_elems(NBLK, B.kblocks), panels, 0, blocks);
  }
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

static inline bool rp_nblk_supported(int N) { return N > 0 && N <= 320; }
// NBLK instantiations: 13 (N <= 208: the additive-attention projection, Q = 200), 19 (N <= 304: D = 300), 20
static inline int rp_nblk_for(int N) { return N <= 208 ? 13 : (N <= 304 ? 19 : 20); }

}  // namespace nrl
