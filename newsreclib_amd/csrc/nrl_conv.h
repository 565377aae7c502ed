// Operand accessors that turn the LSTUR/NAML text CNN (reference CNNAddAtt, text.py:112-176:
// nn.Conv2d(1, F, kernel (W, D), padding ((W-1)/2, 0)) over the token axis + ReLU) into the same
// tiled GEMMs as the NRMS projections -- no im2col buffer is ever materialised.  With rows
// m = news * L + token of a dense (M, inner) activation, the window of row m is the CONTIGUOUS
// range starting `pad` rows earlier, so a windowed operand is an overlapping-row matrix (ld = inner,
// K = W * inner) plus a mask for taps that cross a news boundary:
//
//   forward   c[m, f]       = b_f + sum_{t, d} x[m + t - pad, d] * Wc[f, t*D + d]        K = W*D
//   dgrad     dx[m, d]      = sum_{t', f} dc[m + t' - pad', f] * Wc[f, (W-1-t')*D + d]    K = W*F
//             (taps reversed, t' = W-1-t, pad' = W-1-pad, so the dc rows are contiguous too)
//   wgrad     dWc[f, t*D+d] = sum_m dc[m, f] * x[m + t - pad, d]                          K = M
//
// The dense buffers carry W*inner floats of slack on both sides (carved by the workspace), so the
// unconditional tile loads of the first / last rows stay inside the allocation.
#pragma once
#include "nrl_gemm.h"
#include "nrl_gemm_bf16x3.h"

namespace nrl {

// k-contiguous windowed rows: element (m, k = t*inner + j) = p[(m + t - pad) * inner + j], or 0 if
// token(m) + t - pad falls outside [0, L)
struct KCWindow {
  static constexpr int kLayout = SRC_KC;
  const float* p;  // dense (rows, inner), with slack
  int64_t rows;
  int inner, L, W, pad;
  struct State {
    const float* ptr;
    int lo, hi;  // valid k range [lo, hi) of this row (whole taps)
  };
  __device__ __forceinline__ State init(int64_t m) const {
    const bool ok = m < rows;
    const int64_t mm = ok ? m : 0;
    const int l = (int)(mm % L);
    // tap t valid iff 0 <= l + t - pad < L
    int t0 = pad - l;
    t0 = t0 < 0 ? 0 : t0;
    int t1 = L + pad - l;  // exclusive
    t1 = t1 > W ? W : t1;
    return State{p + (mm - pad) * (int64_t)inner, ok ? t0 * inner : 0, ok ? t1 * inner : 0};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const {
    return *reinterpret_cast<const float4*>(s.ptr + (k < K ? k : K - 4));
  }
  __device__ __forceinline__ const float* src(const State& s, int k, int K) const { return s.ptr + (k < K ? k : K - 4); }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    if (k >= kend || k < s.lo || k >= s.hi) v = f4zero();
  }
};

// KCWindow over x stored ONLY as (hi, lo) bf16 fragment-block planes padded per news (round 5; the layout the convolution weight
// gradient reads, nrl_wgrad_planes.h: block (news * nrb + row / 16, cb) at ((..) * ncb + cb) * 1024, hi plane then lo plane, row
// r % 16 at 32 (r % 16), rows >= L zero).  The reduction index is tap-padded: k' = tap * (16 ncb) + d, so with an even ncb a tap
// is ncb / 2 whole k-blocks and a lane's 8 consecutive k are 16 contiguous bytes of one plane row -- token row t + tap - pad of the
// same news, or nothing (a tap that crosses the news boundary).  No fp32 copy of x exists on this path and no split is done here.
struct KCWindowPlanes {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kPreSplit = true;
  const unsigned char* p;
  int64_t rows;            // real token rows n_news * L
  int L, nrb, ncb, W, pad;
  struct State {
    const unsigned char* news;   // first block of the row's news
    int t;                       // token position inside the news, or -(1 << 20): a row past the end
  };
  __device__ __forceinline__ State init(int64_t m) const {
    const bool ok = m < rows;
    const int64_t mm = ok ? m : 0;
    const int64_t n = mm / L;
    return State{p + n * nrb * (int64_t)ncb * 1024, ok ? (int)(mm - n * L) : -(1 << 20)};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int) const {
    const int kb = k >> 5, per_tap = ncb >> 1;
    const int tap = kb / per_tap;
    const int d = (k & ~4) - tap * (ncb << 4);
    int src = s.t + tap - pad;
    src = (src >= 0 && src < L) ? src : 0;                 // (an invalid tap reads a valid address; `finish` zeroes it)
    return *reinterpret_cast<const float4*>(s.news + ((src >> 4) * ncb + (d >> 4)) * 1024 + ((k & 4) ? 512 : 0) + (src & 15) * 32 +
                                            ((d >> 3) & 1) * 16);
  }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    const int tap = (k >> 5) / (ncb >> 1);
    const int src = s.t + tap - pad;
    if (k >= kend || src < 0 || src >= L) v = f4zero();
  }
};

// KCWindowPlanes over the LIVE token rows only (the convolution's activation gradient from dc planes; see KCWindowLive below)
struct KCWindowLivePlanes {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kPreSplit = true;
  static constexpr bool kLiveRows = true;
  KCWindowPlanes w;
  const int32_t* list;
  const int32_t* n_live;
  using State = KCWindowPlanes::State;
  __device__ __forceinline__ int64_t live_rows() const { return *n_live; }
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < *n_live;
    State s = w.init(list[ok ? r : 0]);
    if (!ok) s.t = -(1 << 20);
    return s;
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const { return w.load(s, k, K); }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t m, int k, int kend, bool b) const { w.finish(v, s, m, k, kend, b); }
};

// EpiPoolBwd (the additive attention's activation gradient with the pooling term, dropout and the ReLU gate) writing dc ONLY as
// (hi, lo) fragment-block planes padded per news -- block (news * nrb + t / 16, cb), the layout of KCWindowPlanes and of the
// convolution weight gradient -- instead of fp32 rows + a conversion launch.  The planes' padding must be zero (taps rotate into
// it): the piece that holds the last real column of a row also clears the block columns' tail; the pad ROWS of every news are
// cleared by planes_zero_pad_rows_kernel (nrl_wgrad_planes.h) in front of the launch.
struct EpiPoolBwdNewsPlanes {
  EpiPoolBwd inner;          // `c` unused; ldc = n_cols = row length of d_out / relu_src
  unsigned char* planes;
  int ncb, nrb, L, n_cols;
  struct Row {
    EpiPoolBwd::Row in;
    unsigned char* news;     // first block of the row's news
    int t;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    const int64_t n = m / L;
    return Row{inner.row(m), planes + n * nrb * (int64_t)ncb * 1024, (int)(m - n * L)};
  }
  __device__ __forceinline__ unsigned char* at(const Row& r, int t, int n) const {
    return r.news + ((t >> 4) * ncb + (n >> 4)) * 1024 + (t & 15) * 32 + (n & 15) * 2;
  }
  __device__ __forceinline__ void put4(unsigned char* dst, uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1) const {
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 512) = make_uint2(l0, l1);
  }
  // the piece that holds the last real columns of a row clears the tail of the block columns (<= 32 columns: a fixed, predicated
  // unroll -- a data-dependent loop here keeps hipcc from unrolling the output stage around it and sends the accumulators to scratch)
  __device__ __forceinline__ void pad_cols(const Row& r) const {
    const int c0 = (n_cols + 3) & ~3;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + 4 * q;
      if (c < 16 * ncb) put4(at(r, r.t, c), 0u, 0u, 0u, 0u);
    }
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    float x = v + r.in.wm * r.in.g[n];
    if (inner.drop.thresh != 0u) x *= inner.drop.mult(r.in.idx0 + (uint32_t)n);
    if (r.in.src != nullptr && !(r.in.src[n] > 0.0f)) x = 0.0f;
    uint32_t h, l;
    split_pair(x, 0.0f, h, l);
    unsigned char* dst = at(r, r.t, n);
    *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(h & 0xFFFFu);
    *reinterpret_cast<uint16_t*>(dst + 512) = (uint16_t)(l & 0xFFFFu);
    if (n == n_cols - 1) pad_cols(r);
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return inner.vec_ok() && (n_cols & 3) == 0; }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    v = inner.apply4(r.in, n, v);
    uint32_t h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    put4(at(r, r.t, n), h0, h1, l0, l1);
    if (n + 4 >= n_cols) pad_cols(r);
  }
};

// KCWindow over the LIVE token rows only (see KCPlanesLive, nrl_gemm.h): the convolution's activation gradient dx feeds
// nothing but the embedding-table scatter, which has no use for the rows of the padding id.  GEMM row r is the r-th live
// token position (`list`, position order).
struct KCWindowLive {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kLiveRows = true;
  KCWindow w;
  const int32_t* list;
  const int32_t* n_live;
  using State = KCWindow::State;
  __device__ __forceinline__ int64_t live_rows() const { return *n_live; }
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < *n_live;
    State s = w.init(list[ok ? r : 0]);
    if (!ok) { s.lo = 0; s.hi = 0; }
    return s;
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const { return w.load(s, k, K); }
  __device__ __forceinline__ const float* src(const State& s, int k, int K) const { return w.src(s, k, K); }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t m, int k, int kend, bool b) const { w.finish(v, s, m, k, kend, b); }
};

// k-major windowed source for the conv weight gradient: element (k = m, r = t*inner + j) is
// x[m + t - pad][j] (0 outside the news); `ones` appends the bias column as in RCPlain.
// n / d for n < 2^31 as one v_mul_hi_u32 + shift (round-up magic, 31-bit dividend): the windowed accessor below
// divides per 16-byte load, and hipcc's generic 32-bit division is ~40 VALU instructions -- in the loader waves of
// the weight-gradient kernels that was the whole k-tile budget
struct FastDiv {
  uint32_t magic, shift, d;   // q = umulhi(n, magic) >> shift   (d == 1: magic = 0, q = n)
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return magic == 0 ? n : (__umulhi(n, magic) >> shift); }
  __device__ __forceinline__ uint32_t mod(uint32_t n) const { return n - div(n) * d; }
};
static inline FastDiv make_fast_div(uint32_t d) {
  FastDiv f{0u, 0u, d};
  if (d <= 1) return f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;                           // l = ceil(log2 d)
  f.magic = (uint32_t)(((1ull << (31 + l)) / d) + 1);   // in [2^31, 2^32)
  f.shift = l - 1;                                      // (n * magic) >> (31 + l) = umulhi >> (l - 1)
  return f;
}

struct RCWindow {
  static constexpr int kLayout = SRC_RC;
  const float* x;  // (M, inner) with slack
  int inner, L, pad, ones;
  int64_t rows;    // = W * inner
  FastDiv divL, divI;   // by L and by inner (rc_window() fills them)
  struct State {};
  __device__ __forceinline__ float4 load(int64_t k, int64_t r, int64_t K) const {
    const int64_t kk = k < K ? k : K - 1;
    const int64_t rr = r < rows ? r : rows - 4;
    return *reinterpret_cast<const float4*>(x + (kk - pad) * inner + rr);
  }
  // token position of window element (k, r): (k mod L) + tap(r) - pad      (k < 2^31, r < rows)
  __device__ __forceinline__ int token(int64_t k, int64_t r) const {
    return (int)divL.mod((uint32_t)k) + (int)divI.div((uint32_t)r) - pad;
  }
  __device__ __forceinline__ void finish(float4& v, int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend) {
      v = f4zero();
      return;
    }
    if (r >= rows) {
      v = (ones && r == rows) ? make_float4(1.f, 0.f, 0.f, 0.f) : f4zero();
      return;
    }
    const int l = token(k, r);
    if (l < 0 || l >= L) v = f4zero();
  }
  __device__ __forceinline__ const float* src(int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend) return nrl_dma_zero16;
    if (r >= rows) return (ones && r == rows) ? nrl_dma_ones16 : nrl_dma_zero16;
    const int l = token(k, r);
    return (l < 0 || l >= L) ? nrl_dma_zero16 : x + (k - pad) * inner + r;
  }
};
static inline RCWindow rc_window(const float* x, int inner, int L, int pad, int ones, int64_t rows) {
  return RCWindow{x, inner, L, pad, ones, rows, make_fast_div((uint32_t)L), make_fast_div((uint32_t)inner)};
}

// conv weight Wc (F, W*D) as the (K = W*F, N = D) operand of the dgrad with reversed taps:
// element (k = t'*F + f, r = d) = Wc[f][(W-1-t')*D + d]
struct RCConvT {
  static constexpr int kLayout = SRC_RC;
  const float* w;
  int F, D, W;
  int64_t rows;  // = D
  struct State {};
  __device__ __forceinline__ float4 load(int64_t k, int64_t r, int64_t K) const {
    const int64_t kk = k < K ? k : K - 1;
    const int tr = (int)(kk / F), f = (int)(kk - (int64_t)tr * F);
    const int64_t rr = r < rows ? r : rows - 4;
    return *reinterpret_cast<const float4*>(w + ((int64_t)f * W + (W - 1 - tr)) * D + rr);
  }
  __device__ __forceinline__ void finish(float4& v, int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend || r >= rows) v = f4zero();
  }
};

}  // namespace nrl
