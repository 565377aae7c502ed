// Exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_16x16x4_f32), LDS-tiled, double-buffered.
//
//   C(m, n) = sum_k A(m, k) * B(k, n)      ->   epi(m, n, value)
//
// One kernel template serves every projection of the NRMS hot path (forward NT, dgrad NN, wgrad
// TN) through three pluggable pieces:
//   * operand accessors (how a tile is fetched from HBM): k-contiguous rows (optionally GATHERED
//     through token ids with the dropout mask applied = the fused embedding lookup), or k-major
//     ("row-contiguous") matrices;
//   * an epilogue functor applied to every valid output element (bias / tanh / dropout / pooled
//     gradient / atomic accumulate / embedding scatter-add);
//   * the tile shape: WM x WN waves, each owning TM x TN MFMA blocks of 16x16.
//
// LDS image: both operands are stored k-major, As[k][BM + pad], Bs[k][BN + pad], pad chosen so the
// per-MFMA fragment read (lanes 0-15 -> k, lanes 16-31 -> k+1, ...) is a conflict-free ds_read_b32.
// Fragment maps (cdna_hip_programming.md section 3): A lane l holds A[i = l&15][k = l>>4], B holds
// B[k = l>>4][j = l&15], C/D: col = l&15, row = 4*(l>>4) + reg.
//
// Why 16x16x4 and not 32x32x2: same 64 FLOP/clk/SIMD, but N in {900, 300, 200} pads to
// {912, 304, 208} with 16-wide blocks vs {928, 320, 224} with 32-wide ones.
#pragma once
#include <type_traits>

#include "nrl_common.h"

namespace nrl {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int GEMM_BK = 16;
enum { SRC_KC = 0, SRC_RC = 1 };

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// 16-byte store with the streaming (non-temporal) hint.  A write-once activation far larger than the 32 MB of L2 gains
// nothing from write-allocate: with plain stores the attention backward's 0.81 GB of dqkv slabs ran at 2.4 TB/s
// (0.57 ms for the kernel), with the hint the kernel's whole 1.9 GB of traffic moves at 5.3 TB/s (0.37 ms;
// tools/nf_probe.hip, profiles/r02_news_fused_probe.txt).
__device__ __forceinline__ void store4_stream(float* p, const float4& v) {
  typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(nt_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f32x4*>(p));
}
__device__ __forceinline__ void store4(float* p, const float4& v, int stream) {
  if (stream) store4_stream(p, v);
  else *reinterpret_cast<float4*>(p) = v;
}


// 16-byte constants an LDS-DMA lane is pointed at instead of predicating the load (out-of-range rows /
// k, and the virtual ones-column of a weight-gradient operand)
__device__ const float nrl_dma_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ const float nrl_dma_ones16[4] = {1.f, 0.f, 0.f, 0.f};

// ---------------------------------------------------------------------------------------------
// operand accessors
// ---------------------------------------------------------------------------------------------
// Loads are UNCONDITIONAL and branch-free: out-of-range rows / k are clamped to a valid address at
// load time and zeroed by `finish` when the staged registers are written to LDS (after the MFMA
// phase).  A predicated load makes hipcc wrap it in an exec-mask branch with an immediate
// `s_waitcnt vmcnt(0)`, which serialises the whole memory latency into the k-loop (measured:
// -13 % GEMM throughput).
//
// rows x K matrix, K contiguous (row-major activations, nn.Linear weights)
struct KCPlain {
  static constexpr int kLayout = SRC_KC;
  const float* p;
  int64_t ld;
  int64_t rows;
  struct State {
    const float* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    return State{p + (ok ? r : 0) * ld, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const {
    return *reinterpret_cast<const float4*>(s.ptr + (k < K ? k : K - 4));
  }
  // address form of `load` (LDS-DMA staging fetches it without touching VGPRs)
  __device__ __forceinline__ const float* src(const State& s, int k, int K) const { return s.ptr + (k < K ? k : K - 4); }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    if (!s.ok || k >= kend) v = f4zero();
  }
};

// three (rows x kp) row-major matrices side by side along k: element (row r, k) of [P0 | P1 | P2] lives at
// p[(k / kp) * part_stride + r * ld + k % kp] -- the activation gradient of three projections of ONE input (the query / key /
// value projections of a transformer layer) is a single GEMM over the concatenated reduction, round 5.  kp % 32 == 0: the lanes
// of one k-block share their part.
struct KCParts {
  static constexpr int kLayout = SRC_KC;
  const float* p;
  int64_t ld;
  int64_t rows;
  int kp;
  int64_t part_stride;
  struct State {
    const float* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    return State{p + (ok ? r : 0) * ld, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const {
    const int kc = k < K ? k : K - 4;
    const int64_t hop = part_stride - kp;
    return *reinterpret_cast<const float4*>(s.ptr + kc + (kc >= kp ? hop : 0) + (kc >= 2 * kp ? hop : 0));
  }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    if (!s.ok || k >= kend) v = f4zero();
  }
};

// "head-plane" activation of the fused news path (nrl_news_fused.h): logical element (row m, k' = head * 64 + c) of
// a (rows x heads * 64) matrix lives at p[(head * rows + m) * 64 + c] -- one (rows x 64) row-major plane per head,
// so the L rows x 64 floats of a (news, head) pair are ONE contiguous slab for the attention kernels that produce it
// (80-byte pieces of packed (rows, 3D) rows cost them 0.4 ms of partial-line writes at B = 128) and a k-block of 32
// is still 128 contiguous bytes per row for the GEMMs that consume it.
struct KCSlab {
  static constexpr int kLayout = SRC_KC;
  const float* p;
  int64_t rows;
  struct State {
    const float* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    return State{p + (ok ? r : 0) * 64, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const {
    const int kc = k < K ? k : K - 4;
    return *reinterpret_cast<const float4*>(s.ptr + (int64_t)(kc >> 6) * rows * 64 + (kc & 63));
  }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    if (!s.ok || k >= kend) v = f4zero();
  }
};

// dqkv of the fused news path as PRE-SPLIT fragment-block planes (nrl_wgrad_planes.h): token rows padded to 32 per
// news, block (head, mb = row / 16, cb) = [p][16 rows][16 features] bf16.  A row-panel lane's fragment (row, 8
// consecutive features) is 16 contiguous bytes of the hi plane and 16 of the lo plane: `load` returns the raw bits
// (k % 8 == 0: hi, k % 8 == 4: lo) and the kernel skips its split (kPreSplit).  Logical k = head * 64 + c.
struct KCPlanes {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kPreSplit = true;
  const unsigned char* p;
  int64_t rows;            // padded rows (n_news * 32)
  struct State {
    const unsigned char* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    const int64_t rr = ok ? r : 0;
    return State{p + (rr >> 4) * 4096 + (rr & 15) * 32, ok};     // + per-k terms in `load`
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int) const {
    // k = head * 64 + cb * 16 + half * 8 (+ 4: the lo plane)
    const int64_t off = (int64_t)(k >> 6) * (rows >> 4) * 4096 + ((k >> 4) & 3) * 1024 + ((k >> 3) & 1) * 16 + ((k & 4) ? 512 : 0);
    return *reinterpret_cast<const float4*>(s.ptr + off);
  }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int, int, bool) const {
    if (!s.ok) v = f4zero();
  }
};

// KCPlanes over the LIVE token rows only.  The last dgrad of the news path (dx = dqkv W_in) feeds nothing but the embedding-
// table scatter, and rows of the padding id 0 have no gradient there (nn.Embedding(padding_idx=0), text.py:215-217): with
// ~11.5 real tokens in a 30-token title 62 % of the rows are dead.  GEMM row r is the r-th live token position
// in POSITION order (`list`, live_compact in nrl_kernels.hip): the live tokens of a news are neighbours, so the 16 lanes of a
// row block read from one or two 16-row blocks of the planes (in id-sorted order every lane pulled its own cache lines:
// 1.27 GB of traffic for 0.33 GB of operands).
struct KCPlanesLive {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kPreSplit = true;
  static constexpr bool kLiveRows = true;
  const unsigned char* p;
  int64_t rows;            // padded rows (n_news * 32) of the planes
  const int32_t* list;     // live token positions news * L + t, ascending
  const int32_t* n_live;   // device scalar: their number
  int L;
  struct State {
    const unsigned char* ptr;
    bool ok;
  };
  __device__ __forceinline__ int64_t live_rows() const { return *n_live; }
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < *n_live;
    const int64_t m = list[ok ? r : 0];                          // token position news * L + t
    const int64_t news = m / L;
    const int64_t rr = news * 32 + (m - news * L);               // its padded row
    return State{p + (rr >> 4) * 4096 + (rr & 15) * 32, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int) const {
    const int64_t off = (int64_t)(k >> 6) * (rows >> 4) * 4096 + ((k >> 4) & 3) * 1024 + ((k >> 3) & 1) * 16 + ((k & 4) ? 512 : 0);
    return *reinterpret_cast<const float4*>(s.ptr + off);
  }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int, int, bool) const {
    if (!s.ok) v = f4zero();
  }
};
template <class T, class = void>
struct HasLiveRows : std::false_type {};
template <class T>
struct HasLiveRows<T, std::enable_if_t<T::kLiveRows>> : std::true_type {};

// plain fragment-block planes over real rows: block (mb = row / 16, cb = k / 16) at ((mb * ncb + cb) * 2 + p) * 512
// (the fused news path's `o` and `dy`, written by the fused forward / the row-panel epilogue EpiPoolBwdPlanes)
struct KCPlanesG {
  static constexpr int kLayout = SRC_KC;
  static constexpr bool kPreSplit = true;
  const unsigned char* p;
  int64_t rows;
  int ncb;
  struct State {
    const unsigned char* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    const int64_t rr = ok ? r : 0;
    return State{p + (rr >> 4) * ncb * 1024 + (rr & 15) * 32, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int) const {
    int cb = k >> 4;
    cb = cb < ncb ? cb : ncb - 1;                  // k-block tail past the last block column: the image rows there are zero
    return *reinterpret_cast<const float4*>(s.ptr + cb * 1024 + ((k >> 3) & 1) * 16 + ((k & 4) ? 512 : 0));
  }
  // (the block columns may carry unwritten bytes past the last feature: dy has 300 of 304 -- anything there must not
  //  reach the matrix cores, a NaN bit pattern times a zero weight is still NaN)
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t, int k, int kend, bool) const {
    const int k0 = k & ~7;                           // the chunk holds features k0 .. k0 + 7 (kend % 4 == 0)
    if (!s.ok || k0 >= kend) v = f4zero();
    else if (k0 + 4 >= kend) { v.z = 0.f; v.w = 0.f; }
  }
};

// rows gathered from an embedding table through int64 ids (nn.Embedding, text.py:224), times the
// dropout multiplier of text.py:225; the tile column 0 workgroup also saves the post-dropout row
// (needed by the in-projection weight gradient).  ids == nullptr means identity rows: a dense
// activation with an input dropout (the PLM text encoder's first dropout, text.py:92-93).
struct KCGather {
  static constexpr int kLayout = SRC_KC;
  const float* table;
  const int64_t* ids;
  int64_t rows;
  int dim;
  Dropout drop;
  float* save;  // (rows, dim) or nullptr
  struct State {
    const float* ptr;
    bool ok;
  };
  __device__ __forceinline__ State init(int64_t r) const {
    const bool ok = r < rows;
    const int64_t rr = ok ? r : 0;
    return State{table + (ids != nullptr ? ids[rr] : rr) * (int64_t)dim, ok};
  }
  __device__ __forceinline__ float4 load(const State& s, int k, int K) const {
    return *reinterpret_cast<const float4*>(s.ptr + (k < K ? k : K - 4));
  }
  __device__ __forceinline__ const float* src(const State& s, int k, int K) const { return s.ptr + (k < K ? k : K - 4); }
  __device__ __forceinline__ void finish(float4& v, const State& s, int64_t r, int k, int kend,
                                         bool primary) const {
    if (!s.ok || k >= kend) {
      v = f4zero();
      return;
    }
    if (drop.thresh != 0u) {
      const uint32_t idx = (uint32_t)r * (uint32_t)dim + (uint32_t)k;
      v.x *= drop.mult(idx);
      v.y *= drop.mult(idx + 1);
      v.z *= drop.mult(idx + 2);
      v.w *= drop.mult(idx + 3);
    }
    if (save != nullptr && primary) *reinterpret_cast<float4*>(save + r * (int64_t)dim + k) = v;
  }
};

// K x rows matrix, rows contiguous (element (k, r) at p[k * ld + r]); `ones` appends a virtual
// column r == rows that reads 1.0 (turns a weight-gradient GEMM into weight + bias gradient).
struct RCPlain {
  static constexpr int kLayout = SRC_RC;
  const float* p;
  int64_t ld;
  int64_t rows;
  int ones;
  struct State {};  // no per-row state (keeps the kernel's staging arrays uniform)
  __device__ __forceinline__ float4 load(int64_t k, int64_t r, int64_t K) const {
    return *reinterpret_cast<const float4*>(p + (k < K ? k : K - 1) * ld + (r < rows ? r : rows - 4));
  }
  __device__ __forceinline__ void finish(float4& v, int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend)
      v = f4zero();
    else if (r >= rows)
      v = (ones && r == rows) ? make_float4(1.f, 0.f, 0.f, 0.f) : f4zero();
  }
  // address form of load + finish (LDS-DMA staging): where to fetch the 16 bytes of (k, r .. r + 3) from
  __device__ __forceinline__ const float* src(int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend) return nrl_dma_zero16;
    if (r < rows) return p + k * ld + r;
    return (ones && r == rows) ? nrl_dma_ones16 : nrl_dma_zero16;
  }
  // the same matrix as "column pointer + k * stride" (nrl_gemm_ws.h's loaders: the row clamp, the 64-bit multiply and the edge
  // selects once per task instead of once per 16 bytes): (k, r .. r + 3) at col(r) + k * kstride() while live(r), else (fill(r), 0, 0, 0)
  __device__ __forceinline__ const float* col(int64_t r, int64_t) const { return p + (r < rows ? r : rows - 4); }
  __device__ __forceinline__ int64_t kstride(int64_t) const { return ld; }
  __device__ __forceinline__ bool live(int64_t r) const { return r < rows; }
  __device__ __forceinline__ float fill(int64_t r) const { return (ones && r == rows) ? 1.f : 0.f; }
};

// the same head-plane matrix read k-major (weight gradient: k = activation row, r = logical column head * 64 + c)
struct RCSlab {
  static constexpr int kLayout = SRC_RC;
  const float* p;
  int64_t rows;   // logical columns: heads * 64
  struct State {};
  __device__ __forceinline__ float4 load(int64_t k, int64_t r, int64_t K) const {
    const int64_t rc = r < rows ? r : rows - 4;
    return *reinterpret_cast<const float4*>(p + ((rc >> 6) * K + (k < K ? k : K - 1)) * 64 + (rc & 63));
  }
  __device__ __forceinline__ void finish(float4& v, int64_t k, int64_t r, int64_t kend) const {
    if (k >= kend || r >= rows) v = f4zero();
  }
  __device__ __forceinline__ const float* col(int64_t r, int64_t K) const {
    const int64_t rc = r < rows ? r : rows - 4;
    return p + (rc >> 6) * K * 64 + (rc & 63);
  }
  __device__ __forceinline__ int64_t kstride(int64_t) const { return 64; }
  __device__ __forceinline__ bool live(int64_t r) const { return r < rows; }
  __device__ __forceinline__ float fill(int64_t) const { return 0.f; }
};

// ---------------------------------------------------------------------------------------------
// epilogues: `row(m)` is evaluated once per output row a lane owns (per-row loads / index math),
// `operator()(row_state, m, n, v)` once per valid element.
// ---------------------------------------------------------------------------------------------
struct NoRow {};

// C = act(v + bias[n]) * dropout(m, n)      (nn.Linear / conv + optional tanh / relu + optional nn.Dropout)
struct EpiLinear {
  float* c;
  int64_t ldc;
  const float* bias;  // may be null
  int act;            // 0 none, 1 tanh, 2 relu
  Dropout drop;
  int n_cols;  // logical row width for the dropout flat index
  const float* gate_src = nullptr;  // (M, N) saved post-ReLU activation: output zeroed where gate_src <= 0
                                    // (a dgrad flowing back into a ReLU, CNNMHSAAddAtt text.py:297-299)
  int stream = 0;                   // 1: streaming stores (store4_stream)
  struct Row {
    float* out;
    const float* gate;
    uint32_t idx0;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    return Row{c + m * ldc, gate_src != nullptr ? gate_src + m * ldc : nullptr, (uint32_t)m * (uint32_t)n_cols};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    if (bias != nullptr) v += bias[n];
    if (act == 1) v = tanhf(v);
    if (act == 2) v = fmaxf(v, 0.0f);
    if (drop.thresh != 0u) v *= drop.mult(r.idx0 + (uint32_t)n);
    if (r.gate != nullptr && !(r.gate[n] > 0.0f)) v = 0.0f;
    r.out[n] = v;
  }
  // four consecutive columns n .. n + 3 of one row (n % 4 == 0): one 16-B store, one 16-B bias load
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const {
    return (ldc & 3) == 0 && (((uintptr_t)c | (uintptr_t)bias | (uintptr_t)gate_src) & 15) == 0;
  }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const { store4(r.out + n, apply4(r, n, v), stream); }
  __device__ __forceinline__ float4 apply4(const Row& r, int n, float4 v) const {
    if (bias != nullptr) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (act == 1) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
    if (act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (drop.thresh != 0u) {
      const uint32_t idx = r.idx0 + (uint32_t)n;
      v.x *= drop.mult(idx); v.y *= drop.mult(idx + 1); v.z *= drop.mult(idx + 2); v.w *= drop.mult(idx + 3);
    }
    if (r.gate != nullptr) {
      const float4 sv = *reinterpret_cast<const float4*>(r.gate + n);
      v.x = sv.x > 0.f ? v.x : 0.f; v.y = sv.y > 0.f ? v.y : 0.f;
      v.z = sv.z > 0.f ? v.z : 0.f; v.w = sv.w > 0.f ? v.w : 0.f;
    }
    return v;
  }
};

// dx of the live rows, COMPACT in position order: GEMM row r (token position list[r]) -> c[r]; the dropout mask is the one
// of the token position (text.py:225 undone).  embedding_grad_sorted finds the row of a position through cidx.
struct EpiDxLive {
  float* c;
  int64_t ldc;
  Dropout drop;
  const int32_t* list;
  struct Row {
    float* out;
    uint32_t idx0;
  };
  __device__ __forceinline__ Row row(int64_t r) const {
    return Row{c + r * ldc, (uint32_t)list[r] * (uint32_t)ldc};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n_, float v) const {
    if (drop.thresh != 0u) v *= drop.mult(r.idx0 + (uint32_t)n_);
    r.out[n_] = v;
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return (ldc & 3) == 0 && ((uintptr_t)c & 15) == 0; }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n_, float4 v) const {
    if (drop.thresh != 0u) {
      const uint32_t idx = r.idx0 + (uint32_t)n_;
      v.x *= drop.mult(idx); v.y *= drop.mult(idx + 1); v.z *= drop.mult(idx + 2); v.w *= drop.mult(idx + 3);
    }
    store4(r.out + n_, v, 0);
  }
};

// dgrad into a plain buffer
struct EpiStore {
  float* c;
  int64_t ldc;
  int stream = 0;
  struct Row {
    float* out;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{c + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const { r.out[n] = v; }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return (ldc & 3) == 0 && ((uintptr_t)c & 15) == 0; }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const { store4(r.out + n, v, stream); }
};

// GELU (exact: x Phi(x), the `gelu` of a BERT-family feed-forward block, text.py:89 -> HF RobertaIntermediate) around nn.Linear,
// round 5.  Forward epilogue: h = v + bias is stored for the backward AND g = gelu(h) for the next projection -- the framework's
// separate GELU kernel re-read h (0.52 GB per layer at 42k x 3072).  Backward epilogue of the NEXT projection's activation gradient:
// d_h = v * gelu'(h) with v = d_g the product, instead of writing d_g and a GeluBackward pass over it.
// erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) on v_rcp_f32 / v_exp_f32: ~20 issue slots per
// element against ~50 for libm's erff -- with erff in the epilogues the fusion was a wash (89 vs 88-89 ms per config-4 step: the
// epilogues' VALU cost what the two framework kernels had taken).  e2 = exp(-z^2), z = x / sqrt(2), serves BOTH the erf and the
// density term of the derivative (exp(-x^2 / 2) is the same number).
__device__ __forceinline__ float gelu_cdf_e2(float x, float& e2) {
  const float z = x * 0.70710678118654752440f;
  e2 = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, fabsf(z), 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = 1.0f - p * t * e2;                    // erf(|z|)
  return 0.5f * (1.0f + copysignf(erf_abs, z));
}
__device__ __forceinline__ float gelu_f(float x) {
  float e2;
  return x * gelu_cdf_e2(x, e2);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e2;
  const float cdf = gelu_cdf_e2(x, e2);
  return fmaf(x * 0.39894228040143267794f, e2, cdf);
}
struct EpiLinearGelu {
  float* h;            // pre-activation (M, N), saved
  float* g;            // gelu(h) (M, N)
  int64_t ldc;
  const float* bias;
  struct Row {
    float *h, *g;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{h + m * ldc, g + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    v += bias[n];
    r.h[n] = v;
    r.g[n] = gelu_f(v);
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const {
    return (ldc & 3) == 0 && (((uintptr_t)h | (uintptr_t)g | (uintptr_t)bias) & 15) == 0;
  }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    store4(r.h + n, v, 0);
    store4(r.g + n, make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w)), 0);
  }
};
struct EpiGeluBwd {
  float* c;            // d_h (M, N)
  int64_t ldc;
  const float* pre;    // h (M, N): the GELU's input
  struct Row {
    float* out;
    const float* pre;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{c + m * ldc, pre + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const { r.out[n] = v * gelu_grad_f(r.pre[n]); }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return (ldc & 3) == 0 && (((uintptr_t)c | (uintptr_t)pre) & 15) == 0; }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    const float4 p = *reinterpret_cast<const float4*>(r.pre + n);
    store4(r.out + n, make_float4(v.x * gelu_grad_f(p.x), v.y * gelu_grad_f(p.y), v.z * gelu_grad_f(p.z), v.w * gelu_grad_f(p.w)), 0);
  }
};

// three projections of one input in ONE GEMM (round 5): column n of [C0 | C1 | C2], part q = n / np, goes to row m of the
// (M, np) matrix at c + q * part_stride with that part's bias -- the attention kernels read q, k, v as three plain matrices
struct EpiLinearParts {
  float* c;
  int64_t part_stride;
  int np;
  const float* b0;
  const float* b1;
  const float* b2;
  struct Row {
    float* out;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{c + m * np}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    const int q = (n >= np ? 1 : 0) + (n >= 2 * np ? 1 : 0), nn = n - q * np;
    const float* b = q == 0 ? b0 : (q == 1 ? b1 : b2);
    r.out[q * part_stride + nn] = v + b[nn];
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const {
    return (np & 3) == 0 && (part_stride & 3) == 0 && (((uintptr_t)c | (uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0;
  }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    const int q = (n >= np ? 1 : 0) + (n >= 2 * np ? 1 : 0), nn = n - q * np;
    const float* b = q == 0 ? b0 : (q == 1 ? b1 : b2);
    const float4 bv = *reinterpret_cast<const float4*>(b + nn);
    store4(r.out + q * part_stride + nn, make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w), 0);
  }
};
// c = v + add: an activation gradient that lands on a tensor with a second consumer (the residual stream of a transformer layer)
// takes the other branch's gradient in the epilogue instead of leaving the sum to a framework kernel
struct EpiAddStore {
  float* c;
  int64_t ldc;
  const float* add;    // (M, N), same leading dimension
  struct Row {
    float* out;
    const float* add;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{c + m * ldc, add + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const { r.out[n] = v + r.add[n]; }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const { return (ldc & 3) == 0 && (((uintptr_t)c | (uintptr_t)add) & 15) == 0; }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    const float4 a = *reinterpret_cast<const float4*>(r.add + n);
    store4(r.out + n, make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w), 0);
  }
};

// additive-attention backward, fused: dy = (v + w[m] * d_out[group(m)][n]) * dropout(m, n)
// (v = d_pre * W_a is the tanh-branch gradient, w * d_out the weighted-sum branch,
// attention.py:34-40; the multiplier undoes text.py:230).
struct EpiPoolBwd {
  float* c;
  int64_t ldc;
  const float* w;      // (M) softmax weights
  const float* d_out;  // (groups, N)
  int group_len;       // rows per group
  Dropout drop;
  const float* relu_src = nullptr;  // (M, N) post-ReLU(-dropout) activation: gradient gated by src > 0
  int stream = 0;
  struct Row {
    float* out;
    const float* g;
    const float* src;
    float wm;
    uint32_t idx0;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    return Row{c + m * ldc, d_out + (m / group_len) * ldc, relu_src != nullptr ? relu_src + m * ldc : nullptr, w[m],
               (uint32_t)m * (uint32_t)ldc};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    v += r.wm * r.g[n];
    if (drop.thresh != 0u) v *= drop.mult(r.idx0 + (uint32_t)n);
    if (r.src != nullptr && !(r.src[n] > 0.0f)) v = 0.0f;
    r.out[n] = v;
  }
  static constexpr bool kVec4 = true;
  __device__ __forceinline__ bool vec_ok() const {
    return (ldc & 3) == 0 && (((uintptr_t)c | (uintptr_t)d_out | (uintptr_t)relu_src) & 15) == 0;
  }
  __device__ __forceinline__ float4 apply4(const Row& r, int n, float4 v) const {
    const float4 gv = *reinterpret_cast<const float4*>(r.g + n);
    v.x = fmaf(r.wm, gv.x, v.x); v.y = fmaf(r.wm, gv.y, v.y); v.z = fmaf(r.wm, gv.z, v.z); v.w = fmaf(r.wm, gv.w, v.w);
    if (drop.thresh != 0u) {
      const uint32_t idx = r.idx0 + (uint32_t)n;
      v.x *= drop.mult(idx); v.y *= drop.mult(idx + 1); v.z *= drop.mult(idx + 2); v.w *= drop.mult(idx + 3);
    }
    if (r.src != nullptr) {
      const float4 sv = *reinterpret_cast<const float4*>(r.src + n);
      v.x = sv.x > 0.f ? v.x : 0.f; v.y = sv.y > 0.f ? v.y : 0.f;
      v.z = sv.z > 0.f ? v.z : 0.f; v.w = sv.w > 0.f ? v.w : 0.f;
    }
    return v;
  }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const { store4(r.out + n, apply4(r, n, v), stream); }
  // The two operand loads of a 4-column piece, issued AHEAD of the arithmetic by the output stage (kPrefetch): as written above
  // every piece waits for its own loads -- 38 dependent HBM round trips per wave behind a 7-k-block product (K = query_dim),
  // which is what the launch then costs (MFMA-busy 0.14 at 2.9 TB/s, profiles/r04_plm_lstur_pmc.txt).
  static constexpr bool kPrefetch = true;
  struct Pre {
    float4 g, s;
  };
  __device__ __forceinline__ Pre prefetch(const Row& r, int n) const {
    Pre p;
    p.g = *reinterpret_cast<const float4*>(r.g + n);
    p.s = r.src != nullptr ? *reinterpret_cast<const float4*>(r.src + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    return p;
  }
  __device__ __forceinline__ void vec4_pre(const Row& r, int64_t, int n, float4 v, const Pre& p) const {
    v.x = fmaf(r.wm, p.g.x, v.x); v.y = fmaf(r.wm, p.g.y, v.y); v.z = fmaf(r.wm, p.g.z, v.z); v.w = fmaf(r.wm, p.g.w, v.w);
    if (drop.thresh != 0u) {
      const uint32_t idx = r.idx0 + (uint32_t)n;
      v.x *= drop.mult(idx); v.y *= drop.mult(idx + 1); v.z *= drop.mult(idx + 2); v.w *= drop.mult(idx + 3);
    }
    if (r.src != nullptr) {
      v.x = p.s.x > 0.f ? v.x : 0.f; v.y = p.s.y > 0.f ? v.y : 0.f;
      v.z = p.s.z > 0.f ? v.z : 0.f; v.w = p.s.w > 0.f ? v.w : 0.f;
    }
    store4(r.out + n, v, stream);
  }
};

// out += v with atomics: split-K partial sums of the small per-step GRU GEMMs (the k-loop of a 128-row
// GEMM is latency bound, so K is cut into ~200-wide slices that run on different CUs)
struct EpiAtomicAdd {
  float* c;
  int64_t ldc;
  struct Row {
    float* out;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{c + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const { atomicAdd(r.out + n, v); }
};

// split-K weight gradient: dW[m][n] += v for n < n_w, bias gradient db[m] += v for n == n_w
struct EpiAtomicWB {
  float* dw;
  int64_t ldc;
  float* db;  // may be null
  int n_w;
  struct Row {
    float* out;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{dw + m * ldc}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t m, int n, float v) const {
    if (n < n_w)
      atomicAdd(r.out + n, v);
    else if (n == n_w && db != nullptr)
      atomicAdd(db + m, v);
  }
};

// EpiAtomicWB whose output COLUMN runs over the head-permuted slots of the `o` planes (NewsFusedArgs::o_planes): slot
// n = 16 cb + c is feature head * 20 + d, the first free slot after the last head is the ones column (bias gradient)
struct EpiAtomicWBPerm {
  float* dw;
  int64_t ldc;
  float* db;  // may be null
  int heads;
  struct Row {
    float* out;
    int64_t m;
  };
  __device__ __forceinline__ Row row(int64_t m) const { return Row{dw + m * ldc, m}; }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    const int cb = n >> 4, c = n & 15;
    int head, d;
    if (cb < heads) { head = cb; d = c; }
    else { head = 4 * (cb - heads) + (c >> 2); d = 16 + (c & 3); }
    if (head < heads)
      atomicAdd(r.out + head * 20 + d, v);
    else if (head == heads && d == 16 && db != nullptr)
      atomicAdd(db + r.m, v);
  }
};

// EpiAtomicWB for a head-plane left operand: output row m = head * 64 + c is row part * (heads * dh) + head * dh + d
// of the packed in-projection weight gradient (c = part * dh + d < 3 dh; the 64 - 3 dh pad rows are dropped)
struct EpiAtomicWBHeads {
  float* dw;
  int64_t ldc;
  float* db;  // may be null
  int n_w;
  int heads, dh;
  struct Row {
    float* out;
    int64_t m;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    const int head = (int)(m >> 6), c = (int)(m & 63);
    if (head >= heads || c >= 3 * dh) return Row{nullptr, 0};
    const int part = c / dh, d = c - part * dh;
    const int64_t r = (int64_t)part * heads * dh + head * dh + d;
    return Row{dw + r * ldc, r};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    if (r.out == nullptr) return;
    if (n < n_w)
      atomicAdd(r.out + n, v);
    else if (n == n_w && db != nullptr)
      atomicAdd(db + r.m, v);
  }
};

// embedding_dense_backward fused into the in-projection dgrad: d_table[ids[m]][n] += v * dropout
// (padding_idx = 0 rows receive nothing, text.py:215-217)
struct EpiScatter {
  float* d_table;
  const int64_t* ids;
  int dim;
  Dropout drop;
  struct Row {
    float* out;  // null for the pad token
    uint32_t idx0;
  };
  __device__ __forceinline__ Row row(int64_t m) const {
    const int64_t id = ids[m];
    return Row{id == 0 ? nullptr : d_table + id * (int64_t)dim, (uint32_t)m * (uint32_t)dim};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    if (r.out == nullptr) return;
    if (drop.thresh != 0u) v *= drop.mult(r.idx0 + (uint32_t)n);
    atomicAdd(r.out + n, v);
  }
};

// ---------------------------------------------------------------------------------------------
// shared output stage of the MFMA kernels
// ---------------------------------------------------------------------------------------------
template <class Epi, class = void>
struct EpiHasVec4 : std::false_type {};
template <class Epi>
struct EpiHasVec4<Epi, std::enable_if_t<Epi::kVec4>> : std::true_type {};
// epilogues whose pieces read operands of their own (EpiPoolBwd): prefetch(row, n) / vec4_pre(row, m, n, v, pre)
template <class Epi, class = void>
struct EpiHasPrefetch : std::false_type {};
template <class Epi>
struct EpiHasPrefetch<Epi, std::enable_if_t<Epi::kPrefetch>> : std::true_type {};

// Output rows of a GEMM over PADDED token rows (32 per news, KCPlanes) mapped back to the real rows news * L + t;
// the pad rows t >= L produce nothing.
template <class Epi>
struct EpiNewsRows {
  Epi inner;
  int L;
  struct Row {
    typename Epi::Row in;
    int64_t m;
    bool ok;
  };
  __device__ __forceinline__ Row row(int64_t mp) const {
    const int t = (int)(mp & 31);
    const bool ok = t < L;
    const int64_t m = (mp >> 5) * L + (ok ? t : 0);
    return Row{inner.row(m), m, ok};
  }
  __device__ __forceinline__ void operator()(const Row& r, int64_t, int n, float v) const {
    if (r.ok) inner(r.in, r.m, n, v);
  }
  static constexpr bool kVec4 = EpiHasVec4<Epi>::value;
  __device__ __forceinline__ bool vec_ok() const {
    if constexpr (EpiHasVec4<Epi>::value) return inner.vec_ok();
    else return false;
  }
  __device__ __forceinline__ void vec4(const Row& r, int64_t, int n, float4 v) const {
    if constexpr (EpiHasVec4<Epi>::value) {
      if (r.ok) inner.vec4(r.in, r.m, n, v);
    }
  }
};


// 4 x 4 transpose across the four lanes of a quad (DPP quad_perm): lane t of the quad holds column t of
// rows 0..3 in a[0..3] on entry and row t, columns 0..3 on exit.
__device__ __forceinline__ void quad_transpose(float (&a)[4], int t) {
  const bool o1 = t & 1, o2 = t & 2;
  {
    const float s01 = o1 ? a[0] : a[1], s23 = o1 ? a[2] : a[3];
    const float r01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
    const float r23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
    if (o1) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
  }
  {
    const float s02 = o2 ? a[0] : a[2], s13 = o2 ? a[1] : a[3];
    const float r02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
    const float r13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
    if (o2) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
  }
}

// The MFMA result layout gives a lane C[row = 4g + r][col = l15] of every 16 x 16 block: a row-per-lane
// store would be 4 bytes wide, and global stores are issue bound (~170 cycles per store instruction
// measured: 80 dword stores per thread were 15 % of a K = 300 GEMM).  Epilogues that can take four
// consecutive columns (`vec4`) get the block transposed inside each lane quad first, so a lane owns
// C[4g + t][4q .. 4q + 3] and issues ONE 16-byte store (and 16-byte loads of bias / pooled gradient /
// ReLU source) per block: 4x fewer memory instructions, same bytes, same addresses.
template <int TM, int TN, class Epi>
__device__ __forceinline__ void store_accumulators(const Epi& epi, f32x4 (&acc)[TM][TN], int64_t m0, int n0, int wm,
                                                   int wn, int l15, int g, int64_t M, int N) {
  if constexpr (EpiHasVec4<Epi>::value) {
    if ((N & 3) == 0 && epi.vec_ok()) {
      const int t = l15 & 3, q4 = l15 & ~3;
      if constexpr (EpiHasPrefetch<Epi>::value) {
        constexpr int CH = 7;                           // pieces whose operand loads are in flight together (14 x 16 B per lane)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int64_t m = m0 + (wm * TM + i) * 16 + 4 * g + t;
          const bool mok = m < M;
          const typename Epi::Row rs = epi.row(mok ? m : 0);
#pragma unroll
          for (int j0 = 0; j0 < TN; j0 += CH) {
            typename Epi::Pre pre[CH];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
              const int n = n0 + (wn * TN + j0 + jj) * 16 + q4;
              if (j0 + jj < TN && mok && n < N) pre[jj] = epi.prefetch(rs, n);
            }
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
              if (j0 + jj < TN) {
                float a[4] = {acc[i][j0 + jj][0], acc[i][j0 + jj][1], acc[i][j0 + jj][2], acc[i][j0 + jj][3]};
                quad_transpose(a, t);
                const int n = n0 + (wn * TN + j0 + jj) * 16 + q4;
                if (mok && n < N) epi.vec4_pre(rs, m, n, make_float4(a[0], a[1], a[2], a[3]), pre[jj]);
              }
            }
          }
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + (wm * TM + i) * 16 + 4 * g + t;
        const bool mok = m < M;
        const typename Epi::Row rs = epi.row(mok ? m : 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float a[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          quad_transpose(a, t);                       // all lanes take part (DPP), masked only at the store
          const int n = n0 + (wn * TN + j) * 16 + q4;
          if (mok && n < N) epi.vec4(rs, m, n, make_float4(a[0], a[1], a[2], a[3]));
        }
      }
      return;
    }
  }
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

// Split-K reduction in two steps.  With one atomic per (split, output element) a launch issues 6 - 10 M fp32 atomics on
// 0.2 - 0.3 M addresses: ~60 us at the L2's atomic rate, whatever the batch (a third of the kernel at B = 32).  Instead every
// workgroup stores its partial tile (16-byte coalesced stores, 26 - 42 MB per launch) and wgrad_reduce_kernel sums the
// splits of each element in a fixed order and hands ONE value per element to the epilogue.
template <int BM, int BN, class Epi>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ scratch, int nsplit, int tiles_n, int tiles_total,
                                                           int64_t M, int N, const Epi epi) {
  constexpr int Q = BM * BN / 4;                             // float4 per tile
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)tiles_total * Q) return;
  const int t = (int)(i / Q), q = (int)(i - (int64_t)t * Q);
  const int r = q / (BN / 4), c = 4 * (q - r * (BN / 4));
  const float4* src = reinterpret_cast<const float4*>(scratch) + (int64_t)t * Q + q;
  // eight partial tiles in flight per lane, summed in split order (a fixed order: the result does not depend on timing)
  const int64_t stride = (int64_t)tiles_total * Q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 8 <= nsplit; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  for (; s < nsplit; ++s) {
    const float4 v = src[s * stride];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int64_t m = (int64_t)(t / tiles_n) * BM + r;
  const int n = (t % tiles_n) * BN + c;
  if (m < M) {
    const typename Epi::Row row = epi.row(m);
    if (n < N) epi(row, m, n, a.x);
    if (n + 1 < N) epi(row, m, n + 1, a.y);
    if (n + 2 < N) epi(row, m, n + 2, a.z);
    if (n + 3 < N) epi(row, m, n + 3, a.w);
  }
}

template <int BM, int BN, class Epi>
static inline int launch_wgrad_reduce(const float* scratch, int nsplit, int tiles_n, int64_t tiles_total, int64_t M, int N,
                                      const Epi& epi, hipStream_t st) {
  const int64_t threads = tiles_total * BM * BN / 4;
  hipLaunchKernelGGL((wgrad_reduce_kernel<BM, BN, Epi>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, scratch,
                     nsplit, tiles_n, (int)tiles_total, M, N, epi);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int ROWS>
struct LdsLd {
  static constexpr int value = (ROWS + 31) / 32 * 32 + 16;  // == 16 (mod 32): conflict-free b32 reads
};

// ABL: ablation bits for tools/gemm_probe.hip only (1 no barrier, 2 no LDS stores, 4 no global
// loads, 8 no fragment reads in the loop); product code always uses 0.
template <int WM, int WN, int TM, int TN, int BK, class AOp, class BOp, class Epi, int ABL = 0>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_f32_kernel(const AOp A, const BOp B, const Epi epi, const int64_t M, const int N,
                    const int64_t K, const int tiles_n, const int64_t tiles_total,
                    const int64_t k_per_split) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int CPK = BK / 4;  // float4 chunks per row of a k-contiguous tile
  constexpr int LDA = LdsLd<BM>::value, LDB = LdsLd<BN>::value;
  constexpr int NCH_A = (BM * CPK + NT - 1) / NT;
  constexpr int NCH_B = (BN * CPK + NT - 1) / NT;
  static_assert(BK % 4 == 0 && BM % 4 == 0 && BN % 4 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
  float* const As = smem;
  float* const Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // readfirstlane makes the wave index (and everything derived from it) provably wave-uniform,
  // so the block-validity tests below compile to scalar branches instead of exec-mask juggling
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;

  // XCD-aware tile order: workgroups that land on one XCD (blockIdx % 8) walk consecutive tiles,
  // so the n-tiles that share an A row-panel hit the same private L2 (bijective remap).
  int64_t t;
  {
    const int64_t bid = blockIdx.x;
    const int64_t q = tiles_total / 8, rem = tiles_total % 8;
    const int64_t xcd = bid % 8, local = bid / 8;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int64_t m0 = (t / tiles_n) * BM;
  const int n0 = (int)(t % tiles_n) * BN;
  const bool primary = (n0 == 0);
  const int64_t kbeg = (int64_t)blockIdx.y * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  if (kbeg >= kend) return;
  const int ntiles = (int)((kend - kbeg + BK - 1) / BK);

  // how many of this wave's 16x16 blocks hold any valid output (wave-uniform scalars)
  int nvi = 0, nvj = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) nvi += (m0 + (wm * TM + i) * 16 < M) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) nvj += (n0 + (wn * TN + j) * 16 < N) ? 1 : 0;
  const bool full = (nvi == TM) && (nvj == TN);

  // per-thread staging assignment (fixed across k-tiles)
  typename AOp::State sa[NCH_A];
  typename BOp::State sb[NCH_B];
  if constexpr (AOp::kLayout == SRC_KC) {
#pragma unroll
    for (int c = 0; c < NCH_A; ++c) {
      const int ch = tid + c * NT;
      sa[c] = A.init(ch < BM * CPK ? m0 + ch / CPK : (int64_t)1 << 60);
    }
  }
  if constexpr (BOp::kLayout == SRC_KC) {
#pragma unroll
    for (int c = 0; c < NCH_B; ++c) {
      const int ch = tid + c * NT;
      sb[c] = B.init(ch < BN * CPK ? (int64_t)n0 + ch / CPK : (int64_t)1 << 60);
    }
  }

  float4 ra[NCH_A], rb[NCH_B];
  auto load_tiles = [&](int64_t k0) {  // raw, unconditional global loads: nothing here waits
#pragma unroll
    for (int c = 0; c < NCH_A; ++c) {
      const int ch = tid + c * NT;
      if constexpr (AOp::kLayout == SRC_KC) {
        ra[c] = A.load(sa[c], (int)(k0 + 4 * (ch % CPK)), (int)K);
      } else {
        constexpr int CPR = BM / 4;
        const int chc = ch < BK * CPR ? ch : BK * CPR - 1;
        ra[c] = A.load(k0 + chc / CPR, m0 + 4 * (chc % CPR), K);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH_B; ++c) {
      const int ch = tid + c * NT;
      if constexpr (BOp::kLayout == SRC_KC) {
        rb[c] = B.load(sb[c], (int)(k0 + 4 * (ch % CPK)), (int)K);
      } else {
        constexpr int CPR = BN / 4;
        const int chc = ch < BK * CPR ? ch : BK * CPR - 1;
        rb[c] = B.load(k0 + chc / CPR, (int64_t)n0 + 4 * (chc % CPR), K);
      }
    }
  };
  auto store_tiles = [&](int buf, int64_t k0) {  // masking / post-load transform + LDS image
    float* as = As + buf * BK * LDA;
    float* bs = Bs + buf * BK * LDB;
#pragma unroll
    for (int c = 0; c < NCH_A; ++c) {
      const int ch = tid + c * NT;
      if constexpr (AOp::kLayout == SRC_KC) {
        if (ch < BM * CPK) {
          const int row = ch / CPK, kc = (ch % CPK) * 4;
          A.finish(ra[c], sa[c], m0 + row, (int)(k0 + kc), (int)kend, primary);
          as[(kc + 0) * LDA + row] = ra[c].x;
          as[(kc + 1) * LDA + row] = ra[c].y;
          as[(kc + 2) * LDA + row] = ra[c].z;
          as[(kc + 3) * LDA + row] = ra[c].w;
        }
      } else {
        constexpr int CPR = BM / 4;
        if (ch < BK * CPR) {
          A.finish(ra[c], k0 + ch / CPR, m0 + 4 * (ch % CPR), kend);
          *reinterpret_cast<float4*>(as + (ch / CPR) * LDA + 4 * (ch % CPR)) = ra[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH_B; ++c) {
      const int ch = tid + c * NT;
      if constexpr (BOp::kLayout == SRC_KC) {
        if (ch < BN * CPK) {
          const int row = ch / CPK, kc = (ch % CPK) * 4;
          B.finish(rb[c], sb[c], (int64_t)n0 + row, (int)(k0 + kc), (int)kend, false);
          bs[(kc + 0) * LDB + row] = rb[c].x;
          bs[(kc + 1) * LDB + row] = rb[c].y;
          bs[(kc + 2) * LDB + row] = rb[c].z;
          bs[(kc + 3) * LDB + row] = rb[c].w;
        }
      } else {
        constexpr int CPR = BN / 4;
        if (ch < BK * CPR) {
          B.finish(rb[c], k0 + ch / CPR, (int64_t)n0 + 4 * (ch % CPR), kend);
          *reinterpret_cast<float4*>(bs + (ch / CPR) * LDB + 4 * (ch % CPR)) = rb[c];
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline: double-buffered LDS, register-staged global loads issued one k-tile ahead so
  // the HBM/L2 latency sits under the MFMAs; one barrier per k-tile.
  // Default order: loads of tile t+1 at the top of iteration t, their LDS write at the bottom of the
  // SAME iteration (staging registers are not loop-carried, so hipcc inserts no early vmcnt wait).
  // ABL & 16 selects the prefetch-distance-2 order (write at the top of the next iteration), which
  // measured ~10 % slower because hipcc copies the loop-carried staging registers right after the
  // loads and waits for them there.
  constexpr bool kDist2 = (ABL & 16) != 0;
  load_tiles(kbeg);
  store_tiles(0, kbeg);
  if (kDist2 && ntiles > 1) load_tiles(kbeg + BK);
  __syncthreads();

  // Two copies of the k-loop, selected once per wave: interior waves (every block valid) run a
  // branch-free MFMA stream; edge waves test the (scalar) block counts.
  auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int tt = 0; tt < ntiles; ++tt) {
      const int buf = tt & 1;
      const int64_t k0 = kbeg + (int64_t)tt * BK;
      if constexpr (kDist2) {
        if constexpr (ABL & 2) {
#pragma unroll
          for (int c = 0; c < NCH_A; ++c) asm volatile("" ::"v"(ra[c].x), "v"(ra[c].y), "v"(ra[c].z), "v"(ra[c].w));
#pragma unroll
          for (int c = 0; c < NCH_B; ++c) asm volatile("" ::"v"(rb[c].x), "v"(rb[c].y), "v"(rb[c].z), "v"(rb[c].w));
        } else {
          if (tt + 1 < ntiles) store_tiles(buf ^ 1, k0 + BK);
        }
        if constexpr (!(ABL & 4)) {
          if (tt + 2 < ntiles) load_tiles(k0 + 2 * BK);
        }
      } else {
        if (tt + 1 < ntiles) load_tiles(k0 + BK);
      }

      const float* as = As + buf * BK * LDA + (wm * TM * 16 + l15);
      const float* bs = Bs + buf * BK * LDB + (wn * TN * 16 + l15);
      // fragments are double-buffered in registers: the ds_reads of k-step s+1 are issued before
      // the MFMAs of k-step s.  A partial last k-tile is zero-filled by the loaders, so every tile
      // runs all BK/4 k-steps branch-free (K = 300 costs 76 instead of 75 steps).
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = as[g * LDA + i * 16];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = bs[g * LDB + j * 16];
      if constexpr (ABL & 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[1][i] = a[0][i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[1][j] = b[0][j];
      }
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        if (ks + 1 < BK / 4 && !(ABL & 8)) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[(ks + 1) & 1][i] = as[(4 * (ks + 1) + g) * LDA + i * 16];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[(ks + 1) & 1][j] = bs[(4 * (ks + 1) + g) * LDB + j * 16];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (FULL || i < nvi) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (FULL || j < nvj)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
      if constexpr (!kDist2) {
        if (tt + 1 < ntiles) store_tiles(buf ^ 1, k0 + BK);
      }
      if constexpr (!(ABL & 1)) __syncthreads();
    }
  };
  if (full)
    k_loop(std::true_type{});
  else
    k_loop(std::false_type{});

  // epilogue: lane holds C[row = 4g + r][col = l15] of each block
  store_accumulators<TM, TN>(epi, acc, m0, n0, wm, wn, l15, g, M, N);
}

// host launcher.  `splits` > 1 partitions K over blockIdx.y (epilogue must accumulate atomically).
template <int WM, int WN, int TM, int TN, int BK = GEMM_BK, int ABL = 0, class AOp, class BOp, class Epi>
int launch_gemm(const AOp& A, const BOp& B, const Epi& epi, int64_t M, int N, int64_t K, int splits,
                hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  if (M <= 0 || N <= 0 || K <= 0) return NRL_OK;
  const int64_t tiles_m = ceil_div(M, BM);
  const int tiles_n = (int)ceil_div(N, BN);
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  int64_t kps = ceil_div(ceil_div(K, splits), BK) * BK;
  splits = (int)ceil_div(K, kps);
  NRL_REQUIRE(tiles_total < (1LL << 31) && splits < 65536, "gemm grid too large");
  dim3 grid((unsigned)tiles_total, (unsigned)splits, 1);
  hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, BK, AOp, BOp, Epi, ABL>), grid, dim3(WM * WN * 64), 0,
                     stream, A, B, epi, M, N, K, tiles_n, tiles_total, kps);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
