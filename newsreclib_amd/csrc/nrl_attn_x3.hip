// Attention over SHORT sequences of short heads (S <= 128, dh = 20) on the bf16 matrix cores with (hi, lo) split operands
// ("bf16x3": hi*lo + lo*hi + hi*hi into fp32, the arithmetic of every projection of the bf16x3 engine), gfx950.
//
// Where it is used: the seq-first nn.MultiheadAttention of the NRMS user encoder (user/nrms.py:32-41; SURVEY.md headline
// fact 3) attends ACROSS THE USERS of the batch for every history slot and head: 50 x 15 = 750 attentions of S = B = 128
// (64 per rank in BASELINE configs[2]) with dh = 20.  The flash kernels of nrl_attn_mfma.hip serve them in exact fp32 --
// v_mfma_f32_16x16x4_f32, 32 cycles per 4-deep product -- and at this size they are bound by exactly that pipe: 4.0 M MFMAs
// = 52 us of the backward's 72 at B = 128.  Here a 32-deep product costs 3 x 16 cycles.
//
// One workgroup of 8 wavefronts owns one (slot, head) group; wave w owns rows 16w .. 16w + 15.  q (pre-scaled) | k | v
// (| d_o) of the group are split ONCE -- one 8-feature piece per thread -- into (hi, lo) bf16 planes in LDS, laid out as
// 16 x 16 blocks [row block 0..7][feature block 0..1][16 rows][16 features] (features 20 .. 31 zero; rows past S are copies of
// row S - 1, so that a masked probability of 0 never meets a non-finite operand).  Every operand is then a plain read of that
// image, no VALU (the first version kept fp32 rows and split each fragment where it was used: every wave re-split the same K / V
// rows, ~10 splits of ~30 VALU instructions per 24 MFMAs -- 45.7 us backward / 20.5 us forward at B = 128 for 6.5 / 2 us of MFMAs):
//   row form    ua_row: lane (l15, g) <- features 8g .. 8g + 7 of row l15: one ds_read_b128 per plane
//                                                                         (A or B operand of a product over features)
//   column form ua_col: lane (d, g) <- rows kappa(g, e) of feature d: two ds_read_b64_tr_b16 per plane
//                                                                         (B operand of a product over rows)
// with kappa(g, e) = (e < 4 ? 4g + e : 16 + 4g + e - 4) inside a 32-row group -- the transpose read hands lane (d, g) rows
// 4g .. 4g + 3 of a block, two row blocks make the 8 k of a 16 x 16 x 32 operand.  A 16 x 16 accumulator block holds rows 4g + r
// of column l15 per lane, so TWO blocks (rows 32t .. 32t + 31 of one column) are exactly the A fragment of the next product
// under kappa: P, dS never leave registers (forward: S^T = K Q^T -> softmax -> O = P V; backward: both orientations, as two
// roles per wave -- its 16 queries' dQ, then its 16 keys' dK / dV -- each recomputing the scores from the saved log-sum-exp).
#include <math.h>

#include "nrl_kernels.h"
#include "nrl_news_fused.h"

namespace nrl {

constexpr int UA_S = 128;            // keys / queries per group the image holds
constexpr int UA_WAVES = 8;
constexpr int UA_DH = 20;
constexpr int UA_PLANE = 16 * 512;   // bytes per plane: 8 row blocks x 2 feature blocks x (16 x 16 bf16)
constexpr int UA_MATB = 2 * UA_PLANE;  // (hi, lo) of one operand

// group -> workgroup, XCD-aware (see fa_item, nrl_attn_mfma.hip): the heads of a slot share the cache lines of its packed rows
__device__ __forceinline__ int64_t ua_item(int64_t n_items) {
  const int64_t per = (n_items + 7) / 8;
  const int64_t v = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
  return v < n_items ? v : -1;
}

// rows [0, UA_S) of one strided operand -> (hi, lo) block planes: thread = (row, 8-feature piece)
__device__ __forceinline__ void ua_stage(unsigned char* __restrict__ dst, const float* __restrict__ src, int64_t row_stride, int S,
                                         float mul, int tid) {
  static_assert(UA_WAVES * 64 == UA_S * 4, "one piece per thread");
  const int row = tid >> 2, c8 = tid & 3;
  const int r = row < S ? row : S - 1;
  const float* rp = src + (int64_t)r * row_stride;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (c8 < 2) {
    v0 = *reinterpret_cast<const float4*>(rp + 8 * c8);
    v1 = *reinterpret_cast<const float4*>(rp + 8 * c8 + 4);
  } else if (c8 == 2) {
    v0 = *reinterpret_cast<const float4*>(rp + 16);
  }
  v0.x *= mul; v0.y *= mul; v0.z *= mul; v0.w *= mul;
  v1.x *= mul; v1.y *= mul; v1.z *= mul; v1.w *= mul;
  bf16x8 hi, lo;
  rp_split8(v0, v1, hi, lo);
  const int off = ((row >> 4) * 2 + (c8 >> 1)) * 512 + (row & 15) * 32 + (c8 & 1) * 16;
  *reinterpret_cast<bf16x8*>(dst + off) = hi;
  *reinterpret_cast<bf16x8*>(dst + UA_PLANE + off) = lo;
}
__device__ __forceinline__ void ua_row(const unsigned char* __restrict__ mat, int row, int g, bf16x8& hi, bf16x8& lo) {
  const int off = ((row >> 4) * 2 + (g >> 1)) * 512 + (row & 15) * 32 + (g & 1) * 16;
  hi = *reinterpret_cast<const bf16x8*>(mat + off);
  lo = *reinterpret_cast<const bf16x8*>(mat + UA_PLANE + off);
}
// rows 32t + kappa(g, e) of feature db * 16 + l15; lane_off = (4g + (l15 >> 2)) * 32 + (l15 & 3) * 8 (ds_read_b64_tr_b16)
__device__ __forceinline__ void ua_col(uint32_t mat_lds, int t, int db, uint32_t lane_off, bf16x8& hi, bf16x8& lo) {
  typedef short ua_v4i16 __attribute__((ext_vector_type(4)));
  typedef short ua_v8i16 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) ua_v4i16* lds_v4;
  const uint32_t b0 = mat_lds + (uint32_t)((4 * t + db) * 512) + lane_off;        // row block 2t; row block 2t + 1 is 1 KiB on
  const ua_v4i16 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)b0);
  const ua_v4i16 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(b0 + 1024u));
  const ua_v4i16 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(b0 + (uint32_t)UA_PLANE));
  const ua_v4i16 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(b0 + (uint32_t)UA_PLANE + 1024u));
  hi = __builtin_bit_cast(bf16x8, (ua_v8i16)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  lo = __builtin_bit_cast(bf16x8, (ua_v8i16)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
}

#define UA_MFMA3(acc, ah, al, bh, bl)                                          \
  do {                                                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);       \
  } while (0)

// softmax(q k^T) v of one (slot, head) group from the staged (hi, lo) planes: wave w owns queries 16w .. 16w + 15
__device__ __forceinline__ void ua_fwd_attend(const unsigned char* __restrict__ Qs, const unsigned char* __restrict__ Ks,
                                              const uint32_t Vs_lds, float* __restrict__ o, float* __restrict__ lse,
                                              const AttnGeom& G, const int64_t grp, const int64_t outer, const int head, const int S,
                                              const int wave, const int lane) {
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  const int q0 = wave * 16;
  if (q0 >= S) return;

  // ---- S^T = K Q^T: lane (query l15, g) <- keys 16 jb + 4g + r ---------------------------------------------------
  bf16x8 qh, ql;
  ua_row(Qs, q0 + l15, g, qh, ql);
  float e[32];
  float m = -INFINITY;
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    bf16x8 kh, kl;
    ua_row(Ks, 16 * jb + l15, g, kh, kl);
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    UA_MFMA3(s, kh, kl, qh, ql);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * jb + 4 * g + r;
      e[4 * jb + r] = key < S ? s[r] : -INFINITY;
      m = fmaxf(m, e[4 * jb + r]);
    }
  }
  m = fmaxf(m, nf_xor16(m, lane));
  m = fmaxf(m, nf_xor32(m, lane));
  constexpr float LOG2E = 1.4426950408889634f;
  const float m2 = m * LOG2E;
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    e[q] = __builtin_amdgcn_exp2f(fmaf(e[q], LOG2E, -m2));      // masked keys: 2^(-inf) = 0
    sum += e[q];
  }
  sum += nf_xor16(sum, lane);
  sum += nf_xor32(sum, lane);
  const float inv = __builtin_amdgcn_rcpf(sum);
  if (lse != nullptr && g == 0 && q0 + l15 < S) lse[grp * S + q0 + l15] = m + logf(sum);

  // ---- O = P V: A = P (two key blocks of a lane = one fragment under kappa), B = V in column form -------------------
  f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    bf16x8 ph, pl;
    rp_split8(make_float4(e[8 * t] * inv, e[8 * t + 1] * inv, e[8 * t + 2] * inv, e[8 * t + 3] * inv),
              make_float4(e[8 * t + 4] * inv, e[8 * t + 5] * inv, e[8 * t + 6] * inv, e[8 * t + 7] * inv), ph, pl);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      bf16x8 vh, vl;
      ua_col(Vs_lds, t, db, lane_off, vh, vl);
      UA_MFMA3(oacc[db], ph, pl, vh, vl);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + 4 * g + r;
    if (qi < S) {
      float* orow = o + outer * G.o_outer + (int64_t)qi * G.o_seq + head * UA_DH;
      orow[l15] = oacc[0][r];
      if (l15 < UA_DH - 16) orow[16 + l15] = oacc[1][r];
    }
  }
}

__global__ void __launch_bounds__(UA_WAVES * 64)
    ua_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse, const AttnGeom G) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * UA_MATB];
  unsigned char* const Qs = smem;
  unsigned char* const Ks = smem + UA_MATB;
  const uint32_t Vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + 2u * UA_MATB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t grp = ua_item(G.groups);
  if (grp < 0) return;
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const int S = G.S;
  const float* qb = qkv + outer * G.q_outer + head * UA_DH;
  ua_stage(Qs, qb, G.q_seq, S, G.scale, tid);
  ua_stage(Ks, qb + G.D, G.q_seq, S, 1.0f, tid);
  ua_stage(smem + 2 * UA_MATB, qb + 2 * G.D, G.q_seq, S, 1.0f, tid);
  __syncthreads();
  ua_fwd_attend(Qs, Ks, Vs_lds, o, lse, G, grp, outer, head, S, wave, lane);
}

// The same forward with the in-projection INSIDE (round 4; user/nrms.py:34: q|k|v = hist W_in^T + b_in): the group's 128 x 64
// slice [q 20 | k 20 | v 20 | 0 4] of the projection is computed by the workgroup itself -- wave w: rows 16w .. 16w + 15, its input
// rows read straight from global (all ten k-blocks in flight together) and split once, B operand = the head's 64 columns of the
// per-head weight image of the fused news encoder (rp_jobs_add_qkv_heads: bias at k = D against a ones column on the A side), read
// from global by every wave (the eight waves of a workgroup walk the same 80 KB: L1 / L2 hits) -- and goes accumulator -> (hi, lo)
// planes in LDS (each wave fills its own row block) and, for a training forward, -> the packed q|k|v rows the backward reads.
// Replaces the tiled in-projection GEMM launch (33 us at B = 128: one panel's k-loop latency for 3.5 GFLOP) and the weight
// split launch in front of it.
struct UaProjArgs {
  const float* x;            // input rows: row (outer, r) at x + outer * x_outer + r * x_seq, D floats each
  int64_t x_outer, x_seq;
  const uint16_t* img;       // per-head q|k|v image: [k-block][head * 4 + nb][hi | lo][lane][8 bf16], 10 k-blocks
  int nblk;                  // heads * 4
  float* qkv;                // packed q|k|v rows (geometry G.q_outer / G.q_seq), or null: evaluation
};

// (hipcc keeps 76 registers here and sinks the row loads towards their use; pinning all twenty in flight -- sched_barrier, 146
//  registers, one workgroup per CU instead of three -- measured SLOWER: 47 vs 39 us.  Three resident workgroups hide the latency.)
__global__ void __launch_bounds__(UA_WAVES * 64)
    ua_fwd_proj_kernel(const UaProjArgs A, float* __restrict__ o, float* __restrict__ lse, const AttnGeom G) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * UA_MATB];
  const uint32_t Vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + 2u * UA_MATB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t grp = ua_item(G.groups);
  if (grp < 0) return;
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const int S = G.S, D = G.D;
  constexpr int KB = 10;                                 // k-blocks of 32: D + 1 (bias column) <= 320
  // ---- A fragments: row min(16w + l15, S - 1), k = 32 kb + 8g .. + 7 (rows past S: copies of the last row) -------------
  const int arow = 16 * wave + l15 < S ? 16 * wave + l15 : S - 1;
  const float* xr = A.x + outer * A.x_outer + (int64_t)arow * A.x_seq;
  float4 raw[KB][2];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int k = kb * 32 + 8 * g;
    const int k0 = (kb < KB - 1 || k < D) ? k : D - 4, k1 = (kb < KB - 1 || k + 4 < D) ? k + 4 : D - 4;
    raw[kb][0] = *reinterpret_cast<const float4*>(xr + k0);
    raw[kb][1] = *reinterpret_cast<const float4*>(xr + k1);
  }
  // this wave's row block of the three operand images: zero (features 20 .. 31 must read as 0), then the values below
  {
    unsigned char* zb = smem + wave * 1024 + lane * 16;   // row block w = bytes [w * 1024, (w + 1) * 1024) of every plane
#pragma unroll
    for (int m = 0; m < 6; ++m) *reinterpret_cast<uint4*>(zb + m * UA_PLANE) = make_uint4(0u, 0u, 0u, 0u);
  }
  f32x4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(A.img) + (size_t)head * 4 * 2048 + lane * 16;
  auto load_b = [&](int kb, bf16x8 (&bh)[4], bf16x8 (&bl)[4]) {
    const unsigned char* p = wb + (size_t)kb * A.nblk * 2048;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      bh[nb] = *reinterpret_cast<const bf16x8*>(p + nb * 2048);
      bl[nb] = *reinterpret_cast<const bf16x8*>(p + nb * 2048 + 1024);
    }
  };
  bf16x8 bh0[4], bl0[4], bh1[4], bl1[4];
  load_b(0, bh0, bl0);
  auto kstep = [&](int kb, const bf16x8 (&bh)[4], const bf16x8 (&bl)[4]) {
    const int k = kb * 32 + 8 * g;
    float4 v0 = raw[kb][0], v1 = raw[kb][1];
    if (kb == KB - 1) {
      const bool in0 = k < D, in1 = k + 4 < D;
      if (!in0) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!in1) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k == D) v0.x = 1.0f;                            // the ones column: the image's bias row is added by the matrix cores
      if (k + 4 == D) v1.x = 1.0f;
    }
    bf16x8 ah, al;
    rp_split8(v0, v1, ah, al);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) UA_MFMA3(acc[nb], ah, al, bh[nb], bl[nb]);
  };
#pragma unroll
  for (int kb = 0; kb < KB; kb += 2) {
    load_b(kb + 1, bh1, bl1);
    kstep(kb, bh0, bl0);
    if (kb + 2 < KB) load_b(kb + 2, bh0, bl0);
    kstep(kb + 1, bh1, bl1);
  }
  // ---- accumulators -> the operand planes (and the packed rows): lane (l15, g) holds rows 16w + 4g + r of column 16 nb + l15 ----
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int c = 16 * nb + l15;
    const int part = c / UA_DH, f = c - part * UA_DH;       // 0 q | 1 k | 2 v | 3 pad
    if (part < 3) {
      unsigned char* mat = smem + part * UA_MATB;
      const float mul = part == 0 ? G.scale : 1.0f;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int row = 16 * wave + 4 * g + r;
        uint32_t h, l;
        split_pair(acc[nb][r] * mul, acc[nb][r + 1] * mul, h, l);
        const int off = (wave * 2 + (f >> 4)) * 512 + (4 * g + r) * 32 + (f & 15) * 2;
        *reinterpret_cast<uint16_t*>(mat + off) = (uint16_t)(h & 0xffffu);
        *reinterpret_cast<uint16_t*>(mat + off + 32) = (uint16_t)(h >> 16);
        *reinterpret_cast<uint16_t*>(mat + UA_PLANE + off) = (uint16_t)(l & 0xffffu);
        *reinterpret_cast<uint16_t*>(mat + UA_PLANE + off + 32) = (uint16_t)(l >> 16);
        if (A.qkv != nullptr) {
          float* q0p = A.qkv + outer * G.q_outer + (int64_t)row * G.q_seq + part * D + head * UA_DH + f;
          if (row < S) q0p[0] = acc[nb][r];
          if (row + 1 < S) q0p[G.q_seq] = acc[nb][r + 1];
        }
      }
    }
  }
  __syncthreads();
  ua_fwd_attend(smem, smem + UA_MATB, Vs_lds, o, lse, G, grp, outer, head, S, wave, lane);
}

__global__ void __launch_bounds__(UA_WAVES * 64)
    ua_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                  const float* __restrict__ lse, float* __restrict__ dqkv, const AttnGeom G) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * UA_MATB + 2 * UA_S * 4];
  unsigned char* const Qs = smem;                      // scale * q
  unsigned char* const Ks = smem + UA_MATB;
  unsigned char* const Vs = smem + 2 * UA_MATB;
  unsigned char* const dOs = smem + 3 * UA_MATB;
  float* const lse_s = reinterpret_cast<float*>(smem + 4 * UA_MATB);   // rows past S: +inf (probability 0)
  float* const delta_s = lse_s + UA_S;                                  // delta[q] = <dO[q], O[q]>
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t Qs_lds = smem_lds, Ks_lds = smem_lds + UA_MATB, dOs_lds = smem_lds + 3u * UA_MATB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  const int64_t grp = ua_item(G.groups);
  if (grp < 0) return;
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const int S = G.S;
  const float* qb = qkv + outer * G.q_outer + head * UA_DH;
  const float* dob = d_o + outer * G.o_outer + head * UA_DH;
  const float* ob = o + outer * G.o_outer + head * UA_DH;
  ua_stage(Qs, qb, G.q_seq, S, G.scale, tid);
  ua_stage(Ks, qb + G.D, G.q_seq, S, 1.0f, tid);
  ua_stage(Vs, qb + 2 * G.D, G.q_seq, S, 1.0f, tid);
  ua_stage(dOs, dob, G.o_seq, S, 1.0f, tid);
  if (tid < UA_S) {          // row statistics in fp32 from the rows themselves (the d_o row is in the caches by now)
    float dl = 0.f, ls = INFINITY;
    if (tid < S) {
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(dob + (int64_t)tid * G.o_seq + 4 * c);
        const float4 b = *reinterpret_cast<const float4*>(ob + (int64_t)tid * G.o_seq + 4 * c);
        dl = fmaf(a.x, b.x, dl); dl = fmaf(a.y, b.y, dl); dl = fmaf(a.z, b.z, dl); dl = fmaf(a.w, b.w, dl);
      }
      ls = lse[grp * S + tid];
    }
    delta_s[tid] = dl;
    lse_s[tid] = ls;
  }
  __syncthreads();
  const int r0 = wave * 16;
  if (r0 >= S) return;
  constexpr float LOG2E = 1.4426950408889634f;

  // ================= role 1: dQ of queries r0 .. r0 + 15 ==============================================================
  {
    bf16x8 qh, ql, doh, dol;
    ua_row(Qs, r0 + l15, g, qh, ql);
    ua_row(dOs, r0 + l15, g, doh, dol);
    const float lse_q = lse_s[r0 + l15] * LOG2E, delta_q = delta_s[r0 + l15];
    f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float ds[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int jb = 2 * t + half;
        bf16x8 kh, kl, vh, vl;
        ua_row(Ks, 16 * jb + l15, g, kh, kl);
        ua_row(Vs, 16 * jb + l15, g, vh, vl);
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
        UA_MFMA3(st, kh, kl, qh, ql);              // S^T[key][query]
        UA_MFMA3(dpt, vh, vl, doh, dol);           // dP^T[key][query] = V dO^T
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * jb + 4 * g + r;
          const float p = key < S ? __builtin_amdgcn_exp2f(fmaf(st[r], LOG2E, -lse_q)) : 0.f;
          ds[4 * half + r] = p * (dpt[r] - delta_q);
        }
      }
      bf16x8 dsh, dsl;
      rp_split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), dsh, dsl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        bf16x8 bh, bl;
        ua_col(Ks_lds, t, db, lane_off, bh, bl);
        UA_MFMA3(dq[db], dsh, dsl, bh, bl);        // dQ += dS K
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = r0 + 4 * g + r;
      if (qi < S) {
        float* row = dqkv + outer * G.q_outer + (int64_t)qi * G.q_seq + head * UA_DH;
        row[l15] = dq[0][r] * G.scale;
        if (l15 < UA_DH - 16) row[16 + l15] = dq[1][r] * G.scale;
      }
    }
  }

  // ================= role 2: dK, dV of keys r0 .. r0 + 15 =============================================================
  {
    bf16x8 kh, kl, vh, vl;
    ua_row(Ks, r0 + l15, g, kh, kl);
    ua_row(Vs, r0 + l15, g, vh, vl);
    f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float pp[8], ds[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ib = 2 * t + half;
        bf16x8 qh, ql, doh, dol;
        ua_row(Qs, 16 * ib + l15, g, qh, ql);
        ua_row(dOs, 16 * ib + l15, g, doh, dol);
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        UA_MFMA3(sc, qh, ql, kh, kl);              // S[query][key]
        UA_MFMA3(dp, doh, dol, vh, vl);            // dP[query][key] = dO V^T
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * ib + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 16 * ib + 4 * g);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f((sc[r] - lq[r]) * LOG2E);    // queries past S: lse = +inf -> 0
          pp[4 * half + r] = p;
          ds[4 * half + r] = p * (dp[r] - dq4[r]);
        }
      }
      bf16x8 ph, pl, dsh, dsl;
      rp_split8(make_float4(pp[0], pp[1], pp[2], pp[3]), make_float4(pp[4], pp[5], pp[6], pp[7]), ph, pl);
      rp_split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), dsh, dsl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        bf16x8 bh, bl;
        ua_col(dOs_lds, t, db, lane_off, bh, bl);
        UA_MFMA3(dv[db], ph, pl, bh, bl);          // dV += P^T dO
        ua_col(Qs_lds, t, db, lane_off, bh, bl);
        UA_MFMA3(dk[db], dsh, dsl, bh, bl);        // dK += dS^T (scale Q)
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ki = r0 + 4 * g + r;
      if (ki < S) {
        float* row = dqkv + outer * G.q_outer + (int64_t)ki * G.q_seq + head * UA_DH;
        row[G.D + l15] = dk[0][r];
        row[2 * G.D + l15] = dv[0][r];
        if (l15 < UA_DH - 16) {
          row[G.D + 16 + l15] = dk[1][r];
          row[2 * G.D + 16 + l15] = dv[1][r];
        }
      }
    }
  }
}

bool attn_x3_ok(const AttnGeom& G) {
  static const bool on = [] {
    const char* e = getenv("NRL_ATTN_X3");
    return !(e != nullptr && e[0] == '0');
  }();
  // (rows are read as float4: 16-byte aligned strides and head offsets)
  return on && G.dh == UA_DH && G.S >= 1 && G.S <= UA_S && G.D % 4 == 0 && G.q_seq % 4 == 0 && G.q_outer % 4 == 0 &&
         G.o_seq % 4 == 0 && G.o_outer % 4 == 0;
}

int attn_fwd_x3(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream) {
  if (G.groups == 0) return NRL_OK;
  NRL_REQUIRE(attn_x3_ok(G), "attn_fwd_x3: unsupported geometry");
  NRL_REQUIRE(G.groups + 7 < (1LL << 31), "attention grid too large");
  NRL_REQUIRE((((uintptr_t)qkv | (uintptr_t)o) & 15) == 0, "attn_fwd_x3: 16-byte alignment");
  hipLaunchKernelGGL(ua_fwd_kernel, dim3((unsigned)(8 * ((G.groups + 7) / 8))), dim3(UA_WAVES * 64), 0, stream, qkv, o, lse, G);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

bool attn_x3_proj_ok(const AttnGeom& G, int nblk, int kblocks) {
  return attn_x3_ok(G) && G.dh == UA_DH && nblk == G.heads * 4 && kblocks == 10 && G.D % 4 == 0 && G.D >= 288 && G.D <= 316;
}

int attn_fwd_x3_proj(const float* x, int64_t x_outer, int64_t x_seq, const uint16_t* img, int nblk, float* qkv, float* o,
                     float* lse, const AttnGeom& G, hipStream_t stream) {
  if (G.groups == 0) return NRL_OK;
  NRL_REQUIRE(attn_x3_ok(G) && nblk == G.heads * 4, "attn_fwd_x3_proj: unsupported geometry");
  NRL_REQUIRE(G.groups + 7 < (1LL << 31), "attention grid too large");
  NRL_REQUIRE((((uintptr_t)x | (uintptr_t)o | (uintptr_t)img) & 15) == 0 && x_outer % 4 == 0 && x_seq % 4 == 0,
              "attn_fwd_x3_proj: 16-byte alignment");
  UaProjArgs A{x, x_outer, x_seq, img, nblk, qkv};
  hipLaunchKernelGGL(ua_fwd_proj_kernel, dim3((unsigned)(8 * ((G.groups + 7) / 8))), dim3(UA_WAVES * 64), 0, stream, A, o, lse, G);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int attn_bwd_x3(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv, const AttnGeom& G,
                hipStream_t stream) {
  if (G.groups == 0) return NRL_OK;
  NRL_REQUIRE(attn_x3_ok(G), "attn_bwd_x3: unsupported geometry");
  NRL_REQUIRE(lse != nullptr, "attention backward needs the saved log-sum-exp");
  NRL_REQUIRE(G.groups + 7 < (1LL << 31), "attention grid too large");
  NRL_REQUIRE((((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15) == 0, "attn_bwd_x3: 16-byte alignment");
  hipLaunchKernelGGL(ua_bwd_kernel, dim3((unsigned)(8 * ((G.groups + 7) / 8))), dim3(UA_WAVES * 64), 0, stream, qkv, o, d_o, lse,
                     dqkv, G);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
