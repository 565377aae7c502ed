// Scaled dot-product attention of a transformer BODY (heads of dh = 64 over sequences of <= 128 tokens) on the bf16 matrix cores
// with (hi, lo) split operands -- the self-attention of the PLM text encoder's HF body (reference text.py:89: `self.plm_model(**text)`;
// roberta-base: 12 heads x 64 over the title's tokens), with its key-padding mask and its attention-probability dropout.
//
// Why: on PyTorch-ROCm that attention runs as fp32 flash kernels built for long sequences -- 310 us forward + 598 us backward per
// layer for 12.5 GFLOP at the config-4 shape (440 news x 96 tokens), 22 ms of a 111-ms step.  The design is nrl_attn_x3.hip's
// (the NRMS user encoder's across-users attention), widened to dh = 64: one workgroup of 8 waves per (sequence, head), wave w =
// rows 16w .. 16w + 15; the operands every wave needs are split ONCE into (hi, lo) 16 x 16 block planes in LDS
// [row block 0..7][feature block 0..3][16][16]; row form = one ds_read_b128 per plane and k-step, column form = two
// ds_read_b64_tr_b16; two accumulator blocks of a lane ARE the A fragment of the next product under kappa, so P and dS stay in
// registers.  A wave's OWN rows (its queries' q / dO, its keys' k / v) are fragments read straight from global memory.
//   forward   LDS: K (row form), V (column form).            S^T = K Q^T -> + key mask -> softmax -> dropout -> O = P V
//   backward  two roles, one launch (workgroup parity):
//             dQ role   LDS: K (row + column form), V (row form);    dS^T = P^T o (M o (V dO^T) - delta),  dQ = scale dS K
//             dK/dV role LDS: scale Q, dO (row + column form each);  dV = (P o M)^T dO,  dK = dS^T (scale Q)
// with M the dropout multiplier of element (sequence, head, query, key) -- a counter-based hash of its flat index, the same
// in all three places -- and delta[q] = <dO[q], O[q]> (unchanged by the dropout: O = (P o M) V).
#include <math.h>

#include "nrl_kernels.h"
#include "nrl_news_fused.h"

namespace nrl {

constexpr int SA_S = 128, SA_DH = 64, SA_FB = 4, SA_KS = 2, SA_WAVES = 8;
constexpr int SA_PLANE = 8 * SA_FB * 512;        // 16 KB
constexpr int SA_MATB = 2 * SA_PLANE;            // (hi, lo)

struct SdpaArgs {
  const float *q, *k, *v;          // (n_batch, S, heads * 64) each
  const uint8_t* key_keep;         // (n_batch, S): 1 = attend to the key; or null
  const float *o, *d_o, *lse_in;   // backward inputs
  float *out, *lse, *dq, *dk, *dv;
  int64_t n_batch;
  int S, heads, D;                 // D = heads * 64 = row stride
  float scale;
  Dropout drop;
};

__device__ __forceinline__ int64_t sa_item(int64_t n_items) {      // XCD-aware (fa_item, nrl_attn_mfma.hip)
  const int64_t per = (n_items + 7) / 8;
  const int64_t v = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
  return v < n_items ? v : -1;
}

// rows [0, 128) x 64 features of one operand -> block planes: thread = two (row, 8-feature piece) items
__device__ __forceinline__ void sa_stage(unsigned char* __restrict__ dst, const float* __restrict__ src, int64_t row_stride, int S,
                                         float mul, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * SA_WAVES * 64;
    const int row = idx >> 3, c8 = idx & 7;
    const int r = row < S ? row : S - 1;
    const float* rp = src + (int64_t)r * row_stride + 8 * c8;
    float4 v0 = *reinterpret_cast<const float4*>(rp), v1 = *reinterpret_cast<const float4*>(rp + 4);
    v0.x *= mul; v0.y *= mul; v0.z *= mul; v0.w *= mul;
    v1.x *= mul; v1.y *= mul; v1.z *= mul; v1.w *= mul;
    bf16x8 hi, lo;
    rp_split8(v0, v1, hi, lo);
    const int off = ((row >> 4) * SA_FB + (c8 >> 1)) * 512 + (row & 15) * 32 + (c8 & 1) * 16;
    *reinterpret_cast<bf16x8*>(dst + off) = hi;
    *reinterpret_cast<bf16x8*>(dst + SA_PLANE + off) = lo;
  }
}
// row form, k-step s (features 32s .. 32s + 31): lane (l15, g) <- features 32s + 8g .. + 7 of `row`
__device__ __forceinline__ void sa_row(const unsigned char* __restrict__ mat, int row, int s, int g, bf16x8& hi, bf16x8& lo) {
  const int off = ((row >> 4) * SA_FB + 2 * s + (g >> 1)) * 512 + (row & 15) * 32 + (g & 1) * 16;
  hi = *reinterpret_cast<const bf16x8*>(mat + off);
  lo = *reinterpret_cast<const bf16x8*>(mat + SA_PLANE + off);
}
// column form: rows 32t + kappa(g, e) of feature 16 db + l15
__device__ __forceinline__ void sa_col(uint32_t mat_lds, int t, int db, uint32_t lane_off, bf16x8& hi, bf16x8& lo) {
  typedef short sa_v4i16 __attribute__((ext_vector_type(4)));
  typedef short sa_v8i16 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) sa_v4i16* lds_v4;
  const uint32_t b0 = mat_lds + (uint32_t)((2 * t * SA_FB + db) * 512) + lane_off;
  const uint32_t b1 = b0 + (uint32_t)(SA_FB * 512);
  const sa_v4i16 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)b0);
  const sa_v4i16 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)b1);
  const sa_v4i16 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(b0 + (uint32_t)SA_PLANE));
  const sa_v4i16 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(uintptr_t)(b1 + (uint32_t)SA_PLANE));
  hi = __builtin_bit_cast(bf16x8, (sa_v8i16)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  lo = __builtin_bit_cast(bf16x8, (sa_v8i16)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
}
// a wave's own row, straight from global: features 32s + 8g .. + 7 of the row at `rowp`, times mul
__device__ __forceinline__ void sa_own(const float* __restrict__ rowp, int s, int g, float mul, bf16x8& hi, bf16x8& lo) {
  float4 v0 = *reinterpret_cast<const float4*>(rowp + 32 * s + 8 * g), v1 = *reinterpret_cast<const float4*>(rowp + 32 * s + 8 * g + 4);
  v0.x *= mul; v0.y *= mul; v0.z *= mul; v0.w *= mul;
  v1.x *= mul; v1.y *= mul; v1.z *= mul; v1.w *= mul;
  rp_split8(v0, v1, hi, lo);
}

#define SA_MFMA3(acc, ah, al, bh, bl)                                          \
  do {                                                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);       \
  } while (0)

// additive key mask of the group in LDS: 0 for a key that takes part, -inf for padding / keys past S
__device__ __forceinline__ void sa_keymask(float* __restrict__ add_s, const SdpaArgs& P, int64_t batch, int tid) {
  if (tid < SA_S) {
    const bool keep = tid < P.S && (P.key_keep == nullptr || P.key_keep[batch * P.S + tid] != 0);
    add_s[tid] = keep ? 0.f : -INFINITY;
  }
}

__global__ void __launch_bounds__(SA_WAVES * 64) sa_fwd_kernel(const SdpaArgs P) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SA_MATB + SA_S * 4];
  unsigned char* const Ks = smem;
  float* const add_s = reinterpret_cast<float*>(smem + 2 * SA_MATB);
  const uint32_t Vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)SA_MATB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  const int64_t grp = sa_item(P.n_batch * P.heads);
  if (grp < 0) return;
  const int64_t batch = grp / P.heads;
  const int head = (int)(grp % P.heads);
  const int S = P.S;
  const int64_t base = batch * (int64_t)S * P.D + head * SA_DH;
  sa_stage(Ks, P.k + base, P.D, S, 1.0f, tid);
  sa_stage(smem + SA_MATB, P.v + base, P.D, S, 1.0f, tid);
  sa_keymask(add_s, P, batch, tid);
  __syncthreads();
  const int q0 = wave * 16;
  if (q0 >= S) return;

  bf16x8 qh[SA_KS], ql[SA_KS];
  {
    const int qi = q0 + l15 < S ? q0 + l15 : S - 1;
    const float* qrow = P.q + base + (int64_t)qi * P.D;
#pragma unroll
    for (int s = 0; s < SA_KS; ++s) sa_own(qrow, s, g, P.scale, qh[s], ql[s]);
  }
  float e[32];
  float m = -INFINITY;
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (16 * jb < S) {                       // (wave-uniform: key blocks past the sequence cost nothing; their scores are masked below)
#pragma unroll
      for (int s = 0; s < SA_KS; ++s) {
        bf16x8 kh, kl;
        sa_row(Ks, 16 * jb + l15, s, g, kh, kl);
        SA_MFMA3(sc, kh, kl, qh[s], ql[s]);
      }
    }
    const float4 a4 = *reinterpret_cast<const float4*>(add_s + 16 * jb + 4 * g);
    e[4 * jb + 0] = sc[0] + a4.x; e[4 * jb + 1] = sc[1] + a4.y; e[4 * jb + 2] = sc[2] + a4.z; e[4 * jb + 3] = sc[3] + a4.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, e[4 * jb + r]);
  }
  m = fmaxf(m, nf_xor16(m, lane));
  m = fmaxf(m, nf_xor32(m, lane));
  constexpr float LOG2E = 1.4426950408889634f;
  const float m2 = m * LOG2E;
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    e[q] = __builtin_amdgcn_exp2f(fmaf(e[q], LOG2E, -m2));
    sum += e[q];
  }
  sum += nf_xor16(sum, lane);
  sum += nf_xor32(sum, lane);
  const float inv = __builtin_amdgcn_rcpf(sum);
  if (P.lse != nullptr && g == 0 && q0 + l15 < S) P.lse[grp * S + q0 + l15] = m + logf(sum);
  // attention-probability dropout: element (group, query, key) <-> flat index (group * 128 + query) * 128 + key
  const uint32_t idx_q = ((uint32_t)grp * (uint32_t)SA_S + (uint32_t)(q0 + l15)) * (uint32_t)SA_S;
  const bool dropping = P.drop.thresh != 0u;

  f32x4 oacc[SA_FB];
#pragma unroll
  for (int db = 0; db < SA_FB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (32 * t >= S) continue;               // (their probabilities are exactly 0)
    float p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int key = 32 * t + (q < 4 ? 4 * g + q : 16 + 4 * g + (q - 4));
      p[q] = e[8 * t + q] * inv;
      if (dropping) p[q] *= P.drop.mult(idx_q + (uint32_t)key);
    }
    bf16x8 ph, pl;
    rp_split8(make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]), ph, pl);
#pragma unroll
    for (int db = 0; db < SA_FB; ++db) {
      bf16x8 vh, vl;
      sa_col(Vs_lds, t, db, lane_off, vh, vl);
      SA_MFMA3(oacc[db], ph, pl, vh, vl);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + 4 * g + r;
    if (qi < S) {
      float* orow = P.out + base + (int64_t)qi * P.D;
#pragma unroll
      for (int db = 0; db < SA_FB; ++db) orow[16 * db + l15] = oacc[db][r];
    }
  }
}

__global__ void __launch_bounds__(SA_WAVES * 64) sa_bwd_kernel(const SdpaArgs P) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SA_MATB + 3 * SA_S * 4];
  unsigned char* const M0 = smem;                  // dQ role: K      dK/dV role: scale Q
  unsigned char* const M1 = smem + SA_MATB;        // dQ role: V      dK/dV role: dO
  float* const add_s = reinterpret_cast<float*>(smem + 2 * SA_MATB);
  float* const lse_s = add_s + SA_S;
  float* const delta_s = lse_s + SA_S;
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t M0_lds = smem_lds, M1_lds = smem_lds + (uint32_t)SA_MATB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)((4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8);
  const int64_t item = sa_item(2 * P.n_batch * P.heads);
  if (item < 0) return;
  const int64_t grp = item >> 1;
  const bool role_dq = (item & 1) == 0;
  const int64_t batch = grp / P.heads;
  const int head = (int)(grp % P.heads);
  const int S = P.S;
  const int64_t base = batch * (int64_t)S * P.D + head * SA_DH;
  constexpr float LOG2E = 1.4426950408889634f;
  const bool dropping = P.drop.thresh != 0u;
  const int r0 = wave * 16;

  if (role_dq) {
    sa_stage(M0, P.k + base, P.D, S, 1.0f, tid);
    sa_stage(M1, P.v + base, P.D, S, 1.0f, tid);
    sa_keymask(add_s, P, batch, tid);
    __syncthreads();
    if (r0 >= S) return;
    const int qi = r0 + l15 < S ? r0 + l15 : S - 1;
    bf16x8 qh[SA_KS], ql[SA_KS], doh[SA_KS], dol[SA_KS];
    float delta_q = 0.f;
    {
      const float* qrow = P.q + base + (int64_t)qi * P.D;
      const float* dorow = P.d_o + base + (int64_t)qi * P.D;
      const float* orow = P.o + base + (int64_t)qi * P.D;
#pragma unroll
      for (int s = 0; s < SA_KS; ++s) {
        sa_own(qrow, s, g, P.scale, qh[s], ql[s]);
        const float4 a0 = *reinterpret_cast<const float4*>(dorow + 32 * s + 8 * g), a1 = *reinterpret_cast<const float4*>(dorow + 32 * s + 8 * g + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(orow + 32 * s + 8 * g), b1 = *reinterpret_cast<const float4*>(orow + 32 * s + 8 * g + 4);
        delta_q = fmaf(a0.x, b0.x, delta_q); delta_q = fmaf(a0.y, b0.y, delta_q); delta_q = fmaf(a0.z, b0.z, delta_q); delta_q = fmaf(a0.w, b0.w, delta_q);
        delta_q = fmaf(a1.x, b1.x, delta_q); delta_q = fmaf(a1.y, b1.y, delta_q); delta_q = fmaf(a1.z, b1.z, delta_q); delta_q = fmaf(a1.w, b1.w, delta_q);
        rp_split8(a0, a1, doh[s], dol[s]);
      }
      delta_q += nf_xor16(delta_q, lane);
      delta_q += nf_xor32(delta_q, lane);
    }
    const float lse_q = P.lse_in[grp * S + qi] * LOG2E;
    const uint32_t idx_q = ((uint32_t)grp * (uint32_t)SA_S + (uint32_t)(r0 + l15)) * (uint32_t)SA_S;
    f32x4 dq[SA_FB];
#pragma unroll
    for (int db = 0; db < SA_FB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (32 * t >= S) continue;             // keys past the sequence: dS = 0
      float ds[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int jb = 2 * t + half;
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SA_KS; ++s) {
          bf16x8 kh, kl, vh, vl;
          sa_row(M0, 16 * jb + l15, s, g, kh, kl);
          sa_row(M1, 16 * jb + l15, s, g, vh, vl);
          SA_MFMA3(st, kh, kl, qh[s], ql[s]);         // S^T[key][query]
          SA_MFMA3(dpt, vh, vl, doh[s], dol[s]);      // (V dO^T)[key][query]
        }
        const float4 a4 = *reinterpret_cast<const float4*>(add_s + 16 * jb + 4 * g);
        const float add[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * jb + 4 * g + r;
          const float p = __builtin_amdgcn_exp2f(fmaf(st[r], LOG2E, -lse_q) + add[r]);      // masked key: 2^(-inf) = 0
          float dp = dpt[r];
          if (dropping) dp *= P.drop.mult(idx_q + (uint32_t)key);
          ds[4 * half + r] = p * (dp - delta_q);
        }
      }
      bf16x8 dsh, dsl;
      rp_split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), dsh, dsl);
#pragma unroll
      for (int db = 0; db < SA_FB; ++db) {
        bf16x8 bh, bl;
        sa_col(M0_lds, t, db, lane_off, bh, bl);
        SA_MFMA3(dq[db], dsh, dsl, bh, bl);           // dQ += dS K
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = r0 + 4 * g + r;
      if (q < S) {
        float* row = P.dq + base + (int64_t)q * P.D;
#pragma unroll
        for (int db = 0; db < SA_FB; ++db) row[16 * db + l15] = dq[db][r] * P.scale;
      }
    }
    return;
  }

  // ---- dK / dV role ---------------------------------------------------------------------------------------------------
  if (tid < SA_S) {
    delta_s[tid] = 0.f;
    lse_s[tid] = tid < S ? P.lse_in[grp * S + tid] : INFINITY;        // rows past S: probability 0
  }
  sa_stage(M0, P.q + base, P.D, S, P.scale, tid);
  sa_stage(M1, P.d_o + base, P.D, S, 1.0f, tid);
  __syncthreads();
  // delta[q] = <dO[q], O[q]>: each thread's two 8-feature pieces, summed per row through LDS atomics
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * SA_WAVES * 64;
    const int row = idx >> 3, c8 = idx & 7;
    if (row < S) {
      const float* a = P.d_o + base + (int64_t)row * P.D + 8 * c8;
      const float* b = P.o + base + (int64_t)row * P.D + 8 * c8;
      const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
      float part = a0.x * b0.x;
      part = fmaf(a0.y, b0.y, part); part = fmaf(a0.z, b0.z, part); part = fmaf(a0.w, b0.w, part);
      part = fmaf(a1.x, b1.x, part); part = fmaf(a1.y, b1.y, part); part = fmaf(a1.z, b1.z, part); part = fmaf(a1.w, b1.w, part);
      atomicAdd(delta_s + row, part);
    }
  }
  __syncthreads();
  if (r0 >= S) return;
  const int ki = r0 + l15 < S ? r0 + l15 : S - 1;
  const bool key_ok = r0 + l15 < S && (P.key_keep == nullptr || P.key_keep[batch * S + ki] != 0);
  bf16x8 kh[SA_KS], kl[SA_KS], vh[SA_KS], vl[SA_KS];
  {
    const float* krow = P.k + base + (int64_t)ki * P.D;
    const float* vrow = P.v + base + (int64_t)ki * P.D;
#pragma unroll
    for (int s = 0; s < SA_KS; ++s) {
      sa_own(krow, s, g, 1.0f, kh[s], kl[s]);
      sa_own(vrow, s, g, 1.0f, vh[s], vl[s]);
    }
  }
  const float key_add = key_ok ? 0.f : -INFINITY;
  f32x4 dk[SA_FB], dv[SA_FB];
#pragma unroll
  for (int db = 0; db < SA_FB; ++db) {
    dk[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (32 * t >= S) continue;               // queries past the sequence: P = dS = 0
    float pp[8], ds[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int ib = 2 * t + half;
      f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SA_KS; ++s) {
        bf16x8 qh, ql, doh, dol;
        sa_row(M0, 16 * ib + l15, s, g, qh, ql);
        sa_row(M1, 16 * ib + l15, s, g, doh, dol);
        SA_MFMA3(sc, qh, ql, kh[s], kl[s]);           // S[query][key]
        SA_MFMA3(dp, doh, dol, vh[s], vl[s]);         // (dO V^T)[query][key]
      }
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * ib + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 16 * ib + 4 * g);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int query = 16 * ib + 4 * g + r;
        const float p = __builtin_amdgcn_exp2f((sc[r] - lq[r]) * LOG2E + key_add);
        float mlt = 1.0f;
        if (dropping) mlt = P.drop.mult(((uint32_t)grp * (uint32_t)SA_S + (uint32_t)query) * (uint32_t)SA_S + (uint32_t)(r0 + l15));
        pp[4 * half + r] = p * mlt;
        ds[4 * half + r] = p * (dp[r] * mlt - dl[r]);
      }
    }
    bf16x8 ph, pl, dsh, dsl;
    rp_split8(make_float4(pp[0], pp[1], pp[2], pp[3]), make_float4(pp[4], pp[5], pp[6], pp[7]), ph, pl);
    rp_split8(make_float4(ds[0], ds[1], ds[2], ds[3]), make_float4(ds[4], ds[5], ds[6], ds[7]), dsh, dsl);
#pragma unroll
    for (int db = 0; db < SA_FB; ++db) {
      bf16x8 bh, bl;
      sa_col(M1_lds, t, db, lane_off, bh, bl);
      SA_MFMA3(dv[db], ph, pl, bh, bl);               // dV += (P o M)^T dO
      sa_col(M0_lds, t, db, lane_off, bh, bl);
      SA_MFMA3(dk[db], dsh, dsl, bh, bl);             // dK += dS^T (scale Q)
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = r0 + 4 * g + r;
    if (key < S) {
      float* krow = P.dk + base + (int64_t)key * P.D;
      float* vrow = P.dv + base + (int64_t)key * P.D;
#pragma unroll
      for (int db = 0; db < SA_FB; ++db) {
        krow[16 * db + l15] = dk[db][r];
        vrow[16 * db + l15] = dv[db][r];
      }
    }
  }
}

bool sdpa_x3_ok(int64_t n_batch, int S, int heads, int dh) {
  return n_batch >= 0 && S >= 1 && S <= SA_S && heads >= 1 && dh == SA_DH && n_batch * heads < (1LL << 24);
}

int sdpa_x3_fwd(const SdpaArgs& a, hipStream_t st) {
  const int64_t groups = a.n_batch * a.heads;
  if (groups == 0) return NRL_OK;
  hipLaunchKernelGGL(sa_fwd_kernel, dim3((unsigned)(8 * ((groups + 7) / 8))), dim3(SA_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int sdpa_x3_bwd(const SdpaArgs& a, hipStream_t st) {
  const int64_t items = 2 * a.n_batch * a.heads;
  if (items == 0) return NRL_OK;
  hipLaunchKernelGGL(sa_bwd_kernel, dim3((unsigned)(8 * ((items + 7) / 8))), dim3(SA_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
using namespace nrl;

extern "C" {

int32_t nrl_sdpa_supported(int64_t n_batch, int32_t seq_len, int32_t num_heads, int32_t head_dim) {
  return sdpa_x3_ok(n_batch, seq_len, num_heads, head_dim) ? 1 : 0;
}

int nrl_sdpa_fwd(const float* q, const float* k, const float* v, const uint8_t* key_keep, int64_t n_batch, int32_t seq_len,
                 int32_t num_heads, int32_t head_dim, float scale, double p_drop, uint64_t seed, uint32_t stream0, float* out,
                 float* lse, void* stream) {
  NRL_REQUIRE(q && k && v && out, "sdpa_fwd: null argument");
  NRL_REQUIRE(sdpa_x3_ok(n_batch, seq_len, num_heads, head_dim), "sdpa_fwd: unsupported geometry (seq_len <= 128, head_dim == 64)");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  // the dropout counter of element (group, query, key) is the 32-bit index (group * 128 + query) * 128 + key: it would wrap -- masks
  // repeating across groups -- from 2^18 groups on
  NRL_REQUIRE(p_drop == 0.0 || n_batch * num_heads < (1LL << 18), "sdpa: attention dropout covers batch * heads < 2^18");
  NRL_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "sdpa_fwd: 16-byte alignment");
  SdpaArgs a{};
  a.q = q; a.k = k; a.v = v; a.key_keep = key_keep; a.out = out; a.lse = lse;
  a.n_batch = n_batch; a.S = seq_len; a.heads = num_heads; a.D = num_heads * head_dim; a.scale = scale;
  a.drop = make_dropout(p_drop, seed, stream0);
  return sdpa_x3_fwd(a, (hipStream_t)stream);
}

int nrl_sdpa_bwd(const float* q, const float* k, const float* v, const uint8_t* key_keep, const float* out, const float* d_out,
                 const float* lse, int64_t n_batch, int32_t seq_len, int32_t num_heads, int32_t head_dim, float scale,
                 double p_drop, uint64_t seed, uint32_t stream0, float* dq, float* dk, float* dv, void* stream) {
  NRL_REQUIRE(q && k && v && out && d_out && lse && dq && dk && dv, "sdpa_bwd: null argument");
  NRL_REQUIRE(sdpa_x3_ok(n_batch, seq_len, num_heads, head_dim), "sdpa_bwd: unsupported geometry (seq_len <= 128, head_dim == 64)");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  // the dropout counter of element (group, query, key) is the 32-bit index (group * 128 + query) * 128 + key: it would wrap -- masks
  // repeating across groups -- from 2^18 groups on
  NRL_REQUIRE(p_drop == 0.0 || n_batch * num_heads < (1LL << 18), "sdpa: attention dropout covers batch * heads < 2^18");
  NRL_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dq | (uintptr_t)dk |
                (uintptr_t)dv) & 15) == 0, "sdpa_bwd: 16-byte alignment");
  SdpaArgs a{};
  a.q = q; a.k = k; a.v = v; a.key_keep = key_keep; a.o = out; a.d_o = d_out; a.lse_in = lse; a.dq = dq; a.dk = dk; a.dv = dv;
  a.n_batch = n_batch; a.S = seq_len; a.heads = num_heads; a.D = num_heads * head_dim; a.scale = scale;
  a.drop = make_dropout(p_drop, seed, stream0);
  return sdpa_x3_bwd(a, (hipStream_t)stream);
}

}  // extern "C"
