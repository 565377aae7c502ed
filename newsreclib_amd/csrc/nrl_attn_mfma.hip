// Flash-attention forward on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for LONG sequences, gfx950.
//
// Where it is used: the seq-first nn.MultiheadAttention quirk (SURVEY.md headline fact 3) makes the PLM
// encoder tail (text.py:92-96) attend ACROSS THE NEWS of a call for every token position and head: S = number
// of news (7040 at B = 128), 1536 attentions of 7040 x 7040 x 48 = 14.6 TFLOP forward -- genuinely dense, unlike
// the 30 x 30 x 20 title attentions, which stay on the vector ALU (nrl_kernels.hip).  Also serves the NRMS user
// encoder (S = users of the batch) once S >= 64.
//
// One workgroup = 4 waves = 64 queries of one (outer, head) group; keys/values stream through LDS in blocks of
// 64.  Per wave (16 queries) and 16-key sub-block:
//   S^T (16 keys x 16 queries) = K_sub (16 x dh) * Q^T (dh x 16)           dh / 4 MFMAs
// The TRANSPOSED scores are computed on purpose: their accumulator layout (lane (l15, g) holds keys 4g + r,
// r = 0..3, of query l15) is exactly the A-operand layout of the next product with the MFMA k index running over
// the keys {r, 4 + r, 8 + r, 12 + r}, so P never leaves the registers:
//   O (16 queries x dh) += P (16 x 16 keys) * V_sub (16 x dh)               4 * ceil(dh / 16) MFMAs
// Online softmax per query: statistics live in the lanes of the query column (replicated over g); the O
// accumulator has queries along (g, r), so its rescale factors are fetched with 4 lane shuffles per 64 keys.
// fp32 in, fp32 MFMA (bitwise an fmaf chain), fp32 out: same arithmetic type as the vector-ALU kernels.
#include <math.h>

#include "nrl_kernels.h"

namespace nrl {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FA_BQ = 64, FA_BK = 64;

// Workgroup id -> work item, XCD-aware.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own
// L2, and consecutive items here share cache lines: the heads of one (outer, row) sit side by side in a packed q|k|v row
// (dh = 20: 80-byte pieces of 128-byte lines), the query tiles of a group re-read the same K / V.  XCD x takes the contiguous
// range [x * per, (x + 1) * per) of the items.  Grid = 8 * per workgroups; ids past the end return at once.
// (What these kernels are bound by at the user encoder's S = 128 is the fp32 matrix pipe itself: 64 tiles x (36 + 48) MFMAs x
//  750 groups = 4.0 M v_mfma_f32_16x16x4_f32 of 32 cycles on 1024 SIMDs = 52 us of the backward's 72; neither this order, nor
//  one launch for both roles, nor fetching the K | V blocks a block ahead into registers -- 27.6 / 75.5 vs 27.9 / 72.3 us,
//  not kept -- moves that.  A bf16x3 form as in nrl_news_fused.h would: 3 x 16 cycles per 32-deep product instead of 8 x 32.)
__device__ __forceinline__ int64_t fa_item(int64_t n_items) {
  const int64_t per = (n_items + 7) / 8;
  const int64_t v = (int64_t)(blockIdx.x % 8) * per + blockIdx.x / 8;
  return v < n_items ? v : -1;
}
static inline unsigned fa_grid(int64_t n_items) { return (unsigned)(8 * ((n_items + 7) / 8)); }

template <int DH>
__global__ void __launch_bounds__(256)
    attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                         const AttnGeom G, const int q_tiles) {
  constexpr int DHP = DH + 4;               // LDS row pitch (16-B aligned rows, 2-way worst-case bank conflict)
  constexpr int KS = DH / 4;                // MFMA k-steps of the score product
  constexpr int DB = (DH + 15) / 16;        // 16-column blocks of the output
  __shared__ __attribute__((aligned(16))) float Ks[FA_BK * DHP];
  __shared__ __attribute__((aligned(16))) float Vs[FA_BK * DHP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t item = fa_item(G.groups * q_tiles);
  if (item < 0) return;
  const int64_t grp = item / q_tiles;
  const int qt = (int)(item % q_tiles);
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const float* qb = qkv + outer * G.q_outer + head * DH;
  const int q0 = qt * FA_BQ + wave * 16;    // first query of this wave
  const int S = G.S;

  // B operand of the score product: Q[query l15][d = 4 s + g], scaled by 1/sqrt(dh) as torch does
  float qreg[KS];
  {
    const int qi = q0 + l15 < S ? q0 + l15 : S - 1;
    const float* qrow = qb + (int64_t)qi * G.q_seq;
#pragma unroll
    for (int s = 0; s < KS; ++s) qreg[s] = qrow[4 * s + g] * G.scale;
  }

  f32x4 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;             // running max / (partial, per g) sum of query l15

  for (int k0 = 0; k0 < S; k0 += FA_BK) {
    __syncthreads();                        // previous block fully consumed
    // stage K and V rows k0 .. k0 + 63 (rows past S: clamped, masked below)
    constexpr int C4 = DH / 4;
    for (int idx = tid; idx < FA_BK * C4; idx += 256) {
      const int row = idx / C4, c4 = idx - row * C4;
      const int kr = k0 + row < S ? k0 + row : S - 1;
      const float* src = qb + (int64_t)kr * G.q_seq + 4 * c4;
      *reinterpret_cast<float4*>(Ks + row * DHP + 4 * c4) = *reinterpret_cast<const float4*>(src + G.D);
      *reinterpret_cast<float4*>(Vs + row * DHP + 4 * c4) = *reinterpret_cast<const float4*>(src + 2 * G.D);
    }
    __syncthreads();

    // ---- scores, transposed: st[kb][r] = S^T[key = k0 + 16 kb + 4 g + r][query = q0 + l15]
    f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* krow = Ks + (16 * kb + l15) * DHP + g;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[4 * s], qreg[s], acc, 0, 0, 0);
      st[kb] = acc;
    }
    // mask the tail keys, block maximum of this query
    float mloc = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * kb + 4 * g + r;
        const float v = key < S ? st[kb][r] : -INFINITY;
        st[kb][r] = v;
        mloc = fmaxf(mloc, v);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m, mloc);
    const float alpha = expf(m - m_new);    // first block: exp(-inf) = 0
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(st[kb][r] - m_new);   // masked keys: exp(-inf) = 0
        st[kb][r] = p;
        lsum += p;
      }
    l = l * alpha + lsum;
    m = m_new;
    // rescale O: its rows are queries 4 g + r, whose alpha lives in lane (l15 = 4 g + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a_r = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
      for (int db = 0; db < DB; ++db) oacc[db][r] *= a_r;
    }
    // ---- O += P V : A = P (registers), MFMA k index <-> keys {r, 4 + r, 8 + r, 12 + r} of the sub-block
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vrow = Vs + (16 * kb + 4 * g + r) * DHP;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int d = 16 * db + l15;
          const float b = (DH % 16 == 0 || d < DH) ? vrow[d < DH ? d : 0] : 0.f;
          oacc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[kb][r], b, oacc[db], 0, 0, 0);
        }
      }
  }

  // ---- finish: total sum over the four g copies, normalise, store
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (lse != nullptr && g == 0 && q0 + l15 < S) lse[grp * S + q0 + l15] = m + logf(l);
  const float inv = 1.0f / l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv_r = __shfl(inv, 4 * g + r, 64);
    const int qi = q0 + 4 * g + r;
    if (qi < S) {
      float* orow = o + outer * G.o_outer + (int64_t)qi * G.o_seq + head * DH;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int d = 16 * db + l15;
        if (DH % 16 == 0 || d < DH) orow[d] = oacc[db][r] * inv_r;
      }
    }
  }
}

#define NRL_FA_DISPATCH(dh, ...)                            \
  switch (dh) {                                             \
    case 16: { constexpr int DH = 16; __VA_ARGS__; } break; \
    case 20: { constexpr int DH = 20; __VA_ARGS__; } break; \
    case 32: { constexpr int DH = 32; __VA_ARGS__; } break; \
    case 48: { constexpr int DH = 48; __VA_ARGS__; } break; \
    case 64: { constexpr int DH = 64; __VA_ARGS__; } break; \
    default: set_error("unsupported head dim %d", dh); return NRL_E_INVALID; \
  }

int attn_fwd_mfma(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream) {
  if (G.groups == 0) return NRL_OK;
  const int q_tiles = (G.S + FA_BQ - 1) / FA_BQ;
  const int64_t blocks = G.groups * q_tiles;
  NRL_REQUIRE(blocks < (1LL << 31), "attention grid too large");
  NRL_FA_DISPATCH(G.dh, {
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<DH>), dim3(fa_grid(blocks)), dim3(256), 0, stream, qkv, o, lse, G, q_tiles);
  });
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// ---------------------------------------------------------------------------------------------------
// Backward, two kernels (no atomics): (B) one workgroup per 64 QUERIES -> dQ, streaming keys; (A) one workgroup
// per 64 KEYS -> dK, dV, streaming queries.  Both recompute the scores from the saved log-sum-exp:
//   P = exp(S - lse[q]),  dP = dO V^T,  dS = P o (dP - delta[q]),  delta[q] = <dO[q], O[q]>
//   dQ = scale * dS K        dK = dS^T (scale Q)        dV = P^T dO
// Kernel B computes S^T and dP^T (accumulator: keys along (g, r), query along l15 = the A layout of dS K);
// kernel A computes S and dP (accumulator: queries along (g, r), key along l15 = the A layout of P^T dO and
// dS^T Q).  Per 16 x 16 tile: 36 + 48 MFMAs (forward: 24).
// ---------------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_dq_body(const float* __restrict__ qkv, const float* __restrict__ o,
                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                 float* __restrict__ dqkv, const AttnGeom& G, const int q_tiles,
                                                 const int64_t bid, float* __restrict__ smem) {
  constexpr int DHP = DH + 4, KS = DH / 4, DB = (DH + 15) / 16;
  float* const Ks = smem;
  float* const Vs = smem + FA_BK * DHP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t grp = bid / q_tiles;
  const int qt = (int)(bid % q_tiles);
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const float* qb = qkv + outer * G.q_outer + head * DH;
  const int q0 = qt * FA_BQ + wave * 16;
  const int S = G.S;

  // per-query operands of this lane's query column (l15): scaled Q and dO as B operands, lse, delta
  float qreg[KS], doreg[KS];
  float lse_q, delta_q;
  {
    const int qi = q0 + l15 < S ? q0 + l15 : S - 1;
    const float* qrow = qb + (int64_t)qi * G.q_seq;
    const float* dorow = d_o + outer * G.o_outer + (int64_t)qi * G.o_seq + head * DH;
    const float* orow = o + outer * G.o_outer + (int64_t)qi * G.o_seq + head * DH;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qreg[s] = qrow[4 * s + g] * G.scale;
      doreg[s] = dorow[4 * s + g];
      part = fmaf(doreg[s], orow[4 * s + g], part);
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    delta_q = part;
    lse_q = lse[grp * S + qi];
  }
  f32x4 dq[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < S; k0 += FA_BK) {
    __syncthreads();
    constexpr int C4 = DH / 4;
    for (int idx = tid; idx < FA_BK * C4; idx += 256) {
      const int row = idx / C4, c4 = idx - row * C4;
      const int kr = k0 + row < S ? k0 + row : S - 1;
      const float* src = qb + (int64_t)kr * G.q_seq + 4 * c4;
      *reinterpret_cast<float4*>(Ks + row * DHP + 4 * c4) = *reinterpret_cast<const float4*>(src + G.D);
      *reinterpret_cast<float4*>(Vs + row * DHP + 4 * c4) = *reinterpret_cast<const float4*>(src + 2 * G.D);
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* krow = Ks + (16 * kb + l15) * DHP + g;
      const float* vrow = Vs + (16 * kb + l15) * DHP + g;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[4 * s], qreg[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[4 * s], doreg[s], dpt, 0, 0, 0);
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * kb + 4 * g + r;
        const float p = key < S ? expf(st[r] - lse_q) : 0.f;
        ds[r] = p * (dpt[r] - delta_q);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* kr = Ks + (16 * kb + 4 * g + r) * DHP;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int d = 16 * db + l15;
          const float b = (DH % 16 == 0 || d < DH) ? kr[d < DH ? d : 0] : 0.f;
          dq[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], b, dq[db], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + 4 * g + r;
    if (qi < S) {
      float* row = dqkv + outer * G.q_outer + (int64_t)qi * G.q_seq + head * DH;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int d = 16 * db + l15;
        if (DH % 16 == 0 || d < DH) row[d] = dq[db][r] * G.scale;
      }
    }
  }
}

template <int DH>
__device__ __forceinline__ void attn_bwd_dkv_body(const float* __restrict__ qkv, const float* __restrict__ o,
                                                  const float* __restrict__ d_o, const float* __restrict__ lse,
                                                  float* __restrict__ dqkv, const AttnGeom& G, const int k_tiles,
                                                  const int64_t bid, float* __restrict__ smem) {
  constexpr int DHP = DH + 4, KS = DH / 4, DB = (DH + 15) / 16;
  float* const Qs = smem;                          // scaled queries of the current block
  float* const dOs = smem + FA_BQ * DHP;
  float* const lse_s = smem + 2 * FA_BQ * DHP;
  float* const delta_s = lse_s + FA_BQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int64_t grp = bid / k_tiles;
  const int kt = (int)(bid % k_tiles);
  const int64_t outer = grp / G.heads;
  const int head = (int)(grp % G.heads);
  const float* qb = qkv + outer * G.q_outer + head * DH;
  const float* ob = o + outer * G.o_outer + head * DH;
  const float* dob = d_o + outer * G.o_outer + head * DH;
  const int key0 = kt * FA_BK + wave * 16;
  const int S = G.S;

  // B operands of the score / dP products: K and V rows of this lane's key column (l15)
  float kreg[KS], vreg[KS];
  {
    const int ki = key0 + l15 < S ? key0 + l15 : S - 1;
    const float* row = qb + (int64_t)ki * G.q_seq;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kreg[s] = row[G.D + 4 * s + g];
      vreg[s] = row[2 * G.D + 4 * s + g];
    }
  }
  f32x4 dk[DB], dv[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    dk[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int qq0 = 0; qq0 < S; qq0 += FA_BQ) {
    __syncthreads();
    constexpr int C4 = DH / 4;
    for (int idx = tid; idx < FA_BQ * C4; idx += 256) {
      const int row = idx / C4, c4 = idx - row * C4;
      const int qr = qq0 + row < S ? qq0 + row : S - 1;
      float4 qv = *reinterpret_cast<const float4*>(qb + (int64_t)qr * G.q_seq + 4 * c4);
      qv.x *= G.scale; qv.y *= G.scale; qv.z *= G.scale; qv.w *= G.scale;
      *reinterpret_cast<float4*>(Qs + row * DHP + 4 * c4) = qv;
      *reinterpret_cast<float4*>(dOs + row * DHP + 4 * c4) =
          *reinterpret_cast<const float4*>(dob + (int64_t)qr * G.o_seq + 4 * c4);
    }
    __syncthreads();
    if (tid < FA_BQ) {   // row statistics of the staged queries (rows past S contribute nothing: lse = +inf)
      const int qr = qq0 + tid;
      float dl = 0.f;
      if (qr < S) {
        const float* orow = ob + (int64_t)qr * G.o_seq;
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
          const float4 a = *reinterpret_cast<const float4*>(dOs + tid * DHP + 4 * c4);
          const float4 b = *reinterpret_cast<const float4*>(orow + 4 * c4);
          dl = fmaf(a.x, b.x, dl); dl = fmaf(a.y, b.y, dl); dl = fmaf(a.z, b.z, dl); dl = fmaf(a.w, b.w, dl);
        }
      }
      delta_s[tid] = dl;
      lse_s[tid] = qr < S ? lse[grp * S + qr] : INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int qbk = 0; qbk < 4; ++qbk) {
      f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* qrow = Qs + (16 * qbk + l15) * DHP + g;
      const float* drow = dOs + (16 * qbk + l15) * DHP + g;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qrow[4 * s], kreg[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(drow[4 * s], vreg[s], dp, 0, 0, 0);
      }
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = 16 * qbk + 4 * g + r;
        p[r] = expf(sc[r] - lse_s[qi]);            // query past S: exp(-inf) = 0
        ds[r] = p[r] * (dp[r] - delta_s[qi]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* dor = dOs + (16 * qbk + 4 * g + r) * DHP;
        const float* qr = Qs + (16 * qbk + 4 * g + r) * DHP;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const int d = 16 * db + l15;
          const bool ok = DH % 16 == 0 || d < DH;
          const float bdo = ok ? dor[d < DH ? d : 0] : 0.f;
          const float bq = ok ? qr[d < DH ? d : 0] : 0.f;
          dv[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], bdo, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], bq, dk[db], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ki = key0 + 4 * g + r;
    if (ki < S) {
      float* row = dqkv + outer * G.q_outer + (int64_t)ki * G.q_seq + head * DH;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int d = 16 * db + l15;
        if (DH % 16 == 0 || d < DH) {
          row[G.D + d] = dk[db][r];
          row[2 * G.D + d] = dv[db][r];
        }
      }
    }
  }
}

// ONE launch for both roles: item 2i computes dQ of (group, query tile) i, item 2i + 1 the dK / dV of (group, key
// tile) i (items -> workgroups: fa_item).  The two read the same q|k|v / dO rows (the neighbour's fetches hit the L2) and
// neither depends on the other; as two launches they ran back to back (36 + 44 us over the 1500 tiles of the NRMS user
// encoder at B = 128, 72-75 us as one: the drain of one and the ramp of the other are all there was to gain, see fa_item).
template <int DH>
__global__ void __launch_bounds__(256)
    attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ d_o,
                         const float* __restrict__ lse, float* __restrict__ dqkv, const AttnGeom G, const int tiles) {
  constexpr int DHP = DH + 4;
  __shared__ __attribute__((aligned(16))) float smem[2 * FA_BQ * DHP + 2 * FA_BQ];
  const int64_t item = fa_item(2 * G.groups * tiles);
  if (item < 0) return;
  const int64_t bid = item >> 1;
  if ((item & 1) == 0) attn_bwd_dq_body<DH>(qkv, o, d_o, lse, dqkv, G, tiles, bid, smem);
  else attn_bwd_dkv_body<DH>(qkv, o, d_o, lse, dqkv, G, tiles, bid, smem);
}

int attn_bwd_mfma(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                  const AttnGeom& G, hipStream_t stream) {
  if (G.groups == 0) return NRL_OK;
  const int tiles = (G.S + FA_BQ - 1) / FA_BQ;
  const int64_t blocks = G.groups * tiles;
  NRL_REQUIRE(blocks < (1LL << 31), "attention grid too large");
  NRL_REQUIRE(2 * blocks < (1LL << 31), "attention grid too large");
  NRL_FA_DISPATCH(G.dh, {
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<DH>), dim3(fa_grid(2 * blocks)), dim3(256), 0, stream, qkv, o, d_o, lse,
                       dqkv, G, tiles);
  });
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
