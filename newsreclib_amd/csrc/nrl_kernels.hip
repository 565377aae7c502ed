// Non-GEMM kernels of the NRMS hot path for gfx950 (wave64).
//
//  * attn_*: the (news x head) L x L and (slot x head) B x B attentions.  d_h = 20 and L = 30 are
//    far too small/odd for MFMA tiles to pay (a 30x30x20 product fills 2x2 16x16 blocks at 5 k-steps
//    with 12 % padding, and the softmax between the two products needs the scores in a per-row
//    layout anyway), so they run on the vector ALU: one lane owns one query row, K/V of the group
//    sit in LDS and are read as wave-wide broadcasts, softmax is online and in registers.  The
//    scores (N*h*L*L, 380 MB at B=128 in the reference) never exist in memory.
//  * pool_*: additive-attention softmax + weighted sum and its backward.
//  * dense batching, dot-product scorer, probability-target cross entropy, dense Adam.
#include <math.h>
#include <stdlib.h>

#include "nrl_kernels.h"
#include <cfloat>

namespace nrl {

// =============================================================================================
// attention
// =============================================================================================
template <int DH>
__device__ __forceinline__ float dot_row(const float (&q)[DH], const float* row) {
  const float4* r4 = reinterpret_cast<const float4*>(row);
  float acc = 0.f;
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    const float4 kv = r4[d4];
    acc = fmaf(q[4 * d4 + 0], kv.x, acc);
    acc = fmaf(q[4 * d4 + 1], kv.y, acc);
    acc = fmaf(q[4 * d4 + 2], kv.z, acc);
    acc = fmaf(q[4 * d4 + 3], kv.w, acc);
  }
  return acc;
}

template <int DH>
__device__ __forceinline__ void axpy_row(float a, const float* row, float (&o)[DH]) {
  const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    const float4 vv = r4[d4];
    o[4 * d4 + 0] = fmaf(a, vv.x, o[4 * d4 + 0]);
    o[4 * d4 + 1] = fmaf(a, vv.y, o[4 * d4 + 1]);
    o[4 * d4 + 2] = fmaf(a, vv.z, o[4 * d4 + 2]);
    o[4 * d4 + 3] = fmaf(a, vv.w, o[4 * d4 + 3]);
  }
}

template <int DH>
__device__ __forceinline__ void load_row(const float* src, float (&r)[DH], float mul) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    const float4 v = s4[d4];
    r[4 * d4 + 0] = v.x * mul;
    r[4 * d4 + 1] = v.y * mul;
    r[4 * d4 + 2] = v.z * mul;
    r[4 * d4 + 3] = v.w * mul;
  }
}

template <int DH>
__device__ __forceinline__ void store_row(float* dst, const float (&r)[DH], float mul) {
  float4* d4p = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4)
    d4p[d4] = make_float4(r[4 * d4] * mul, r[4 * d4 + 1] * mul, r[4 * d4 + 2] * mul, r[4 * d4 + 3] * mul);
}

// online-softmax accumulation of one query row against `nrows` keys/values held in LDS
template <int DH>
__device__ __forceinline__ void attend_rows(const float (&q)[DH], const float* Ks, const float* Vs,
                                            int nrows, float& m, float& l, float (&o)[DH]) {
  for (int cb = 0; cb < nrows; cb += 8) {
    float s[8];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = cb + u;
      const float acc = dot_row<DH>(q, Ks + (c < nrows ? c : 0) * DH);
      s[u] = c < nrows ? acc : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    const float m_new = fmaxf(m, mx);
    const float alpha = expf(m - m_new);  // m = -inf on the first block -> 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] *= alpha;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = cb + u;
      const float p = expf(s[u] - m_new);  // padded slots: exp(-inf) = 0
      l += p;
      axpy_row<DH>(p, Vs + (c < nrows ? c : 0) * DH, o);
    }
    m = m_new;
  }
}

// cooperative copy of `nrows` rows of DH floats (row stride `stride` elements) into LDS
template <int DH>
__device__ __forceinline__ void stage_rows(const float* src, int64_t stride, int nrows, float* dst,
                                           int tid, int nthreads, float mul = 1.0f) {
  constexpr int C4 = DH / 4;
  for (int idx = tid; idx < nrows * C4; idx += nthreads) {
    const int row = idx / C4, c4 = idx % C4;
    float4 v = *reinterpret_cast<const float4*>(src + row * stride + 4 * c4);
    v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    reinterpret_cast<float4*>(dst)[row * C4 + c4] = v;
  }
}

// ---- S <= LPG: LPG (32 or 64) lanes per group -- two groups per wavefront or one --, GPW groups per workgroup;
// lane = query row (then key row), K/V (then Q/dO) of the group in LDS.  S in (32, 64) used to fall to the
// one-workgroup-per-group kernel below, 3.5x slower per token (profiles/r01_mins_x3_kernel_stats.csv: the
// 50-token abstracts of MINS) ---------------------------------------------------------------------------------
template <int DH, int GPW, int LPG = 32>
__global__ void __launch_bounds__(GPW * LPG)
    attn_fwd_small(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                   const AttnGeom G) {
  __shared__ float4 smem4[GPW * 2 * LPG * DH / 4];
  const int tid = threadIdx.x, li = tid % LPG, grp = tid / LPG;
  const int64_t g = (int64_t)blockIdx.x * GPW + grp;
  const bool gvalid = g < G.groups;
  float* Ks = reinterpret_cast<float*>(smem4) + grp * (2 * LPG * DH);
  float* Vs = Ks + LPG * DH;
  const int64_t outer = gvalid ? g / G.heads : 0;
  const int head = gvalid ? (int)(g % G.heads) : 0;
  const float* qb = qkv + outer * G.q_outer + head * DH;
  if (gvalid) {
    stage_rows<DH>(qb + G.D, G.q_seq, G.S, Ks, li, LPG);
    stage_rows<DH>(qb + 2 * G.D, G.q_seq, G.S, Vs, li, LPG);
  }
  __syncthreads();
  if (gvalid && li < G.S) {
    float q[DH], acc[DH];
    load_row<DH>(qb + li * G.q_seq, q, G.scale);
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    attend_rows<DH>(q, Ks, Vs, G.S, m, l, acc);
    store_row<DH>(o + outer * G.o_outer + li * G.o_seq + head * DH, acc, 1.0f / l);
    if (lse != nullptr) lse[g * G.S + li] = m + logf(l);
  }
}

template <int DH, int GPW, int LPG = 32>
__global__ void __launch_bounds__(GPW * LPG)
    attn_bwd_small(const float* __restrict__ qkv, const float* __restrict__ o,
                   const float* __restrict__ d_o, const float* __restrict__ lse,
                   float* __restrict__ dqkv, const AttnGeom G) {
  // LDS per group: two row buffers (K,V in phase 1; scaled Q, dO in phase 2) + row statistics.
  // Re-using the buffers across the phases halves the footprint -> twice the resident waves.
  constexpr int PER = 2 * LPG * DH + 2 * LPG;
  __shared__ float4 smem4[GPW * PER / 4];
  const int tid = threadIdx.x, li = tid % LPG, grp = tid / LPG;
  const int64_t g = (int64_t)blockIdx.x * GPW + grp;
  const bool gvalid = g < G.groups;
  float* Ra = reinterpret_cast<float*>(smem4) + grp * PER;
  float* Rb = Ra + LPG * DH;
  float* lse_s = Rb + LPG * DH;
  float* dl_s = lse_s + LPG;
  const int64_t outer = gvalid ? g / G.heads : 0;
  const int head = gvalid ? (int)(g % G.heads) : 0;
  const float* qb = qkv + outer * G.q_outer + head * DH;
  const float* ob = o + outer * G.o_outer + head * DH;
  const float* dob = d_o + outer * G.o_outer + head * DH;
  float* dqb = dqkv + outer * G.q_outer + head * DH;
  const bool active = gvalid && li < G.S;
  if (gvalid) {
    stage_rows<DH>(qb + G.D, G.q_seq, G.S, Ra, li, LPG);     // K
    stage_rows<DH>(qb + 2 * G.D, G.q_seq, G.S, Rb, li, LPG); // V
  }
  __syncthreads();
  // phase 1: lane = query row -> dq, and the row statistics phase 2 needs
  float k[DH], v[DH], q[DH], dout[DH];
  if (active) {
    float dq[DH];
    load_row<DH>(qb + li * G.q_seq, q, G.scale);
    load_row<DH>(dob + li * G.o_seq, dout, 1.0f);
    const float dl = dot_row<DH>(dout, ob + li * G.o_seq);  // rowsum(dO * O) = rowsum(P * dP)
    const float ls = lse[g * G.S + li];
    lse_s[li] = ls;
    dl_s[li] = dl;
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.f;
    for (int c = 0; c < G.S; ++c) {
      const float p = expf(dot_row<DH>(q, Ra + c * DH) - ls);
      const float dp = dot_row<DH>(dout, Rb + c * DH);
      axpy_row<DH>(p * (dp - dl), Ra + c * DH, dq);
    }
    store_row<DH>(dqb + li * G.q_seq, dq, G.scale);
    load_row<DH>(Ra + li * DH, k, 1.0f);  // this lane's key/value rows for phase 2
    load_row<DH>(Rb + li * DH, v, 1.0f);
  }
  __syncthreads();
  if (active) {  // the row buffers now take scaled Q and dO, straight from registers (no re-read)
    store_row<DH>(Ra + li * DH, q, 1.0f);
    store_row<DH>(Rb + li * DH, dout, 1.0f);
  }
  __syncthreads();
  // phase 2: lane = key row -> dk, dv
  if (active) {
    float dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int r = 0; r < G.S; ++r) {
      const float p = expf(dot_row<DH>(k, Ra + r * DH) - lse_s[r]);
      const float dp = dot_row<DH>(v, Rb + r * DH);
      axpy_row<DH>(p * (dp - dl_s[r]), Ra + r * DH, dk);
      axpy_row<DH>(p, Rb + r * DH, dv);
    }
    store_row<DH>(dqb + li * G.q_seq + G.D, dk, 1.0f);
    store_row<DH>(dqb + li * G.q_seq + 2 * G.D, dv, 1.0f);
  }
}

// ---- any S: one group per 256-thread workgroup, keys (or queries) streamed through LDS chunks ----
constexpr int ATT_CHUNK = 128;

template <int DH>
__global__ void __launch_bounds__(256)
    attn_fwd_general(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ lse,
                     const AttnGeom G) {
  __shared__ float4 smem4[2 * ATT_CHUNK * DH / 4];
  float* Ks = reinterpret_cast<float*>(smem4);
  float* Vs = Ks + ATT_CHUNK * DH;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t outer = g / G.heads;
  const int head = (int)(g % G.heads);
  const float* qb = qkv + outer * G.q_outer + head * DH;
  for (int q0 = 0; q0 < G.S; q0 += 256) {
    const int qi = q0 + tid;
    const bool active = qi < G.S;
    float q[DH], acc[DH];
    if (active) load_row<DH>(qb + (int64_t)qi * G.q_seq, q, G.scale);
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < G.S; c0 += ATT_CHUNK) {
      const int nrows = min(ATT_CHUNK, G.S - c0);
      __syncthreads();
      stage_rows<DH>(qb + G.D + (int64_t)c0 * G.q_seq, G.q_seq, nrows, Ks, tid, 256);
      stage_rows<DH>(qb + 2 * G.D + (int64_t)c0 * G.q_seq, G.q_seq, nrows, Vs, tid, 256);
      __syncthreads();
      if (active) attend_rows<DH>(q, Ks, Vs, nrows, m, l, acc);
    }
    if (active) {
      store_row<DH>(o + outer * G.o_outer + (int64_t)qi * G.o_seq + head * DH, acc, 1.0f / l);
      if (lse != nullptr) lse[g * G.S + qi] = m + logf(l);
    }
  }
}

template <int DH>
__global__ void __launch_bounds__(256)
    attn_bwd_general(const float* __restrict__ qkv, const float* __restrict__ o,
                     const float* __restrict__ d_o, const float* __restrict__ lse,
                     float* __restrict__ dqkv, const AttnGeom G) {
  __shared__ float4 smem4[(2 * ATT_CHUNK * DH + 2 * ATT_CHUNK) / 4];
  float* As = reinterpret_cast<float*>(smem4);  // K chunk (phase 1) / scaled-Q chunk (phase 2)
  float* Bs = As + ATT_CHUNK * DH;              // V chunk (phase 1) / dO chunk (phase 2)
  float* lse_s = Bs + ATT_CHUNK * DH;
  float* dl_s = lse_s + ATT_CHUNK;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t outer = g / G.heads;
  const int head = (int)(g % G.heads);
  const float* qb = qkv + outer * G.q_outer + head * DH;
  const float* ob = o + outer * G.o_outer + head * DH;
  const float* dob = d_o + outer * G.o_outer + head * DH;
  float* dqb = dqkv + outer * G.q_outer + head * DH;

  // phase 1: thread = query row
  for (int q0 = 0; q0 < G.S; q0 += 256) {
    const int qi = q0 + tid;
    const bool active = qi < G.S;
    float q[DH], dout[DH], dq[DH];
    float ls = 0.f, dl = 0.f;
    if (active) {
      load_row<DH>(qb + (int64_t)qi * G.q_seq, q, G.scale);
      load_row<DH>(dob + (int64_t)qi * G.o_seq, dout, 1.0f);
      dl = dot_row<DH>(dout, ob + (int64_t)qi * G.o_seq);
      ls = lse[g * G.S + qi];
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.f;
    for (int c0 = 0; c0 < G.S; c0 += ATT_CHUNK) {
      const int nrows = min(ATT_CHUNK, G.S - c0);
      __syncthreads();
      stage_rows<DH>(qb + G.D + (int64_t)c0 * G.q_seq, G.q_seq, nrows, As, tid, 256);
      stage_rows<DH>(qb + 2 * G.D + (int64_t)c0 * G.q_seq, G.q_seq, nrows, Bs, tid, 256);
      __syncthreads();
      if (active) {
        for (int c = 0; c < nrows; ++c) {
          const float p = expf(dot_row<DH>(q, As + c * DH) - ls);
          const float dp = dot_row<DH>(dout, Bs + c * DH);
          axpy_row<DH>(p * (dp - dl), As + c * DH, dq);
        }
      }
    }
    if (active) store_row<DH>(dqb + (int64_t)qi * G.q_seq, dq, G.scale);
  }
  // phase 2: thread = key row
  for (int k0 = 0; k0 < G.S; k0 += 256) {
    const int ki = k0 + tid;
    const bool active = ki < G.S;
    float k[DH], v[DH], dk[DH], dv[DH];
    if (active) {
      load_row<DH>(qb + G.D + (int64_t)ki * G.q_seq, k, 1.0f);
      load_row<DH>(qb + 2 * G.D + (int64_t)ki * G.q_seq, v, 1.0f);
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int r0 = 0; r0 < G.S; r0 += ATT_CHUNK) {
      const int nrows = min(ATT_CHUNK, G.S - r0);
      __syncthreads();
      stage_rows<DH>(qb + (int64_t)r0 * G.q_seq, G.q_seq, nrows, As, tid, 256, G.scale);
      stage_rows<DH>(dob + (int64_t)r0 * G.o_seq, G.o_seq, nrows, Bs, tid, 256);
      if (tid < nrows) lse_s[tid] = lse[g * G.S + r0 + tid];
      __syncthreads();
      if (tid < nrows) {
        float dout[DH];
        load_row<DH>(Bs + tid * DH, dout, 1.0f);
        dl_s[tid] = dot_row<DH>(dout, ob + (int64_t)(r0 + tid) * G.o_seq);
      }
      __syncthreads();
      if (active) {
        for (int r = 0; r < nrows; ++r) {
          const float p = expf(dot_row<DH>(k, As + r * DH) - lse_s[r]);
          const float dp = dot_row<DH>(v, Bs + r * DH);
          axpy_row<DH>(p * (dp - dl_s[r]), As + r * DH, dk);
          axpy_row<DH>(p, Bs + r * DH, dv);
        }
      }
    }
    if (active) {
      store_row<DH>(dqb + (int64_t)ki * G.q_seq + G.D, dk, 1.0f);
      store_row<DH>(dqb + (int64_t)ki * G.q_seq + 2 * G.D, dv, 1.0f);
    }
  }
}

bool attn_head_dim_supported(int dh) { return dh == 16 || dh == 20 || dh == 32 || dh == 48 || dh == 64; }

#define NRL_DISPATCH_DH(dh, ...)                         \
  switch (dh) {                                          \
    case 16: { constexpr int DH = 16; __VA_ARGS__; } break; \
    case 20: { constexpr int DH = 20; __VA_ARGS__; } break; \
    case 32: { constexpr int DH = 32; __VA_ARGS__; } break; \
    case 48: { constexpr int DH = 48; __VA_ARGS__; } break; \
    case 64: { constexpr int DH = 64; __VA_ARGS__; } break; \
    default: set_error("unsupported head dim %d", dh); return NRL_E_INVALID; \
  }

static int check_geom(const AttnGeom& G) {
  NRL_REQUIRE(G.S > 0 && G.groups >= 0 && G.heads > 0, "bad attention geometry");
  NRL_REQUIRE(G.D % 4 == 0 && G.q_seq % 4 == 0 && G.q_outer % 4 == 0 && G.o_seq % 4 == 0 &&
                  G.o_outer % 4 == 0, "attention strides must be multiples of 4 floats");
  NRL_REQUIRE(G.groups < (1LL << 31), "too many attention groups");
  return NRL_OK;
}

int attn_fwd(const float* qkv, float* o, float* lse, const AttnGeom& G, hipStream_t stream) {
  NRL_TRY(check_geom(G));
  if (G.groups == 0) return NRL_OK;
  // long sequences (the seq-first attention across users / across the news of a PLM call) are genuinely dense:
  // matrix cores.  NRL_ATTN_MFMA=0 keeps them on the vector-ALU kernel (A/B measurements).
  static const bool use_mfma = [] {
    const char* e = getenv("NRL_ATTN_MFMA");
    return !(e != nullptr && e[0] == '0');
  }();
  if (use_mfma && G.S >= 64) return attn_fwd_mfma(qkv, o, lse, G, stream);
  NRL_DISPATCH_DH(G.dh, {
    if (G.S <= 32) {
      constexpr int GPW = DH <= 32 ? 8 : 4;
      hipLaunchKernelGGL((attn_fwd_small<DH, GPW>), dim3((unsigned)ceil_div(G.groups, GPW)),
                         dim3(GPW * 32), 0, stream, qkv, o, lse, G);
    } else if (G.S <= 64) {
      constexpr int GPW = DH <= 32 ? 4 : 2;
      hipLaunchKernelGGL((attn_fwd_small<DH, GPW, 64>), dim3((unsigned)ceil_div(G.groups, GPW)),
                         dim3(GPW * 64), 0, stream, qkv, o, lse, G);
    } else {
      hipLaunchKernelGGL((attn_fwd_general<DH>), dim3((unsigned)G.groups), dim3(256), 0, stream, qkv,
                         o, lse, G);
    }
  });
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int attn_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
             const AttnGeom& G, hipStream_t stream) {
  NRL_TRY(check_geom(G));
  NRL_REQUIRE(lse != nullptr, "attention backward needs the saved log-sum-exp");
  if (G.groups == 0) return NRL_OK;
  static const bool use_mfma = [] {
    const char* e = getenv("NRL_ATTN_MFMA");
    return !(e != nullptr && e[0] == '0');
  }();
  if (use_mfma && G.S >= 64) return attn_bwd_mfma(qkv, o, d_o, lse, dqkv, G, stream);
  NRL_DISPATCH_DH(G.dh, {
    if (G.S <= 32) {
      constexpr int GPW = DH <= 32 ? 8 : 4;
      hipLaunchKernelGGL((attn_bwd_small<DH, GPW>), dim3((unsigned)ceil_div(G.groups, GPW)),
                         dim3(GPW * 32), 0, stream, qkv, o, d_o, lse, dqkv, G);
    } else if (G.S <= 64) {
      constexpr int GPW = DH <= 32 ? 4 : 2;
      hipLaunchKernelGGL((attn_bwd_small<DH, GPW, 64>), dim3((unsigned)ceil_div(G.groups, GPW)),
                         dim3(GPW * 64), 0, stream, qkv, o, d_o, lse, dqkv, G);
    } else {
      hipLaunchKernelGGL((attn_bwd_general<DH>), dim3((unsigned)G.groups), dim3(256), 0, stream, qkv,
                         o, d_o, lse, dqkv, G);
    }
  });
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// additive-attention pooling
// =============================================================================================
// One workgroup (4 waves) per group of S rows.  All global traffic is float4: a row of t (Q floats) or y
// (D floats) is one coalesced wave access, and the column passes split the 256 threads into
// RG = 256 / (D/4) row-groups x (D/4) float4 columns (D = 300: 3 x 75) whose partial sums meet in LDS.
__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  return fmaf(a.w, b.w, acc);
}

__global__ void __launch_bounds__(256)
    pool_fwd_kernel(const float* __restrict__ t, const float* __restrict__ q_a,
                    const float* __restrict__ y, int S, int Q, int D, float* __restrict__ w,
                    float* __restrict__ out) {
  extern __shared__ float sm[];  // a[S] -> w[S]
  __shared__ float4 part[256];
  float* a_s = sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g = blockIdx.x;
  const int64_t row0 = g * S;
  const int Q4 = Q >> 2, D4 = D >> 2;
  const float4* q4 = reinterpret_cast<const float4*>(q_a);
  for (int l = wave; l < S; l += 4) {
    const float4* tr = reinterpret_cast<const float4*>(t + (row0 + l) * Q);
    float acc = 0.f;
    for (int n = lane; n < Q4; n += 64) acc = dot4(tr[n], q4[n], acc);
    acc = wave_sum(acc);
    if (lane == 0) a_s[l] = acc;
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int l = lane; l < S; l += 64) mx = fmaxf(mx, a_s[l]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int l = lane; l < S; l += 64) sum += expf(a_s[l] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int l = lane; l < S; l += 64) {
      const float wl = expf(a_s[l] - mx) * inv;
      a_s[l] = wl;
      w[row0 + l] = wl;
    }
  }
  __syncthreads();
  // out[d] = sum_l w_l y[l][d]
  const int cpp = D4 < 256 ? D4 : 256;           // float4 columns per pass
  const int RG = 256 / cpp, rg = tid / cpp, cc = tid - rg * cpp;
  const float4* y4 = reinterpret_cast<const float4*>(y + row0 * D);
  float4* o4 = reinterpret_cast<float4*>(out + g * D);
  for (int c4 = cc; c4 < D4; c4 += cpp) {        // one iteration unless D > 1024
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rg < RG) {
      for (int l = rg; l < S; l += RG) {
        const float wl = a_s[l];
        const float4 v = y4[(int64_t)l * D4 + c4];
        acc.x = fmaf(wl, v.x, acc.x); acc.y = fmaf(wl, v.y, acc.y);
        acc.z = fmaf(wl, v.z, acc.z); acc.w = fmaf(wl, v.w, acc.w);
      }
    }
    if (RG == 1) {
      o4[c4] = acc;
    } else {
      part[tid] = acc;
      __syncthreads();
      if (rg == 0) {
        for (int r = 1; r < RG; ++r) {
          const float4 p = part[r * cpp + cc];
          acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        o4[c4] = acc;
      }
      __syncthreads();
    }
  }
}

// (a, b) -> packed bf16 (hi_a | hi_b << 16), (lo_a | lo_b << 16): the bf16x3 split of nrl_gemm_bf16x3.h (same builtins,
// same round-to-nearest-even values; repeated here because this translation unit does not see the GEMM headers)
typedef __bf16 pool_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pool_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const pool_bf16x2 h = __builtin_convertvector(pool_f32x2{a, b}, pool_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  const pool_bf16x2 l = __builtin_convertvector(pool_f32x2{a - ha, b - hb}, pool_bf16x2);
  lo = __builtin_bit_cast(uint32_t, l);
}

__global__ void __launch_bounds__(256)
    pool_bwd_pre_kernel(const float* __restrict__ d_out, const float* __restrict__ y,
                        const float* __restrict__ w, float* __restrict__ t_dpre,
                        const float* __restrict__ q_a, float* __restrict__ dq_a, int64_t groups, int S, int Q,
                        int D, unsigned char* __restrict__ dpre_planes, int ncb_q,
                        const unsigned char* __restrict__ y_planes, int ncb_y) {
  // y_planes != nullptr (y == nullptr): y exists only as (hi, lo) bf16 fragment-block planes over the rows (the fused news
  // tail, nrl_news_tail.h, never writes it as fp32) and enters the dot products as hi + lo
  // dpre_planes != nullptr: d_pre goes out as (hi, lo) bf16 fragment-block planes over the rows (KCPlanesG; its readers
  // are the additive-attention dgrad and weight-gradient GEMMs only) and t is left as it is
  extern __shared__ float sm[];  // c[S] -> da[S]
  __shared__ float4 part[256];
  float* c_s = sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Q4 = Q >> 2, D4 = D >> 2;
  const int cpp = Q4 < 256 ? Q4 : 256;
  const int RG = 256 / cpp, rg = tid / cpp, cc = tid - rg * cpp;
  const float4* q4 = reinterpret_cast<const float4*>(q_a);
  // dq_a partial sums live in registers across ALL groups of this workgroup (persistent over groups) and
  // are flushed with one coalesced atomic per element at the end: the query gradient is a reduction over
  // every row of the batch onto Q addresses, and atomics are served per (instruction, cache line)
  float4 accq[4];                                 // Q4 <= 4 * 256
#pragma unroll
  for (int i = 0; i < 4; ++i) accq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t row0 = g * S;
    const float4* dr = reinterpret_cast<const float4*>(d_out + g * D);
    // c_l = d_out . y_l: sixteen lanes per row, sixteen rows of the group at a time (one wave per row left every row a
    // serial load -> reduce chain: the kernel sat at ~45 us per group however few groups there were)
    {
      const int gi = tid >> 4, li = tid & 15;
      for (int l = gi; l < S; l += 16) {
        float acc = 0.f;
        if (y_planes != nullptr) {
          const int64_t m = row0 + l;
          const unsigned char* blk = y_planes + (m >> 4) * ncb_y * 1024 + (m & 15) * 32;
          for (int d = li; d < D4; d += 16) {
            const unsigned char* src = blk + (d >> 2) * 1024 + (d & 3) * 8;
            const uint2 h = *reinterpret_cast<const uint2*>(src), lo = *reinterpret_cast<const uint2*>(src + 512);
            const float4 yv = make_float4(__builtin_bit_cast(float, h.x << 16) + __builtin_bit_cast(float, lo.x << 16),
                                          __builtin_bit_cast(float, h.x & 0xffff0000u) + __builtin_bit_cast(float, lo.x & 0xffff0000u),
                                          __builtin_bit_cast(float, h.y << 16) + __builtin_bit_cast(float, lo.y << 16),
                                          __builtin_bit_cast(float, h.y & 0xffff0000u) + __builtin_bit_cast(float, lo.y & 0xffff0000u));
            acc = dot4(dr[d], yv, acc);
          }
        } else {
          const float4* yr = reinterpret_cast<const float4*>(y + (row0 + l) * D);
          for (int d = li; d < D4; d += 16) acc = dot4(dr[d], yr[d], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 8, 64);
        if (li == 0) c_s[l] = acc;
      }
    }
    __syncthreads();
    if (wave == 0) {
      float dbar = 0.f;
      for (int l = lane; l < S; l += 64) dbar = fmaf(w[row0 + l], c_s[l], dbar);
      dbar = wave_sum(dbar);
      for (int l = lane; l < S; l += 64) c_s[l] = w[row0 + l] * (c_s[l] - dbar);  // da_l
    }
    __syncthreads();
    // t -> d_pre = da * q * (1 - t^2) in place; dq_a[n] += sum_l da_l t[l][n]
    float4* t4 = reinterpret_cast<float4*>(t_dpre + row0 * Q);
    if (rg < RG) {
      int it = 0;
      for (int c4 = cc; c4 < Q4; c4 += cpp, ++it) {
        const float4 qn = q4[c4];
        float4 acc = accq[it & 3];
        // eight rows of t in flight per lane (one row at a time, the in-place store of row l ordered the load of row
        // l + RG behind it: a serial chain of ~2 us memory round trips per row, ~40 us per group whatever the batch)
        for (int l0 = rg; l0 < S; l0 += 8 * RG) {
          float4 tv8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int l = l0 + u * RG;
            tv8[u] = l < S ? t4[(int64_t)l * Q4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int l = l0 + u * RG;
            if (l < S) {
              const float da = c_s[l];
              const float4 tv = tv8[u];
              acc.x = fmaf(da, tv.x, acc.x); acc.y = fmaf(da, tv.y, acc.y);
              acc.z = fmaf(da, tv.z, acc.z); acc.w = fmaf(da, tv.w, acc.w);
              const float4 dp = make_float4(da * qn.x * (1.0f - tv.x * tv.x), da * qn.y * (1.0f - tv.y * tv.y),
                                            da * qn.z * (1.0f - tv.z * tv.z), da * qn.w * (1.0f - tv.w * tv.w));
              if (dpre_planes != nullptr) {
                const int64_t m = row0 + l;
                uint32_t h0, l0_, h1, l1;
                split_pair(dp.x, dp.y, h0, l0_);
                split_pair(dp.z, dp.w, h1, l1);
                unsigned char* dst = dpre_planes + ((m >> 4) * ncb_q + (c4 >> 2)) * 1024 + (m & 15) * 32 + (c4 & 3) * 8;
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + 512) = make_uint2(l0_, l1);
              } else {
                t4[(int64_t)l * Q4 + c4] = dp;
              }
            }
          }
        }
        accq[it & 3] = acc;
      }
    }
    __syncthreads();                              // c_s is rewritten by the next group
  }
  // flush: row-groups meet in LDS, then Q contiguous atomics
  int it = 0;
  for (int c4 = cc; c4 < Q4; c4 += cpp, ++it) {
    float4 acc = accq[it & 3];
    if (RG > 1) {
      part[tid] = rg < RG ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      if (rg == 0)
        for (int r = 1; r < RG; ++r) {
          const float4 p = part[r * cpp + cc];
          acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
      __syncthreads();
      if (rg == 0) part[cc] = acc;
      __syncthreads();
      const float* pf = reinterpret_cast<const float*>(part);
      if (tid < 4 * cpp) atomicAdd(dq_a + 4 * (c4 - cc) + tid, pf[tid]);
      __syncthreads();
    } else {
      atomicAdd(dq_a + 4 * c4 + 0, acc.x);
      atomicAdd(dq_a + 4 * c4 + 1, acc.y);
      atomicAdd(dq_a + 4 * c4 + 2, acc.z);
      atomicAdd(dq_a + 4 * c4 + 3, acc.w);
    }
  }
}

int pool_fwd(const float* t, const float* q_a, const float* y, int64_t groups, int S, int Q, int D,
             float* w, float* out, hipStream_t stream) {
  if (groups == 0) return NRL_OK;
  NRL_REQUIRE(S > 0 && S <= 8192 && groups < (1LL << 31), "pool_fwd: bad shape");
  NRL_REQUIRE(Q % 4 == 0 && D % 4 == 0, "pool_fwd: Q and D must be multiples of 4");
  hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)groups), dim3(256), S * sizeof(float), stream, t,
                     q_a, y, S, Q, D, w, out);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int pool_bwd_pre(const float* d_out, const float* y, const float* w, float* t_dpre, const float* q_a,
                 float* dq_a, int64_t groups, int S, int Q, int D, hipStream_t stream, void* dpre_planes,
                 const void* y_planes) {
  if (groups == 0) return NRL_OK;
  NRL_REQUIRE((y != nullptr) != (y_planes != nullptr), "pool_bwd_pre: y as fp32 rows or as planes");
  NRL_REQUIRE(S > 0 && S <= 8192 && groups < (1LL << 31), "pool_bwd_pre: bad shape");
  NRL_REQUIRE(Q % 4 == 0 && D % 4 == 0, "pool_bwd_pre: Q and D must be multiples of 4");
  NRL_REQUIRE(Q <= 4096, "pool_bwd_pre: query_dim > 4096 unsupported");
  const unsigned grid = (unsigned)(groups < 2048 ? groups : 2048);  // 8 workgroups per CU, persistent over groups
  hipLaunchKernelGGL(pool_bwd_pre_kernel, dim3(grid), dim3(256), S * sizeof(float), stream, d_out, y, w, t_dpre,
                     q_a, dq_a, groups, S, Q, D, (unsigned char*)dpre_planes, (Q + 15) / 16, (const unsigned char*)y_planes,
                     (D + 16) / 16);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// dense batching (to_dense_batch as an index map) -- float4 granularity, D % 4 == 0
// =============================================================================================
__global__ void to_dense_fwd_kernel(const float4* __restrict__ x, const int64_t* __restrict__ offsets,
                                    int64_t total4, int64_t max_len, int D4, float4* __restrict__ dense) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const int64_t slot = i / D4;
    const int64_t b = slot / max_len, h = slot % max_len;
    const int64_t beg = offsets[b], cnt = offsets[b + 1] - beg;
    dense[i] = h < cnt ? x[(beg + h) * D4 + d4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__global__ void to_dense_bwd_kernel(const float4* __restrict__ d_dense,
                                    const int64_t* __restrict__ offsets, int64_t total4,
                                    int64_t max_len, int D4, float4* __restrict__ d_x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const int64_t slot = i / D4;
    const int64_t b = slot / max_len, h = slot % max_len;
    const int64_t beg = offsets[b], cnt = offsets[b + 1] - beg;
    if (h < cnt) d_x[(beg + h) * D4 + d4] = d_dense[i];
  }
}

static unsigned grid_for(int64_t n, int block) {
  int64_t b = ceil_div(n, block);
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int to_dense_fwd(const float* x, const int64_t* offsets, int64_t B, int64_t max_len, int D,
                 float* dense, hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "to_dense: dim must be a multiple of 4");
  const int64_t total4 = B * max_len * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(to_dense_fwd_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream,
                     (const float4*)x, offsets, total4, max_len, D / 4, (float4*)dense);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int to_dense_bwd(const float* d_dense, const int64_t* offsets, int64_t B, int64_t max_len, int D,
                 int64_t /*n_rows*/, float* d_x, hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "to_dense: dim must be a multiple of 4");
  const int64_t total4 = B * max_len * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(to_dense_bwd_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream,
                     (const float4*)d_dense, offsets, total4, max_len, D / 4, (float4*)d_x);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// late fusion: mean of the clicked-news vectors (division by the TRUE history size, sum over all slots)
// =============================================================================================
__global__ void hist_mean_fwd_kernel(const float4* __restrict__ hist, const int64_t* __restrict__ offsets,
                                     int64_t total4, int64_t max_len, int D4, float4* __restrict__ user) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / D4;
    const int d4 = (int)(i % D4);
    const float inv = 1.0f / (float)(offsets[b + 1] - offsets[b]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t h = 0; h < max_len; ++h) {
      const float4 v = hist[(b * max_len + h) * D4 + d4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    user[i] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

__global__ void hist_mean_bwd_kernel(const float4* __restrict__ d_user, const int64_t* __restrict__ offsets,
                                     int64_t total4, int64_t max_len, int D4, float4* __restrict__ d_hist) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const int64_t b = (i / D4) / max_len;
    const float inv = 1.0f / (float)(offsets[b + 1] - offsets[b]);
    const float4 g = d_user[b * D4 + d4];
    d_hist[i] = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
  }
}

int hist_mean_fwd(const float* hist, const int64_t* offsets, int64_t B, int64_t max_len, int D, float* user,
                  hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "hist_mean: dim must be a multiple of 4");
  const int64_t total4 = B * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(hist_mean_fwd_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream, (const float4*)hist,
                     offsets, total4, max_len, D / 4, (float4*)user);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int hist_mean_bwd(const float* d_user, const int64_t* offsets, int64_t B, int64_t max_len, int D, float* d_hist,
                  hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "hist_mean: dim must be a multiple of 4");
  const int64_t total4 = B * max_len * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(hist_mean_bwd_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream, (const float4*)d_user,
                     offsets, total4, max_len, D / 4, (float4*)d_hist);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// scorer + loss
// =============================================================================================
// one wavefront per (b, c): scores[b, c] = <user[b], cand[b, c]>
__global__ void __launch_bounds__(256)
    dot_scores_fwd_kernel(const float* __restrict__ user, const float* __restrict__ cand, int64_t BC,
                          int64_t C, int D, float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= BC) return;
  const float* u = user + (pair / C) * D;
  const float* cv = cand + pair * D;
  float part = 0.f;
  for (int d = lane; d < D; d += 64) part = fmaf(u[d], cv[d], part);
  part = wave_sum(part);
  if (lane == 0) scores[pair] = part;
}

// thread per (b, d): d_user = sum_c ds * cand; d_cand = ds * user
__global__ void dot_scores_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ user,
                                      const float* __restrict__ cand, int64_t B, int64_t C, int D,
                                      float* __restrict__ d_user, float* __restrict__ d_cand) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int64_t b = i / D;
  const int d = (int)(i % D);
  const float u = user[i];
  float acc = 0.f;
  for (int64_t c = 0; c < C; ++c) {
    const float s = ds[b * C + c];
    const int64_t idx = (b * C + c) * D + d;
    acc = fmaf(s, cand[idx], acc);
    d_cand[idx] = s * u;
  }
  d_user[i] = acc;
}

// CrossEntropyLoss with probability targets, mean over rows; one workgroup, deterministic sum
__global__ void __launch_bounds__(256)
    ce_kernel(const float* __restrict__ scores, const float* __restrict__ y, int64_t B, int64_t C,
              float grad_scale, float* __restrict__ loss, float* __restrict__ d_scores) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float local = 0.f;
  const float invB = 1.0f / (float)B;
  for (int64_t b = tid; b < B; b += 256) {
    const float* s = scores + b * C;
    const float* yy = y + b * C;
    float mx = -INFINITY;
    for (int64_t c = 0; c < C; ++c) mx = fmaxf(mx, s[c]);
    float sum = 0.f, ysum = 0.f;
    for (int64_t c = 0; c < C; ++c) { sum += expf(s[c] - mx); ysum += yy[c]; }
    const float lse = mx + logf(sum);
    float lb = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float logp = s[c] - lse;
      lb -= yy[c] * logp;
      if (d_scores != nullptr) d_scores[b * C + c] = (expf(logp) * ysum - yy[c]) * invB * grad_scale;
    }
    local += lb;
  }
  local = wave_sum(local);
  if ((tid & 63) == 0) red[tid >> 6] = local;
  __syncthreads();
  if (tid == 0) *loss = (red[0] + red[1] + red[2] + red[3]) * invB;
}

// =============================================================================================
// supervised contrastive loss over the (B, C) score matrix (reference losses.py:6-40 on top of
// pytorch-metric-learning 2.2.0's GenericPairLoss.mat_based_loss + AvgNonZeroReducer; index construction
// nrms_module.py:289-304).  Row b: positives = {c : y != 0}, negatives = the other REAL candidates (c < size_b);
// x = s / T;  loss_b = -(1 / (npos + tiny)) sum_{p} (x_p - logsumexp_{c < size_b} x_c);
// loss = mean of the loss_b that are > 0.  One workgroup: B is a batch size.
// =============================================================================================
struct SupConRow {
  float loss, lse, mx;
  int npos, n;
};

__device__ __forceinline__ SupConRow supcon_row(const float* s, const float* y, int64_t C, int64_t n, float inv_t) {
  SupConRow r;
  r.n = (int)n;
  // (the reference first shifts by the max of the whole row, pads included, and torch.logsumexp then shifts again
  //  by the max of the kept entries; shifting once by the latter is the same value without the underflow window)
  float mx = n > 0 ? -INFINITY : 0.f;
  for (int64_t c = 0; c < n; ++c) mx = fmaxf(mx, s[c] * inv_t);
  float sum = 0.f, possum = 0.f;
  int npos = 0;
  for (int64_t c = 0; c < n; ++c) {
    const float x = s[c] * inv_t - mx;
    sum += expf(x);
    if (y[c] != 0.f) { ++npos; possum += x; }
  }
  for (int64_t c = n; c < C; ++c) npos += y[c] != 0.f;                 // (cannot happen with to_dense_batch labels)
  r.mx = mx;
  r.npos = npos;
  r.lse = n > 0 ? logf(sum) : 0.f;                                     // masked logsumexp: empty set -> 0
  r.loss = -(possum - (float)npos * r.lse) / ((float)npos + FLT_MIN);
  return r;
}

__global__ void __launch_bounds__(256)
    supcon_kernel(const float* __restrict__ scores, const float* __restrict__ y, const int64_t* __restrict__ sizes,
                  int64_t B, int64_t C, float inv_t, float grad_scale, float* __restrict__ loss,
                  float* __restrict__ d_scores) {
  __shared__ float red[4][4];
  const int tid = threadIdx.x;
  float lsum = 0.f, lcnt = 0.f, tpos = 0.f, tneg = 0.f;
  for (int64_t b = tid; b < B; b += 256) {
    const SupConRow r = supcon_row(scores + b * C, y + b * C, C, sizes[b], inv_t);
    if (r.loss > 0.f) { lsum += r.loss; lcnt += 1.f; }
    tpos += (float)r.npos;
    tneg += (float)(r.n - r.npos);
  }
  float v[4] = {wave_sum(lsum), wave_sum(lcnt), wave_sum(tpos), wave_sum(tneg)};
  if ((tid & 63) == 0)
    for (int i = 0; i < 4; ++i) red[i][tid >> 6] = v[i];
  __syncthreads();
  float tot[4];
  for (int i = 0; i < 4; ++i) tot[i] = red[i][0] + red[i][1] + red[i][2] + red[i][3];
  // losses.py:13-15 (fewer than two pairs of either kind) and :20 (no positive or no negative pair): zero loss
  const bool zero = (tot[2] <= 1.f && tot[3] <= 1.f) || !(tot[2] > 0.f && tot[3] > 0.f) || tot[1] < 1.f;
  if (tid == 0) *loss = zero ? 0.f : tot[0] / tot[1];
  if (d_scores == nullptr) return;
  const float k = zero ? 0.f : inv_t * grad_scale / tot[1];
  for (int64_t b = tid; b < B; b += 256) {
    const float* s = scores + b * C;
    const float* yy = y + b * C;
    const int64_t n = sizes[b];
    const SupConRow r = supcon_row(s, yy, C, n, inv_t);
    const float live = (r.loss > 0.f) ? k / ((float)r.npos + FLT_MIN) : 0.f;
    for (int64_t c = 0; c < C; ++c) {
      float g = 0.f;
      if (c < n && live != 0.f) {
        const float p = expf(s[c] * inv_t - r.mx - r.lse);
        g = ((float)r.npos * p - (yy[c] != 0.f ? 1.f : 0.f)) * live;
      }
      d_scores[b * C + c] = g;
    }
  }
}

int supcon_loss_fwd_bwd(const float* scores, const float* y, const int64_t* sizes, int64_t B, int64_t C,
                        float temperature, float grad_scale, float* loss, float* d_scores, hipStream_t stream) {
  NRL_REQUIRE(B > 0 && C > 0 && temperature > 0.f, "supcon: bad arguments");
  hipLaunchKernelGGL(supcon_kernel, dim3(1), dim3(256), 0, stream, scores, y, sizes, B, C, 1.0f / temperature,
                     grad_scale, loss, d_scores);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int dot_scores_fwd(const float* user, const float* cand, int64_t B, int64_t C, int D, float* scores,
                   hipStream_t stream) {
  const int64_t BC = B * C;
  if (BC == 0) return NRL_OK;
  hipLaunchKernelGGL(dot_scores_fwd_kernel, dim3((unsigned)ceil_div(BC, 4)), dim3(256), 0, stream, user,
                     cand, BC, C, D, scores);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int dot_scores_bwd(const float* d_scores, const float* user, const float* cand, int64_t B, int64_t C,
                   int D, float* d_user, float* d_cand, hipStream_t stream) {
  if (B * D == 0) return NRL_OK;
  hipLaunchKernelGGL(dot_scores_bwd_kernel, dim3((unsigned)ceil_div(B * D, 256)), dim3(256), 0, stream,
                     d_scores, user, cand, B, C, D, d_user, d_cand);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int ce_loss_fwd_bwd(const float* scores, const float* y, int64_t B, int64_t C, float grad_scale,
                    float* loss, float* d_scores, hipStream_t stream) {
  NRL_REQUIRE(B > 0 && C > 0, "ce: empty batch");
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, stream, scores, y, B, C, grad_scale, loss,
                     d_scores);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// dense Adam (the update rule of torch _single_tensor_adam; see adam_elem for the arithmetic), HBM-bound: 4 reads + 4 writes per element
// =============================================================================================
struct AdamConst {
  float b1, b2, one_m_b1, one_m_b2, step_size, inv_sqrt_bc2, eps, grad_scale;
};

// One element, one step.  Every operation is spelled out (no contraction left to the compiler), so the dense kernel and the
// lazy row kernel below execute the SAME instruction sequence and a replayed zero-gradient step reproduces the dense one
// bit for bit.  sqrt and the reciprocal are the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32: the relative error they add to an
// update of size <= lr is ~2e-7 (the IEEE sequences cost ~20 instructions per element, which is what a replayed step pays).
__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, const AdamConst& k) {
  const float gg = __fmul_rn(g, k.grad_scale);
  m = __fmaf_rn(k.one_m_b1, gg, __fmul_rn(m, k.b1));
  v = __fmaf_rn(__fmul_rn(k.one_m_b2, gg), gg, __fmul_rn(v, k.b2));
  const float denom = __fmaf_rn(__builtin_amdgcn_sqrtf(v), k.inv_sqrt_bc2, k.eps);
  p = __fmaf_rn(-k.step_size, __fmul_rn(m, __builtin_amdgcn_rcpf(denom)), p);
}

__global__ void __launch_bounds__(256)
    adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                float* __restrict__ v, int64_t n, AdamConst k, int zero_grad) {
  const int64_t n4 = n / 4;
  float4* p4 = (float4*)p; float4* g4 = (float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam_elem(pp.x, gg.x, mm.x, vv.x, k);
    adam_elem(pp.y, gg.y, mm.y, vv.y, k);
    adam_elem(pp.z, gg.z, mm.z, vv.z, k);
    adam_elem(pp.w, gg.w, mm.w, vv.w, k);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
      adam_elem(pp, gg, mm, vv, k);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (zero_grad) g[i] = 0.f;
    }
  }
}

int adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
              double eps, int64_t step, float grad_scale, int zero_grad, hipStream_t stream) {
  if (n == 0) return NRL_OK;
  NRL_REQUIRE(step >= 1, "adam: step is 1-based");
  NRL_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
              "adam: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  AdamConst k;
  k.b1 = (float)b1; k.b2 = (float)b2;
  k.one_m_b1 = (float)(1.0 - b1); k.one_m_b2 = (float)(1.0 - b2);
  k.step_size = (float)(lr / bc1);
  k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.eps = (float)eps; k.grad_scale = grad_scale;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, stream, p, g, m, v, n, k,
                     zero_grad);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// Lazy dense Adam for a (rows, dim) table whose gradient is zero outside the rows a step touched
// =============================================================================================
// Dense Adam moves EVERY row every step (abstract_recommender.py:96: torch.optim.Adam over all parameters; a row with a zero
// gradient still drifts through its decaying moments).  For a row with g = 0 that update is a deterministic fp32 recurrence
// in (p, m, v, step), so it can be applied LATER -- when the row is next read (the forward's gather) or written (a non-zero
// gradient), or when its state is exported -- with the same instruction sequence (adam_elem with g = 0) and therefore the same
// bits.  `last[r]` = the step up to which row r has been advanced.  One kernel, three uses:
//   catch-up  rows with mark[r] == tag -> advanced to step upto0 = t - 1 (g = 0), before the forward of step t reads them;
//   update    the same rows -> (advanced to t - 1 if another rank touched them, then) step t with their gradient row, which is
//             cleared; last = t;
//   flush     a slice (rows r = offset + j * stride: the rolling flush that bounds every row's lag) or all rows -> upto0.
//   early catch-up (round 5)  the catch-up of step t + 1, issued on a side stream WHILE step t runs, for a caller that knows
//             the next batch's ids: rows marked for t + 1 (a second mark array) are advanced to step t -- except the rows step t
//             itself owns (excl[r] == t: its update brings them to t anyway, and must not be raced).  Sound on one rank only: a
//             row outside the running step's batch has a zero gradient at that step.
// The bias corrections of the 128 steps up to the current one travel as kernel arguments (computed on the host exactly as
// adam_step computes them), indexed step & 127: a row may lag by at most 127 steps (status[0] is set if one lags further --
// the trainer's rolling flush keeps every lag <= its period).
constexpr int ADAM_WIN = 128;
struct AdamRowsArgs {
  float *p, *g, *m, *v;
  int32_t* last;
  const int32_t* mark;     // or null: every candidate row
  const int32_t* excl;     // or null; else rows with excl[r] == excl_tag are skipped (the rows the RUNNING step owns, see below)
  int32_t excl_tag;
  int32_t* status;
  int64_t n_cand, stride, offset;
  int32_t tag, upto0, with_grad, D;
  int32_t scan;            // with_grad and no mark: a candidate row is updated iff its gradient row has a non-zero element (a dense
                           // all-reduce leaves no list of touched rows; an all-zero row's update IS the replay it gets later)
  AdamConst base;          // b1, b2, 1 - b1, 1 - b2, eps, grad_scale (step_size / inv_sqrt_bc2 come from the window)
  float step_size[ADAM_WIN], inv_sqrt_bc2[ADAM_WIN];
};

// CAND candidate rows per 256-thread workgroup: 32 where a mark selects a sparse subset (several rows per wave), 4 for the
// flushes (one row per wave: a flushed row replays up to `period` steps, a long dependent chain).  A lane owns one float4 and
// one single float of every 320-float chunk of the row (D = 300: 5 elements per lane, one pass).
template <int CAND>
__global__ void __launch_bounds__(256) adam_rows_kernel(const AdamRowsArgs A) {
  __shared__ int s_rows[CAND];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 64) {
    // candidates are dealt round-robin over the workgroups: the token ids of a batch crowd the low rows (Zipf), and contiguous
    // blocks of candidates would hand a few workgroups 32 marked rows and most of them none
    const int64_t j = (int64_t)blockIdx.x + (int64_t)tid * gridDim.x;
    bool take = false;
    if (tid < CAND && j < A.n_cand) {
      const int64_t r = A.offset + j * A.stride;
      take = (A.mark == nullptr || A.mark[r] == A.tag) && (A.with_grad || A.last[r] < A.upto0);     // (scan: every candidate)
      if (A.excl != nullptr && A.excl[r] == A.excl_tag) take = false;
    }
    const unsigned long long b = __ballot(take);
    if (take) s_rows[__popcll(b & ((1ull << lane) - 1ull))] = tid;
    if (lane == 0) s_n = __popcll(b);
  }
  __syncthreads();
  const int n = s_n;
  const int D = A.D;
  for (int i = wave; i < n; i += 4) {
    const int64_t r = A.offset + ((int64_t)blockIdx.x + (int64_t)s_rows[i] * gridDim.x) * A.stride;
    float* const gr = A.g + r * D;
    if (A.scan) {
      // any non-zero gradient element in this row?  (D <= 320 * 4 chunks is plenty for the tables this serves)
      bool nz = false;
      for (int base = 0; base < D; base += 256) {
        const int q = base + 4 * lane;
        if (q + 3 < D) {
          const float4 gv = *reinterpret_cast<const float4*>(gr + q);
          nz = nz || gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f;
        }
      }
      if (__ballot(nz) == 0ull) continue;
    }
    const int from = __builtin_amdgcn_readfirstlane(A.last[r]);
    if (A.upto0 - from >= ADAM_WIN && lane == 0) A.status[0] = 1;
    float* const pr = A.p + r * D;
    float* const mr = A.m + r * D;
    float* const vr = A.v + r * D;
    for (int base = 0; base < D; base += 320) {
      const int q = base + 4 * lane;                       // this lane's float4 ...
      const bool has4 = q + 3 < D && q < base + 256;
      const int e = base + 256 + lane;                     // ... and its single float of the chunk
      const bool has1 = e < D && e < base + 320;
      float4 pp = make_float4(0.f, 0.f, 0.f, 0.f), mm = pp, vv = pp;
      float p1 = 0.f, m1 = 0.f, v1 = 0.f;
      if (has4) { pp = *reinterpret_cast<const float4*>(pr + q); mm = *reinterpret_cast<const float4*>(mr + q); vv = *reinterpret_cast<const float4*>(vr + q); }
      if (has1) { p1 = pr[e]; m1 = mr[e]; v1 = vr[e]; }
      AdamConst k = A.base;
      for (int s = from + 1; s <= A.upto0; ++s) {
        k.step_size = A.step_size[s & (ADAM_WIN - 1)];
        k.inv_sqrt_bc2 = A.inv_sqrt_bc2[s & (ADAM_WIN - 1)];
        float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f, z4 = 0.f;   // the row's gradient at a missed step
        adam_elem(pp.x, z0, mm.x, vv.x, k);
        adam_elem(pp.y, z1, mm.y, vv.y, k);
        adam_elem(pp.z, z2, mm.z, vv.z, k);
        adam_elem(pp.w, z3, mm.w, vv.w, k);
        adam_elem(p1, z4, m1, v1, k);
      }
      if (A.with_grad) {
        const int s = A.upto0 + 1;
        k.step_size = A.step_size[s & (ADAM_WIN - 1)];
        k.inv_sqrt_bc2 = A.inv_sqrt_bc2[s & (ADAM_WIN - 1)];
        float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
        float g1 = 0.f;
        if (has4) { gg = *reinterpret_cast<const float4*>(gr + q); *reinterpret_cast<float4*>(gr + q) = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (has1) { g1 = gr[e]; gr[e] = 0.f; }
        adam_elem(pp.x, gg.x, mm.x, vv.x, k);
        adam_elem(pp.y, gg.y, mm.y, vv.y, k);
        adam_elem(pp.z, gg.z, mm.z, vv.z, k);
        adam_elem(pp.w, gg.w, mm.w, vv.w, k);
        adam_elem(p1, g1, m1, v1, k);
      }
      if (has4) { *reinterpret_cast<float4*>(pr + q) = pp; *reinterpret_cast<float4*>(mr + q) = mm; *reinterpret_cast<float4*>(vr + q) = vv; }
      if (has1) { pr[e] = p1; mr[e] = m1; vr[e] = v1; }
    }
    if (lane == 0) A.last[r] = A.upto0 + (A.with_grad ? 1 : 0);
  }
}

__global__ void __launch_bounds__(256) adam_mark_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t rows,
                                                         int32_t* __restrict__ mark, int32_t tag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[i];
    // (plain stores of one value: racing writers agree; a load first keeps the hot ids' lines clean)
    if (id >= 0 && id < rows && mark[id] != tag) mark[id] = tag;
  }
}

int adam_rows_mark(const int64_t* ids, int64_t n, int64_t rows, int32_t* mark, int64_t step, hipStream_t stream) {
  if (n == 0) return NRL_OK;
  NRL_REQUIRE(step >= 1 && step < (1LL << 31), "adam_rows_mark: step out of range");
  hipLaunchKernelGGL(adam_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ids, n, rows, mark, (int32_t)step);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

int adam_rows_advance(float* p, float* g, float* m, float* v, int64_t rows, int dim, int32_t* last, const int32_t* mark,
                      int32_t* status, int64_t stride, int64_t offset, int64_t upto0, int with_grad, double lr, double b1,
                      double b2, double eps, float grad_scale, hipStream_t stream, const int32_t* excl, int32_t excl_tag) {
  // with_grad == 2: "scan" -- no mark; every candidate row whose gradient row is non-zero is updated (dense all-reduce)
  const int scan = with_grad == 2 ? 1 : 0;
  NRL_REQUIRE(!scan || mark == nullptr, "adam_rows: the scan update takes no mark");
  NRL_REQUIRE(p && g && m && v && last && status && rows >= 0 && dim > 0 && dim % 4 == 0, "adam_rows: bad arguments");
  NRL_REQUIRE(stride >= 1 && offset >= 0 && upto0 >= 0 && upto0 + 1 < (1LL << 31), "adam_rows: bad slice / step");
  NRL_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (dim * sizeof(float)) % 16 == 0,
              "adam_rows: buffers must be 16-byte aligned");
  if (offset >= rows) return NRL_OK;
  AdamRowsArgs A;
  NRL_REQUIRE(excl == nullptr || (mark != nullptr && !with_grad), "adam_rows: an exclusion mark goes with a marked catch-up only");
  A.p = p; A.g = g; A.m = m; A.v = v; A.last = last; A.mark = mark; A.status = status; A.excl = excl; A.excl_tag = excl_tag;
  A.n_cand = (rows - offset + stride - 1) / stride; A.stride = stride; A.offset = offset;
  A.tag = (int32_t)(upto0 + 1);      // the step the marks were written for: catch-up and update both run during step upto0 + 1
  A.upto0 = (int32_t)upto0; A.with_grad = with_grad ? 1 : 0; A.D = dim; A.scan = scan;
  A.base.b1 = (float)b1; A.base.b2 = (float)b2; A.base.one_m_b1 = (float)(1.0 - b1); A.base.one_m_b2 = (float)(1.0 - b2);
  A.base.eps = (float)eps; A.base.grad_scale = grad_scale; A.base.step_size = 0.f; A.base.inv_sqrt_bc2 = 0.f;
  const int64_t hi = upto0 + 1;
  for (int64_t s = hi; s > hi - ADAM_WIN; --s) {
    float ss = 0.f, ib = 0.f;
    if (s >= 1) {
      const double bc1 = 1.0 - pow(b1, (double)s), bc2 = 1.0 - pow(b2, (double)s);     // (exactly adam_step's constants)
      ss = (float)(lr / bc1);
      ib = (float)(1.0 / sqrt(bc2));
    }
    A.step_size[s & (ADAM_WIN - 1)] = ss;
    A.inv_sqrt_bc2[s & (ADAM_WIN - 1)] = ib;
  }
  static const int cand = [] { const char* e = getenv("NRL_ADAM_ROWS_CAND"); return e ? atoi(e) : 32; }();   // (probe: 8 / 16 / 32 / 64)
  if (mark != nullptr && cand == 8) hipLaunchKernelGGL(adam_rows_kernel<8>, dim3((unsigned)((A.n_cand + 7) / 8)), dim3(256), 0, stream, A);
  else if (mark != nullptr && cand == 16) hipLaunchKernelGGL(adam_rows_kernel<16>, dim3((unsigned)((A.n_cand + 15) / 16)), dim3(256), 0, stream, A);
  else if (mark != nullptr && cand == 64) hipLaunchKernelGGL(adam_rows_kernel<64>, dim3((unsigned)((A.n_cand + 63) / 64)), dim3(256), 0, stream, A);
  else if (mark != nullptr || scan) hipLaunchKernelGGL(adam_rows_kernel<32>, dim3((unsigned)((A.n_cand + 31) / 32)), dim3(256), 0, stream, A);
  else hipLaunchKernelGGL(adam_rows_kernel<4>, dim3((unsigned)((A.n_cand + 3) / 4)), dim3(256), 0, stream, A);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// embedding gather (bit-exact copy of table rows) and the dropout-mask probe
// =============================================================================================
__global__ void gather_kernel(const float4* __restrict__ table, const int64_t* __restrict__ ids,
                              int64_t total4, int D4, float4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D4;
    out[i] = table[ids[row] * D4 + (i % D4)];
  }
}

int embedding_gather(const float* table, const int64_t* ids, int64_t n_ids, int D, float* out,
                     hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "embedding dim must be a multiple of 4");
  const int64_t total4 = n_ids * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream,
                     (const float4*)table, ids, total4, D / 4, (float4*)out);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// embedding_dense_backward without hot-row contention: rows are visited in id-sorted order
// (`order` = argsort of the flat id vector), each workgroup owns a run of consecutive sorted
// positions, accumulates rows of equal id in registers and flushes once per (id, workgroup) with an
// atomicAdd.  A token that fills 13 % of the batch ("the") costs ~170 atomics per address instead of
// ~11 000 serialised ones.  Rows of id 0 (padding_idx) are skipped; they sort first.
constexpr int EMB_SEG_ROWS = 64;
__global__ void __launch_bounds__(256)
    embedding_grad_sorted_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                 const int64_t* __restrict__ order, int64_t n_rows, int D,
                                 float* __restrict__ d_table, const int32_t* __restrict__ cidx) {
  // cidx != nullptr: dx holds ONLY the live rows (id != 0), compact in position order -- the row of token position p is
  // dx[cidx[p]] (live_compact below); the segments then cover the live part of the sorted order, [n_zero, n_rows), with
  // n_zero = order[n_rows]
  int64_t skip = 0;
  const bool compact = cidx != nullptr;
  if (compact) {
    skip = order[n_rows];
    if ((int64_t)blockIdx.x * EMB_SEG_ROWS >= n_rows - skip) return;
    order += skip;
    n_rows -= skip;
  }
  // The segment's (position, id) pairs are fetched by 64 lanes at once and parked in LDS; the row loads of eight
  // positions are then in flight together.  (One position at a time, each iteration was a chain of three dependent
  // global loads -- order -> id -> row: ~50 us per launch however small the batch.)  Rows are still ADDED in sorted order.
  __shared__ int64_t s_pos[EMB_SEG_ROWS], s_id[EMB_SEG_ROWS];
  const int64_t beg = (int64_t)blockIdx.x * EMB_SEG_ROWS;
  const int64_t end = beg + EMB_SEG_ROWS < n_rows ? beg + EMB_SEG_ROWS : n_rows;
  const int n = (int)(end - beg);
  const int tid = threadIdx.x;
  if (tid < EMB_SEG_ROWS) {
    const int64_t pos = tid < n ? order[beg + tid] : 0;
    s_pos[tid] = pos;
    s_id[tid] = tid < n ? ids[pos] : 0;                 // slots past the end behave like padding rows
  }
  __syncthreads();
  if (s_id[n - 1] == 0) return;  // sorted ascending: the whole segment is padding
  const bool d0 = tid < D, d1 = tid + 256 < D;
  float acc0 = 0.f, acc1 = 0.f;  // dims tid and tid + 256 (D <= 512)
  int64_t cur = -1;
  for (int j0 = 0; j0 < n; j0 += 8) {
    float r0[8], r1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < EMB_SEG_ROWS ? j0 + u : EMB_SEG_ROWS - 1;
      const bool live = j0 + u < n && s_id[j] != 0;
      const float* row = dx + (compact ? (int64_t)cidx[s_pos[j]] : s_pos[j]) * D;
      r0[u] = (live && d0) ? row[tid] : 0.f;
      r1[u] = (live && d1) ? row[tid + 256] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u < n) {
        const int64_t id = s_id[j0 + u];
        if (id != cur) {
          if (cur > 0) {
            if (d0) atomicAdd(d_table + cur * D + tid, acc0);
            if (d1) atomicAdd(d_table + cur * D + tid + 256, acc1);
          }
          cur = id;
          acc0 = acc1 = 0.f;
        }
        acc0 += r0[u];
        acc1 += r1[u];
      }
    }
  }
  if (cur > 0) {
    if (d0) atomicAdd(d_table + cur * D + tid, acc0);
    if (d1) atomicAdd(d_table + cur * D + tid + 256, acc1);
  }
}

int embedding_grad_sorted(const float* dx, const int64_t* ids, const int64_t* order, int64_t n_rows, int D,
                          float* d_table, hipStream_t stream, const int32_t* cidx) {
  if (n_rows == 0) return NRL_OK;
  NRL_REQUIRE(D <= 512, "embedding_grad_sorted: dim > 512 unsupported");
  hipLaunchKernelGGL(embedding_grad_sorted_kernel, dim3((unsigned)ceil_div(n_rows, EMB_SEG_ROWS)), dim3(256), 0,
                     stream, dx, ids, order, n_rows, D, d_table, cidx);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// The same reduction for ANY nn.Embedding (round 5: the word / position / token-type tables of a transformer body, text.py:89 --
// ATen's embedding_dense_backward there is a merge sort + three segment kernels, 2.6 ms of a config-4 step): dim <= 1024 (four
// accumulators per thread), `padding_idx` (< 0: none) names the one row that receives no gradient -- it may sit anywhere in the
// order, so it is skipped per position instead of by the "zeros sort first" shortcut of the kernel above.
__global__ void __launch_bounds__(256)
    embedding_grad_any_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids, const int64_t* __restrict__ order,
                              int64_t n_rows, int D, int64_t padding_idx, float* __restrict__ d_table) {
  __shared__ int64_t s_pos[EMB_SEG_ROWS], s_id[EMB_SEG_ROWS];
  const int64_t beg = (int64_t)blockIdx.x * EMB_SEG_ROWS;
  const int64_t end = beg + EMB_SEG_ROWS < n_rows ? beg + EMB_SEG_ROWS : n_rows;
  const int n = (int)(end - beg);
  const int tid = threadIdx.x;
  if (tid < EMB_SEG_ROWS) {
    const int64_t pos = tid < n ? order[beg + tid] : 0;
    s_pos[tid] = pos;
    s_id[tid] = tid < n ? ids[pos] : -1;
  }
  __syncthreads();
  bool dk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dk[q] = tid + 256 * q < D;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t cur = -1;
  auto flush = [&]() {
    if (cur >= 0 && cur != padding_idx) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (dk[q]) atomicAdd(d_table + cur * D + tid + 256 * q, acc[q]);
    }
  };
  for (int j0 = 0; j0 < n; j0 += 4) {
    float r[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u < EMB_SEG_ROWS ? j0 + u : EMB_SEG_ROWS - 1;
      const bool live = j0 + u < n && s_id[j] != padding_idx;
      const float* row = dx + s_pos[j] * D;
#pragma unroll
      for (int q = 0; q < 4; ++q) r[u][q] = (live && dk[q]) ? row[tid + 256 * q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u < n) {
        const int64_t id = s_id[j0 + u];
        if (id != cur) {
          flush();
          cur = id;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += r[u][q];
      }
    }
  }
  flush();
}

int embedding_grad_any(const float* dx, const int64_t* ids, const int64_t* order, int64_t n_rows, int D, int64_t padding_idx,
                       float* d_table, hipStream_t stream) {
  if (n_rows == 0) return NRL_OK;
  NRL_REQUIRE(D > 0 && D <= 1024, "embedding_grad: dim > 1024 unsupported");
  hipLaunchKernelGGL(embedding_grad_any_kernel, dim3((unsigned)ceil_div(n_rows, EMB_SEG_ROWS)), dim3(256), 0, stream, dx, ids, order,
                     n_rows, D, padding_idx, d_table);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// ---- compaction of the LIVE token positions (id != 0) in position order -------------------------------------------------
// list[c] = c-th live position, cidx[p] = number of live positions before p, counts[nb] = their total.  The last dgrad of a
// text encoder runs over list (rows of one news stay together: its fragment loads share the 16-row blocks of the planes);
// the table-gradient pass finds the dx row of a sorted position through cidx.
constexpr int LC_THREADS = 256, LC_ITEMS = 8, LC_BLOCK = LC_THREADS * LC_ITEMS;
__global__ void __launch_bounds__(LC_THREADS) live_count_kernel(const int64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ counts) {
  __shared__ int wsum[LC_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * LC_BLOCK + (int64_t)tid * LC_ITEMS;
  int c = 0;
#pragma unroll
  for (int q = 0; q < LC_ITEMS; ++q) c += (p0 + q < n && ids[p0 + q] != 0) ? 1 : 0;
  c = (int)wave_sum((float)c);                   // (<= 512 per wave: exact in fp32)
  if ((tid & 63) == 0) wsum[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void __launch_bounds__(LC_THREADS) live_scatter_kernel(const int64_t* __restrict__ ids, int64_t n, int nb,
                                                                  int32_t* __restrict__ counts, int32_t* __restrict__ list,
                                                                  int32_t* __restrict__ cidx) {
  __shared__ int red[LC_THREADS];
  __shared__ int wbase[LC_THREADS / 64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // live positions in the blocks before this one (and, block 0, the grand total for the consumers)
  int before = 0, total = 0;
  for (int j = tid; j < nb; j += LC_THREADS) {
    const int c = counts[j];
    total += c;
    if (j < (int)blockIdx.x) before += c;
  }
  red[tid] = before;
  __syncthreads();
  for (int off = LC_THREADS / 2; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  const int base = red[0];
  __syncthreads();
  if (blockIdx.x == 0) {
    red[tid] = total;
    __syncthreads();
    for (int off = LC_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) counts[nb] = red[0];
    __syncthreads();
  }
  // this thread's 8 consecutive positions; exclusive scan of the per-thread counts across the workgroup
  const int64_t p0 = (int64_t)blockIdx.x * LC_BLOCK + (int64_t)tid * LC_ITEMS;
  bool live[LC_ITEMS];
  int c = 0;
#pragma unroll
  for (int q = 0; q < LC_ITEMS; ++q) {
    live[q] = p0 + q < n && ids[p0 + q] != 0;
    c += live[q] ? 1 : 0;
  }
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wbase[wave + 1] = incl;
  __syncthreads();
  if (tid == 0) {
    wbase[0] = 0;
    for (int w = 1; w <= LC_THREADS / 64; ++w) wbase[w] += wbase[w - 1];
  }
  __syncthreads();
  int pos = base + wbase[wave] + incl - c;
#pragma unroll
  for (int q = 0; q < LC_ITEMS; ++q) {
    if (p0 + q < n) {
      cidx[p0 + q] = pos;
      if (live[q]) list[pos++] = (int32_t)(p0 + q);
    }
  }
}

size_t live_compact_ints(int64_t n) { return (size_t)(2 * n + ceil_div(n, LC_BLOCK) + 8); }
int live_compact(const int64_t* ids, int64_t n, int32_t* scratch, const int32_t** list, const int32_t** cidx,
                 const int32_t** n_live, hipStream_t stream) {
  NRL_REQUIRE(n >= 0 && n < (1LL << 31), "live_compact: n < 2^31");
  const int nb = (int)ceil_div(n > 0 ? n : 1, LC_BLOCK);
  int32_t* l = scratch;
  int32_t* c = scratch + n;
  int32_t* counts = scratch + 2 * n;
  *list = l; *cidx = c; *n_live = counts + nb;
  hipLaunchKernelGGL(live_count_kernel, dim3((unsigned)nb), dim3(LC_THREADS), 0, stream, ids, n, counts);
  NRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(live_scatter_kernel, dim3((unsigned)nb), dim3(LC_THREADS), 0, stream, ids, n, nb, counts, l, c);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__global__ void dropout_mask_kernel(uint8_t* keep, int64_t n, Dropout d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    keep[i] = d.mult((uint32_t)i) != 0.0f ? 1 : 0;
}

int dropout_mask(uint8_t* keep, int64_t n, Dropout d, hipStream_t stream) {
  if (n == 0) return NRL_OK;
  NRL_REQUIRE(n < (1LL << 32), "dropout index space is 32-bit");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, keep, n, d);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =============================================================================================
// LSTUR path: row-masked embedding lookups, axis swap, GRU cell
// =============================================================================================
__global__ void embedding_rows_fwd_kernel(const float4* __restrict__ table, const int64_t* __restrict__ ids,
                                          int64_t total4, int D4, Dropout drop, int row_mode,
                                          float4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D4;
    const int c4 = (int)(i % D4);
    float4 v = table[ids[row] * D4 + c4];
    if (drop.thresh != 0u) {
      if (row_mode) {
        const float m = drop.mult((uint32_t)row);
        v.x *= m; v.y *= m; v.z *= m; v.w *= m;
      } else {
        const uint32_t idx = (uint32_t)row * (uint32_t)(4 * D4) + 4u * (uint32_t)c4;
        v.x *= drop.mult(idx); v.y *= drop.mult(idx + 1); v.z *= drop.mult(idx + 2); v.w *= drop.mult(idx + 3);
      }
    }
    out[i] = v;
  }
}

int embedding_rows_fwd(const float* table, const int64_t* ids, int64_t n_ids, int D, Dropout drop, int row_mode,
                       float* out, hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "embedding dim must be a multiple of 4");
  NRL_REQUIRE(n_ids * D < (1LL << 32), "dropout index space is 32-bit");
  const int64_t total4 = n_ids * (D / 4);
  if (total4 == 0) return NRL_OK;
  hipLaunchKernelGGL(embedding_rows_fwd_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, stream,
                     (const float4*)table, ids, total4, D / 4, drop, row_mode, (float4*)out);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__global__ void embedding_rows_bwd_kernel(const float* __restrict__ d_out, const int64_t* __restrict__ ids,
                                          int64_t total, int D, Dropout drop, float* __restrict__ d_table) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D;
    const int64_t id = ids[row];
    if (id == 0) continue;  // padding_idx
    const float m = drop.thresh != 0u ? drop.mult((uint32_t)row) : 1.0f;
    if (m != 0.0f) atomicAdd(d_table + id * D + (i % D), d_out[i] * m);
  }
}

int embedding_rows_bwd(const float* d_out, const int64_t* ids, int64_t n_ids, int D, Dropout drop, float* d_table,
                       hipStream_t stream) {
  const int64_t total = n_ids * D;
  if (total == 0) return NRL_OK;
  hipLaunchKernelGGL(embedding_rows_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, d_out, ids,
                     total, D, drop, d_table);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__global__ void transpose01_kernel(const float4* __restrict__ src, int64_t A, int64_t Bd, int D4,
                                   float4* __restrict__ dst) {
  const int64_t total = A * Bd * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D4;  // destination row = b * A + a
    const int64_t b = row / A, a = row % A;
    dst[i] = src[(a * Bd + b) * D4 + (i % D4)];
  }
}

int transpose01(const float* src, int64_t A, int64_t Bd, int D, float* dst, hipStream_t stream) {
  NRL_REQUIRE(D % 4 == 0, "transpose01: dim must be a multiple of 4");
  const int64_t total = A * Bd * (D / 4);
  if (total == 0) return NRL_OK;
  hipLaunchKernelGGL(transpose01_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const float4*)src, A,
                     Bd, D / 4, (float4*)dst);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void gru_gate_fwd_kernel(const float* gi, float* __restrict__ gh, const float* __restrict__ b_hh,
                                    const float* __restrict__ h_prev, const int64_t* __restrict__ len, int t,
                                    int64_t total, int Hd, float* gates, float* __restrict__ ghn,
                                    float* __restrict__ h_new) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / Hd;
    const int j = (int)(i % Hd);
    const int64_t o = b * 3 * Hd + j;
    const float hn = gh[o + 2 * Hd] + b_hh[j + 2 * Hd];
    const float r = sigmoidf_(gi[o] + gh[o] + b_hh[j]);
    const float z = sigmoidf_(gi[o + Hd] + gh[o + Hd] + b_hh[j + Hd]);
    const float n = tanhf(gi[o + 2 * Hd] + r * hn);
    gh[o] = 0.f;
    gh[o + Hd] = 0.f;
    gh[o + 2 * Hd] = 0.f;
    const float hp = h_prev[i];
    h_new[i] = (int64_t)t < len[b] ? (1.0f - z) * n + z * hp : hp;
    if (ghn != nullptr) {
      gates[o] = r;
      gates[o + Hd] = z;
      gates[o + 2 * Hd] = n;
      ghn[i] = hn;
    }
  }
}

int gru_gate_fwd(const float* gi, float* gh, const float* b_hh, const float* h_prev, const int64_t* len, int t,
                 int64_t B, int Hd, float* gates, float* ghn, float* h_new, hipStream_t stream) {
  const int64_t total = B * Hd;
  if (total == 0) return NRL_OK;
  hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, gi, gh, b_hh, h_prev, len,
                     t, total, Hd, gates, ghn, h_new);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

__global__ void gru_gate_bwd_kernel(const float* gates, const float* __restrict__ ghn,
                                    const float* __restrict__ h_prev, const int64_t* __restrict__ len, int t,
                                    int64_t total, int Hd, float* __restrict__ dh, float* dgi,
                                    float* __restrict__ dgh) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / Hd;
    const int j = (int)(i % Hd);
    const int64_t o = b * 3 * Hd + j;
    float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dn_r = 0.f;
    if ((int64_t)t < len[b]) {
      const float r = gates[o], z = gates[o + Hd], n = gates[o + 2 * Hd];
      const float d = dh[i];
      dn_pre = d * (1.0f - z) * (1.0f - n * n);
      dz_pre = d * (h_prev[i] - n) * z * (1.0f - z);
      dr_pre = dn_pre * ghn[i] * r * (1.0f - r);
      dn_r = dn_pre * r;
      dh[i] = d * z;
    }
    dgi[o] = dr_pre;
    dgi[o + Hd] = dz_pre;
    dgi[o + 2 * Hd] = dn_pre;
    dgh[o] = dr_pre;
    dgh[o + Hd] = dz_pre;
    dgh[o + 2 * Hd] = dn_r;
  }
}

int gru_gate_bwd(const float* gates, const float* ghn, const float* h_prev, const int64_t* len, int t, int64_t B,
                 int Hd, float* dh, float* dgi, float* dgh, hipStream_t stream) {
  const int64_t total = B * Hd;
  if (total == 0) return NRL_OK;
  hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, gates, ghn, h_prev, len,
                     t, total, Hd, dh, dgi, dgh);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
