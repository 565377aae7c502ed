// NOT PRODUCT CODE (moved out of newsreclib_amd/csrc/nrl_gru_fused.h in round 5): one BPTT step of the GRU as ONE kernel.
// Measured SLOWER than the two launches it replaced (28.6 us per step against 15.5 + 5.1 us, profiles/r04_ab.txt): the whole reduction
// of dgh_t W_hh inside one workgroup serialises what split-K spreads over the chip.  It was wired into nrl_gru_bwd as
//   gru_gate_bwd(last step); for t = T - 1 .. 0: gru_step_bwd_fused(dgh_t, W_hh^T hi plane, ld, dh, gates_{t-1}, ghn_{t-1}, h_{t-2}, dgh_{t-1}, ...)
// under NRL_GRU_BWD_FUSED=1.  Include after nrl_gru_fused.h.
#pragma once
#include "nrl_gru_fused.h"

namespace nrl {

// ---- backward: one BPTT step as ONE kernel (round 4) -----------------------------------------------------------------------
//   dh_{t-1} = dhz_t + dgh_t W_hh          dhz_t = dh_t * z_t (the direct term the gate backward of step t left), dgh_t (B, 3Hd)
//   gate backward of step t - 1 at dh_{t-1}  -> dgi_{t-1} (over the saved gates, in place), dgh_{t-1}, dhz_{t-1}
// The two-launch form (gate-adjoint kernel, then a split-K GEMM with atomic accumulation into dh) cost 5.0 + 15.6 us per step in
// a dependent chain of 100 launches.  Here a workgroup owns 16 TR sequences x 16 hidden units and the WHOLE reduction (K = 3 Hd
// = 66 k-tiles at Hd = 700, dealt to its 16 waves: <= GRU_BWD_MAXK each), so dh_{t-1} of its tile is final inside the kernel and
// the gate backward of step t - 1 -- elementwise in (sequence, unit) -- runs in the epilogue: 51 launches instead of 100, no atomics.
// B operand: the k-contiguous planes of W_hh^T (split_weight's transposed image), row = hidden unit.
constexpr int GRU_BWD_KTILES = 80;     // k-tiles a workgroup covers at most: GRU_BWD_WAVES * GRU_BWD_MAXK for every shape below

template <int TR, int GRU_BWD_WAVES>
__global__ void __launch_bounds__(GRU_BWD_WAVES * 64)
    gru_step_bwd_fused_kernel(const float* __restrict__ dgh_t, const uint16_t* __restrict__ wt_hi, const int64_t ldt,
                              float* dhz, float* gates_prev, const float* __restrict__ ghn_prev, const float* __restrict__ hprev_prev,
                              float* __restrict__ dgh_prev, const int64_t* __restrict__ len, const int t_prev, const int B,
                              const int Hd) {
  constexpr int GRU_BWD_MAXK = GRU_BWD_KTILES / GRU_BWD_WAVES;
  __shared__ float red[GRU_BWD_WAVES][TR][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  int tile;
  {
    const int bid = blockIdx.x, total = gridDim.x;
    const int xcd = bid % 8, local = bid / 8;
    const int q = total / 8, rem = total % 8;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int row_tiles = (B + 16 * TR - 1) / (16 * TR);
  const int m0 = (tile % row_tiles) * 16 * TR, u0 = (tile / row_tiles) * 16;
  const int K = 3 * Hd;
  const int nk = (K + 31) >> 5;
  const int unit = u0 + l15 < Hd ? u0 + l15 : Hd - 1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  uint4 bh[GRU_BWD_MAXK], bl[GRU_BWD_MAXK];
  float4 a0[GRU_BWD_MAXK][TR], a1[GRU_BWD_MAXK][TR];
#pragma unroll
  for (int s = 0; s < GRU_BWD_MAXK; ++s) {
    const int kt = wave + GRU_BWD_WAVES * s;
    if (kt < nk) {
      const uint16_t* p = wt_hi + (int64_t)unit * ldt + 64 * kt + 8 * g;     // [32 hi | 32 lo] per k-tile of a row
      bh[s] = *reinterpret_cast<const uint4*>(p);
      bl[s] = *reinterpret_cast<const uint4*>(p + 32);
      const int k = kt * 32 + 8 * g;
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int row = m0 + 16 * i + l15 < B ? m0 + 16 * i + l15 : B - 1;
        const float* arow = dgh_t + (int64_t)row * K;
        a0[s][i] = k + 3 < K ? *reinterpret_cast<const float4*>(arow + k) : zero4;       // K % 4 == 0
        a1[s][i] = k + 7 < K ? *reinterpret_cast<const float4*>(arow + k + 4) : zero4;
      }
    }
  }
  f32x4 acc[TR];
#pragma unroll
  for (int i = 0; i < TR; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < GRU_BWD_MAXK; ++s) {
    if (wave + GRU_BWD_WAVES * s < nk) {
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        uint32_t h[4], l[4];
        split_pair(a0[s][i].x, a0[s][i].y, h[0], l[0]);
        split_pair(a0[s][i].z, a0[s][i].w, h[1], l[1]);
        split_pair(a1[s][i].x, a1[s][i].y, h[2], l[2]);
        split_pair(a1[s][i].z, a1[s][i].w, h[3], l[3]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
        const bf16x8 al = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bl[s]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, bh[s]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, bh[s]), acc[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][i][r][lane] = acc[i][r];
  __syncthreads();
  // thread (i, r, lane) of the first TR * 4 waves finishes sequence m0 + 16 i + 4 (lane >> 4) + r, unit u0 + (lane & 15)
  for (int o_ = tid; o_ < TR * 256; o_ += GRU_BWD_WAVES * 64) {
    const int i = o_ >> 8, r = (o_ >> 6) & 3;      // (o_ & 63 == lane: the stride is a multiple of 64)
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < GRU_BWD_WAVES; ++w) sum += red[w][i][r][lane];
    const int m = m0 + 16 * i + 4 * g + r, u = u0 + l15;
    if (m < B && u < Hd) {
      const int64_t idx = (int64_t)m * Hd + u;
      const float d = dhz[idx] + sum;                       // dh_{t-1}
      if (t_prev < 0) {
        dhz[idx] = d;                                       // (t = 0: this is the gradient of the initial state)
      } else {
        const int64_t o = (int64_t)m * K + u;
        float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dn_r = 0.f, keep = d;
        if ((int64_t)t_prev < len[m]) {
          const float rg = gates_prev[o], z = gates_prev[o + Hd], n = gates_prev[o + 2 * Hd];
          dn_pre = d * (1.0f - z) * (1.0f - n * n);
          dz_pre = d * (hprev_prev[idx] - n) * z * (1.0f - z);
          dr_pre = dn_pre * ghn_prev[idx] * rg * (1.0f - rg);
          dn_r = dn_pre * rg;
          keep = d * z;
        }
        dhz[idx] = keep;
        gates_prev[o] = dr_pre;
        gates_prev[o + Hd] = dz_pre;
        gates_prev[o + 2 * Hd] = dn_pre;
        dgh_prev[o] = dr_pre;
        dgh_prev[o + Hd] = dz_pre;
        dgh_prev[o + 2 * Hd] = dn_r;
      }
    }
  }
}

inline bool gru_step_bwd_fused_ok(int Hd) {
  // OFF by default (NRL_GRU_BWD_FUSED=1 enables): measured SLOWER -- 28.6 us per step against 15.5 + 5.1 us for the split-K product
  // + gate launch (LSTUR step 9.31 vs 8.88 ms; GRU forward + backward 2.79-3.03 ms over 4 / 8 / 16 waves and 16 / 32 rows per
  // workgroup against 2.40).  With the whole reduction in one workgroup a CU has to pull 400 KB of operands through its own
  // outstanding-miss capacity (the forward kernel's ~1.6 us per k-tile, times 66 k-tiles); split-K spreads the same lines over
  // 11x more CUs.  What would win is split-K with a last-arriver epilogue, whose cross-workgroup release is exactly what made
  // the persistent forward slow on this multi-XCD part (gru_grid_barrier above).
  static const bool on = [] { const char* e = getenv("NRL_GRU_BWD_FUSED"); return e != nullptr && e[0] == '1'; }();
  return on && Hd % 4 == 0 && (3 * Hd + 31) / 32 <= GRU_BWD_KTILES;
}

// one BPTT step: dh_{t-1} into `dhz` (in place over dh_t * z_t), and -- t_prev >= 0 -- the gate backward of step t_prev = t - 1
inline int gru_step_bwd_fused(const float* dgh_t, const uint16_t* wt_hi, int64_t ldt, float* dhz, float* gates_prev,
                              const float* ghn_prev, const float* hprev_prev, float* dgh_prev, const int64_t* len, int t_prev,
                              int64_t B, int Hd, hipStream_t stream) {
  if (B == 0) return NRL_OK;
  static const int tr_env = [] { const char* e = getenv("NRL_GRU_BWD_TR"); return e ? atoi(e) : 0; }();
  static const int wv_env = [] { const char* e = getenv("NRL_GRU_BWD_WAVES"); return e ? atoi(e) : 16; }();
  const int tr = tr_env ? tr_env : (Hd >= 512 ? 2 : 1);
#define NRL_GRU_BWD_LAUNCH(TRV, WV)                                                                                              \
  hipLaunchKernelGGL((gru_step_bwd_fused_kernel<TRV, WV>), dim3((unsigned)(ceil_div(B, 16 * TRV) * ceil_div(Hd, 16))), dim3(WV * 64), \
                     0, stream, dgh_t, wt_hi, ldt, dhz, gates_prev, ghn_prev, hprev_prev, dgh_prev, len, t_prev, (int)B, Hd)
  if (tr == 2 && wv_env == 16) NRL_GRU_BWD_LAUNCH(2, 16);
  else if (tr == 2 && wv_env == 8) NRL_GRU_BWD_LAUNCH(2, 8);
  else if (tr == 2) NRL_GRU_BWD_LAUNCH(2, 4);
  else if (wv_env == 16) NRL_GRU_BWD_LAUNCH(1, 16);
  else if (wv_env == 8) NRL_GRU_BWD_LAUNCH(1, 8);
  else NRL_GRU_BWD_LAUNCH(1, 4);
#undef NRL_GRU_BWD_LAUNCH
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
