// One GRU time step as ONE kernel (bf16x3 engine): gh = h_{t-1} W_hh^T on the matrix cores and the gate
// arithmetic in the epilogue, replacing a split-K GEMM launch (+ atomics into a gh buffer) and a gate launch
// per step.  The recurrence is a chain of B x 3Hd x Hd products with B ~ 128: each is latency-, not
// throughput-bound (12.5 + 4.6 us per step for the two launches at Hd = 700, ~4.5 us being the floor of ANY launch
// in a dependent chain), so the kernel is built to be short rather than dense:
//   * a workgroup owns 16 TR sequences x 16 hidden units (all three gates of those units: 48 columns of W_hh^T);
//     its 4 waves split K among themselves (k-tiles w, w + 4, ...; at most GRU_FUSED_MAXK each) and load their
//     operand fragments straight into registers in MFMA fragment layout (no LDS staging), split A to (hi, lo)
//     bf16 and issue 3 products x 3 gates per k-tile;
//   * the four partial accumulators meet in LDS and wave r finishes row-quad r: gates, h_t, and the saved
//     r | z | n and W_hn h + b_hn the backward needs (same buffers and meanings as gru_gate_fwd).
// What it does NOT fix: the step still costs ~1.6 us per k-tile of a wave's share (14.5 us at Hd = 700, 5 us at
// Hd = 52) -- W_hh is re-fetched every step (L2 does not survive the kernel boundary on this multi-XCD part) and a
// CU's outstanding-miss capacity, not bandwidth, paces the fetch.  The persistent form (PERSIST = 1 below: W_hh
// fragments resident in registers, one cooperative launch, a grid-wide barrier per step) is built and correct but
// slower -- see gru_persistent_fwd.
// Grid = ceil(B / (16 TR)) x ceil(Hd / 16) workgroups, flattened and XCD-ordered.  hidden_dim <= 768; beyond that,
// and under the exact-fp32 engine, nrl_gru_fwd keeps the two-launch path.
#pragma once
#include <hip/hip_cooperative_groups.h>

#include "nrl_gemm_bf16x3.h"

namespace nrl {

constexpr int GRU_FUSED_MAXK = 6;

// Grid-wide barrier of a co-resident (cooperative) launch on one device counter.  The hidden state is exchanged through
// agent-coherent (sc1, write-through) stores, so the arriving side only has to wait for ITS stores to complete
// (vmcnt(0)) -- no `buffer_wbl2` of the XCD's whole L2, which is what made a fence-based barrier (and
// cooperative_groups::grid_group::sync()) cost ~30 us per step; the leaving side invalidates its L2 (`buffer_inv sc1`,
// cheap) so that the h_t lines written by the other XCDs are fetched from memory.
__device__ __forceinline__ void gru_grid_barrier(unsigned* ctr, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's sc1 stores of h_t have reached memory
  __syncthreads();                                         // ... and every wave's of this workgroup
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// PERSIST = 1: the whole recurrence in ONE cooperative launch -- the workgroup's 48 columns of W_hh^T stay in registers
// (144 VGPRs of (hi, lo) fragments, loaded once), every time step reloads only its h_{t-1} rows and ends in a grid-wide
// barrier; `t` is then the number of steps and h_prev / gi_gates / ghn / h_new point at step 0 (step strides B*Hd /
// 3*B*Hd floats, h_new = h_prev + B*Hd as nrl_gru_fwd lays the saved states out).
template <int TR, int PERSIST = 0>   // 16-row tiles per workgroup: W_hh is streamed once per workgroup ROW
__global__ void __launch_bounds__(256)
    gru_step_fused_kernel(const float* h_prev, const uint16_t* __restrict__ w_hi, const int64_t ld,
                          float* gi_gates, const float* __restrict__ b_hh, const int64_t* __restrict__ len,
                          int t, const int B, const int Hd, const int save, float* ghn,
                          float* h_new, unsigned* barrier_ctr) {
  __shared__ float red[4][TR][3][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  // XCD-aware tile order (workgroup id % 8 = XCD): each XCD walks a CONTIGUOUS eighth of the (unit block, row
  // tile) list, so it only ever touches its own eighth of W_hh (0.74 of 5.9 MB at Hd = 700) -- resident in that
  // XCD's 4 MB L2 from the second time step on, the grid being the same for every step
  int tile;
  {
    const int bid = blockIdx.x, total = gridDim.x;
    const int xcd = bid % 8, local = bid / 8;
    const int q = total / 8, rem = total % 8;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  }
  const int row_tiles = (B + 16 * TR - 1) / (16 * TR);
  const int m0 = (tile % row_tiles) * 16 * TR, u0 = (tile / row_tiles) * 16;
  const int nk = (Hd + 31) >> 5;
  const int unit = u0 + l15 < Hd ? u0 + l15 : Hd - 1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  uint4 bh[GRU_FUSED_MAXK][3], bl[GRU_FUSED_MAXK][3];
#pragma unroll
  for (int s = 0; s < GRU_FUSED_MAXK; ++s) {
    const int kt = wave + 4 * s;
    if (kt < nk) {
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) {
        // interleaved planes: the k-tile of a row is [32 hi | 32 lo]; this lane's 8 k's are 16 bytes of each
        const uint16_t* p = w_hi + ((int64_t)gate * Hd + unit) * ld + 64 * kt + 8 * g;
        bh[s][gate] = *reinterpret_cast<const uint4*>(p);
        bl[s][gate] = *reinterpret_cast<const uint4*>(p + 32);
      }
    }
  }
  const int steps = PERSIST ? t : 1;
  for (int step = 0; step < steps; ++step) {
  if constexpr (PERSIST) t = step;
  float4 a0[GRU_FUSED_MAXK][TR], a1[GRU_FUSED_MAXK][TR];
#pragma unroll
  for (int s = 0; s < GRU_FUSED_MAXK; ++s) {
    const int kt = wave + 4 * s;
    if (kt < nk) {
      const int k = kt * 32 + 8 * g;
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int row = m0 + 16 * i + l15 < B ? m0 + 16 * i + l15 : B - 1;
        const float* arow = h_prev + (int64_t)row * Hd;
        a0[s][i] = k + 3 < Hd ? *reinterpret_cast<const float4*>(arow + k) : zero4;       // Hd % 4 == 0
        a1[s][i] = k + 7 < Hd ? *reinterpret_cast<const float4*>(arow + k + 4) : zero4;
      }
    }
  }
  f32x4 acc[TR][3];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) acc[i][gate] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < GRU_FUSED_MAXK; ++s) {
    if (wave + 4 * s < nk) {
      bf16x8 ah[TR], al[TR];
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        uint32_t h[4], l[4];
        split_pair(a0[s][i].x, a0[s][i].y, h[0], l[0]);
        split_pair(a0[s][i].z, a0[s][i].w, h[1], l[1]);
        split_pair(a1[s][i].x, a1[s][i].y, h[2], l[2]);
        split_pair(a1[s][i].z, a1[s][i].w, h[3], l[3]);
        ah[i] = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
        al[i] = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
      }
      // per accumulator: lo-terms first, hi * hi last (the order of the tiled kernels)
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
            acc[i][gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                pass == 1 ? al[i] : ah[i], __builtin_bit_cast(bf16x8, pass == 0 ? bl[s][gate] : bh[s][gate]),
                acc[i][gate], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i][gate][r][lane] = acc[i][gate][r];
  __syncthreads();
  // accumulator layout: column (hidden unit) = lane & 15, row (sequence) = 4 * (lane >> 4) + r; wave r finishes r
  const int r = wave;
  const int u = u0 + l15;
#pragma unroll
  for (int i = 0; i < TR; ++i) {
    float gh[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
      gh[gate] = (red[0][i][gate][r][lane] + red[1][i][gate][r][lane]) +
                 (red[2][i][gate][r][lane] + red[3][i][gate][r][lane]);
    const int m = m0 + 16 * i + 4 * g + r;
    if (m < B && u < Hd) {
      const int64_t o = (int64_t)m * 3 * Hd + u, idx = (int64_t)m * Hd + u;
      const float hn = gh[2] + b_hh[u + 2 * Hd];
      const float rg = 1.0f / (1.0f + expf(-(gi_gates[o] + gh[0] + b_hh[u])));
      const float z = 1.0f / (1.0f + expf(-(gi_gates[o + Hd] + gh[1] + b_hh[u + Hd])));
      const float n = tanhf(gi_gates[o + 2 * Hd] + rg * hn);
      const float hp = h_prev[idx];
      const float hv = (int64_t)t < len[m] ? (1.0f - z) * n + z * hp : hp;
      if constexpr (PERSIST) __hip_atomic_store(h_new + idx, hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
      else h_new[idx] = hv;
      if (save) {
        gi_gates[o] = rg;
        gi_gates[o + Hd] = z;
        gi_gates[o + 2 * Hd] = n;
        ghn[idx] = hn;
      }
    }
  }
  if constexpr (PERSIST) {
    h_prev += (int64_t)B * Hd;
    h_new += (int64_t)B * Hd;
    ghn += (int64_t)B * Hd;
    gi_gates += (int64_t)B * 3 * Hd;
    gru_grid_barrier(barrier_ctr, (unsigned)(step + 1) * gridDim.x);   // every workgroup's h_t visible before anyone reads it
  }
  }
}

inline bool gru_step_fused_ok(int Hd) {
  static const bool on = [] { const char* e = getenv("NRL_GRU_FUSED"); return !(e != nullptr && e[0] == '0'); }();
  return on && Hd % 4 == 0 && Hd <= 4 * GRU_FUSED_MAXK * 32;
}

inline int gru_step_fused(const float* h_prev, const uint16_t* w_hi, int64_t ld, float* gi_gates, const float* b_hh,
                          const int64_t* len, int t, int64_t B, int Hd, bool save, float* ghn, float* h_new,
                          hipStream_t stream) {
  if (B == 0) return NRL_OK;
  NRL_REQUIRE(ceil_div(B, 16) * ceil_div(Hd, 16) < (1LL << 31), "gru_step_fused: grid too large");
  // wide hidden states: 32 sequences per workgroup, so that W_hh (5.9 MB of planes at Hd = 700) is fetched B / 32
  // times per step instead of B / 16; narrow ones: 16, for more workgroups.  Measured (tools/gru_time.py, forward
  // of the whole GRU, two-launch path -> fused): Hd 700 / B 128 / T 50: 1.01 -> 0.85 ms (TR = 2; 0.94 with 1, 1.31
  // with 4); Hd 400 / B 128 / T 20: 0.30 -> 0.22 ms; Hd 52 / B 768 / T 50: 0.51 -> 0.23 ms
  static const int tr_env = [] { const char* e = getenv("NRL_GRU_FUSED_TR"); return e ? atoi(e) : 0; }();
  const int tr = tr_env ? tr_env : (Hd >= 512 ? 2 : 1);
  const dim3 block(256);
  if (tr >= 4) {
    hipLaunchKernelGGL(gru_step_fused_kernel<4>, dim3((unsigned)(ceil_div(B, 64) * ceil_div(Hd, 16))), block, 0,
                       stream, h_prev, w_hi, ld, gi_gates, b_hh, len, t, (int)B, Hd, save ? 1 : 0, ghn, h_new, (unsigned*)nullptr);
  } else if (tr == 2) {
    hipLaunchKernelGGL(gru_step_fused_kernel<2>, dim3((unsigned)(ceil_div(B, 32) * ceil_div(Hd, 16))), block, 0,
                       stream, h_prev, w_hi, ld, gi_gates, b_hh, len, t, (int)B, Hd, save ? 1 : 0, ghn, h_new, (unsigned*)nullptr);
  } else {
    hipLaunchKernelGGL(gru_step_fused_kernel<1>, dim3((unsigned)(ceil_div(B, 16) * ceil_div(Hd, 16))), block, 0,
                       stream, h_prev, w_hi, ld, gi_gates, b_hh, len, t, (int)B, Hd, save ? 1 : 0, ghn, h_new, (unsigned*)nullptr);
  }
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// (The one-launch BPTT step of round 4 -- gru_step_bwd_fused_kernel: 28.6 us per step against 15.5 + 5.1 us for the split-K GEMM + gate
//  kernel -- lost its A/B and lives in tools/experimental/nrl_gru_bwd_fused.h since round 5.)

// The whole forward recurrence in one cooperative launch (see PERSIST above).  Returns NRL_E_INVALID-free "not
// applicable" as *done = false when the grid cannot be co-resident (the caller then launches step by step).
inline int gru_persistent_fwd(const float* hs0, const uint16_t* w_hi, int64_t ld, float* g0, const float* b_hh,
                              const int64_t* len, int64_t T, int64_t B, int Hd, float* ghn0, unsigned* barrier_ctr,
                              hipStream_t stream, bool* done) {
  *done = false;
  // OFF by default (NRL_GRU_PERSISTENT=1 enables): measured SLOWER than one launch per step -- LSTUR step 10.15 vs 9.55 ms at
  // B = 128, Hd = 700, T = 50, i.e. ~27 us per time step against 15.4 us.  History: with a fence-based barrier (and with
  // cooperative_groups' grid.sync()) a step cost ~36 us -- the device-scope release writes back the XCD's whole L2; exchanging
  // h_t through sc1 stores + an acquire-only barrier (gru_grid_barrier) brought it to ~27 us.  What is left is the round trip
  // of 176 workgroups on 8 XCDs through one memory-side counter plus a cold L2 after the invalidate, against the ~5 us a
  // kernel boundary costs in a back-to-back stream.
  static const bool on = [] { const char* e = getenv("NRL_GRU_PERSISTENT"); return e != nullptr && e[0] == '1'; }();
  if (!on || B == 0 || T <= 1) return NRL_OK;
  const int tr = Hd >= 512 ? 2 : 1;
  const int64_t blocks = ceil_div(B, 16 * tr) * ceil_div(Hd, 16);
  const void* fn = tr == 2 ? reinterpret_cast<const void*>(&gru_step_fused_kernel<2, 1>)
                           : reinterpret_cast<const void*>(&gru_step_fused_kernel<1, 1>);
  static int cus = 0, per_cu[3] = {0, -1, -1};
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    NRL_HIP(hipGetDevice(&dev));
    NRL_HIP(hipGetDeviceProperties(&prop, dev));
    if (!prop.cooperativeLaunch) { cus = -1; return NRL_OK; }
    cus = prop.multiProcessorCount;
  }
  if (cus < 0) return NRL_OK;
  if (per_cu[tr] < 0) {
    int n = 0;
    NRL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, 0));
    per_cu[tr] = n;
  }
  if (blocks > (int64_t)cus * per_cu[tr]) return NRL_OK;   // would not be co-resident: a grid barrier would hang
  const float* h_prev = hs0;
  float* h_new = const_cast<float*>(hs0) + B * Hd;
  int steps = (int)T, Bi = (int)B, save = 1;
  NRL_HIP(hipMemsetAsync(barrier_ctr, 0, sizeof(unsigned), stream));
  void* args[] = {(void*)&h_prev, (void*)&w_hi, (void*)&ld, (void*)&g0, (void*)&b_hh, (void*)&len, (void*)&steps,
                  (void*)&Bi, (void*)&Hd, (void*)&save, (void*)&ghn0, (void*)&h_new, (void*)&barrier_ctr};
  NRL_HIP(hipLaunchCooperativeKernel(fn, dim3((unsigned)blocks), dim3(256), args, 0, stream));
  *done = true;
  return NRL_OK;
}

}  // namespace nrl
