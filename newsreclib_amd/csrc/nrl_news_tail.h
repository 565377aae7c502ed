// Fused back half of the NRMS news encoder (gfx950):  attention output `o` (planes) -> news vector
//
//   y = dropout(o W_o^T + b_o)                    text.py:229-230   (out-projection of nn.MultiheadAttention + nn.Dropout)
//   a = tanh(y W_a^T + b_a) . q_a                 attention.py:34-36
//   w = softmax_L(a);  out = sum_l w_l y_l        attention.py:37-40
//
// One wavefront owns ONE news (L <= 32 tokens = 2 MFMA column blocks), eight news per workgroup.  Everything between the
// `o` planes the fused front half wrote (nrl_news_fused.h) and the (n_news, D) output stays on the chip; the row-panel
// pipeline it replaces (out-projection GEMM -> y fp32 + y planes -> additive-attention GEMM -> t -> pool_fwd) wrote y
// twice, t once and read y three times: 1.9 GB of HBM traffic per forward at B = 128.
//
// Both projections are computed TRANSPOSED so that no operand ever changes layout:
//   phase 1   y^T (features x tokens) = W_o (A operand: rows = features, from the LDS weight ring) * o^T (B operand: a lane
//             loads the 8 consecutive plane slots of its token straight from global -- the (hi, lo) planes ARE fragments).
//             Accumulator block (fb, tb): lane (l15, g) holds features 16 fb + 4g + r of token 16 tb + l15.  The bias
//             rides in the image at the ones slot of the `o` planes (slot D).
//   epilogue  dropout on the accumulators, then ONE split: the accumulators of feature blocks (2s, 2s + 1) of a token are
//             8 values = a bf16x3 B fragment of k-step s under the feature permutation kappa(g, e) = (e < 4 ? 4g + e :
//             16 + 4g + e - 4).  The W_a image is built in that order (rp_jobs_add_kappa), so y never leaves registers;
//             the same registers are the (hi, lo) planes of y the backward GEMMs want (training: 8-byte stores).
//   phase 2   pre^T (queries x tokens) = W_a (A, from the ring) * y^T (B, registers), four query-block groups so that the
//             accumulators fit beside the 160 fragment registers; tanh (v_exp + v_rcp), dot with q_a in the lane, two
//             cross-lane steps -> a[token] in every lane of the token's column.
//   pooling   softmax over the 32 token lanes (DPP row reductions), out[f] = sum_t w_t (hi + lo)[f][t].
// y enters the pooling sum as hi + lo (16 mantissa bits, the precision it has as a bf16x3 operand everywhere else).
//
// Two workgroup shapes (template parameter WAVES, NtCfg below), both two wavefronts per SIMD (<= 256 VGPRs):
//   8 waves, one workgroup per CU: 3 x 40 KB weight ring, chunk c + 2 issued when chunk c starts; every weight byte is
//     fetched once per 8 news, but the two waves of a SIMD run in lock step (same barriers), so their VALU epilogues
//     (dropout hash, split, tanh, pooling) coincide and the matrix pipe idles meanwhile;
//   4 waves, two workgroups per CU: 2 x 38 KB ring each, chunk c + 1 issued when chunk c starts; twice the L2 -> LDS
//     weight traffic, but the two waves of a SIMD belong to free-running workgroups and one's epilogue hides under the
//     other's MFMAs.
// Sync: one barrier + one vmcnt(0) per chunk; training stores go out right after that barrier so the next wait finds them done.
#pragma once
#include <type_traits>
#include <utility>

#include "nrl_news_fused.h"
#include "nrl_news_tail_api.h"

namespace nrl {

template <int WAVES>
struct NtCfg;
template <>
struct NtCfg<8> {                      // one workgroup per CU
  static constexpr int SLOT = 40 * 1024, SLOTS = 3, LOOK = 2;
  static constexpr int KPARTS = 2;     // forward phase 2: k-steps {0-4, 5-9} per query group (<= 40 pieces of 1 KiB)
  static constexpr int KPC = 2;        // backward phase C: k-steps per chunk
  static constexpr int BSLOT = 40 * 1024;
};
template <>
struct NtCfg<4> {                      // two workgroups per CU
  static constexpr int SLOT = 38 * 1024, SLOTS = 2, LOOK = 1;
  static constexpr int KPARTS = 3;     // k-steps {0-3, 4-6, 7-9} (<= 32 pieces)
  static constexpr int KPC = 1;
  static constexpr int BSLOT = 26 * 1024;
};
// compile-time loop: f(integral_constant<int, 0>{}), f(<1>), ...  (hipcc gives up `#pragma unroll` on the largest bodies and then
// indexes the register arrays dynamically -- they land in scratch)
template <class F, int... I>
__device__ __forceinline__ void nt_static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nt_static_for(F&& f) {
  nt_static_for(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__host__ __device__ constexpr int nt_kp0(int kparts, int p) {   // first k-step of part p of the phase-2 reduction
  return kparts == 2 ? 5 * p : (p == 0 ? 0 : p == 1 ? 4 : p == 2 ? 7 : 10);
}

__device__ __forceinline__ float nt_dpp(float v, int ctrl_sel) {
  const int i = __builtin_bit_cast(int, v);
  int r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1, 0, 3, 2]
    case 1: r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2, 3, 0, 1]
    case 2: r = __builtin_amdgcn_mov_dpp(i, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
    default: r = __builtin_amdgcn_mov_dpp(i, 0x140, 0xF, 0xF, true); break;  // row_mirror
  }
  return __builtin_bit_cast(float, r);
}
// all-reduce over the 16 lanes of a DPP row (= the 16 tokens of a column block held by one lane group g)
__device__ __forceinline__ float nt_row_sum(float v) {
  v += nt_dpp(v, 0);
  v += nt_dpp(v, 1);
  v += nt_dpp(v, 2);
  v += nt_dpp(v, 3);
  return v;
}
__device__ __forceinline__ float nt_row_max(float v) {
  v = fmaxf(v, nt_dpp(v, 0));
  v = fmaxf(v, nt_dpp(v, 1));
  v = fmaxf(v, nt_dpp(v, 2));
  v = fmaxf(v, nt_dpp(v, 3));
  return v;
}
// 8-byte plane store; NT = 1: streaming (write-once activations far larger than the L2 gain nothing from write-allocate)
template <int NT>
__device__ __forceinline__ void nt_store8(unsigned char* dst, uint32_t a, uint32_t b) {
  typedef uint32_t nt_u32x2 __attribute__((ext_vector_type(2)));
  if constexpr (NT) __builtin_nontemporal_store(nt_u32x2{a, b}, reinterpret_cast<nt_u32x2*>(dst));
  else *reinterpret_cast<uint2*>(dst) = make_uint2(a, b);
}
// tanh(x) = 1 - 2 / (exp(2x) + 1): v_exp_f32 + v_rcp_f32 (1 ulp each; absolute error ~1e-7, saturates cleanly at +-1)
__device__ __forceinline__ float nt_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// ABL (tools/nt_probe.hip only; product code uses 0): 1 = no dropout hash, 2 = no training stores, 4 = no weight DMA,
// 8 = no phase-1 MFMAs, 16 = no phase-2 MFMAs, 32 = no pooling reduction, 64 = plain y stores, 128 = `o` from cache-resident planes
// SAVE: 0 = evaluation (nothing but `out`), 1 = training (y planes + w; the fused backward recomputes tanh), 2 = training with
// the tanh output t as well (backward through pool_bwd_pre)
// The kernel's work for one wave's news in two compile-time shapes: NTB = 2 token blocks, and (SH: pad-row sharing, a short
// news of an evaluation call) NTB = 1.  Both execute the same barriers and the same share of the weight DMA, so waves of
// either shape can share a workgroup.  (A function template rather than a lambda inside the kernel: as a nested closure the
// evaluation shapes kept 40-odd captured values in scratch.)
template <int SAVE, int WAVES, int ABL, bool SH>
__device__ __forceinline__ void news_tail_fwd_body(const NewsTailArgs& P, unsigned char* const smem, const int64_t news,
                                                   const bool news_ok) {
  using Cfg = NtCfg<WAVES>;
  constexpr int NT_SLOT = Cfg::SLOT, NT_SLOTS = Cfg::SLOTS, LOOK = Cfg::LOOK, KPARTS = Cfg::KPARTS;
  constexpr int NT_NCHUNK = NT_KB + 4 * KPARTS;       // phase 2: 4 query groups x KPARTS parts of the reduction
  constexpr int PPW = (40 + WAVES - 1) / WAVES;       // DMA pieces per wave and chunk (<= 40 pieces per chunk)
  constexpr int STREAM = (ABL & 64) ? 0 : 1;          // y planes: streaming stores (probe: ABL 64 = plain)
  constexpr int NTB = SH ? 1 : 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* const qa_s = reinterpret_cast<float*>(smem + NT_SLOTS * NT_SLOT);
  if (tid < 256) qa_s[tid] = tid < P.Q ? P.q_a[tid] : 0.f;
  __syncthreads();

  const int L = P.L, D = P.D, Q = P.Q;
  const int64_t row0 = news * L;

  // ---- weight DMA: chunk c < NT_KB = k-block c of the W_o image (38 pieces of 1 KiB); chunk NT_KB + KPARTS grp + part =
  // k-steps of part `part` x query blocks of group grp x (hi, lo) of the W_a image.  Pieces wave, wave + WAVES, ...
  auto issue_chunk = [&](int c) {
    if constexpr (ABL & 4) return;
    const uint32_t dst = smem_base + (uint32_t)(c % NT_SLOTS) * (uint32_t)NT_SLOT;
    if (c < NT_KB) {
      // (wave-uniform SGPR base + one shared lane offset: per-piece VGPR address pairs cost 2 x PPW registers)
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img_o) + (size_t)c * (NT_FB * 2048);
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        int piece = wave + q * WAVES;
        piece = piece < 2 * NT_FB ? piece : 2 * NT_FB - 1;
        glds16_saddr(src + piece * 1024, lane_off, dst + (uint32_t)piece * 1024u);
      }
    } else if (c < NT_NCHUNK) {
      const int j = c - NT_KB, grp = j / KPARTS, part = j - grp * KPARTS;
      const int nb0 = grp == 0 ? 0 : 1 + 3 * grp, nbs = grp == 0 ? 4 : 3;
      const int ks0 = nt_kp0(KPARTS, part), nks = nt_kp0(KPARTS, part + 1) - ks0;
      const int pieces = 2 * nks * nbs;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img_a);
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        int piece = wave + q * WAVES;
        piece = piece < pieces ? piece : pieces - 1;
        const int ks = piece / (2 * nbs), rem = piece - ks * 2 * nbs;     // rem = nb * 2 + plane
        glds16_saddr(src + ((size_t)((ks0 + ks) * NT_QB + nb0) * 2 + rem) * 1024, lane_off, dst + (uint32_t)piece * 1024u);
      }
    }
  };

  // ---- `o` fragments (B operand of phase 1): token 16 tb + l15 (clamped into the news: the pad columns are computed on a
  // copy of the last token and masked at the softmax), slots 32 kb + 8g .. + 7 = block column 2 kb + (g >> 1), half g & 1
  const unsigned char* orow[2];
  int64_t mrow[2];
  bool tok_ok[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int t = tb * 16 + l15;
    tok_ok[tb] = t < L;
    mrow[tb] = row0 + (tok_ok[tb] ? t : L - 1);
    // (probe, ABL 128: every news reads the `o` planes of one of the first 64 news -- 2.3 MB, cache-resident: the loads are issued,
    //  nothing comes from HBM.  The upper bound of what handing `o` over on chip could save this kernel, tools/nt_probe.hip)
    const int64_t mo = (ABL & 128) ? (news & 63) * L + (tok_ok[tb] ? t : L - 1) : mrow[tb];
    orow[tb] = P.o_planes + ((mo >> 4) * NT_FB + (g >> 1)) * 1024 + (mo & 15) * 32 + (g & 1) * 16;
  }
  auto load_o = [&](int kb, bf16x8 (&oh)[2], bf16x8 (&ol)[2]) {
    // block column 19 (k-block 9, g >= 2) does not exist: read a valid address, zero the fragment
    const bool dead = kb == NT_KB - 1 && g >= 2;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const unsigned char* p = orow[tb] + (dead ? 0 : kb * 2048);
      uint4 h = *reinterpret_cast<const uint4*>(p), l = *reinterpret_cast<const uint4*>(p + 512);
      if (dead) { h = make_uint4(0u, 0u, 0u, 0u); l = h; }
      oh[tb] = __builtin_bit_cast(bf16x8, h);
      ol[tb] = __builtin_bit_cast(bf16x8, l);
    }
  };

  // =============================== phase 1: y^T = W_o o^T ===============================================
  f32x4 acc[NT_FB][2];
#pragma unroll
  for (int fb = 0; fb < NT_FB; ++fb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) acc[fb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int P1_PAIRS = (NT_FB + 1) / 2;
  auto p1_read = [&](const unsigned char* base, int p, bf16x8 (&wh)[2], bf16x8 (&wl)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int fb = 2 * p + jj < NT_FB ? 2 * p + jj : NT_FB - 1;
      wh[jj] = *reinterpret_cast<const bf16x8*>(base + fb * 2048);
      wl[jj] = *reinterpret_cast<const bf16x8*>(base + fb * 2048 + 1024);
    }
  };
  auto p1_mfma = [&](int p, const bf16x8 (&wh)[2], const bf16x8 (&wl)[2], const bf16x8 (&oh)[2], const bf16x8 (&ol)[2]) {
    if constexpr (ABL & 8) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) asm volatile("" ::"v"(wh[jj]), "v"(wl[jj]));
      return;
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (2 * p + jj < NT_FB)
#pragma unroll
          for (int tb = 0; tb < NTB; ++tb)
            acc[2 * p + jj][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? wl[jj] : wh[jj], pass == 0 ? ol[tb] : oh[tb],
                                                                          acc[2 * p + jj][tb], 0, 0, 0);
  };
  auto p1_chunk = [&](int slot, const bf16x8 (&oh)[2], const bf16x8 (&ol)[2]) {
    const unsigned char* base = smem + slot * NT_SLOT + lane * 16;
    bf16x8 wh0[2], wl0[2], wh1[2], wl1[2];
    p1_read(base, 0, wh0, wl0);
#pragma unroll
    for (int p = 0; p < P1_PAIRS; p += 2) {
      if (p + 1 < P1_PAIRS) p1_read(base, p + 1, wh1, wl1);
      p1_mfma(p, wh0, wl0, oh, ol);
      if (p + 1 < P1_PAIRS) {
        if (p + 2 < P1_PAIRS) p1_read(base, p + 2, wh0, wl0);
        p1_mfma(p + 1, wh1, wl1, oh, ol);
      }
    }
    if constexpr (!(ABL & 8)) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int p = 0; p < P1_PAIRS; ++p) {
        if (p + 1 < P1_PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        if (2 * p + 1 < NT_FB) {
          __builtin_amdgcn_sched_group_barrier(0x008, 6 * NTB, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTB, 0);
        }
      }
    }
  };

#pragma unroll
  for (int c = 0; c < LOOK; ++c) issue_chunk(c);
  {
    bf16x8 oha[2], ola[2], ohb[2], olb[2];
    load_o(0, oha, ola);
    auto step = [&](int kb, const bf16x8 (&ch)[2], const bf16x8 (&cl)[2], bf16x8 (&nh)[2], bf16x8 (&nl)[2]) {
      wait_vmcnt<0>();                     // chunk kb has landed for this wave (issued LOOK chunks ago) ...
      __builtin_amdgcn_s_barrier();        // ... and for all; everyone is done with the slot chunk kb + LOOK goes to
      issue_chunk(kb + LOOK);
      load_o(kb + 1 < NT_KB ? kb + 1 : kb, nh, nl);
      __builtin_amdgcn_sched_barrier(0);
      p1_chunk(kb % NT_SLOTS, ch, cl);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kb = 0; kb < NT_KB; kb += 2) {
      step(kb, oha, ola, ohb, olb);
      step(kb + 1, ohb, olb, oha, ola);
    }
  }

  // =============================== epilogue 1: dropout, split, (training) y planes =======================
  bf16x8 yh[NT_KS][2], yl[NT_KS][2];
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb) {
    const uint32_t idx_row = (uint32_t)mrow[tb] * (uint32_t)D;
#pragma unroll
    for (int s = 0; s < NT_KS; ++s) {
      f32x4 v0 = acc[2 * s][tb];
      f32x4 v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (2 * s + 1 < NT_FB) v1 = acc[2 * s + 1][tb];
      const int f0 = 32 * s + 4 * g;
      if (!(ABL & 1) && P.drop2.thresh != 0u) {
        const uint32_t i0 = idx_row + (uint32_t)f0, i1 = i0 + 16u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] *= P.drop2.mult(i0 + r);
          if (2 * s + 1 < NT_FB) v1[r] *= P.drop2.mult(i1 + r);
        }
      }
      // ones column at feature D (block 18, row 12): the bias row of the W_a image / the bias gradient's operand
      if (2 * s == NT_FB - 1) {
        if (f0 == D) v0[0] = 1.0f;
      }
      rp_split8(make_float4(v0[0], v0[1], v0[2], v0[3]), make_float4(v1[0], v1[1], v1[2], v1[3]), yh[s][tb], yl[s][tb]);
    }
  }
  // y planes of k-step s: block columns 2s, 2s + 1, 8 bytes per lane and plane
  auto store_y = [&](int s) {
    if (!SAVE || (ABL & 2) || !news_ok) return;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      // (no token mask: the pad columns hold a bit-identical copy of the last token and store it to the same address;
      //  opaque row: hipcc otherwise hoists the 80 store addresses out of phase 2 and spills them)
      int64_t m = mrow[tb];
      asm volatile("" : "+v"(m));
      unsigned char* dst = P.y_planes + ((m >> 4) * NT_FB + 2 * s) * 1024 + (m & 15) * 32 + 8 * g;
      const uint4 h = __builtin_bit_cast(uint4, yh[s][tb]), l = __builtin_bit_cast(uint4, yl[s][tb]);
      nt_store8<STREAM>(dst, h.x, h.y);
      nt_store8<STREAM>(dst + 512, l.x, l.y);
      if (2 * s + 1 < NT_FB) {
        nt_store8<STREAM>(dst + 1024, h.z, h.w);
        nt_store8<STREAM>(dst + 1536, l.z, l.w);
      }
    }
  };

  // =============================== phase 2: pre^T = W_a y^T, tanh, . q_a =================================
  float apart[2] = {0.f, 0.f};
  nt_static_for<4>([&](auto grp_c) {
    constexpr int grp = decltype(grp_c)::value;
    constexpr int nb0 = grp == 0 ? 0 : 1 + 3 * grp, nbs = grp == 0 ? 4 : 3;
    f32x4 pacc[4][2];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) pacc[nb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    nt_static_for<KPARTS>([&](auto part_c) {
      constexpr int part = decltype(part_c)::value;
      constexpr int c = NT_KB + KPARTS * grp + part;
      constexpr int ks0 = nt_kp0(KPARTS, part), nks = nt_kp0(KPARTS, part + 1) - ks0;
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      issue_chunk(c + LOOK);
      // training stores under this chunk's MFMAs: k-step j of the y planes in the j-th chunk of phase 2 (the first chunks
      // take the k-steps that are left over when there are fewer chunks than k-steps)
      {
        constexpr int j = c - NT_KB, nch = 4 * KPARTS;
        if constexpr (j < NT_KS) store_y(j);
        if constexpr (nch + j < NT_KS) store_y(nch + j);
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* base = smem + (c % NT_SLOTS) * NT_SLOT + lane * 16;
      constexpr int nsteps = nks * nbs;              // step i = (k-step ks0 + i / nbs, query block nb0 + i % nbs)
      constexpr int npairs = (nsteps + 1) / 2;
      auto p2_read = [&](int p, bf16x8 (&wh)[2], bf16x8 (&wl)[2]) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int i = 2 * p + jj < nsteps ? 2 * p + jj : nsteps - 1;
          wh[jj] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
          wl[jj] = *reinterpret_cast<const bf16x8*>(base + i * 2048 + 1024);
        }
      };
      auto p2_mfma = [&](int p, const bf16x8 (&wh)[2], const bf16x8 (&wl)[2]) {
        if constexpr (ABL & 16) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) asm volatile("" ::"v"(wh[jj]), "v"(wl[jj]));
          return;
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int i = 2 * p + jj;
            if (i < nsteps) {
              const int s = ks0 + i / nbs, nb = i % nbs;
#pragma unroll
              for (int tb = 0; tb < NTB; ++tb)
                pacc[nb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? wl[jj] : wh[jj],
                                                                       pass == 0 ? yl[s][tb] : yh[s][tb], pacc[nb][tb], 0, 0, 0);
            }
          }
      };
      bf16x8 wh0[2], wl0[2], wh1[2], wl1[2];
      p2_read(0, wh0, wl0);
#pragma unroll
      for (int p = 0; p < npairs; p += 2) {
        if (p + 1 < npairs) p2_read(p + 1, wh1, wl1);
        p2_mfma(p, wh0, wl0);
        if (p + 1 < npairs) {
          if (p + 2 < npairs) p2_read(p + 2, wh0, wl0);
          p2_mfma(p + 1, wh1, wl1);
        }
      }
      if constexpr (!(ABL & 16)) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int p = 0; p < npairs; ++p) {
          if (p + 1 < npairs) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          if (2 * p + 1 < nsteps) {
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NTB, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NTB, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // group epilogue: t = tanh(pre) (b_a came in through the ones feature), a += t . q_a over this lane's 4 queries
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      if (nb >= nbs) continue;
      const int q0 = 16 * (nb0 + nb) + 4 * g;
      const float4 qv = *reinterpret_cast<const float4*>(qa_s + q0);
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb) {
        const f32x4 pv = pacc[nb][tb];
        const float4 tv = make_float4(nt_tanh(pv[0]), nt_tanh(pv[1]), nt_tanh(pv[2]), nt_tanh(pv[3]));
        apart[tb] = fmaf(tv.x, qv.x, apart[tb]);
        apart[tb] = fmaf(tv.y, qv.y, apart[tb]);
        apart[tb] = fmaf(tv.z, qv.z, apart[tb]);
        apart[tb] = fmaf(tv.w, qv.w, apart[tb]);
        if (SAVE == 2 && !(ABL & 2) && news_ok && q0 < Q) {   // (pad columns: same value, same address)
          int64_t m = mrow[tb];
          asm volatile("" : "+v"(m));
          *reinterpret_cast<float4*>(P.t + m * Q + q0) = tv;
        }
      }
      // (pinned: the conditional stores split this epilogue into basic blocks, and hipcc sinks the dot products -- whose
      //  result is only read after the last group -- down to the pooling, keeping every tanh value and q_a alive: spills)
      asm volatile("" : "+v"(apart[0]), "+v"(apart[1]));
    }
  });

  // =============================== softmax over the tokens, pooled sum ====================================
  float wt[2];
  {
    float a[2];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      float v = apart[tb];
      v += nf_xor16(v, lane);
      v += nf_xor32(v, lane);
      a[tb] = tok_ok[tb] ? v : -INFINITY;
    }
    if constexpr (SH) {
      // tokens 16 .. L - 1 of a short news ARE token 15 (same o row -> same y -> same logit): the value the lanes l15 == 15 hold
      const float a15 = __shfl(a[0], lane | 15, 64);
      a[1] = tok_ok[1] ? a15 : -INFINITY;
    }
    const float mx = nt_row_max(fmaxf(a[0], a[1]));
    constexpr float LOG2E = 1.4426950408889634f;
    const float e0 = __builtin_amdgcn_exp2f((a[0] - mx) * LOG2E), e1 = __builtin_amdgcn_exp2f((a[1] - mx) * LOG2E);
    const float inv = 1.0f / nt_row_sum(e0 + e1);
    wt[0] = e0 * inv;
    wt[1] = e1 * inv;
    if (SAVE && !(ABL & 2) && news_ok && g == 0) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
        if (tok_ok[tb]) P.w[mrow[tb]] = wt[tb];
    }
  }
  // out[f] = sum_t w_t y[t][f], y = hi + lo; feature 16 fb + 4g + r sits in k-step fb / 2, elements 4 (fb & 1) + r
  float* const outp = P.out + news * D;
#pragma unroll
  for (int fb = 0; fb < NT_FB; ++fb) {
    const int s = fb >> 1, wsel = (fb & 1) * 2;
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = 0.f;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const uint4 h = __builtin_bit_cast(uint4, yh[s][SH ? 0 : tb]), l = __builtin_bit_cast(uint4, yl[s][SH ? 0 : tb]);
      uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
      if constexpr (SH) {
        if (tb == 1) {   // y of tokens 16 .. L - 1 = y of token 15: this lane's features as the lane (l15 = 15, same g) holds them
          hw[wsel] = __shfl(hw[wsel], lane | 15, 64);
          hw[wsel + 1] = __shfl(hw[wsel + 1], lane | 15, 64);
          lw[wsel] = __shfl(lw[wsel], lane | 15, 64);
          lw[wsel + 1] = __shfl(lw[wsel + 1], lane | 15, 64);
        }
      }
      // (opaque: hipcc otherwise recognises `word << 16` as the hi half the split already computed and keeps ~90 unpacked
      //  floats alive -- spilled -- across all of phase 2)
      asm volatile("" : "+v"(hw[wsel]), "+v"(hw[wsel + 1]), "+v"(lw[wsel]), "+v"(lw[wsel + 1]));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t hv = hw[wsel + (r >> 1)], lv = lw[wsel + (r >> 1)];
        const float yv = __builtin_bit_cast(float, (r & 1) ? (hv & 0xFFFF0000u) : (hv << 16)) +
                         __builtin_bit_cast(float, (r & 1) ? (lv & 0xFFFF0000u) : (lv << 16));
        pr[r] = fmaf(wt[tb], yv, pr[r]);
      }
    }
    if constexpr (!(ABL & 32)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = nt_row_sum(pr[r]);
    }
    const int f0 = 16 * fb + 4 * g;
    if (news_ok && l15 == 0 && f0 < D) *reinterpret_cast<float4*>(outp + f0) = make_float4(pr[0], pr[1], pr[2], pr[3]);
  }
  // (every DMA issued has been waited for: the last chunk's wait covers chunks up to NT_NCHUNK - 1, none is issued later)
  static_assert(2 * 5 * 4 <= NT_SLOT / 1024 || KPARTS == 3, "phase-2 chunk must fit a ring slot");
}

template <int SAVE, int WAVES, int ABL = 0, bool SHARE = false>
__global__ void __launch_bounds__(WAVES * 64, 2) news_tail_fwd_kernel(const NewsTailArgs P) {
  static_assert(!(SHARE && SAVE), "pad-row sharing is for evaluation forwards");
  using Cfg = NtCfg<WAVES>;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[Cfg::SLOTS * Cfg::SLOT + 1024];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t news_raw = (int64_t)blockIdx.x * WAVES + wave;
  const bool news_ok = news_raw < P.n_news;
  int64_t news = news_ok ? news_raw : 0;            // idle waves of the last workgroup recompute news 0 and store nothing
  if constexpr (SHARE) {
    // the wave's news comes from the short-first list; position < n_short <=> a short news (wave-uniform)
    bool short_w = false;
    if (news_ok) {
      news = __builtin_amdgcn_readfirstlane(P.perm[news_raw]);
      short_w = news_raw < (int64_t)__builtin_amdgcn_readfirstlane(*P.n_short);
    }
    if (short_w) news_tail_fwd_body<SAVE, WAVES, ABL, true>(P, smem, news, news_ok);
    else news_tail_fwd_body<SAVE, WAVES, ABL, false>(P, smem, news, news_ok);
  } else {
    news_tail_fwd_body<SAVE, WAVES, ABL, false>(P, smem, news, news_ok);
  }
}

// (Tail balancing measured and NOT adopted: cutting the grid at the last multiple of the 512 workgroup slots and sending the
// remainder -- 224 workgroups at B = 128 -- as a second launch forced to one workgroup per CU by 32 KB of unused dynamic LDS.
// Step 3.05-3.08 vs 3.01 ms; forward 2 x 130 vs 250 us, backward 2 x 177 vs ~300 us: the drain of the first launch and the
// ramp of the second cost more than the SIMD sharing in the last round.)
template <int WAVES = 4, int ABL = 0>
static inline int launch_news_tail_fwd(const NewsTailArgs& a, hipStream_t st) {
  if (a.n_news <= 0) return NRL_OK;
  const int64_t blocks = ceil_div(a.n_news, WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  NRL_REQUIRE(a.D == 300 && a.Q <= 16 * NT_QB && a.L >= 1 && a.L <= 32, "fused news tail: unsupported geometry");
  NRL_REQUIRE((((uintptr_t)a.out | (uintptr_t)a.t | (uintptr_t)a.q_a) & 15) == 0, "fused news tail: 16-byte alignment");
  if (a.y_planes != nullptr) {
    NRL_REQUIRE(a.w != nullptr, "fused news tail: save y planes and w (and optionally t), or nothing");
    NRL_REQUIRE(a.perm == nullptr, "fused news tail: pad-row sharing is for evaluation forwards");
    if (a.t != nullptr) hipLaunchKernelGGL((news_tail_fwd_kernel<2, WAVES, ABL>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a);
    else hipLaunchKernelGGL((news_tail_fwd_kernel<1, WAVES, ABL>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a);
  } else {
    NRL_REQUIRE(a.t == nullptr && a.w == nullptr, "fused news tail: save y planes and w (and optionally t), or nothing");
    if (a.perm != nullptr) {
      NRL_REQUIRE(a.n_short != nullptr && a.drop2.thresh == 0u, "fused news tail: pad-row sharing needs the short-first list and no dropout");
      hipLaunchKernelGGL((news_tail_fwd_kernel<0, WAVES, ABL, true>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a);
    } else {
      hipLaunchKernelGGL((news_tail_fwd_kernel<0, WAVES, ABL>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a);
    }
  }
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

// =====================================================================================================
// Backward of the additive attention of the news path in ONE kernel (replaces pool_bwd_pre + the additive-attention
// activation-gradient GEMM; the tanh output is RECOMPUTED from the y planes, so the forward never writes it):
//
//   t = tanh(y W_a^T + b_a)                       recomputed (phase A: the forward's phase 2, all 13 query blocks kept)
//   c_l = d_out . y_l                             one extra row of the A operand: d_out rides as query row Q of block Q / 16
//   da_l = w_l (c_l - sum_l' w_l' c_l');  d_pre = da * q_a * (1 - t^2);  dq_a += sum_l da_l t_l        attention.py:34-40
//   dy = (d_pre W_a + w_l d_out) * dropout2       phase C: dy^T = W_a^T (A: rows = features, reduction = queries in kappa
//                                                  order) * d_pre^T (B: the accumulators of phase A, split once), two halves
// Outputs: d_pre and dy as (hi, lo) planes over the real rows (operands of the two weight gradients and of the
// out-projection's activation gradient), dq_a through LDS + one atomic per query and workgroup.
// (Round 4's phase D -- the out-projection's activation gradient d_o = dy W_o in the same kernel -- grew it 337 -> 488 us to remove a
//  135 us launch: tools/experimental/nrl_news_tail_phase_d.inc.)
template <int WAVES, int ABL = 0>
__global__ void __launch_bounds__(WAVES * 64, 2) news_tail_bwd_kernel(const NewsTailBwdArgs P) {
  using Cfg = NtCfg<WAVES>;
  constexpr int NT_SLOT = Cfg::BSLOT, NT_SLOTS = Cfg::SLOTS, LOOK = Cfg::LOOK, KPC = Cfg::KPC;
  constexpr int NPC = (NT_QS + KPC - 1) / KPC;         // phase-C chunks per half of the features
  constexpr int NT_NCHUNK_C = NT_KS + 2 * NPC;         // chunks of phases A and C
  constexpr int PPW = (2 * KPC * 10 + WAVES - 1) / WAVES > (2 * NT_QB + WAVES - 1) / WAVES ? (2 * KPC * 10 + WAVES - 1) / WAVES
                                                                                             : (2 * NT_QB + WAVES - 1) / WAVES;
  static_assert(2 * NT_QB * 1024 <= NT_SLOT && 2 * KPC * 10 * 1024 <= NT_SLOT, "chunk must fit a ring slot");
  constexpr int STREAM = (ABL & 64) ? 0 : 1;          // d_pre / dy planes: streaming stores (probe: ABL 64 = plain)
  constexpr int STREAM_DY = STREAM;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NT_SLOTS * NT_SLOT + 2048 + WAVES * NT_DROW * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* const qa_s = reinterpret_cast<float*>(smem + NT_SLOTS * NT_SLOT);          // 256 floats
  float* const dq_s = qa_s + 256;                                                   // 256 floats
  float* const d_s = dq_s + 256 + wave * NT_DROW;                                   // this wave's d_out row

  const int L = P.L, D = P.D, Q = P.Q;
  const int64_t news_raw = (int64_t)blockIdx.x * WAVES + wave;
  const bool news_ok = news_raw < P.n_news;
  const int64_t news = news_ok ? news_raw : 0;
  const int64_t row0 = news * L;

  if (tid < 256) {
    qa_s[tid] = tid < Q ? P.q_a[tid] : 0.f;
    dq_s[tid] = 0.f;
  }
#pragma unroll
  for (int i = lane; i < NT_DROW; i += 64) d_s[i] = i < D ? P.d_out[news * D + i] : 0.f;
  __syncthreads();

  // chunk c < NT_KS: k-step c of the W_a image (26 pieces); chunk NT_KS + NPC half + p: k-steps KPC p .. of the W_a^T image
  // x feature blocks of the half (10 or 9) x (hi, lo)
  auto issue_chunk = [&](int c) {
    if constexpr (ABL & 4) return;
    const uint32_t dst = smem_base + (uint32_t)(c % NT_SLOTS) * (uint32_t)NT_SLOT;
    if (c < NT_KS) {
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img_a) + (size_t)c * (NT_QB * 2048);
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        int piece = wave + q * WAVES;
        piece = piece < 2 * NT_QB ? piece : 2 * NT_QB - 1;
        glds16_saddr(src + piece * 1024, lane_off, dst + (uint32_t)piece * 1024u);
      }
    } else if (c < NT_NCHUNK_C) {
      const int j = c - NT_KS, half = j / NPC, p = j - half * NPC;
      const int fb0 = half ? 10 : 0, nfb = half ? 9 : 10;
      const int nks = (p + 1) * KPC <= NT_QS ? KPC : NT_QS - p * KPC;
      const int pieces = nks * nfb * 2;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img_ad);
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        int piece = wave + q * WAVES;
        piece = piece < pieces ? piece : pieces - 1;
        const int ks = piece / (2 * nfb), rem = piece - ks * 2 * nfb;      // rem = fb * 2 + plane
        glds16_saddr(src + ((size_t)((KPC * p + ks) * NT_FB + fb0) * 2 + rem) * 1024, lane_off, dst + (uint32_t)piece * 1024u);
      }
    }
  };

  const unsigned char* yrow[2];
  int64_t mrow[2];
  bool tok_ok[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int t = tb * 16 + l15;
    tok_ok[tb] = t < L;
    mrow[tb] = row0 + (tok_ok[tb] ? t : L - 1);
    yrow[tb] = P.y_planes + (mrow[tb] >> 4) * (NT_FB * 1024) + (mrow[tb] & 15) * 32 + 8 * g;
  }
  // B fragment of k-step s in kappa order: features 32 s + 4g .. + 3 (block column 2s) and 32 s + 16 + 4g .. + 3 (2s + 1)
  auto load_y = [&](int s, bf16x8 (&yh)[2], bf16x8 (&yl)[2]) {
    const bool dead = 2 * s + 1 >= NT_FB;            // block column 19 does not exist
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const unsigned char* p = yrow[tb] + s * 2048;
      const uint2 h0 = *reinterpret_cast<const uint2*>(p), l0 = *reinterpret_cast<const uint2*>(p + 512);
      uint2 h1 = *reinterpret_cast<const uint2*>(p + (dead ? 0 : 1024)), l1 = *reinterpret_cast<const uint2*>(p + (dead ? 512 : 1536));
      if (dead) { h1 = make_uint2(0u, 0u); l1 = h1; }
      yh[tb] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
      yl[tb] = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
  };

  // =============================== phase A: pre^T = W_a y^T (+ the d_out row) =============================
  const int qc_blk = Q >> 4, qc_row = Q & 15;       // d_out rides as query row Q (a zero row of the image: Q < 16 NT_QB)
  f32x4 pacc[NT_QB][2];
#pragma unroll
  for (int nb = 0; nb < NT_QB; ++nb)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) pacc[nb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int PA_PAIRS = (NT_QB + 1) / 2;
#pragma unroll
  for (int c = 0; c < LOOK; ++c) issue_chunk(c);
  {
    bf16x8 yha[2], yla[2], yhb[2], ylb[2];
    load_y(0, yha, yla);
    auto step = [&](int s, const bf16x8 (&ch)[2], const bf16x8 (&cl)[2], bf16x8 (&nh)[2], bf16x8 (&nl)[2]) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      issue_chunk(s + LOOK);
      load_y(s + 1 < NT_KS ? s + 1 : s, nh, nl);
      // this k-step's slice of d_out in kappa order, split: the A fragment of row qc_row in block qc_blk
      bf16x8 dh, dl;
      {
        const float4 v0 = *reinterpret_cast<const float4*>(d_s + 32 * s + 4 * g);
        const float4 v1 = *reinterpret_cast<const float4*>(d_s + 32 * s + 16 + 4 * g);
        rp_split8(v0, v1, dh, dl);
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* base = smem + (s % NT_SLOTS) * NT_SLOT + lane * 16;
      auto rd = [&](int p, bf16x8 (&wh)[2], bf16x8 (&wl)[2]) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int nb = 2 * p + jj < NT_QB ? 2 * p + jj : NT_QB - 1;
          wh[jj] = *reinterpret_cast<const bf16x8*>(base + nb * 2048);
          wl[jj] = *reinterpret_cast<const bf16x8*>(base + nb * 2048 + 1024);
          if (nb == qc_blk && l15 == qc_row) { wh[jj] = dh; wl[jj] = dl; }
        }
      };
      auto mm = [&](int p, const bf16x8 (&wh)[2], const bf16x8 (&wl)[2]) {
        if constexpr (ABL & 8) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) asm volatile("" ::"v"(wh[jj]), "v"(wl[jj]));
          return;
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (2 * p + jj < NT_QB)
#pragma unroll
              for (int tb = 0; tb < 2; ++tb)
                pacc[2 * p + jj][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? wl[jj] : wh[jj], pass == 0 ? cl[tb] : ch[tb],
                                                                               pacc[2 * p + jj][tb], 0, 0, 0);
      };
      bf16x8 wh0[2], wl0[2], wh1[2], wl1[2];
      rd(0, wh0, wl0);
#pragma unroll
      for (int p = 0; p < PA_PAIRS; p += 2) {
        if (p + 1 < PA_PAIRS) rd(p + 1, wh1, wl1);
        mm(p, wh0, wl0);
        if (p + 1 < PA_PAIRS) {
          if (p + 2 < PA_PAIRS) rd(p + 2, wh0, wl0);
          mm(p + 1, wh1, wl1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < NT_KS; s += 2) {
      step(s, yha, yla, yhb, ylb);
      step(s + 1, yhb, ylb, yha, yla);
    }
  }

  // =============================== phase B: the attention's backward, elementwise ========================
  float wt[2], da[2];
  {
    // c_l sits in accumulator (qc_blk, tb), row qc_row = lanes g == qc_row / 4, element qc_row % 4
    float cl[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      float v = 0.f;
#pragma unroll
      for (int nb = 0; nb < NT_QB; ++nb)
        if (nb == qc_blk) {
          const f32x4 pv = pacc[nb][tb];
          v = (qc_row & 3) == 0 ? pv[0] : (qc_row & 3) == 1 ? pv[1] : (qc_row & 3) == 2 ? pv[2] : pv[3];
        }
      cl[tb] = __shfl(v, (qc_row >> 2) * 16 + l15, 64);
      wt[tb] = tok_ok[tb] ? P.w[mrow[tb]] : 0.f;
    }
    float dbar = nt_row_sum(wt[0] * cl[0] + wt[1] * cl[1]);
    da[0] = wt[0] * (cl[0] - dbar);
    da[1] = wt[1] * (cl[1] - dbar);
    // (pad columns: w = 0 -> da = 0 -> d_pre = 0; their dy is a copy of nothing and is never stored)
  }
  // t = tanh(pre); dq_a partials; d_pre in place
#pragma unroll
  for (int nb = 0; nb < NT_QB; ++nb) {
    const int q0 = 16 * nb + 4 * g;
    const float4 qv = *reinterpret_cast<const float4*>(qa_s + q0);
    const float qq[4] = {qv.x, qv.y, qv.z, qv.w};
    float dqp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dqp[r] = 0.f;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      f32x4 pv = pacc[nb][tb];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tv = nt_tanh(pv[r]);
        dqp[r] = fmaf(da[tb], tv, dqp[r]);
        pv[r] = da[tb] * qq[r] * (1.0f - tv * tv);
      }
      pacc[nb][tb] = pv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = nt_row_sum(dqp[r]);
      if (l15 == 0 && news_ok && q0 + r < Q) atomicAdd(dq_s + q0 + r, v);
    }
  }
  // split once: B fragments of phase C (k-step s' = query blocks 2s', 2s' + 1) = the d_pre planes
  bf16x8 ph[NT_QS][2], pl[NT_QS][2];
#pragma unroll
  for (int s = 0; s < NT_QS; ++s)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const f32x4 v0 = pacc[2 * s][tb];
      f32x4 v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (2 * s + 1 < NT_QB) v1 = pacc[2 * s + 1][tb];
      rp_split8(make_float4(v0[0], v0[1], v0[2], v0[3]), make_float4(v1[0], v1[1], v1[2], v1[3]), ph[s][tb], pl[s][tb]);
    }
  // d_pre planes: (Q + 15) / 16 block columns per 16-row block -- what the weight gradient that reads them was sized for
  const int ncb_q = (Q + 15) >> 4;
  auto store_dpre = [&](int s) {
    if ((ABL & 2) || !news_ok || 2 * s >= ncb_q) return;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      if (!tok_ok[tb]) continue;              // (pad columns hold zeros, not a copy: masked)
      int64_t m = mrow[tb];
      asm volatile("" : "+v"(m));
      unsigned char* dst = P.dpre_planes + ((m >> 4) * ncb_q + 2 * s) * 1024 + (m & 15) * 32 + 8 * g;
      const uint4 h = __builtin_bit_cast(uint4, ph[s][tb]), l = __builtin_bit_cast(uint4, pl[s][tb]);
      nt_store8<STREAM>(dst, h.x, h.y);
      nt_store8<STREAM>(dst + 512, l.x, l.y);
      if (2 * s + 1 < NT_QB && 2 * s + 1 < ncb_q) {
        nt_store8<STREAM>(dst + 1024, h.z, h.w);
        nt_store8<STREAM>(dst + 1536, l.z, l.w);
      }
    }
  };

  // =============================== phase C: dy^T = W_a^T d_pre^T, two halves of the features ===============
  auto phase_c = [&](auto half_c) {
    constexpr int half = decltype(half_c)::value;
    constexpr int fb0 = half ? 10 : 0, nfb = half ? 9 : 10;
    f32x4 acc[10][2];
#pragma unroll
    for (int fb = 0; fb < 10; ++fb)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) acc[fb][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      const int c = NT_KS + NPC * half + p;
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      issue_chunk(c + LOOK);
      const int nks = (p + 1) * KPC <= NT_QS ? KPC : NT_QS - p * KPC;
      if (half == 0) {                         // the d_pre planes go out under the first half's MFMAs
#pragma unroll
        for (int ks = 0; ks < KPC; ++ks)
          if (ks < nks) store_dpre(KPC * p + ks);
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* base = smem + (c % NT_SLOTS) * NT_SLOT + lane * 16;
      const int nsteps = nks * nfb;              // step i = (k-step KPC p + i / nfb, feature block fb0 + i % nfb)
      const int npairs = (nsteps + 1) / 2;
      auto rd = [&](int pp, bf16x8 (&wh)[2], bf16x8 (&wl)[2]) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int i = 2 * pp + jj < nsteps ? 2 * pp + jj : nsteps - 1;
          wh[jj] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
          wl[jj] = *reinterpret_cast<const bf16x8*>(base + i * 2048 + 1024);
        }
      };
      auto mm = [&](int pp, const bf16x8 (&wh)[2], const bf16x8 (&wl)[2]) {
        if constexpr (ABL & 16) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) asm volatile("" ::"v"(wh[jj]), "v"(wl[jj]));
          return;
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int i = 2 * pp + jj;
            if (i < nsteps) {
              const int s = KPC * p + i / nfb, fb = i % nfb;
#pragma unroll
              for (int tb = 0; tb < 2; ++tb)
                acc[fb][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 1 ? wl[jj] : wh[jj], pass == 0 ? pl[s][tb] : ph[s][tb],
                                                                      acc[fb][tb], 0, 0, 0);
            }
          }
      };
      bf16x8 wh0[2], wl0[2], wh1[2], wl1[2];
      rd(0, wh0, wl0);
#pragma unroll
      for (int pp = 0; pp < npairs; pp += 2) {
        if (pp + 1 < npairs) rd(pp + 1, wh1, wl1);
        mm(pp, wh0, wl0);
        if (pp + 1 < npairs) {
          if (pp + 2 < npairs) rd(pp + 2, wh0, wl0);
          mm(pp + 1, wh1, wl1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue of the half: dy = (acc + w_l d_out) * dropout2 -> planes (8 bytes per lane and plane)
#pragma unroll
    for (int fb = 0; fb < 10; ++fb) {
      if (fb >= nfb) continue;
      const int f0 = 16 * (fb0 + fb) + 4 * g;
      const float4 dv = *reinterpret_cast<const float4*>(d_s + f0);
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        f32x4 v = acc[fb][tb];
        v[0] = fmaf(wt[tb], dv.x, v[0]); v[1] = fmaf(wt[tb], dv.y, v[1]);
        v[2] = fmaf(wt[tb], dv.z, v[2]); v[3] = fmaf(wt[tb], dv.w, v[3]);
        int64_t m = mrow[tb];
        asm volatile("" : "+v"(m));
        if (!(ABL & 1) && P.drop2.thresh != 0u) {
          const uint32_t i0 = (uint32_t)m * (uint32_t)D + (uint32_t)f0;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= P.drop2.mult(i0 + r);
        }
        uint32_t h0, l0, h1, l1;
        split_pair(v[0], v[1], h0, l0);
        split_pair(v[2], v[3], h1, l1);
        if (!(ABL & 2) && news_ok && tok_ok[tb]) {
          unsigned char* dst = P.dy_planes + ((m >> 4) * NT_FB + fb0 + fb) * 1024 + (m & 15) * 32 + 8 * g;
          nt_store8<STREAM_DY>(dst, h0, h1);
          nt_store8<STREAM_DY>(dst + 512, l0, l1);
        }
      }
    }
  };
  phase_c(std::integral_constant<int, 0>{});
  phase_c(std::integral_constant<int, 1>{});

  // dq_a: the workgroup's sums -> one atomic per query
  __syncthreads();
  if (tid < Q) atomicAdd(P.dq_a + tid, dq_s[tid]);
}

template <int WAVES = 4, int ABL = 0>
static inline int launch_news_tail_bwd(const NewsTailBwdArgs& a, hipStream_t st) {
  if (a.n_news <= 0) return NRL_OK;
  const int64_t blocks = ceil_div(a.n_news, WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  NRL_REQUIRE(a.D == 300 && a.Q < 16 * NT_QB && a.Q % 4 == 0 && a.L >= 1 && a.L <= 32, "fused news tail backward: unsupported geometry");
  NRL_REQUIRE((((uintptr_t)a.d_out | (uintptr_t)a.q_a) & 15) == 0, "fused news tail backward: 16-byte alignment");
  hipLaunchKernelGGL((news_tail_bwd_kernel<WAVES, ABL>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
