// NOT PRODUCT CODE (moved out of newsreclib_amd/csrc/nrl_news_fused.h in round 5): the recomputing token-attention backward of round 2.
// q|k|v are recomputed per head from the re-gathered rows inside the matrix-core attention backward, so the forward saves no
// q|k|v slabs (-1.9 GB of traffic per step).  Correct (gradients within 2e-5 of the oracle when it was wired in) and SLOWER:
// the 160 VGPRs of A fragments are held through the attention backward (78 spilled registers): 1.69 ms against 0.60 + 0.22 ms,
// step 5.56 vs 4.76 ms (profiles/r02_fused_bwd_ab.txt).  DESIGN.md section 8.0 (round 5) prices the half-news-per-wave variant:
// break-even needs a matrix-pipe occupancy >= 0.54, the fused forward reaches 0.34-0.49.
// Include after nrl_news_fused.h; `launch_news_fused_bwd(NewsFusedBwdArgs, stream)`.
#pragma once
#include "nrl_news_fused.h"

namespace nrl {

// =====================================================================================================
// Backward counterpart: d_o (gradient of the attention output) -> dq|dk|dv, with q|k|v RECOMPUTED per head.
//
// The forward above no longer has to save q|k|v (0.76 GB at B = 128) and the separate attention-backward kernel no
// longer reads them (attn_bwd_small: 2.4 GB of HBM traffic, 0.6 ms, the largest line of the step): a wave
// re-gathers its news' rows (same dropout draw), redoes the head's slice of the in-projection exactly as the
// forward did, and runs the whole 32 x 32 attention backward of that head on the matrix cores from its LDS image:
//   S = Q K^T, S^T = K Q^T, dP = dO V^T, dP^T = V dO^T     (both orientations: an accumulator block holds
//   P = exp(S - lse), delta = rowsum(P o dP), dS = P o (dP - delta)   4 rows x 1 column per lane, which is the A-operand
//   dV = P^T dO,  dK = dS^T (scale Q),  dQ = scale dS K             layout of the NEXT product under the slot
//                                                                   permutation kappa, so P / dS never leave registers)
// Only d_o (0.25 GB), lse and the ids are read, only dq|dk|dv (the operand of the in-projection's dgrad / wgrad
// GEMMs) is written.  Workgroup = 7 waves (LDS: 80 KB weight ring + 7 x 11.25 KB image / dO / row-vector areas).
struct NewsFusedBwdArgs {
  const float* table;
  const int64_t* ids;
  const uint16_t* img;
  int64_t n_news;
  int L, D, heads, dh;
  float scale;
  Dropout drop1;
  const float* d_o;   // (n_news * L, D)
  const float* lse;   // (n_news * heads, L) from the forward
  float* dqkv;        // (n_news * L, 3D): dq at head * dh, dk at D + .., dv at 2D + ..
};

constexpr int NFB_WAVES = 7;
constexpr int NFB_DO_FLOATS = 32 * 20;
constexpr int NFB_VEC_FLOATS = 64;                         // [0, 32): lse of the query, [32, 64): delta of the query
constexpr int NFB_WAVE_FLOATS = NF_IMG_FLOATS + NFB_DO_FLOATS + NFB_VEC_FLOATS;

template <int DH>
__global__ void __launch_bounds__(NFB_WAVES * 64, 2) news_fused_bwd_kernel(const NewsFusedBwdArgs P) {
  static_assert(DH == 20, "image packing assumes dh = 20");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NF_RING + NFB_WAVES * NFB_WAVE_FLOATS * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  float* const image = reinterpret_cast<float*>(smem + NF_RING) + wave * NFB_WAVE_FLOATS;
  float* const dO_s = image + NF_IMG_FLOATS;              // [32][20]
  float* const vec = dO_s + NFB_DO_FLOATS;                // lse | delta

  const int L = P.L, D = P.D, heads = P.heads;
  const int nblk = heads * 4;
  const int64_t news = (int64_t)blockIdx.x * NFB_WAVES + wave;
  const bool news_ok = news < P.n_news;
  const int64_t row0 = (news_ok ? news : 0) * L;

  // 16 one-KiB pieces per chunk over 7 waves: three per wave (pieces >= 16 are re-issues of piece 15)
  auto issue_chunk = [&](int h, int c) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int piece = wave * 3 + q;
      piece = piece < 16 ? piece : 15;
      const int kbi = piece >> 3, nb = (piece >> 1) & 3, plane = piece & 1;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img) +
                                 ((size_t)(((2 * c + kbi) * nblk + h * 4 + nb) * 2 + plane)) * 1024 + lane * 16;
      glds16_asm(src, smem_base + (uint32_t)c * 16384u + (uint32_t)piece * 1024u);
    }
  };
#pragma unroll
  for (int c = 0; c < 5; ++c) issue_chunk(0, c);

  // ---- gather + dropout + split (identical to the forward: same rows, same mask) ------------------------
  bf16x8 ah[2][NF_KB], al[2][NF_KB];
  {
    float4 raw[2][NF_KB][2];
    bool okr[2];
    int64_t growr[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int t = rb * 16 + l15;
      okr[rb] = news_ok && t < L;
      growr[rb] = row0 + (okr[rb] ? t : 0);
      const float* rowp = P.table + P.ids[growr[rb]] * (int64_t)D;
#pragma unroll
      for (int kb = 0; kb < NF_KB; ++kb) {
        const int k = kb * 32 + 8 * g;
        const int k0 = (kb < NF_KB - 1 || k < D) ? k : D - 4;
        const int k1 = (kb < NF_KB - 1 || k + 4 < D) ? k + 4 : D - 4;
        raw[rb][kb][0] = *reinterpret_cast<const float4*>(rowp + k0);
        raw[rb][kb][1] = *reinterpret_cast<const float4*>(rowp + k1);
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const uint32_t idx0 = (uint32_t)growr[rb] * (uint32_t)D;
      const float live = okr[rb] ? 1.0f : 0.0f;
#pragma unroll
      for (int kb = 0; kb < NF_KB; ++kb) {
        const int k = kb * 32 + 8 * g;
        float4 v0 = raw[rb][kb][0], v1 = raw[rb][kb][1];
        const bool in0 = kb < NF_KB - 1 || k < D, in1 = kb < NF_KB - 1 || k + 4 < D;
        const float m0 = in0 ? live : 0.0f, m1 = in1 ? live : 0.0f;
        const uint32_t idx = idx0 + (uint32_t)k;
        v0.x *= m0 * P.drop1.mult(idx);     v0.y *= m0 * P.drop1.mult(idx + 1);
        v0.z *= m0 * P.drop1.mult(idx + 2); v0.w *= m0 * P.drop1.mult(idx + 3);
        v1.x *= m1 * P.drop1.mult(idx + 4); v1.y *= m1 * P.drop1.mult(idx + 5);
        v1.z *= m1 * P.drop1.mult(idx + 6); v1.w *= m1 * P.drop1.mult(idx + 7);
        if (kb == NF_KB - 1) {
          if (k == D) v0.x = 1.0f;
          if (k + 4 == D) v1.x = 1.0f;
        }
        rp_split8(v0, v1, ah[rb][kb], al[rb][kb]);
      }
    }
  }

  // d_o slice (32 x 20, rows >= L zero) and lse of head `hd`: global -> registers; `put_head_inputs` parks them in LDS
  const float* const do_base = P.d_o + row0 * (int64_t)D;
  auto get_head_inputs = [&](int hd, float4 (&dv)[3], float& ls) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + ln;
      const int row = slot_i / 5, c4 = slot_i - row * 5;
      const bool ok = news_ok && slot_i < 160 && row < L;
      dv[pass] = *reinterpret_cast<const float4*>(do_base + (ok ? row * D + hd * DH + 4 * c4 : 0));
      if (!ok) dv[pass] = f4zero();
    }
    const bool lok = news_ok && ln < L;
    ls = P.lse[(news_ok ? news * heads + hd : 0) * L + (lok ? ln : 0)];
    if (!lok) ls = 1e30f;                                  // pad queries: P = exp(s - 1e30) = 0
  };
  auto put_head_inputs = [&](const float4 (&dv)[3], float ls) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const int slot_i = pass * 64 + lane;
      if (slot_i < 160) *reinterpret_cast<float4*>(dO_s + 4 * slot_i) = dv[pass];
    }
    if (lane < 32) vec[lane] = ls;
  };
  {
    float4 dv[3];
    float ls;
    get_head_inputs(0, dv, ls);
    put_head_inputs(dv, ls);
  }

  for (int h = 0; h < heads; ++h) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int hn = h + 1 < heads ? h + 1 : h;
    auto read_step = [&](int t, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
      const int c = t >> 2, kbi = (t >> 1) & 1, np = t & 1;
      const unsigned char* slot = smem + c * 16384 + lane * 16;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        bh[jj] = *reinterpret_cast<const bf16x8*>(slot + ((kbi * 4 + 2 * np + jj) * 2) * 1024);
        bl[jj] = *reinterpret_cast<const bf16x8*>(slot + ((kbi * 4 + 2 * np + jj) * 2 + 1) * 1024);
      }
    };
    auto mfma_step = [&](int t, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
      const int kb = t >> 1, np = t & 1;
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][2 * np + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                pass == 1 ? al[i][kb] : ah[i][kb], pass == 0 ? bl[jj] : bh[jj], acc[i][2 * np + jj], 0, 0, 0);
    };
    bf16x8 bh0[2], bl0[2], bh1[2], bl1[2];
    read_step(0, bh0, bl0);
#pragma unroll
    for (int t = 0; t < 20; t += 2) {
      read_step(t + 1, bh1, bl1);
      mfma_step(t, bh0, bl0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      if (t + 2 < 20) read_step(t + 2, bh0, bl0);
      mfma_step(t + 1, bh1, bl1);
      if (t + 2 < 20) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      if ((t & 3) == 2) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        issue_chunk(hn, t >> 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // accumulators -> image [token][q 20 | k 20 | v 20 | 0 4]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) image[(i * 16 + 4 * g + r) * NF_IMG_LD + nb * 16 + l15] = acc[i][nb][r];

    // the NEXT head's d_o / lse start their trip now and are parked in LDS after this head's attention backward
    float4 nx_dv[3];
    float nx_ls;
    get_head_inputs(hn, nx_dv, nx_ls);

    // ---- fragment readers -----------------------------------------------------------------------------
    auto frag8 = [&](const float* src, int ld, int row, int col0, float mul, bf16x8& hi, bf16x8& lo) {
      float4 v0 = f4zero(), v1 = f4zero();                 // 8 consecutive features 8g .. 8g + 7 (>= dh: zero)
      if (g < 2) {
        v0 = *reinterpret_cast<const float4*>(src + row * ld + col0 + 8 * g);
        v1 = *reinterpret_cast<const float4*>(src + row * ld + col0 + 8 * g + 4);
      } else if (g == 2) {
        v0 = *reinterpret_cast<const float4*>(src + row * ld + col0 + 16);
      }
      v0.x *= mul; v0.y *= mul; v0.z *= mul; v0.w *= mul;
      v1.x *= mul; v1.y *= mul; v1.z *= mul; v1.w *= mul;
      rp_split8(v0, v1, hi, lo);
    };
    // lane (d = db * 16 + l15, g) <- src[kappa(g, e)][col0 + d], e = 0..7  (B operand of a product over rows)
    auto kfrag = [&](const float* src, int ld, int col0, int db, float mul, bf16x8& hi, bf16x8& lo) {
      const int d = db * 16 + l15;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rowk = (q < 4) ? 4 * g + q : 16 + 4 * g + (q - 4);
        v[q] = d < DH ? src[rowk * ld + col0 + d] * mul : 0.f;
      }
      rp_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
    };
    auto mm3 = [&](f32x4& c, const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, c, 0, 0, 0);
    };
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- scores in both orientations: s[ib][jb] = (queries x keys), sT[jb][ib] = (keys x queries) --------
    f32x4 s[2][2], sT[2][2];
    {
      bf16x8 qh[2], ql[2], kh[2], kl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        frag8(image, NF_IMG_LD, b * 16 + l15, 0, P.scale, qh[b], ql[b]);
        frag8(image, NF_IMG_LD, b * 16 + l15, DH, 1.0f, kh[b], kl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          s[a][b] = z4; sT[a][b] = z4;
          mm3(s[a][b], qh[a], ql[a], kh[b], kl[b]);
          mm3(sT[a][b], kh[a], kl[a], qh[b], ql[b]);
        }
    }
    // ---- dP = dO V^T (queries x keys), dP^T = V dO^T --------------------------------------------------------
    f32x4 dp[2][2], dpT[2][2];
    {
      bf16x8 oh[2], ol[2], vh[2], vl[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        frag8(dO_s, DH, b * 16 + l15, 0, 1.0f, oh[b], ol[b]);
        frag8(image, NF_IMG_LD, b * 16 + l15, 2 * DH, 1.0f, vh[b], vl[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          dp[a][b] = z4; dpT[a][b] = z4;
          mm3(dp[a][b], oh[a], ol[a], vh[b], vl[b]);
          mm3(dpT[a][b], vh[a], vl[a], oh[b], ol[b]);
        }
    }
    // ---- P^T, delta (per query = per column of the transposed orientation), dS^T ---------------------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float lse_c = vec[ib * 16 + l15];
      float dl = 0.f;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = jb * 16 + 4 * g + r;
          const float p = key < L ? expf(sT[jb][ib][r] - lse_c) : 0.f;
          dl += p * dpT[jb][ib][r];
          sT[jb][ib][r] = p;
        }
      dl += nf_shfl_xor(dl, 16);
      dl += nf_shfl_xor(dl, 32);
      if (g == 0) vec[32 + ib * 16 + l15] = dl;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sT[jb][ib][r] *= dpT[jb][ib][r] - dl;          // sT now holds dS^T
    }
    // ---- P, dS in the (queries x keys) orientation: rows 4g + r need lse / delta of THEIR query ------------
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float4 lse_r = *reinterpret_cast<const float4*>(vec + ib * 16 + 4 * g);
      const float4 dl_r = *reinterpret_cast<const float4*>(vec + 32 + ib * 16 + 4 * g);
      const float ls4[4] = {lse_r.x, lse_r.y, lse_r.z, lse_r.w}, dl4[4] = {dl_r.x, dl_r.y, dl_r.z, dl_r.w};
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const bool kok = jb * 16 + l15 < L;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = kok ? expf(s[ib][jb][r] - ls4[r]) : 0.f;
          s[ib][jb][r] = p;                                                          // s now holds P
          dp[ib][jb][r] = p * (dp[ib][jb][r] - dl4[r]);                              // dp now holds dS
        }
      }
    }
    // ---- dV = P^T dO: A = P with the query slots kappa-permuted (exactly what a key-column lane holds) --------
    f32x4 dv_[2][2], dk_[2][2], dq_[2][2];
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(dO_s, DH, 0, db, 1.0f, bh[db], bl[db]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(s[0][jb][0], s[0][jb][1], s[0][jb][2], s[0][jb][3]),
                  make_float4(s[1][jb][0], s[1][jb][1], s[1][jb][2], s[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv_[jb][db] = z4;
          mm3(dv_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    // dV -> image V columns (V is dead: its row fragments fed dP / dP^T above)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (db * 16 + l15 < DH) image[(jb * 16 + 4 * g + r) * NF_IMG_LD + 2 * DH + db * 16 + l15] = dv_[jb][db][r];
    // ---- dK = dS^T (scale Q): A = dS in the same key-column form; dQ = scale dS K: A = dS^T (query-column form) ---
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(image, NF_IMG_LD, 0, db, P.scale, bh[db], bl[db]);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(dp[0][jb][0], dp[0][jb][1], dp[0][jb][2], dp[0][jb][3]),
                  make_float4(dp[1][jb][0], dp[1][jb][1], dp[1][jb][2], dp[1][jb][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dk_[jb][db] = z4;
          mm3(dk_[jb][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    {
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int db = 0; db < 2; ++db) kfrag(image, NF_IMG_LD, DH, db, 1.0f, bh[db], bl[db]);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        bf16x8 a_hi, a_lo;
        rp_split8(make_float4(sT[0][ib][0], sT[0][ib][1], sT[0][ib][2], sT[0][ib][3]),
                  make_float4(sT[1][ib][0], sT[1][ib][1], sT[1][ib][2], sT[1][ib][3]), a_hi, a_lo);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dq_[ib][db] = z4;
          mm3(dq_[ib][db], a_hi, a_lo, bh[db], bl[db]);
        }
      }
    }
    // dQ (x scale) -> Q columns, dK -> K columns (every strided read of Q and K is done)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (db * 16 + l15 < DH) {
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + db * 16 + l15] = dq_[b][db][r] * P.scale;
            image[(b * 16 + 4 * g + r) * NF_IMG_LD + DH + db * 16 + l15] = dk_[b][db][r];
          }
    // ---- image [token][dq | dk | dv] -> dqkv (row, 3D) ------------------------------------------------------
    if (news_ok) {
      float* out = P.dqkv + row0 * (int64_t)(3 * D);
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int slot_i = pass * 64 + ln;
        const int row = slot_i >> 4, ch = slot_i & 15;
        if (ch < 15 && row < L) {
          const int part = ch / 5, c4 = ch - part * 5;
          *reinterpret_cast<float4*>(out + (row * 3 * D + part * D + h * DH + 4 * c4)) =
              *reinterpret_cast<const float4*>(image + row * NF_IMG_LD + part * DH + 4 * c4);
        }
      }
    }
    put_head_inputs(nx_dv, nx_ls);
  }
  wait_vmcnt<0>();
}

static inline int launch_news_fused_bwd(const NewsFusedBwdArgs& a, hipStream_t st) {
  if (a.n_news <= 0) return NRL_OK;
  const int64_t blocks = ceil_div(a.n_news, NFB_WAVES);
  NRL_REQUIRE(blocks < (1LL << 31), "news grid too large");
  hipLaunchKernelGGL(news_fused_bwd_kernel<20>, dim3((unsigned)blocks), dim3(NFB_WAVES * 64), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
