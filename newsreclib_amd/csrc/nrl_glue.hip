// Elementwise glue of a transformer body around this library's GEMMs (config 4, the PLM news encoder: text.py:89-109 runs a
// roberta-base-shaped encoder whose every layer ends its attention and its feed-forward block with
//     y = LayerNorm(dropout(dense_out) + residual)          (RobertaSelfOutput / RobertaOutput)
// On PyTorch-ROCm that line is a dropout kernel (reads x, writes x' and a byte mask), an add (reads two, writes one) and a layer
// norm (reads, writes, statistics): seven passes over the (tokens, 768) activation forward, and -- LayerNorm backward, masked
// scale, gradient accumulation of the residual branch -- eight backward.  Here: ONE kernel each way.
//   forward   z = x * keep * 1/(1-p) + r;  y = (z - mean) * rstd * gamma + beta;  saves z, mean, rstd        (4 passes)
//   backward  xhat = (z - mean) * rstd;  g = dy * gamma;  dz = rstd * (g - mean(g) - xhat * mean(g * xhat));
//             d_res = dz;  d_x = dz * keep * 1/(1-p);  d_gamma += sum_rows dy * xhat;  d_beta += sum_rows dy   (4 passes)
// The keep mask is the library's counter-based one (nrl_common.h: element index row * dim + col), evaluated again in the
// backward instead of stored.  One wavefront owns a row (dim <= 1024 * ... : 4 floats per lane and pass, any dim % 4 == 0 up to
// NG_MAXV * 256); HBM-bound by construction.
#include <math.h>

#include "nrl_common.h"

namespace nrl {

constexpr int NG_MAXV = 8;          // float4 slots per lane: dim <= 8 * 256 = 2048

__device__ __forceinline__ float ng_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct GlueArgs {
  const float* x;        // fwd: dense output (rows, dim);   bwd: dy
  const float* res;      // fwd: residual (rows, dim);        bwd: z (saved pre-norm sum)
  const float* gamma;
  const float* beta;
  float* z;              // fwd out (saved), or null: evaluation;   bwd: d_res (may alias nothing; written always)
  float* y;              // fwd out;                                  bwd: d_x (null when p == 0: d_x == d_res)
  float* mean;           // (rows) fwd out / bwd in
  float* rstd;           // (rows)
  float* dgamma;         // bwd: accumulated, or null (frozen)
  float* dbeta;
  int64_t rows;
  int dim;
  float eps;
  Dropout drop;
};

template <int NV>
__global__ void __launch_bounds__(256) glue_ln_fwd_kernel(const GlueArgs A) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= A.rows) return;
  const int dim = A.dim;
  const float* xr = A.x + row * dim;
  const float* rr = A.res + row * dim;
  float4 z[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dim) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + c), rv = *reinterpret_cast<const float4*>(rr + c);
      const uint32_t idx = (uint32_t)(row * dim + c);
      z[i].x = fmaf(xv.x, A.drop.mult(idx), rv.x);
      z[i].y = fmaf(xv.y, A.drop.mult(idx + 1), rv.y);
      z[i].z = fmaf(xv.z, A.drop.mult(idx + 2), rv.z);
      z[i].w = fmaf(xv.w, A.drop.mult(idx + 3), rv.w);
      sum += (z[i].x + z[i].y) + (z[i].z + z[i].w);
    }
  }
  const float inv = 1.0f / (float)dim;
  const float mean = ng_wave_sum(sum) * inv;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < dim) {
      const float a = z[i].x - mean, b = z[i].y - mean, cc = z[i].z - mean, d = z[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(ng_wave_sum(sq) * inv + A.eps);
  float* yr = A.y + row * dim;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < dim) {
      const float4 gv = *reinterpret_cast<const float4*>(A.gamma + c), bv = *reinterpret_cast<const float4*>(A.beta + c);
      float4 o;
      o.x = fmaf((z[i].x - mean) * rstd, gv.x, bv.x);
      o.y = fmaf((z[i].y - mean) * rstd, gv.y, bv.y);
      o.z = fmaf((z[i].z - mean) * rstd, gv.z, bv.z);
      o.w = fmaf((z[i].w - mean) * rstd, gv.w, bv.w);
      *reinterpret_cast<float4*>(yr + c) = o;
      if (A.z != nullptr) *reinterpret_cast<float4*>(A.z + row * dim + c) = z[i];
    }
  }
  if (A.z != nullptr && lane == 0) {
    A.mean[row] = mean;
    A.rstd[row] = rstd;
  }
}

// RPW rows per wave, one after the other: the column sums of the two parameter gradients stay in registers across them, meet
// in LDS once per workgroup (LDS float atomics into ONE 2 x dim image) and leave as one global atomic per column and workgroup.
// WAVES = 16 with parameter gradients: the global atomics serialise per address at ~100 ns, so a launch costs (workgroups x 0.1 us)
// whatever else it does -- 960 four-wave workgroups at 38400 x 768: 146 us against 60 without parameter gradients, 513 us at 3840
// workgroups (round 5, tools/glue_bwd_time.py); sixteen waves per workgroup keep the same waves in flight behind 240 atomics per address.
template <int NV, bool PARAMS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) glue_ln_bwd_kernel(const GlueArgs A, const int rpw) {
  __shared__ float s_red[PARAMS ? 2 * NV * 256 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (PARAMS) {
    for (int i = threadIdx.x; i < 2 * NV * 256; i += WAVES * 64) s_red[i] = 0.f;
    __syncthreads();
  }
  const int dim = A.dim;
  const float inv = 1.0f / (float)dim;
  float4 gsum[NV], bsum[NV], gv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    gsum[i] = bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gv[i] = c < dim ? *reinterpret_cast<const float4*>(A.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * rpw;
  // the next row's d_y / z are in flight while this row is reduced and written (a wave walks its rows one after the other:
  // without the prefetch every row paid a full memory latency -- 146 us against 42 for the one-row-per-wave form at 38400 x 768)
  float4 ndy[NV], nz[NV];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](int64_t row) {
    const bool on = row < A.rows;
    const int64_t r = on ? row : A.rows - 1;
    const float* dyr = A.x + r * dim;
    const float* zr = A.res + r * dim;
    nmean = A.mean[r]; nrstd = A.rstd[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      ndy[i] = nz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < dim) {
        ndy[i] = *reinterpret_cast<const float4*>(dyr + c);
        nz[i] = *reinterpret_cast<const float4*>(zr + c);
      }
    }
  };
  if (row0 < A.rows) fetch(row0);
  for (int q = 0; q < rpw; ++q) {
    const int64_t row = row0 + q;
    if (row >= A.rows) break;
    float4 cdy[NV], cz[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { cdy[i] = ndy[i]; cz[i] = nz[i]; }
    const float mean = nmean, rstd = nrstd;
    if (q + 1 < rpw) fetch(row + 1);
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      g[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < dim) {
        const float4 dy = cdy[i], zv = cz[i];
        xh[i] = make_float4((zv.x - mean) * rstd, (zv.y - mean) * rstd, (zv.z - mean) * rstd, (zv.w - mean) * rstd);
        g[i] = make_float4(dy.x * gv[i].x, dy.y * gv[i].y, dy.z * gv[i].z, dy.w * gv[i].w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        if constexpr (PARAMS) {
          gsum[i].x = fmaf(dy.x, xh[i].x, gsum[i].x); gsum[i].y = fmaf(dy.y, xh[i].y, gsum[i].y);
          gsum[i].z = fmaf(dy.z, xh[i].z, gsum[i].z); gsum[i].w = fmaf(dy.w, xh[i].w, gsum[i].w);
          bsum[i].x += dy.x; bsum[i].y += dy.y; bsum[i].z += dy.z; bsum[i].w += dy.w;
        }
      }
    }
    const float c1 = ng_wave_sum(s1) * inv, c2 = ng_wave_sum(s2) * inv;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < dim) {
        float4 dz;
        dz.x = rstd * (g[i].x - c1 - xh[i].x * c2);
        dz.y = rstd * (g[i].y - c1 - xh[i].y * c2);
        dz.z = rstd * (g[i].z - c1 - xh[i].z * c2);
        dz.w = rstd * (g[i].w - c1 - xh[i].w * c2);
        *reinterpret_cast<float4*>(A.z + row * dim + c) = dz;
        if (A.y != nullptr) {
          const uint32_t idx = (uint32_t)(row * dim + c);
          float4 dx;
          dx.x = dz.x * A.drop.mult(idx);
          dx.y = dz.y * A.drop.mult(idx + 1);
          dx.z = dz.z * A.drop.mult(idx + 2);
          dx.w = dz.w * A.drop.mult(idx + 3);
          *reinterpret_cast<float4*>(A.y + row * dim + c) = dx;
        }
      }
    }
  }
  if constexpr (PARAMS) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < dim) {
        atomicAdd(s_red + c, gsum[i].x); atomicAdd(s_red + c + 1, gsum[i].y); atomicAdd(s_red + c + 2, gsum[i].z); atomicAdd(s_red + c + 3, gsum[i].w);
        float* sb = s_red + NV * 256 + c;
        atomicAdd(sb, bsum[i].x); atomicAdd(sb + 1, bsum[i].y); atomicAdd(sb + 2, bsum[i].z); atomicAdd(sb + 3, bsum[i].w);
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += WAVES * 64) {
      atomicAdd(A.dgamma + c, s_red[c]);
      atomicAdd(A.dbeta + c, s_red[NV * 256 + c]);
    }
  }
}

static int glue_check(const GlueArgs& a) {
  NRL_REQUIRE(a.rows >= 0 && a.dim > 0 && a.dim % 4 == 0 && a.dim <= NG_MAXV * 256, "dropout_add_layernorm: dim must be a multiple of 4, <= %d", NG_MAXV * 256);
  NRL_REQUIRE(a.rows * a.dim < (1LL << 32), "dropout_add_layernorm: activation too large for the 32-bit dropout index space");
  return NRL_OK;
}

template <int NV>
static int glue_fwd_nv(const GlueArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(glue_ln_fwd_kernel<NV>, dim3((unsigned)((a.rows + 3) / 4)), dim3(256), 0, st, a);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}
template <int NV>
static int glue_bwd_nv(const GlueArgs& a, hipStream_t st) {
  if (a.dgamma == nullptr) {
    const int64_t blocks = (a.rows + 3) / 4;
    hipLaunchKernelGGL((glue_ln_bwd_kernel<NV, false, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, 1);
    NRL_LAUNCH_CHECK();
    return NRL_OK;
  }
  // with parameter gradients: ~256 workgroups (one global atomic per column and workgroup), WAVES x rpw rows each
  constexpr int WAVES = NV <= 3 ? 16 : (NV == 4 ? 8 : 4);       // (registers: 16 waves of a CU share 512 per lane and SIMD)
  static const int forced = [] { const char* e = getenv("NRL_GLUE_RPW"); return e ? atoi(e) : 0; }();      // (A/B runs)
  int rpw = forced > 0 ? forced : (int)((a.rows + 256 * WAVES - 1) / (256 * WAVES));
  rpw = rpw < 1 ? 1 : (rpw > 256 ? 256 : rpw);
  const int64_t blocks = (a.rows + WAVES * rpw - 1) / (WAVES * rpw);
  hipLaunchKernelGGL((glue_ln_bwd_kernel<NV, true, WAVES>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a, rpw);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // namespace nrl
using namespace nrl;

extern "C" {

int nrl_dropout_add_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, int64_t rows,
                                  int32_t dim, float eps, double p_drop, uint64_t seed, uint32_t stream0, float* z_save,
                                  float* mean_save, float* rstd_save, float* y, void* stream) {
  NRL_REQUIRE(x && residual && gamma && beta && y, "dropout_add_layernorm_fwd: null argument");
  NRL_REQUIRE((z_save == nullptr) == (mean_save == nullptr) && (z_save == nullptr) == (rstd_save == nullptr),
              "dropout_add_layernorm_fwd: save z, mean and rstd, or none of them");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  NRL_REQUIRE((((uintptr_t)x | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)z_save) & 15) == 0,
              "dropout_add_layernorm_fwd: 16-byte alignment");
  GlueArgs a{};
  a.x = x; a.res = residual; a.gamma = gamma; a.beta = beta; a.z = z_save; a.y = y; a.mean = mean_save; a.rstd = rstd_save;
  a.rows = rows; a.dim = dim; a.eps = eps; a.drop = make_dropout(p_drop, seed, stream0);
  NRL_TRY(glue_check(a));
  if (rows == 0) return NRL_OK;
  const int nv = (dim + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  switch (nv) {
    case 1: return glue_fwd_nv<1>(a, st);
    case 2: return glue_fwd_nv<2>(a, st);
    case 3: return glue_fwd_nv<3>(a, st);
    case 4: return glue_fwd_nv<4>(a, st);
    default: return glue_fwd_nv<NG_MAXV>(a, st);
  }
}

int nrl_dropout_add_layernorm_bwd(const float* d_y, const float* z_saved, const float* gamma, const float* mean_saved,
                                  const float* rstd_saved, int64_t rows, int32_t dim, double p_drop, uint64_t seed,
                                  uint32_t stream0, float* d_x, float* d_residual, float* d_gamma, float* d_beta, void* stream) {
  NRL_REQUIRE(d_y && z_saved && gamma && mean_saved && rstd_saved && d_residual, "dropout_add_layernorm_bwd: null argument");
  NRL_REQUIRE((d_gamma == nullptr) == (d_beta == nullptr), "dropout_add_layernorm_bwd: both parameter gradients or neither");
  NRL_REQUIRE(p_drop >= 0.0 && p_drop < 1.0, "dropout probability must be in [0, 1)");
  NRL_REQUIRE(p_drop == 0.0 || d_x != nullptr, "dropout_add_layernorm_bwd: d_x is needed when p_drop > 0");
  NRL_REQUIRE((((uintptr_t)d_y | (uintptr_t)z_saved | (uintptr_t)gamma | (uintptr_t)d_x | (uintptr_t)d_residual) & 15) == 0,
              "dropout_add_layernorm_bwd: 16-byte alignment");
  GlueArgs a{};
  a.x = d_y; a.res = z_saved; a.gamma = gamma; a.z = d_residual; a.y = d_x; a.mean = const_cast<float*>(mean_saved);
  a.rstd = const_cast<float*>(rstd_saved); a.dgamma = d_gamma; a.dbeta = d_beta; a.rows = rows; a.dim = dim;
  a.drop = make_dropout(p_drop, seed, stream0);
  NRL_TRY(glue_check(a));
  if (rows == 0) return NRL_OK;
  const int nv = (dim + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  switch (nv) {
    case 1: return glue_bwd_nv<1>(a, st);
    case 2: return glue_bwd_nv<2>(a, st);
    case 3: return glue_bwd_nv<3>(a, st);
    case 4: return glue_bwd_nv<4>(a, st);
    default: return glue_bwd_nv<NG_MAXV>(a, st);
  }
}

}  // extern "C"
