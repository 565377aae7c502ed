// Shared device/host helpers for the gfx950 NRMS kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/newsreclib_amd.h"

namespace nrl {

// ---- error plumbing (thread-local message behind nrl_last_error) -----------------------------
void set_error(const char* fmt, ...);

#define NRL_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::nrl::set_error(__VA_ARGS__);      \
      return NRL_E_INVALID;               \
    }                                     \
  } while (0)

#define NRL_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      ::nrl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return NRL_E_HIP;                                                            \
    }                                                                              \
  } while (0)

#define NRL_LAUNCH_CHECK() NRL_HIP(hipGetLastError())

#define NRL_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != NRL_OK) return rc__; \
  } while (0)

// ---- dropout keep mask (normative statement in oracle/nrms_oracle.py) -------------------------
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

inline uint32_t dropout_key(uint64_t seed, uint32_t stream) {
  uint32_t s = lowbias32(stream + 0x9E3779B9u);
  s = lowbias32((uint32_t)(seed >> 32) ^ s);
  s = lowbias32((uint32_t)(seed & 0xFFFFFFFFu) ^ s);
  return s;
}

struct Dropout {
  uint32_t key;
  uint32_t thresh;  // keep iff hash >= thresh
  float scale;      // 1/(1-p); p == 0 -> thresh 0, scale 1 (every element kept)
  __device__ __forceinline__ float mult(uint32_t flat_idx) const {
    return lowbias32(flat_idx * 0x9E3779B1u + key) >= thresh ? scale : 0.0f;
  }
};

inline Dropout make_dropout(double p, uint64_t seed, uint32_t stream) {
  Dropout d;
  d.key = dropout_key(seed, stream);
  if (p <= 0.0) {
    d.thresh = 0u;
    d.scale = 1.0f;
  } else {
    d.thresh = (uint32_t)(p * 4294967296.0);
    d.scale = (float)(1.0 / (1.0 - p));
  }
  return d;
}

// ---- small device helpers ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace nrl
