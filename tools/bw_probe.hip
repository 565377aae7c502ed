// Probe (not product code): how fast can a CU pull a k-contiguous fp32 operand panel out of HBM/L2 with the
// access pattern of the GEMM staging (8 lanes x 16 B per row segment), versus row alignment and segment width?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bw_probe.hip -o tools/bin/bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

// Each workgroup (512 threads) owns a panel of BM rows and walks K in steps of BKF floats; per step a thread
// loads float4 chunks: chunk c -> (row = c / (BKF/4), kc = c % (BKF/4)).  REPS workgroups read the same panel
// (like the n-tiles of a GEMM).  UNROLL independent steps are in flight per thread.
template <int BM, int BKF, int UNROLL>
__global__ void __launch_bounds__(512) panel_read(const float* __restrict__ a, int64_t ld, int K, int reps,
                                                  float* __restrict__ sink) {
  constexpr int CPR = BKF / 4, NCH = BM * CPR / 512;
  const int64_t bid = blockIdx.x;
  const int64_t xcd = bid % 8, local = bid / 8;
  const int64_t total = gridDim.x;
  const int64_t q = total / 8, rem = total % 8;
  const int64_t t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
  const int64_t m0 = (t / reps) * BM;
  float4 acc = make_float4(0, 0, 0, 0);
  const float* p[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = threadIdx.x + c * 512;
    p[c] = a + (m0 + ch / CPR) * ld + 4 * (ch % CPR);
  }
  const int steps = K / BKF;
  for (int s = 0; s < steps; s += UNROLL) {
    float4 v[UNROLL][NCH];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[u][c] = *reinterpret_cast<const float4*>(p[c] + (int64_t)(s + u) * BKF);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        acc.x += v[u][c].x; acc.y += v[u][c].y; acc.z += v[u][c].z; acc.w += v[u][c].w;
      }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int BM, int BKF, int UNROLL>
float run(const float* a, int64_t M, int64_t ld, int K, int reps, float* sink, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((panel_read<BM, BKF, UNROLL>), dim3((unsigned)(M / BM * reps)), dim3(512), 0, st, a, ld, K, reps, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  const int64_t M = 211200;
  float *a, *sink;
  CK(hipMalloc(&a, M * 1024 * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, M * 1024 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
#define T(BM, BKF, U, ld, K, reps)                                                                              \
  {                                                                                                             \
    const float ms = run<BM, BKF, U>(a, M, ld, K, reps, sink, st);                                              \
    printf("BM=%3d seg=%3dB inflight=%d ld=%4d K=%4d reps=%d : %7.3f ms  L2->CU %6.2f TB/s  unique %6.2f TB/s\n", BM, BKF * 4, \
           U, ld, K, reps, ms, (double)M * K * 4 * reps / ms / 1e9, (double)M * K * 4 / ms / 1e9);                \
  }
  T(256, 32, 1, 900, 896, 1)
  T(256, 32, 1, 900, 896, 2)
  T(256, 32, 1, 900, 896, 6)
  T(256, 32, 2, 900, 896, 2)
  T(256, 32, 4, 900, 896, 2)
  T(256, 32, 1, 928, 896, 2)
  T(256, 32, 2, 928, 896, 2)
  T(256, 32, 4, 928, 896, 2)
  T(256, 64, 1, 900, 896, 2)
  T(256, 64, 2, 900, 896, 2)
  T(256, 64, 1, 928, 896, 2)
  T(256, 64, 2, 928, 896, 2)
  T(128, 32, 2, 900, 896, 2)
  T(128, 32, 4, 900, 896, 2)
  T(128, 64, 4, 928, 896, 2)
  T(256, 32, 2, 300, 288, 2)
  T(256, 32, 2, 320, 288, 2)
  T(256, 32, 2, 300, 288, 6)
  T(256, 32, 2, 320, 288, 6)
  return 0;
}
