// Probe (not product code): issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD vs two / four, with 2 .. 16
// independent accumulators per wave (dependent MFMAs that far apart).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(1024) mfma_loop(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f / (1 + i)); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(int waves_per_simd, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;          // 4 SIMDs x waves x 64 lanes, one workgroup per CU
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(256), dim3(threads), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas_per_simd = (double)iters * NACC * waves_per_simd;
  const double tf = mfmas_per_simd * 1024 * 16384.0 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d  independent accumulators %2d : %.3f ms  %.0f TFLOP/s bf16  (%.1f ns per MFMA per SIMD)\n", waves_per_simd, NACC, ms, tf,
         ms * 1e6 / mfmas_per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  for (int w : {1, 2, 4}) {
    run<2>(w, out);
    run<4>(w, out);
    run<8>(w, out);
    run<16>(w, out);
  }
  return 0;
}
