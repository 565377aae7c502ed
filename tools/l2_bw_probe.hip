// Probe (not product code): the ceiling of L2 -> CU traffic.  Every workgroup walks the SAME small buffer (1, 2 or 16 MiB: L2-resident
// per XCD, resp. MALL / HBM) again and again -- 1 KiB contiguous per wave instruction (global_load_dwordx4), or LDS-DMA
// (global_load_lds_dwordx4) -- with 8 / 16 / 32 KiB in flight per wave; 4, 8 or 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_bw_probe.hip -o tools/bin/l2_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                               \
    }                                                        \
  } while (0)

template <int UNROLL, int DMA>
__global__ void __launch_bounds__(1024) walk(const unsigned char* __restrict__ buf, size_t bytes, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  // wave w of workgroup b starts at its own offset; consecutive instructions of a wave are 1 KiB apart x (all waves of the chip)
  // a CU walks the WHOLE buffer cyclically, its waves interleaved KiB by KiB (no KiB comes back before `bytes` >> L1 went by: every
  // access misses L1), every CU from another phase; all CUs of an XCD share the buffer (L2-resident when it is <= 2 MiB)
  const size_t stride = (size_t)nw * 1024;
  size_t off = (((size_t)blockIdx.x * 37 * nw + wave) * 1024) & (bytes - 1);
  float4 acc = make_float4(0, 0, 0, 0);
  const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (UNROLL * 1024);
  for (int it = 0; it < iters; ++it) {
    if constexpr (DMA) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const unsigned char* p = buf + off + lane * 16;
        asm volatile("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, off" ::"v"(p), "s"(lds + u * 1024) : "memory");
        off = (off + stride) & (bytes - 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        v[u] = *reinterpret_cast<const float4*>(buf + off + lane * 16);
        off = (off + stride) & (bytes - 1);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int UNROLL, int DMA>
static void run(const unsigned char* buf, size_t bytes, int waves, float* sink) {
  hipStream_t st = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 2000 / UNROLL * 8;
  const size_t lds = DMA ? (size_t)waves * UNROLL * 1024 : 0;
  if (lds > 160 * 1024) return;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk<UNROLL, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((walk<UNROLL, DMA>), dim3(256), dim3(waves * 64), lds, st, buf, bytes, iters, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (r && t < best) best = t;
  }
  const double total = 256.0 * waves * iters * UNROLL * 1024.0;
  printf("buffer %5.1f MiB  %s  %2d waves/CU  %2d KiB in flight per wave : %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", bytes / 1048576.0,
         DMA ? "LDS-DMA " : "to VGPRs", waves, UNROLL, best, total / best / 1e9, total / (best * 1e-3) / 256 / 2.4e9);
  fflush(stdout);
}

int main() {
  unsigned char* buf;
  float* sink;
  CK(hipMalloc(&buf, 64u << 20));
  CK(hipMemset(buf, 1, 64u << 20));
  CK(hipMalloc(&sink, 64));
  for (size_t mb : {1, 2, 16}) {
    const size_t bytes = mb << 20;
    for (int waves : {4, 8, 16}) {
      run<8, 0>(buf, bytes, waves, sink);
      run<16, 0>(buf, bytes, waves, sink);
      run<32, 0>(buf, bytes, waves, sink);
      run<8, 1>(buf, bytes, waves, sink);
      run<16, 1>(buf, bytes, waves, sink);
    }
  }
  return 0;
}
