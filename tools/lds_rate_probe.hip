// Probe (not product code): LDS read throughput per CU of ds_read_b128 vs ds_read_b64_tr_b16 (16 x 16 bf16 blocks), 4 or 8
// waves per CU, nothing but reads in the loop.   hipcc --offload-arch=gfx950 -O3 tools/lds_rate_probe.hip -o tools/bin/lds_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(512) k(int* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const int l15 = lane & 15, g = lane >> 4;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + wave * 8192;
  const unsigned a128 = base + lane * 16;
  const unsigned atr = base + (4 * g + (l15 >> 2)) * 32 + (l15 & 3) * 8;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(
          "ds_read_b128 v[100:103], %0\n ds_read_b128 v[104:107], %0 offset:1024\n ds_read_b128 v[108:111], %0 offset:2048\n"
          "ds_read_b128 v[112:115], %0 offset:3072\n ds_read_b128 v[116:119], %0 offset:4096\n ds_read_b128 v[120:123], %0 offset:5120\n"
          "ds_read_b128 v[124:127], %0 offset:6144\n ds_read_b128 v[128:131], %0 offset:7168\n s_waitcnt lgkmcnt(0)\n"
          :: "v"(a128) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115",
             "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", "memory");
    } else {
      asm volatile(
          "ds_read_b64_tr_b16 v[100:101], %0\n ds_read_b64_tr_b16 v[102:103], %0 offset:512\n ds_read_b64_tr_b16 v[104:105], %0 offset:1024\n"
          "ds_read_b64_tr_b16 v[106:107], %0 offset:1536\n ds_read_b64_tr_b16 v[108:109], %0 offset:2048\n ds_read_b64_tr_b16 v[110:111], %0 offset:2560\n"
          "ds_read_b64_tr_b16 v[112:113], %0 offset:3072\n ds_read_b64_tr_b16 v[114:115], %0 offset:3584\n s_waitcnt lgkmcnt(0)\n"
          :: "v"(atr) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", "memory");
    }
  }
  out[blockIdx.x * blockDim.x + tid] = tid;
}

template <int MODE>
static void run(const char* name, int waves, int bytes_per_read) {
  int* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double reads = (double)iters * 8 * waves;   // wave-reads per CU
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("%-22s %d waves/CU: %.1f CU-cycles per wave-read, %.0f B/clk/CU (at 2.4 GHz)\n", name, waves, cyc / reads,
         reads * bytes_per_read / cyc);
  hipFree(out);
}

int main() {
  run<0>("ds_read_b128", 4, 1024);
  run<0>("ds_read_b128", 8, 1024);
  run<1>("ds_read_b64_tr_b16", 4, 512);
  run<1>("ds_read_b64_tr_b16", 8, 512);
  return 0;
}
