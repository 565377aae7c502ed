// Output-stage probe: how fast can 4-wave workgroups write a 128 x 160 fp32 tile of a (M, N) row-major matrix,
// as a function of the footprint of ONE store instruction?  (M = 211200; N = 900 and 300: the NRMS shapes.)
//   pat 0: 16 rows x 64 B per instruction  (the MFMA accumulator layout after the quad transpose: what the GEMMs do)
//   pat 1: 8 rows x 128 B                   (two adjacent 16-column tiles paired)
//   pat 2: 4 rows x 256 B
//   pat 3: 1 row x 640 B on 40 of 64 lanes  (whole tile rows, e.g. after staging the tile through LDS)
//   pat 4: 16 rows x 16 B... per lane dword  (the original 4-byte accumulator stores), for reference
//   pat 5/6: the workgroup's 128 x 160 tile stored as ONE contiguous 80 KB chunk (a tile-contiguous activation layout)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT>
__global__ void __launch_bounds__(256) store_kernel(float* __restrict__ c, int64_t M, int N, int tiles_n) {
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)tile_m * 128;
  const int n0 = tile_n * 160;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 v = make_float4((float)lane, (float)wave, (float)tile_m, (float)tile_n);
  if (PAT == 5 || PAT == 6) {
    // same 80 KB per workgroup, but CONTIGUOUS in memory: pat 5 = tile-contiguous ("blocked") layout, the
    // workgroup's tile is one 81920-byte chunk; pat 6 = the same chunks visited in a scrambled workgroup order
    int64_t chunk = blockIdx.x;
    if (PAT == 6) chunk = (int64_t)((uint64_t)blockIdx.x * 2654435761ull % (uint64_t)gridDim.x);
    float4* base = reinterpret_cast<float4*>(c) + chunk * (128 * 160 / 4);
    const int64_t total4 = M * (int64_t)N / 4;
    for (int i = threadIdx.x; i < 128 * 160 / 4; i += 256)
      if (chunk * (128 * 160 / 4) + i < total4) base[i] = v;
    return;
  }
  if (PAT == 3) {
    // wave w writes rows w, w+4, ...: 40 lanes x 16 B = one 640-byte row segment per instruction
    for (int r = wave; r < 128; r += 4) {
      const int64_t row = m0 + r;
      const int col = n0 + lane * 4;
      if (lane < 40 && row < M && col + 3 < N) *reinterpret_cast<float4*>(c + row * N + col) = v;
    }
    return;
  }
  // wave (wm, wn) owns a 64 x 80 sub-tile
  const int wm = wave >> 1, wn = wave & 1;
  if (PAT == 4) {
    for (int tm = 0; tm < 4; ++tm)
      for (int tn = 0; tn < 5; ++tn)
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * 64 + tm * 16 + 4 * (lane >> 4) + r;
          const int col = n0 + wn * 80 + tn * 16 + (lane & 15);
          if (row < M && col < N) c[row * N + col] = v.x;
        }
    return;
  }
  constexpr int RB = PAT == 0 ? 64 : (PAT == 1 ? 128 : 256);   // bytes per row per instruction
  constexpr int LPR = RB / 16;                                 // lanes per row
  constexpr int RPI = 64 / LPR;                                // rows per instruction
  // 64 x 80 floats = 64 rows x 320 B: instructions walk the sub-tile in (RPI rows) x (RB bytes) pieces
  for (int rb = 0; rb < 64; rb += RPI)
    for (int cb = 0; cb < 320; cb += RB) {
      const int64_t row = m0 + wm * 64 + rb + lane / LPR;
      const int colb = cb + (lane % LPR) * 16;
      const int col = n0 + wn * 80 + colb / 4;
      if (colb < 320 && row < M && col + 3 < N) *reinterpret_cast<float4*>(c + row * N + col) = v;
    }
}

template <int PAT>
float run(float* c, int64_t M, int N, hipStream_t st) {
  const int tiles_n = (N + 159) / 160;
  const int64_t tiles_m = (M + 127) / 128;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(store_kernel<PAT>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st, c, M, N, tiles_n);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const int64_t M = 211200;
  float* c;
  CK(hipMalloc(&c, M * 1024 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  const char* names[7] = {"16 rows x  64 B (current)", " 8 rows x 128 B", " 4 rows x 256 B", " 1 row  x 640 B (40 lanes)",
                          "dword stores (16 x 64 B x4)", "tile-contiguous 80 KB chunks", "80 KB chunks, scrambled order"};
  const int64_t M1 = M;
  for (int N : {900, 960, 928, 320, -300}) {
    // N = -300: three (M, 300) planes back to back = one (3M, 300) matrix -- the same 760 MB as N = 900
    const int64_t M = N < 0 ? 3 * M1 : M1;
    if (N < 0) N = -N;
    const double bytes = (double)M * (N / 4 * 4) * 4;
    float t[7] = {run<0>(c, M, N, st), run<1>(c, M, N, st), run<2>(c, M, N, st), run<3>(c, M, N, st), run<4>(c, M, N, st),
                  run<5>(c, M, N, st), run<6>(c, M, N, st)};
    for (int p = 0; p < 7; ++p)
      printf("N=%d  %-28s median %7.3f ms  %7.1f GB/s\n", N, names[p], t[p], bytes / t[p] / 1e6);
  }
  return 0;
}
