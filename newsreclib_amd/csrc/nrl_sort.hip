// Token-id grouping for the embedding-table gradient (embedding_dense_backward of text.py:215-217,224): the
// positions of the flat (N * L) id vector in ascending id order.
//
// torch.argsort on int64 runs a 14-pass merge sort, and so does rocPRIM's radix_sort_pairs at this size (it
// dispatches to its merge sort below ~1 M keys: 40 launches, 0.23 ms of a 5 ms step at B = 128,
// profiles/r02_x3_kernel_stats.csv).  The ids are < vocab, so this is a COUNTING sort in three launches:
//   1. per workgroup, equal ids are merged in an LDS hash table (a Zipf-head token fills 13 % of a batch: one
//      global atomic per (workgroup, id) instead of one per occurrence); the atomic's return value is the
//      workgroup's base inside the id's run, the LDS counter the position inside the workgroup;
//   2. exclusive scan of the vocab-sized histogram inside 1024-entry blocks (+ block totals);
//   3. order[prefix of the block totals + start[id] + rank] = position.
// Ids ascend; the order INSIDE an id's run follows the arrival order of the workgroups' atomics (the consumer
// reduces a run with fp32 adds and atomics, whose order is not fixed either).
#include "nrl_common.h"

namespace nrl {

// 2048 positions per workgroup: the RETURNING global atomics of one hot id (the padding id, a Zipf head) are
// serialised by the L2 at ~100 ns each, so their count -- one per (workgroup, id) -- is what the kernel's time follows
// (256 positions per workgroup: 825 workgroups, 82 us at B = 128; 2048: 104 workgroups)
constexpr int CS_THREADS = 256, CS_ITEMS = 8, CS_BLOCK = CS_THREADS * CS_ITEMS, CS_SLOTS = 2 * CS_BLOCK;

// ids outside [0, vocab) cannot index the histogram (and -1 is the hash table's empty-slot sentinel): they are grouped
// with the nearest valid id, consistently in both passes, so the result stays a permutation of the positions and every
// write stays in bounds.  (The embedding gather of such an id is the caller's error; the sort must not add memory
// corruption to it.)
__device__ __forceinline__ int cs_clamp_id(int64_t id, int vocab) {
  return id < 0 ? 0 : (id >= vocab ? vocab - 1 : (int)id);
}

__global__ void __launch_bounds__(CS_THREADS) cs_rank_kernel(const int64_t* __restrict__ ids, int64_t n, int vocab,
                                                            int* __restrict__ hist, int* __restrict__ rank) {
  __shared__ int keys[CS_SLOTS], cnt[CS_SLOTS], base[CS_SLOTS];
  const int tid = threadIdx.x;
  for (int s = tid; s < CS_SLOTS; s += CS_THREADS) {
    keys[s] = -1;
    cnt[s] = 0;
  }
  __syncthreads();
  const int64_t p0 = (int64_t)blockIdx.x * CS_BLOCK + tid;
  int slot[CS_ITEMS], lrank[CS_ITEMS], idv[CS_ITEMS];
  int64_t raw[CS_ITEMS];
#pragma unroll
  for (int q = 0; q < CS_ITEMS; ++q) {                    // all eight id loads in flight before the first hash probe
    const int64_t p = p0 + q * CS_THREADS;
    raw[q] = ids[p < n ? p : n - 1];                      // (clamped address, unconditional load: no branch, no wait per item)
  }
#pragma unroll
  for (int q = 0; q < CS_ITEMS; ++q) idv[q] = p0 + q * CS_THREADS < n ? cs_clamp_id(raw[q], vocab) : -1;
#pragma unroll
  for (int q = 0; q < CS_ITEMS; ++q) {
    slot[q] = 0;
    lrank[q] = 0;
    if (idv[q] >= 0) {
      const int id = idv[q];
      int s = (int)(((uint32_t)id * 0x9E3779B1u) >> 20) & (CS_SLOTS - 1);
      while (true) {                                      // <= CS_BLOCK distinct ids per workgroup in 2 x CS_BLOCK slots
        const int prev = atomicCAS(&keys[s], -1, id);
        if (prev == -1 || prev == id) break;
        s = (s + 1) & (CS_SLOTS - 1);
      }
      slot[q] = s;
      lrank[q] = atomicAdd(&cnt[s], 1);
    }
  }
  __syncthreads();
  // one RETURNING global atomic per occupied slot.  All 16 of a thread are issued before the first result is used: as a
  // loop of `if (occupied) base[s] = atomicAdd(..)` each LDS store waited for its own atomic, 16 serial L2 round trips
  // (~1.5 us each) that were most of this kernel's time.
  int got[CS_SLOTS / CS_THREADS];
#pragma unroll
  for (int i = 0; i < CS_SLOTS / CS_THREADS; ++i) {
    const int s = tid + i * CS_THREADS;
    const int k = keys[s];
    got[i] = 0;
    if (k != -1) got[i] = atomicAdd(&hist[k], cnt[s]);
  }
#pragma unroll
  for (int i = 0; i < CS_SLOTS / CS_THREADS; ++i) base[tid + i * CS_THREADS] = got[i];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < CS_ITEMS; ++q) {
    const int64_t p = p0 + q * CS_THREADS;
    if (p < n) rank[p] = base[slot[q]] + lrank[q];
  }
}

// hist (vocab) -> exclusive prefix sums INSIDE each 1024-entry block (in place) + the block totals
__global__ void __launch_bounds__(1024) cs_scan_kernel(int* __restrict__ hist, int vocab, int* __restrict__ totals) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 1024 + tid;
  const int c = i < vocab ? hist[i] : 0;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int wbase = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) wbase += w < wave ? wsum[w] : 0;
  if (i < vocab) hist[i] = wbase + incl - c;
  if (tid == 1023) totals[blockIdx.x] = wbase + incl;
}

// order[block prefix + start-in-block[id] + rank] = position; the prefix of the <= 1024 block totals is rebuilt per
// workgroup in LDS (a few hundred cached loads)
__global__ void __launch_bounds__(256) cs_scatter_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                         int vocab, const int* __restrict__ start,
                                                         const int* __restrict__ totals, int nblocks,
                                                         const int* __restrict__ rank, int64_t* __restrict__ order) {
  __shared__ int pre[1024];
  const int tid = threadIdx.x;
  for (int b = tid; b < 1024; b += 256) pre[b] = b < nblocks ? totals[b] : 0;
  __syncthreads();
  if (tid < 64) {                                        // one wave: exclusive scan of 1024 totals, 16 per lane
    int loc[16], sum = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      loc[q] = sum;
      sum += pre[tid * 16 + q];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (tid >= off) incl += v;
    }
    const int lbase = incl - sum;
#pragma unroll
    for (int q = 0; q < 16; ++q) pre[tid * 16 + q] = lbase + loc[q];
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + tid;
  // order[n] = number of positions holding id 0 (the padding id, whose embedding row has no gradient): the sorted order puts
  // them first, so order[n_zero .. n) are the LIVE positions -- the news-encoder backward computes dx for those only
  if (p == 0) order[n] = vocab > 1 ? start[1] : n;
  if (p < n) {
    const int id = cs_clamp_id(ids[p], vocab);
    order[pre[id >> 10] + start[id] + rank[p]] = p;
  }
}

}  // namespace nrl
using namespace nrl;

extern "C" {

size_t nrl_sort_positions_workspace_bytes(int64_t n, int64_t vocab) {
  return align_up((size_t)(vocab > 0 ? vocab : 1) * sizeof(int), 256) + align_up((size_t)(n > 0 ? n : 1) * sizeof(int), 256) +
         4096;  // + block totals
}

int nrl_sort_positions(const int64_t* ids, int64_t n, int64_t vocab, int64_t* order, void* ws, size_t ws_bytes,
                       void* stream) {
  NRL_REQUIRE(n >= 0 && n < (1LL << 31) && vocab > 0 && vocab <= (1LL << 20),
              "sort_positions: bad arguments (n < 2^31, 0 < vocab <= 2^20)");
  NRL_REQUIRE(order != nullptr, "sort_positions: null argument");
  if (n == 0) {                       // order[0] = 0 positions of id 0
    NRL_HIP(hipMemsetAsync(order, 0, sizeof(int64_t), (hipStream_t)stream));
    return NRL_OK;
  }
  NRL_REQUIRE(ids && order, "sort_positions: null argument");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  if (ws_bytes < nrl_sort_positions_workspace_bytes(n, vocab)) {
    set_error("workspace too small: %zu bytes", ws_bytes);
    return NRL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int* hist = (int*)ws;
  int* rank = (int*)((unsigned char*)ws + align_up((size_t)vocab * sizeof(int), 256));
  int* totals = (int*)((unsigned char*)rank + align_up((size_t)n * sizeof(int), 256));
  const int sblocks = (int)ceil_div(vocab, 1024);
  NRL_HIP(hipMemsetAsync(hist, 0, (size_t)vocab * sizeof(int), st));
  const unsigned blocks = (unsigned)ceil_div(n, CS_BLOCK);
  hipLaunchKernelGGL(cs_rank_kernel, dim3(blocks), dim3(CS_THREADS), 0, st, ids, n, (int)vocab, hist, rank);
  NRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(cs_scan_kernel, dim3((unsigned)sblocks), dim3(1024), 0, st, hist, (int)vocab, totals);
  NRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(cs_scatter_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, ids, n, (int)vocab, hist, totals,
                     sblocks, rank, order);
  NRL_LAUNCH_CHECK();
  return NRL_OK;
}

}  // extern "C"
