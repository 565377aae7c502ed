// Token-id grouping for the embedding-table gradient (embedding_dense_backward of text.py:215-217,224): the
// positions of the flat (N * L) id vector in id-sorted order.  torch.argsort on int64 runs a 14-pass merge sort
// (0.125 ms of a 5 ms step at B = 128); ids are < vocab, so a stable LSD radix sort over ceil(log2 vocab) bits of
// a 32-bit key does it in 3 passes.  Stable => deterministic order inside a token's segment.
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "nrl_common.h"

namespace nrl {

struct IdKey {
  __host__ __device__ uint32_t operator()(const int64_t& v) const { return (uint32_t)v; }
};

static unsigned key_bits(int64_t vocab) {
  unsigned b = 1;
  while (b < 32 && ((int64_t)1 << b) < vocab) ++b;
  return b;
}

}  // namespace nrl
using namespace nrl;

extern "C" {

size_t nrl_sort_positions_workspace_bytes(int64_t n, int64_t vocab) {
  size_t temp = 0;
  auto keys_in = rocprim::make_transform_iterator((const int64_t*)nullptr, IdKey{});
  (void)rocprim::radix_sort_pairs(nullptr, temp, keys_in, (uint32_t*)nullptr, rocprim::counting_iterator<int64_t>(0),
                                  (int64_t*)nullptr, (size_t)(n > 0 ? n : 1), 0u, key_bits(vocab), (hipStream_t)0);
  return align_up(temp, 256) + align_up((size_t)(n > 0 ? n : 1) * sizeof(uint32_t), 256);
}

int nrl_sort_positions(const int64_t* ids, int64_t n, int64_t vocab, int64_t* order, void* ws, size_t ws_bytes,
                       void* stream) {
  NRL_REQUIRE(n >= 0 && vocab > 0 && vocab <= ((int64_t)1 << 32), "sort_positions: bad arguments");
  if (n == 0) return NRL_OK;
  NRL_REQUIRE(ids && order, "sort_positions: null argument");
  NRL_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  const size_t keys_bytes = align_up((size_t)n * sizeof(uint32_t), 256);
  if (ws_bytes < nrl_sort_positions_workspace_bytes(n, vocab)) {
    set_error("workspace too small: %zu bytes", ws_bytes);
    return NRL_E_WORKSPACE;
  }
  uint32_t* keys_out = (uint32_t*)ws;
  void* temp = (unsigned char*)ws + keys_bytes;
  size_t temp_bytes = ws_bytes - keys_bytes;
  auto keys_in = rocprim::make_transform_iterator(ids, IdKey{});
  NRL_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<int64_t>(0), order,
                                    (size_t)n, 0u, key_bits(vocab), (hipStream_t)stream));
  return NRL_OK;
}

}  // extern "C"
