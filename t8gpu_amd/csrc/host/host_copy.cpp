// host_copy.cpp -- a memcpy spread over the planner's OpenMP threads.
//
// The adaptive loop uploads ~300 MB of connectivity and plan arrays per adapt cycle. From pageable memory the runtime stages
// them through its own pinned buffers at ~10 GB/s; copied by several threads into a pinned staging buffer the application
// owns (t8gpu_amd/hostmem.py: PinnedUploader) they move at the memory system's rate and the DMA engine does the rest
// asynchronously.
#include <cstddef>
#include <cstring>

#include "host_threads.hpp"

extern "C" void t8gpu_host_parallel_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t kBlock = size_t(1) << 20;
  const long       nb     = static_cast<long>((bytes + kBlock - 1) / kBlock);
  if (nb <= 1) {
    std::memcpy(dst, src, bytes);
    return;
  }
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (long b = 0; b < nb; b++) {
    const size_t a = static_cast<size_t>(b) * kBlock, n = a + kBlock <= bytes ? kBlock : bytes - a;
    std::memcpy(static_cast<char*>(dst) + a, static_cast<const char*>(src) + a, n);
  }
}
