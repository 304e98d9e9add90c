// host_threads.hpp -- how many OpenMP threads the host-side planners use.
//
// A GPU node shows every hardware thread of the host to every process (256 on the MI355X boxes) while a rank is entitled to
// a fraction of them; left to the OpenMP default the tile planner and the synthetic mesh provider ran 2.5-4x SLOWER on 256
// threads than on 16 (round 3, scripts/host_cycle_time.py). The parallel regions of this library therefore carry
// num_threads(host_threads()): T8GPU_HOST_THREADS if set, else min(omp_get_max_threads(), 16). A clause, not
// omp_set_num_threads(): the setting must not leak into other users of the OpenMP runtime in the process.
#ifndef T8GPU_HOST_THREADS_HPP
#define T8GPU_HOST_THREADS_HPP

#include <omp.h>

#include <cstdlib>

inline int host_threads() {
  static const int n = [] {
    if (const char* e = std::getenv("T8GPU_HOST_THREADS")) {
      const int v = std::atoi(e);
      if (v > 0) return v;
    }
    const int m = omp_get_max_threads();
    return m < 16 ? (m < 1 ? 1 : m) : 16;
  }();
  return n;
}

#endif  // T8GPU_HOST_THREADS_HPP
