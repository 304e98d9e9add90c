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

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

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

// std::vector whose resize() leaves new elements of a trivial type uninitialised: for the large output arrays that a parallel
// loop overwrites completely right away (a value-initialising resize is a serial pass over hundreds of MB).
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U>
  void construct(U* p) {
    ::new (static_cast<void*>(p)) U;   // default-initialisation: nothing for trivial types
  }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
template <class T>
using uvector = std::vector<T, NoInitAlloc<T>>;

// T8GPU_PLAN_VERBOSE=1: the planners print how long each of their phases took (scripts/host_cycle_time.py)
struct PhaseTimer {
  const char*                           who;
  bool                                  on;
  std::chrono::steady_clock::time_point prev;
  explicit PhaseTimer(const char* w) : who(w), on(std::getenv("T8GPU_PLAN_VERBOSE") != nullptr), prev(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[%s] %-28s %.3f s\n", who, what, std::chrono::duration<double>(now - prev).count());
    prev = now;
  }
};

#endif  // T8GPU_HOST_THREADS_HPP
