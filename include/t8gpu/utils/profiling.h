// t8gpu/utils/profiling.h (MI355X backend) -- wall-clock helpers with the reference's macro names
// (t8gpu/utils/profiling.h:7-36), on std::chrono::steady_clock.
#ifndef T8GPU_HIP_UTILS_PROFILING_H
#define T8GPU_HIP_UTILS_PROFILING_H

#include <chrono>
#include <cstdio>

#define T8GPU_TIME(expr)                                                                                      \
  do {                                                                                                        \
    const auto t8gpu_t0_ = std::chrono::steady_clock::now();                                                  \
    (expr);                                                                                                   \
    const double t8gpu_dt_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t8gpu_t0_).count(); \
    std::fprintf(stderr, "%20.20s:%5d       %-40.40s %.5e sec \n", __FUNCTION__, __LINE__, #expr, t8gpu_dt_);   \
  } while (0)

#define T8GPU_TIMER_START(name)            \
  const int  t8gpu_line_##name = __LINE__; \
  const auto t8gpu_start_##name = std::chrono::steady_clock::now()

#define T8GPU_TIMER_STOP(name)                                                                                   \
  do {                                                                                                           \
    const double t8gpu_dt_ =                                                                                     \
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t8gpu_start_##name).count();            \
    std::fprintf(stderr, "%20.20s:%5d-%-5d %-40.40s %.5e sec \n", __FUNCTION__, t8gpu_line_##name, __LINE__, #name, \
                 t8gpu_dt_);                                                                                     \
  } while (0)

#endif  // T8GPU_HIP_UTILS_PROFILING_H
