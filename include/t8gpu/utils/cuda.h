// t8gpu/utils/cuda.h (MI355X backend) -- error-check macros of the reference (t8gpu/utils/cuda.h:7-33)
// on the HIP runtime. Same macro names so that user code keeps compiling; same policy: print
// file:line + the runtime's message, then abort the process (SC_ABORT when libsc is present).
#ifndef T8GPU_HIP_UTILS_CUDA_H
#define T8GPU_HIP_UTILS_CUDA_H

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <iostream>

#if defined(SC_ABORT)
#define T8GPU_ABORT(msg) SC_ABORT(msg)
#else
#define T8GPU_ABORT(msg)            \
  do {                              \
    std::cerr << (msg) << std::endl; \
    std::abort();                   \
  } while (0)
#endif

#define T8GPU_CUDA_CHECK_ERROR(expr)                                                                         \
  do {                                                                                                       \
    const hipError_t t8gpu_status_ = (expr);                                                                 \
    if (t8gpu_status_ != hipSuccess) {                                                                       \
      std::cerr << "caught HIP runtime error at: " << __FILE__ << ":" << __LINE__ << "\n"                    \
                << hipGetErrorString(t8gpu_status_) << std::endl;                                            \
      T8GPU_ABORT("HIP error caught");                                                                       \
    }                                                                                                        \
  } while (0)

// return codes of the C-ABI (include/t8gpu_hip.h): 0 ok, hipError_t, or 10000 + ncclResult_t
#define T8GPU_HIP_CHECK_ABI(expr)                                                                            \
  do {                                                                                                       \
    const int t8gpu_code_ = (expr);                                                                          \
    if (t8gpu_code_ != 0) {                                                                                  \
      std::cerr << "t8gpu_hip call failed at: " << __FILE__ << ":" << __LINE__ << " code " << t8gpu_code_    \
                << std::endl;                                                                                \
      T8GPU_ABORT("t8gpu_hip error caught");                                                                 \
    }                                                                                                        \
  } while (0)

// Debug builds serialise after every launch to surface asynchronous faults (reference cuda.h:20-30).
#ifndef NDEBUG
#define T8GPU_CUDA_CHECK_LAST_ERROR()                 \
  do {                                                \
    T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());   \
    T8GPU_CUDA_CHECK_ERROR(hipGetLastError());        \
  } while (0)
#else
#define T8GPU_CUDA_CHECK_LAST_ERROR()
#endif

#endif  // T8GPU_HIP_UTILS_CUDA_H
