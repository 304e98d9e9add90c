// kernels_probe.hip -- diagnostics: evaluates the fast-tier scalar helpers of flux_math.hpp element-wise so
// that tests can pin their accuracy against the host libm (tests/test_gpu_fastmath.py). Not on the hot path.
#include <hip/hip_runtime.h>

#include "flux_math.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
__global__ void k_math_probe(int op, int n, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T x = a[i], y = b ? b[i] : T(1);
  T r;
  switch (op) {
    case T8GPU_PROBE_RCP: r = t8_rcp(x); break;
    case T8GPU_PROBE_DIV: r = t8_div(x, y); break;
    case T8GPU_PROBE_SQRT: r = t8_sqrt_fast(x); break;
    case T8GPU_PROBE_LOG: r = t8_log_fast(x); break;
    case T8GPU_PROBE_LOG_TAB: r = t8_log_tab(x, kLogTab); break;   // (the tile kernels read an LDS copy of the table)
    case T8GPU_PROBE_SQRT_RATIO: r = t8_sqrt_ratio(x, y); break;
    case T8GPU_PROBE_DIV_SHARED: r = t8_div_by(x, y, t8_rcp_shared(y)); break;
    case T8GPU_PROBE_LN_MEAN: r = ln_mean_dlog(x, y, t8_log_fast(y) - t8_log_fast(x)); break;
    case T8GPU_PROBE_LN_MEAN_REF: r = ln_mean_ref<T>(x, y); break;
    default: r = T(0); break;
  }
  out[i] = r;
}

template <class T>
int math_probe(int op, int n, const T* a, const T* b, T* out, void* stream) {
  if (op < 0 || op > T8GPU_PROBE_DIV_SHARED || n < 0 || !a || !out) return static_cast<int>(hipErrorInvalidValue);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_math_probe<T>, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), op, n, a, b, out);
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_math_probe_f32(int op, int n, const float* a, const float* b, float* out, void* stream) {
  return t8gpu_hip::math_probe<float>(op, n, a, b, out, stream);
}
int t8gpu_hip_math_probe_f64(int op, int n, const double* a, const double* b, double* out, void* stream) {
  return t8gpu_hip::math_probe<double>(op, n, a, b, out, stream);
}
}
