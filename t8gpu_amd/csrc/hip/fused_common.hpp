// fused_common.hpp -- small pieces shared by the fused tile kernels (kernels_fused.hip, kernels_fused_persistent.hip).
#ifndef T8GPU_HIP_FUSED_COMMON_HPP
#define T8GPU_HIP_FUSED_COMMON_HPP

#include <hip/hip_runtime.h>

#include "flux_math.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
struct FVars {
  T* p[5];
};

template <class T>
struct vec4;
template <>
struct vec4<float> {
  using type = float4;
};
template <>
struct vec4<double> {
  using type = double4;
};

// XCD-aware bijection block -> position in [0, nb): XCD x (= b % 8) owns a contiguous run.
T8_DEV int xcd_position(int b, int nb) {
  const int q = nb >> 3, rem = nb & 7, x = b & 7, k = b >> 3;
  return x * q + (x < rem ? x : rem) + k;
}

template <class T, class V>
FVars<T> fmk(const V& v) {
  FVars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

// persistent, software-pipelined tile kernel (kernels_fused_persistent.hip). Returns -1 when the plan is outside what
// that kernel takes (the caller then uses the one-tile-per-workgroup kernels), otherwise 0 or a hipError_t.
template <class T>
int plain_persistent_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, FVars<T> prev,
                           FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, hipStream_t stream);

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_FUSED_COMMON_HPP
