// fused_common.hpp -- small pieces shared by the fused tile kernels (kernels_fused.hip, kernels_fused_persistent.hip).
#ifndef T8GPU_HIP_FUSED_COMMON_HPP
#define T8GPU_HIP_FUSED_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "flux_math.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

// Compute units of the current device, asked once (0: no device). Function-local statics: the launchers are called from two
// host threads of a rank (stepper.hip: the step driver's lanes), so their lazily initialised settings must be race-free.
inline int device_cu_count() {
  static const int n = [] {
    int             dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}
// a tuning variable holding workgroups per CU: 1 .. 8, anything else (or unset) = 0 = "the kernel's own default"
inline int env_per_cu(const char* name) {
  const char* env = std::getenv(name);
  const int   v   = env ? std::atoi(env) : 0;
  return (v < 0 || v > 8) ? 0 : v;
}

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

template <class T>
struct vec2;
template <>
struct vec2<float> {
  using type = float2;
};
template <>
struct vec2<double> {
  using type = double2;
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

// ---- ghost window (T8gpuPlainPlan::ghost_buf / send_map: the multi-rank step driver's zero-copy exchange) -------------
// State value `k` of slot `slot`: ghosts (slot >= n_owned) come from the exchange's receive buffer in its wire format.
template <class T>
T8_DEV T ghost_window_load(const T8gpuPlainPlan& P, const FVars<T>& src, int slot, int k) {
  const T* gb = static_cast<const T*>(P.ghost_buf);
  const T* p  = slot >= P.n_owned ? gb + (5 * static_cast<size_t>(slot - P.n_owned) + k) : src.p[k] + slot;
  return *p;
}
// The five new values of owned element e also go to its send slots (a no-op for the elements no peer mirrors).
template <class T>
T8_DEV void ghost_window_send(const T8gpuPlainPlan& P, int e, const T v[5]) {
  const int m = P.send_map[e];
  if (m == -1) return;
  T* const sb = static_cast<T*>(P.send_buf);
  if (m >= 0) {
#pragma unroll
    for (int k = 0; k < 5; k++) sb[5 * static_cast<size_t>(m) + k] = v[k];
    return;
  }
  const int32_t* l = P.send_list + (-m - 2);
  for (;;) {
    const int ent = *l++;
    const size_t t = static_cast<size_t>(ent & 0x7FFFFFFF);
#pragma unroll
    for (int k = 0; k < 5; k++) sb[5 * t + k] = v[k];
    if (ent < 0) break;
  }
}

// ---- LDS records of the persistent kernels (kernels_fused_persistent.hip, kernels_fused_patch.hip) ---------------
template <class T>
struct vec16;
template <>
struct vec16<double> {
  using type = double2;
  static constexpr int lanes = 2;
};
template <>
struct vec16<float> {
  using type = float4;
  static constexpr int lanes = 4;
};

// words per LDS record: NW payload words padded so that (a) 16-byte pieces stay aligned and (b) consecutive slots
// start in different bank groups (record size / 16 B is odd: 5 or 3)
template <class T, int NW>
constexpr int rec_words() {
  return sizeof(T) == 8 ? (NW > 5 ? 10 : 6) : 12;
}

template <class T, int NW>
T8_DEV void rec_store(T* rec, const T* w) {
  using V         = typename vec16<T>::type;
  constexpr int L = vec16<T>::lanes;
#pragma unroll
  for (int c = 0; c + L <= NW; c += L) {
    V v;
    T* vv = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < L; j++) vv[j] = w[c + j];
    *reinterpret_cast<V*>(rec + c) = v;
  }
#pragma unroll
  for (int c = NW / L * L; c < NW; c++) rec[c] = w[c];
}

template <class T, int NW>
T8_DEV void rec_load(const T* rec, T* w) {
  using V         = typename vec16<T>::type;
  constexpr int L = vec16<T>::lanes;
#pragma unroll
  for (int c = 0; c + L <= NW; c += L) {
    const V  v  = *reinterpret_cast<const V*>(rec + c);
    const T* vv = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int j = 0; j < L; j++) w[c + j] = vv[j];
  }
#pragma unroll
  for (int c = NW / L * L; c < NW; c++) w[c] = rec[c];
}

template <class T>
T8_DEV void prim_words(const T s[5], T w[kPrimWords], const double* logtab) {
#ifdef T8GPU_EXP_NOMATH    // experiment builds only: same loads, LDS traffic, barriers and stores, (almost) no arithmetic
  Prim<T> q;
  q.rho = s[0]; q.vx = s[1]; q.vy = s[2]; q.vz = s[3]; q.p = s[4]; q.beta = s[0]; q.lrho = s[1]; q.lbeta = s[2]; q.v0 = s[3];
#else
  const Prim<T> q = prim_from_state<T, sizeof(T) == 8>(s, logtab);   // fp64: table-driven logarithms (flux_math.hpp)
#endif
  w[0] = q.rho; w[1] = q.vx; w[2] = q.vy; w[3] = q.vz; w[4] = q.p; w[5] = q.beta; w[6] = q.lrho; w[7] = q.lbeta; w[8] = q.v0;
}
template <class T>
T8_DEV void words_prim(const T w[kPrimWords], Prim<T>& q) {
  q.rho = w[0]; q.vx = w[1]; q.vy = w[2]; q.vz = w[3]; q.p = w[4]; q.beta = w[5]; q.lrho = w[6]; q.lbeta = w[7]; q.v0 = w[8];
}

// persistent, software-pipelined tile kernel (kernels_fused_persistent.hip). Returns -1 when the plan is outside what
// that kernel takes (the caller then uses the one-tile-per-workgroup kernels), otherwise 0 or a hipError_t.
template <class T>
int plain_persistent_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, FVars<T> prev,
                           FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, hipStream_t stream);

// structured-patch kernel (kernels_fused_patch.hip): [patch_begin, +patch_count) of tile_order are patch tiles; the generic
// tiles [tile_begin, +tile_count) ride in the same launch where the mixed kernel takes them (-1: it does not; launch apart)
template <class T>
int plain_patch_stage(int kind, int stage, const T8gpuPlainPlan* plan, int patch_begin, int patch_count, int tile_begin, int tile_count,
                      FVars<T> prev, FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, bool persistent, hipStream_t stream);

// 3D structured patches (kernels_fused_patch3.hip): [tile_begin, +tile_count) of tile_order are 8 x 8 x 4 patch tiles, all
// regular or (irregular = true) all irregular ones
template <class T>
int plain_patch3_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, FVars<T> prev, FVars<T> mid,
                       FVars<T> out, const T* volume, T dt, T* speed, bool persistent, bool irregular, hipStream_t stream);

// regular + irregular 3D patches of one class in one persistent launch (-1: launch them one after the other)
template <class T>
int plain_patch3_both_stage(int kind, int stage, const T8gpuPlainPlan* plan, int reg_begin, int reg_count, int irr_begin, int irr_count,
                            FVars<T> prev, FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, hipStream_t stream);

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_FUSED_COMMON_HPP
