// kernels_amr.hip -- device side of SURVEY 8f-3: the AMR indicator and the data transfer of
// MeshManager::adapt / partition for plain elements.
//
//   estimate_gradient        examples/compressible_euler/kernels.cu:471-501   |rho_r - rho_l| added to both cells
//   refinement criteria      examples/compressible_euler/solver.cu:231-241    gradient / cbrt(volume)
//   adapt_variables_and_volume  t8gpu/mesh/mesh_manager.inl:165-193           injection (refine) / mean (coarsen)
//   gather_elements          the device half of partition_data (mesh_manager.inl:626-643): the reference PULLS
//                            new[k][i] = old[k][rank][index] through CUDA-IPC pointers; here the old owner
//                            gathers the elements a peer needs into one contiguous message (RCCL send/recv)
// The reference hard-codes the 3D volume factors 0.125 / 8.0 in the transfer kernel even for 2D
// forests; `dim` selects 1/2^dim and 2^dim here (dim = 3 reproduces the reference).
#include <hip/hip_runtime.h>

#include "flux_math.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
struct AVars {
  T* p[5];
};

template <class T>
__global__ __launch_bounds__(256) void k_estimate_gradient(int F, const int32_t* __restrict__ fn, const int32_t* __restrict__ idx,
                                                           const T* __restrict__ rho, T* __restrict__ gradient) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  int l = fn[2 * (size_t)i], r = fn[2 * (size_t)i + 1];
  if (idx) {
    l = idx[l];
    r = idx[r];
  }
  const T g = t8_abs(rho[r] - rho[l]);
  unsafeAtomicAdd(&gradient[l], g);
  unsafeAtomicAdd(&gradient[r], g);
}

template <class T>
__global__ __launch_bounds__(256) void k_refinement_criteria(int N, const T* __restrict__ gradient, const T* __restrict__ volume,
                                                             T* __restrict__ criteria) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  criteria[i] = gradient[i] / t8_cbrt(volume[i]);
}

template <class T>
__global__ __launch_bounds__(256) void k_adapt_transfer(int n_new, int dim, const int32_t* __restrict__ adapt_data, AVars<T> old_v,
                                                        AVars<T> new_v, const T* __restrict__ vol_old, T* __restrict__ vol_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int first = adapt_data[i];
  const int diff  = adapt_data[i + 1] - first;
  const int nsum  = diff > 1 ? diff : 1;
  const T   down  = dim == 3 ? T(0.125) : T(0.25), up = dim == 3 ? T(8.0) : T(4.0);
  T         v     = vol_old[first] * (diff == 0 ? down : (diff == 1 ? T(1.0) : up));
  if (i > 0 && adapt_data[i - 1] == first) v = vol_old[first] * down;   // last child of a refined element
  vol_new[i] = v;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    T acc = T(0.0);
    for (int j = 0; j < nsum; j++) acc += old_v.p[k][first + j] / static_cast<T>(nsum);
    new_v.p[k][i] = acc;
  }
}

// out = 6 planes of n values (5 variables + volume) for elements [first, first + n)
template <class T>
__global__ __launch_bounds__(256) void k_gather_range(int n, int first, AVars<T> v, const T* __restrict__ vol, T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 5; k++) out[(size_t)k * n + i] = v.p[k][first + i];
  out[(size_t)5 * n + i] = vol[first + i];
}

template <class T>
__global__ __launch_bounds__(256) void k_scatter_range(int n, int first, const T* __restrict__ in, AVars<T> v, T* __restrict__ vol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 5; k++) v.p[k][first + i] = in[(size_t)k * n + i];
  vol[first + i] = in[(size_t)5 * n + i];
}

// ---- Subgrid<4,4> / Subgrid<4,4,4> -----------------------------------------------------------------
// compute_refinement_criteria<Subgrid>, examples/subgrid/kernels.inl:1110-1168: discrete H1 seminorm of the
// density inside a block, divided by the block volume. One lane per block, same loop order.
template <class T, int RANK>
__global__ __launch_bounds__(256) void k_subgrid_criteria(int N, const T* __restrict__ rho, const T* __restrict__ volumes,
                                                          T* __restrict__ criteria) {
  constexpr int S = RANK == 3 ? 64 : 16;
  const int     e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const T* d = rho + (size_t)e * S;
  const T  h = (RANK == 3 ? t8_cbrt(volumes[e]) : t8_sqrt(volumes[e])) / T(4);
  T        acc = T(0.0);
  const int nz = RANK == 3 ? 4 : 1;
  for (int p = 0; p < 3; p++)
    for (int q = 0; q < 4; q++)
      for (int r = 0; r < nz; r++) {
        const T a = d[(p + 1) + 4 * q + 16 * r] - d[p + 4 * q + 16 * r];
        acc += a * a * h;
      }
  for (int p = 0; p < 4; p++)
    for (int q = 0; q < 3; q++)
      for (int r = 0; r < nz; r++) {
        const T a = d[p + 4 * (q + 1) + 16 * r] - d[p + 4 * q + 16 * r];
        acc += a * a * h;
      }
  if (RANK == 3)
    for (int p = 0; p < 4; p++)
      for (int q = 0; q < 4; q++)
        for (int r = 0; r < 3; r++) {
          const T a = d[p + 4 * q + 16 * (r + 1)] - d[p + 4 * q + 16 * r];
          acc += a * a * h;
        }
  criteria[e] = acc / volumes[e];
}

// adapt_variables<Subgrid> + adapt_volume<Subgrid>, t8gpu/mesh/subgrid_mesh_manager.inl:246-425: one lane per
// NEW subcell. Refined block: piecewise-constant injection from the parent's octant/quadrant given by the
// child's position in its family; coarsened block: mean of the 2^rank fine cells; otherwise a copy.
template <class T, int RANK>
__global__ __launch_bounds__(256) void k_subgrid_adapt(int n_new, const int32_t* __restrict__ adapt_data, AVars<T> old_v,
                                                       AVars<T> new_v, const T* __restrict__ vol_old, T* __restrict__ vol_new) {
  constexpr int S = RANK == 3 ? 64 : 16;
  const size_t  t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_new * S) return;
  const int e = (int)(t / S), c = (int)(t % S);
  const int i = c & 3, j = (c >> 2) & 3, k = RANK == 3 ? c >> 4 : 0;
  const int first = adapt_data[e], diff = adapt_data[e + 1] - first;
  const bool refined = diff == 0 || (e > 0 && adapt_data[e - 1] == first);
  if (c == 0) {
    const T down = RANK == 3 ? T(0.125) : T(0.25), up = RANK == 3 ? T(8.0) : T(4.0);
    T       v    = vol_old[first] * (diff == 0 ? down : (diff == 1 ? T(1.0) : up));
    if (e > 0 && adapt_data[e - 1] == first) v = vol_old[first] * down;
    vol_new[e] = v;
  }
  if (refined) {
    int child = 0;
    while (e - child - 1 >= 0 && adapt_data[e - child - 1] == first) child++;   // position in the family, z-order
    const int I = child & 1, J = (child >> 1) & 1, K = (child >> 2) & 1;
    const size_t src = (size_t)first * S + (2 * I + i / 2) + 4 * (2 * J + j / 2) + (RANK == 3 ? 16 * (2 * K + k / 2) : 0);
#pragma unroll
    for (int l = 0; l < 5; l++) new_v.p[l][t] = old_v.p[l][src];
  } else if (diff > 1) {
    const int    z   = (i >> 1) | ((j >> 1) << 1) | (RANK == 3 ? (k >> 1) << 2 : 0);
    const size_t blk = (size_t)(first + z) * S;
#pragma unroll
    for (int l = 0; l < 5; l++) {
      T acc = T(0.0);
      for (int ii = 0; ii < 2; ii++)
        for (int jj = 0; jj < 2; jj++)
          for (int kk = 0; kk < (RANK == 3 ? 2 : 1); kk++)
            acc += old_v.p[l][blk + (2 * (i & 1) + ii) + 4 * (2 * (j & 1) + jj) + (RANK == 3 ? 16 * (2 * (k & 1) + kk) : 0)];
      new_v.p[l][t] = acc / static_cast<T>(1 << RANK);
    }
  } else {
#pragma unroll
    for (int l = 0; l < 5; l++) new_v.p[l][t] = old_v.p[l][(size_t)first * S + c];
  }
}

template <class T, class V>
AVars<T> amk(const V& v) {
  AVars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

inline int status() { return static_cast<int>(hipGetLastError()); }
inline dim3 grid_for(int n) { return dim3((n + 255) / 256); }

}  // namespace t8gpu_hip

using namespace t8gpu_hip;

extern "C" {

#define T8_DEFINE_AMR(SUF, T, V)                                                                                        \
  int t8gpu_hip_estimate_gradient_##SUF(int F, const int32_t* fn, const int32_t* idx, const T* rho, T* gradient,         \
                                        void* stream) {                                                                 \
    if (F <= 0) return 0;                                                                                               \
    hipLaunchKernelGGL((k_estimate_gradient<T>), grid_for(F), dim3(256), 0, static_cast<hipStream_t>(stream), F, fn, idx, rho, gradient); \
    return status();                                                                                                    \
  }                                                                                                                     \
  int t8gpu_hip_refinement_criteria_##SUF(int N, const T* gradient, const T* volume, T* criteria, void* stream) {       \
    if (N <= 0) return 0;                                                                                               \
    hipLaunchKernelGGL((k_refinement_criteria<T>), grid_for(N), dim3(256), 0, static_cast<hipStream_t>(stream), N, gradient, volume, criteria); \
    return status();                                                                                                    \
  }                                                                                                                     \
  int t8gpu_hip_adapt_variables_and_volume_##SUF(int n_new, int dim, const int32_t* adapt_data, V old_v, V new_v,       \
                                                 const T* vol_old, T* vol_new, void* stream) {                          \
    if (n_new <= 0) return 0;                                                                                           \
    if (dim != 2 && dim != 3) return static_cast<int>(hipErrorInvalidValue);                                            \
    hipLaunchKernelGGL((k_adapt_transfer<T>), grid_for(n_new), dim3(256), 0, static_cast<hipStream_t>(stream), n_new, dim, adapt_data, \
                       amk<T>(old_v), amk<T>(new_v), vol_old, vol_new);                                                 \
    return status();                                                                                                    \
  }                                                                                                                     \
  int t8gpu_hip_gather_elements_##SUF(int n, int first, V vars, const T* volume, T* out, void* stream) {                \
    if (n <= 0) return 0;                                                                                               \
    hipLaunchKernelGGL((k_gather_range<T>), grid_for(n), dim3(256), 0, static_cast<hipStream_t>(stream), n, first, amk<T>(vars), volume, out); \
    return status();                                                                                                    \
  }                                                                                                                     \
  int t8gpu_hip_scatter_elements_##SUF(int n, int first, const T* in, V vars, T* volume, void* stream) {                \
    if (n <= 0) return 0;                                                                                               \
    hipLaunchKernelGGL((k_scatter_range<T>), grid_for(n), dim3(256), 0, static_cast<hipStream_t>(stream), n, first, in, amk<T>(vars), volume); \
    return status();                                                                                                    \
  }

#define T8_DEFINE_AMR_SUBGRID(SUF, T, V)                                                                                 \
  int t8gpu_hip_subgrid_refinement_criteria_##SUF(int rank, int N, const T* rho, const T* volumes, T* criteria,            \
                                                  void* stream) {                                                         \
    if (N <= 0) return 0;                                                                                                 \
    hipStream_t s = static_cast<hipStream_t>(stream);                                                                     \
    if (rank == 3)                                                                                                        \
      hipLaunchKernelGGL((k_subgrid_criteria<T, 3>), grid_for(N), dim3(256), 0, s, N, rho, volumes, criteria);            \
    else if (rank == 2)                                                                                                   \
      hipLaunchKernelGGL((k_subgrid_criteria<T, 2>), grid_for(N), dim3(256), 0, s, N, rho, volumes, criteria);            \
    else                                                                                                                  \
      return static_cast<int>(hipErrorInvalidValue);                                                                      \
    return status();                                                                                                      \
  }                                                                                                                       \
  int t8gpu_hip_subgrid_adapt_variables_and_volume_##SUF(int rank, int n_new, const int32_t* adapt_data, V old_v, V new_v, \
                                                         const T* vol_old, T* vol_new, void* stream) {                    \
    if (n_new <= 0) return 0;                                                                                             \
    hipStream_t  s = static_cast<hipStream_t>(stream);                                                                    \
    const size_t n = (size_t)n_new * (rank == 3 ? 64 : 16);                                                               \
    const dim3   g((unsigned)((n + 255) / 256));                                                                          \
    if (rank == 3)                                                                                                        \
      hipLaunchKernelGGL((k_subgrid_adapt<T, 3>), g, dim3(256), 0, s, n_new, adapt_data, amk<T>(old_v), amk<T>(new_v), vol_old, vol_new); \
    else if (rank == 2)                                                                                                   \
      hipLaunchKernelGGL((k_subgrid_adapt<T, 2>), g, dim3(256), 0, s, n_new, adapt_data, amk<T>(old_v), amk<T>(new_v), vol_old, vol_new); \
    else                                                                                                                  \
      return static_cast<int>(hipErrorInvalidValue);                                                                      \
    return status();                                                                                                      \
  }
T8_DEFINE_AMR_SUBGRID(f32, float, T8gpuVars_f32)
T8_DEFINE_AMR_SUBGRID(f64, double, T8gpuVars_f64)

T8_DEFINE_AMR(f32, float, T8gpuVars_f32)
T8_DEFINE_AMR(f64, double, T8gpuVars_f64)

}  // extern "C"
