// kernels_readback.hip -- SURVEY 8f-4: what the reference does between the device arrays and its VTK
// writer, kept on the device so that one D2H copy of ready-to-write doubles is all the host sees:
//   column_major_to_z_order<Subgrid>   t8gpu/mesh/subgrid_mesh_manager.inl:1008-1049
//   get_host_scalar_variable           mesh_manager.inl:516-560, subgrid_mesh_manager.inl:1140-1160 (cast to double)
//   get_host_vector_variable           same files (three planes -> interleaved xyz doubles)
#include <hip/hip_runtime.h>

#include "t8gpu_hip.h"

namespace t8gpu_hip {

// cell (i, j, k) of a 4^rank block -> position in the z-order of the block refined uniformly twice
__device__ __forceinline__ int morton_of_flat(int flat, int rank) {
  int m = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (a < rank) {
      const int c = (flat >> (2 * a)) & 3;
      m |= ((c & 1) << a) | (((c >> 1) & 1) << (rank + a));
    }
  }
  return m;
}

template <class T>
__global__ void k_column_major_to_z_order(int rank, size_t n_cells, const T* __restrict__ from, T* __restrict__ to) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells) return;
  const int    S = rank == 3 ? 64 : 16;
  const size_t e = i / S;
  to[e * S + morton_of_flat((int)(i % S), rank)] = from[i];
}

template <class T>
__global__ void k_scalar_to_f64(size_t n, const T* __restrict__ v, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = static_cast<double>(v[i]);
}

template <class T>
__global__ void k_vector_to_f64(size_t n, const T* __restrict__ v0, const T* __restrict__ v1, const T* __restrict__ v2,
                                double* __restrict__ out) {
  // one lane per output double keeps the store coalesced; the three reads are strided by 3 lanes
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n) return;
  const size_t e = i / 3;
  const int    j = (int)(i % 3);
  out[i] = static_cast<double>(j == 0 ? v0[e] : (j == 1 ? v1[e] : v2[e]));
}

template <class T>
int z_order(int rank, int num_elements, const T* from, T* to, void* stream) {
  if ((rank != 2 && rank != 3) || num_elements < 0 || (num_elements > 0 && (!from || !to || from == to)))
    return static_cast<int>(hipErrorInvalidValue);
  if (num_elements == 0) return 0;
  const size_t n = (size_t)num_elements * (rank == 3 ? 64 : 16);
  hipLaunchKernelGGL(k_column_major_to_z_order<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rank, n,
                     from, to);
  return static_cast<int>(hipGetLastError());
}
template <class T>
int scalar_var(size_t n, const T* v, double* out, void* stream) {
  if (n > 0 && (!v || !out)) return static_cast<int>(hipErrorInvalidValue);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_scalar_to_f64<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, v, out);
  return static_cast<int>(hipGetLastError());
}
template <class T>
int vector_var(size_t n, const T* v0, const T* v1, const T* v2, double* out, void* stream) {
  if (n > 0 && (!v0 || !v1 || !v2 || !out)) return static_cast<int>(hipErrorInvalidValue);
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_vector_to_f64<T>, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, v0, v1, v2,
                     out);
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_column_major_to_z_order_f32(int rank, int num_elements, const float* from, float* to, void* stream) {
  return t8gpu_hip::z_order<float>(rank, num_elements, from, to, stream);
}
int t8gpu_hip_column_major_to_z_order_f64(int rank, int num_elements, const double* from, double* to, void* stream) {
  return t8gpu_hip::z_order<double>(rank, num_elements, from, to, stream);
}
int t8gpu_hip_host_scalar_variable_f32(size_t n, const float* variable, double* out, void* stream) {
  return t8gpu_hip::scalar_var<float>(n, variable, out, stream);
}
int t8gpu_hip_host_scalar_variable_f64(size_t n, const double* variable, double* out, void* stream) {
  return t8gpu_hip::scalar_var<double>(n, variable, out, stream);
}
int t8gpu_hip_host_vector_variable_f32(size_t n, const float* v0, const float* v1, const float* v2, double* out, void* stream) {
  return t8gpu_hip::vector_var<float>(n, v0, v1, v2, out, stream);
}
int t8gpu_hip_host_vector_variable_f64(size_t n, const double* v0, const double* v1, const double* v2, double* out, void* stream) {
  return t8gpu_hip::vector_var<double>(n, v0, v1, v2, out, stream);
}
}
