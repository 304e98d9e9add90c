// kernels_reduce.hip -- the two scalar reductions that sit right after the hot path.
//
//   max wave speed : compute_timestep() reduces the per-face speed estimates with thrust::reduce(max)
//                    and MPI_Allreduce (examples/compressible_euler/solver.cu:213-229);
//   integral       : compute_integral() copies a variable and the volumes to the host and sums
//                    vol * u there (solver.cu:190-211; subgrid: examples/subgrid/solver.inl:281-305).
// Both become a two-level device reduction (grid-stride partials per workgroup -> one workgroup) that
// writes a device scalar, stays on the stream and is bitwise reproducible (fixed tree, no atomics).
// The integral accumulates in double whatever float_type is.
#include <hip/hip_runtime.h>

#include "t8gpu_hip.h"

namespace t8gpu_hip {

constexpr int kReduceBlocks = 1024;

struct OpMax {
  static __device__ double identity() { return 0.0; }  // speeds are >= 0; the reference starts from 0 too
  static __device__ double apply(double a, double b) { return a > b ? a : b; }
};
struct OpSum {
  static __device__ double identity() { return 0.0; }
  static __device__ double apply(double a, double b) { return a + b; }
};

template <class Op>
__device__ double block_reduce(double v) {
  __shared__ double part[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = Op::apply(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = Op::identity();
  if (threadIdx.x == 0) r = Op::apply(Op::apply(part[0], part[1]), Op::apply(part[2], part[3]));
  __syncthreads();
  return r;  // valid in thread 0
}

template <class T>
__global__ __launch_bounds__(256) void k_max_partial(size_t n, const T* __restrict__ x, double* __restrict__ partial) {
  double v = OpMax::identity();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v = OpMax::apply(v, (double)x[i]);
  v = block_reduce<OpMax>(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

template <class T>
__global__ __launch_bounds__(256) void k_integral_partial(size_t ncells, int cells_per_element, const T* __restrict__ u,
                                                          const T* __restrict__ volume, double* __restrict__ partial) {
  double       v   = 0.0;
  const double inv = 1.0 / cells_per_element;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ncells; i += (size_t)gridDim.x * 256)
    v += (double)volume[i / cells_per_element] * inv * (double)u[i];
  v = block_reduce<OpSum>(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

template <class Op>
__global__ __launch_bounds__(256) void k_final(int nparts, const double* __restrict__ partial, double* __restrict__ result) {
  double v = Op::identity();
  for (int i = threadIdx.x; i < nparts; i += 256) v = Op::apply(v, partial[i]);
  v = block_reduce<Op>(v);
  if (threadIdx.x == 0) *result = v;
}

inline int blocks_for(size_t n) {
  size_t b = (n + 255) / 256;
  return static_cast<int>(b < 1 ? 1 : (b > kReduceBlocks ? kReduceBlocks : b));
}

template <class T>
int max_speed(size_t n, const T* speed, void* workspace, double* result, void* stream) {
  if (!workspace || !result) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t s  = static_cast<hipStream_t>(stream);
  const int   nb = blocks_for(n);
  hipLaunchKernelGGL((k_max_partial<T>), dim3(nb), dim3(256), 0, s, n, speed, static_cast<double*>(workspace));
  hipLaunchKernelGGL((k_final<OpMax>), dim3(1), dim3(256), 0, s, nb, static_cast<const double*>(workspace), result);
  return static_cast<int>(hipGetLastError());
}

template <class T>
int integral(size_t ncells, int cpe, const T* u, const T* volume, void* workspace, double* result, void* stream) {
  if (!workspace || !result || cpe < 1) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t s  = static_cast<hipStream_t>(stream);
  const int   nb = blocks_for(ncells);
  hipLaunchKernelGGL((k_integral_partial<T>), dim3(nb), dim3(256), 0, s, ncells, cpe, u, volume, static_cast<double*>(workspace));
  hipLaunchKernelGGL((k_final<OpSum>), dim3(1), dim3(256), 0, s, nb, static_cast<const double*>(workspace), result);
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
size_t t8gpu_hip_reduce_workspace_bytes(void) { return sizeof(double) * t8gpu_hip::kReduceBlocks; }
int t8gpu_hip_max_speed_f32(size_t n, const float* speed, void* workspace, double* result, void* stream) {
  return t8gpu_hip::max_speed<float>(n, speed, workspace, result, stream);
}
int t8gpu_hip_max_speed_f64(size_t n, const double* speed, void* workspace, double* result, void* stream) {
  return t8gpu_hip::max_speed<double>(n, speed, workspace, result, stream);
}
int t8gpu_hip_integral_f32(size_t num_cells, int cells_per_element, const float* variable, const float* volume,
                           void* workspace, double* result, void* stream) {
  return t8gpu_hip::integral<float>(num_cells, cells_per_element, variable, volume, workspace, result, stream);
}
int t8gpu_hip_integral_f64(size_t num_cells, int cells_per_element, const double* variable, const double* volume,
                           void* workspace, double* result, void* stream) {
  return t8gpu_hip::integral<double>(num_cells, cells_per_element, variable, volume, workspace, result, stream);
}
}
