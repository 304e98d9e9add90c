// kernels_halo.hip -- ghost-layer exchange, device side.
//
// The reference has no message exchange: every rank dereferences every other rank's device pointers
// on ONE shared GPU (CUDA IPC, t8gpu/memory/shared_device_vector.inl:15-30) and even atomicAdd's
// into them (examples/compressible_euler/kernels.cu:295-308). One-GPU-per-rank replaces that by
// ghost mirror slots [N, N+G) appended to every plane and, per RK stage, one gather kernel -> one
// grouped RCCL send/recv over xGMI -> one scatter kernel. Wire format: 5 values per element,
// element-major, so that each peer's chunk is one contiguous message.
#include <hip/hip_runtime.h>

#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
struct HVars {
  T* p[5];
};

// `cells` = values per element and variable: 1 for plain elements, Subgrid::size (16 / 64) for blocks,
// whose cells are contiguous (e * cells + c). One lane per cell; the wire format stays element-major.
template <class T>
__global__ __launch_bounds__(256) void k_halo_pack(size_t n, int cells, const int32_t* __restrict__ send_idx, HVars<T> st,
                                                   T* __restrict__ buf) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t src = (size_t)send_idx[t / cells] * cells + t % cells;
#pragma unroll
  for (int k = 0; k < 5; k++) buf[5 * t + k] = st.p[k][src];
}

template <class T>
__global__ __launch_bounds__(256) void k_halo_unpack(size_t g, size_t first_cell, const T* __restrict__ buf, HVars<T> st) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g) return;
#pragma unroll
  for (int k = 0; k < 5; k++) st.p[k][first_cell + t] = buf[5 * t + k];
}

template <class T, class V>
HVars<T> hmk(const V& v) {
  HVars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

template <class T, class V>
int halo_pack(int n_send, int cells, const int32_t* send_idx, V st, T* buf, void* stream) {
  if (n_send <= 0) return 0;
  if (cells < 1) return static_cast<int>(hipErrorInvalidValue);
  const size_t n = (size_t)n_send * cells;
  hipLaunchKernelGGL((k_halo_pack<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, cells,
                     send_idx, hmk<T>(st), buf);
  return static_cast<int>(hipGetLastError());
}

template <class T, class V>
int halo_unpack(int num_ghosts, int first_slot, int cells, const T* buf, V st, void* stream) {
  if (num_ghosts <= 0) return 0;
  if (cells < 1) return static_cast<int>(hipErrorInvalidValue);
  const size_t g = (size_t)num_ghosts * cells;
  hipLaunchKernelGGL((k_halo_unpack<T>), dim3((unsigned)((g + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                     (size_t)first_slot * cells, buf, hmk<T>(st));
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_halo_pack_f32(int n_send, int cells_per_element, const int32_t* send_idx, T8gpuVars_f32 state, float* sendbuf,
                            void* stream) {
  return t8gpu_hip::halo_pack<float>(n_send, cells_per_element, send_idx, state, sendbuf, stream);
}
int t8gpu_hip_halo_pack_f64(int n_send, int cells_per_element, const int32_t* send_idx, T8gpuVars_f64 state, double* sendbuf,
                            void* stream) {
  return t8gpu_hip::halo_pack<double>(n_send, cells_per_element, send_idx, state, sendbuf, stream);
}
int t8gpu_hip_halo_unpack_f32(int num_ghosts, int first_ghost_slot, int cells_per_element, const float* recvbuf,
                              T8gpuVars_f32 state, void* stream) {
  return t8gpu_hip::halo_unpack<float>(num_ghosts, first_ghost_slot, cells_per_element, recvbuf, state, stream);
}
int t8gpu_hip_halo_unpack_f64(int num_ghosts, int first_ghost_slot, int cells_per_element, const double* recvbuf,
                              T8gpuVars_f64 state, void* stream) {
  return t8gpu_hip::halo_unpack<double>(num_ghosts, first_ghost_slot, cells_per_element, recvbuf, state, stream);
}
}
