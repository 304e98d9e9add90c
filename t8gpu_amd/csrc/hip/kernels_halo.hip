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

template <class T>
__global__ __launch_bounds__(256) void k_halo_pack(int n, const int32_t* __restrict__ send_idx, HVars<T> st,
                                                   T* __restrict__ buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int e = send_idx[t];
#pragma unroll
  for (int k = 0; k < 5; k++) buf[5 * (size_t)t + k] = st.p[k][e];
}

template <class T>
__global__ __launch_bounds__(256) void k_halo_unpack(int g, int first_slot, const T* __restrict__ buf, HVars<T> st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g) return;
#pragma unroll
  for (int k = 0; k < 5; k++) st.p[k][first_slot + t] = buf[5 * (size_t)t + k];
}

template <class T, class V>
HVars<T> hmk(const V& v) {
  HVars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

template <class T, class V>
int halo_pack(int n, const int32_t* send_idx, V st, T* buf, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL((k_halo_pack<T>), dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), n, send_idx,
                     hmk<T>(st), buf);
  return static_cast<int>(hipGetLastError());
}

template <class T, class V>
int halo_unpack(int g, int first_slot, const T* buf, V st, void* stream) {
  if (g <= 0) return 0;
  hipLaunchKernelGGL((k_halo_unpack<T>), dim3((g + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                     first_slot, buf, hmk<T>(st));
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_halo_pack_f32(int n_send, const int32_t* send_idx, T8gpuVars_f32 state, float* sendbuf, void* stream) {
  return t8gpu_hip::halo_pack<float>(n_send, send_idx, state, sendbuf, stream);
}
int t8gpu_hip_halo_pack_f64(int n_send, const int32_t* send_idx, T8gpuVars_f64 state, double* sendbuf, void* stream) {
  return t8gpu_hip::halo_pack<double>(n_send, send_idx, state, sendbuf, stream);
}
int t8gpu_hip_halo_unpack_f32(int num_ghosts, int first_ghost_slot, const float* recvbuf, T8gpuVars_f32 state,
                              void* stream) {
  return t8gpu_hip::halo_unpack<float>(num_ghosts, first_ghost_slot, recvbuf, state, stream);
}
int t8gpu_hip_halo_unpack_f64(int num_ghosts, int first_ghost_slot, const double* recvbuf, T8gpuVars_f64 state,
                              void* stream) {
  return t8gpu_hip::halo_unpack<double>(num_ghosts, first_ghost_slot, recvbuf, state, stream);
}
}
