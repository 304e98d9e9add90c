// t8gpu/backend/transport.h -- how the ranks of one job move data for MeshManager::adapt / partition and for the ghost layer.
//
// The reference has no such thing: all its ranks share ONE GPU and read each other's device memory through CUDA-IPC pointers
// (t8gpu/memory/shared_device_vector.inl:15-30), so `partition()` is a kernel that pulls (mesh_manager.inl:626-643) and a ghost
// value is a pointer dereference. With one GPU per rank these three operations are messages. The MeshManager asks a
// Transport for them; RcclTransport is the product (the C-ABI's RCCL entry points); tests drive several ranks of one process
// through a loopback implementation of the same interface (tests/compat/).
#ifndef T8GPU_HIP_BACKEND_TRANSPORT_H
#define T8GPU_HIP_BACKEND_TRANSPORT_H

#include <hip/hip_runtime.h>
#include <t8gpu_hip.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace t8gpu {

  class Transport {
   public:
    virtual ~Transport() = default;
    [[nodiscard]] virtual int rank() const = 0;
    [[nodiscard]] virtual int size() const = 0;
    /// every rank's chunk mine[offsets[rank + 1] - offsets[rank]] to every rank's all[offsets[size]] (device pointers,
    /// offsets on the host); returns when `all` is complete. Collective.
    virtual void allgatherv(double const* mine, double* all, int64_t const* offsets) = 0;
    /// t8gpu_hip_repartition_*: runs of elements from the `src` planes to their new owners' `dst` planes; returns when this
    /// rank's `dst` is complete. Collective.
    virtual void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                             int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f32 src,
                             float const* src_volume, T8gpuVars_f32 dst, float* dst_volume, int cells_per_element) = 0;
    virtual void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                             int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f64 src,
                             double const* src_volume, T8gpuVars_f64 dst, double* dst_volume, int cells_per_element) = 0;
    /// refresh the ghost mirror slots [N, N + G) of the five planes of `state` (t8gpu_hip_halo_exchange_*); returns when
    /// they are current. `halo.comm` is ignored by implementations that have their own channel. Collective.
    virtual void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f32 state) = 0;
    virtual void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f64 state) = 0;
  };

  /// The product transport: one RCCL communicator per process (hip::Communicator::handle()), one rank per GPU.
  class RcclTransport final : public Transport {
   public:
    RcclTransport(void* nccl_comm, int rank, int nranks) : m_comm{nccl_comm}, m_rank{rank}, m_size{nranks} {}
    [[nodiscard]] int rank() const override { return m_rank; }
    [[nodiscard]] int size() const override { return m_size; }
    void allgatherv(double const* mine, double* all, int64_t const* offsets) override {
      check(t8gpu_hip_comm_allgatherv_f64(m_comm, m_rank, m_size, mine, all, offsets, nullptr));
      check(static_cast<int>(hipStreamSynchronize(nullptr)));
    }
    void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                     int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f32 src, float const* src_volume,
                     T8gpuVars_f32 dst, float* dst_volume, int cells) override {
      check(t8gpu_hip_repartition_f32(m_comm, m_rank, n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count, src,
                                      src_volume, dst, dst_volume, cells, nullptr));
      check(static_cast<int>(hipStreamSynchronize(nullptr)));
    }
    void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                     int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f64 src, double const* src_volume,
                     T8gpuVars_f64 dst, double* dst_volume, int cells) override {
      check(t8gpu_hip_repartition_f64(m_comm, m_rank, n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count, src,
                                      src_volume, dst, dst_volume, cells, nullptr));
      check(static_cast<int>(hipStreamSynchronize(nullptr)));
    }
    void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f32 state) override {
      T8gpuHalo h = halo;
      h.comm      = m_comm;
      check(t8gpu_hip_halo_exchange_f32(&h, state, nullptr));
      check(static_cast<int>(hipStreamSynchronize(nullptr)));
    }
    void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f64 state) override {
      T8gpuHalo h = halo;
      h.comm      = m_comm;
      check(t8gpu_hip_halo_exchange_f64(&h, state, nullptr));
      check(static_cast<int>(hipStreamSynchronize(nullptr)));
    }

   private:
    static void check(int code) {
      if (code != 0) {
        std::fprintf(stderr, "t8gpu: transport call failed: %d (%s)\n", code, t8gpu_hip_error_string(code));
        std::abort();
      }
    }
    void* m_comm;
    int   m_rank, m_size;
  };

}  // namespace t8gpu

#endif  // T8GPU_HIP_BACKEND_TRANSPORT_H
