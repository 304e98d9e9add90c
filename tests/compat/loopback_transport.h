// TEST CODE -- not product. A t8gpu::Transport (include/t8gpu/backend/transport.h) for SEVERAL RANKS IN ONE PROCESS on one GPU:
// every rank is a host thread with its own MeshManager; a collective call posts its arguments, meets the other ranks at a
// barrier and copies device-to-device what RCCL would move between GPUs. Lets the C++ adapt -> partition -> connectivity loop
// of a multi-rank run be checked against the single-rank run where only one GPU (and no second RCCL rank) is available.
#ifndef T8GPU_TEST_LOOPBACK_TRANSPORT_H
#define T8GPU_TEST_LOOPBACK_TRANSPORT_H

#include <t8gpu/backend/transport.h>

#include <condition_variable>
#include <mutex>
#include <vector>

namespace t8gpu_test {

  class LoopbackHub {   // what the ranks share
   public:
    explicit LoopbackHub(int nranks) : m_size{nranks}, m_posts(static_cast<size_t>(nranks)) {}
    [[nodiscard]] int size() const { return m_size; }
    void barrier() {
      std::unique_lock<std::mutex> lk(m_mutex);
      const long gen = m_generation;
      if (++m_arrived == m_size) {
        m_arrived = 0;
        m_generation++;
        m_cv.notify_all();
      } else {
        m_cv.wait(lk, [&] { return m_generation != gen; });
      }
    }
    struct Post {   // one rank's arguments of the collective in flight
      void const*         ptr[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
      int                 n_send = 0;
      int32_t const *     send_peer = nullptr, *send_first = nullptr, *send_count = nullptr;
      T8gpuHalo           halo{};
    };
    Post& post(int rank) { return m_posts[static_cast<size_t>(rank)]; }

   private:
    int                     m_size, m_arrived = 0;
    long                    m_generation = 0;
    std::mutex              m_mutex;
    std::condition_variable m_cv;
    std::vector<Post>       m_posts;
  };

  class LoopbackTransport final : public t8gpu::Transport {
   public:
    LoopbackTransport(LoopbackHub& hub, int rank) : m_hub{hub}, m_rank{rank} {}
    [[nodiscard]] int rank() const override { return m_rank; }
    [[nodiscard]] int size() const override { return m_hub.size(); }

    void allgatherv(double const* mine, double* all, int64_t const* offsets) override {
      m_hub.post(m_rank).ptr[0] = mine;
      sync_and_meet();
      for (int q = 0; q < size(); q++) {
        const size_t n = static_cast<size_t>(offsets[q + 1] - offsets[q]);
        if (n) check(hipMemcpy(all + offsets[q], m_hub.post(q).ptr[0], sizeof(double) * n, hipMemcpyDeviceToDevice));
      }
      sync_and_meet();
    }
    void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                     int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f32 src, float const* src_volume,
                     T8gpuVars_f32 dst, float* dst_volume, int cells) override {
      runs<float>(n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count, src.p, src_volume, dst.p, dst_volume, cells);
    }
    void repartition(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv,
                     int32_t const* recv_peer, int32_t const* recv_first, int32_t const* recv_count, T8gpuVars_f64 src, double const* src_volume,
                     T8gpuVars_f64 dst, double* dst_volume, int cells) override {
      runs<double>(n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count, src.p, src_volume, dst.p, dst_volume, cells);
    }
    void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f32 state) override {
      const int cells = halo.cells_per_element > 0 ? halo.cells_per_element : 1;
      code(t8gpu_hip_halo_pack_f32(halo.n_send, cells, halo.send_idx, state, static_cast<float*>(halo.sendbuf), nullptr));
      deliver<float>(halo);
      code(t8gpu_hip_halo_unpack_f32(halo.num_ghosts, halo.num_elements, cells, static_cast<float const*>(halo.recvbuf), state, nullptr));
      sync_and_meet();
    }
    void halo_exchange(T8gpuHalo const& halo, T8gpuVars_f64 state) override {
      const int cells = halo.cells_per_element > 0 ? halo.cells_per_element : 1;
      code(t8gpu_hip_halo_pack_f64(halo.n_send, cells, halo.send_idx, state, static_cast<double*>(halo.sendbuf), nullptr));
      deliver<double>(halo);
      code(t8gpu_hip_halo_unpack_f64(halo.num_ghosts, halo.num_elements, cells, static_cast<double const*>(halo.recvbuf), state, nullptr));
      sync_and_meet();
    }

   private:
    template<class T>
    void deliver(T8gpuHalo const& halo) {   // every send chunk into its peer's receive chunk
      m_hub.post(m_rank).halo = halo;
      sync_and_meet();
      for (int j = 0; j < halo.n_peers; j++) {
        T8gpuHalo const& o  = m_hub.post(halo.peers[j]).halo;
        int              jj = -1;
        for (int k = 0; k < o.n_peers; k++)
          if (o.peers[k] == m_rank) jj = k;
        const size_t n = static_cast<size_t>(halo.recv_off[j + 1] - halo.recv_off[j]);
        const size_t w = 5 * static_cast<size_t>(halo.cells_per_element > 0 ? halo.cells_per_element : 1);   // values per element on the wire
        if (jj < 0 || static_cast<size_t>(o.send_off[jj + 1] - o.send_off[jj]) != n) std::abort();
        check(hipMemcpy(static_cast<T*>(halo.recvbuf) + w * static_cast<size_t>(halo.recv_off[j]),
                        static_cast<T const*>(o.sendbuf) + w * static_cast<size_t>(o.send_off[jj]), sizeof(T) * w * n, hipMemcpyDeviceToDevice));
      }
    }
    template<class T>
    void runs(int n_send, int32_t const* send_peer, int32_t const* send_first, int32_t const* send_count, int n_recv, int32_t const* recv_peer,
              int32_t const* recv_first, int32_t const* recv_count, T* const src[5], T const* src_volume, T* const dst[5], T* dst_volume, int cells) {
      LoopbackHub::Post& mine = m_hub.post(m_rank);
      for (int k = 0; k < 5; k++) mine.ptr[k] = src[k];
      mine.ptr[5] = src_volume;
      mine.n_send = n_send; mine.send_peer = send_peer; mine.send_first = send_first; mine.send_count = send_count;
      sync_and_meet();
      const size_t w = static_cast<size_t>(cells);
      for (int j = 0; j < n_recv; j++) {   // the run rank q sends to me (at most one per pair of ranks: both shares are intervals)
        LoopbackHub::Post const& o = m_hub.post(recv_peer[j]);
        int                      i = -1;
        for (int k = 0; k < o.n_send; k++)
          if (o.send_peer[k] == m_rank) i = k;
        if (i < 0 || o.send_count[i] != recv_count[j]) std::abort();
        const size_t n = static_cast<size_t>(recv_count[j]);
        for (int k = 0; k < 5; k++)
          check(hipMemcpy(dst[k] + w * recv_first[j], static_cast<T const*>(o.ptr[k]) + w * o.send_first[i], sizeof(T) * w * n, hipMemcpyDeviceToDevice));
        check(hipMemcpy(dst_volume + recv_first[j], static_cast<T const*>(o.ptr[5]) + o.send_first[i], sizeof(T) * n, hipMemcpyDeviceToDevice));
      }
      sync_and_meet();
    }
    void sync_and_meet() {
      check(hipDeviceSynchronize());
      m_hub.barrier();
    }
    static void check(hipError_t e) {
      if (e != hipSuccess) std::abort();
    }
    static void code(int c) {
      if (c != 0) std::abort();
    }
    LoopbackHub& m_hub;
    int          m_rank;
  };

}  // namespace t8gpu_test

#endif  // T8GPU_TEST_LOOPBACK_TRANSPORT_H
