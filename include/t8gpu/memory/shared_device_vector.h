// t8gpu/memory/shared_device_vector.h (MI355X backend)
//
// API of the reference's SharedDeviceVector<T> / SharedDeviceVector<std::array<T,N>>
// (t8gpu/memory/shared_device_vector.h:42-163,178-337) on one-GPU-per-rank semantics:
//   * the reference maps EVERY rank's allocation into every rank through CUDA IPC on one shared GPU
//     and re-exchanges handles with MPI_Allgather at construction and on every resize
//     (shared_device_vector.inl:15-30,159-199,249-290). Here a rank only ever dereferences its own
//     allocation: elements owned by other ranks are mirrored in ghost slots appended after the owned
//     ones and refreshed by RCCL send/recv (include/t8gpu_hip.h, T8gpuHalo).
//   * get_all() is kept: it returns a device table with one entry per rank, every entry pointing at
//     THIS rank's allocation, so kernels written as `var[rank][index]` keep working when the
//     connectivity reports (rank, index) = (any, local slot).
//   * resize() keeps the reference's contract (grow by 1.5x, previous contents discarded, accessors
//     invalidated) but is a purely local operation -- no collective, no handle exchange.
#ifndef T8GPU_HIP_MEMORY_SHARED_DEVICE_VECTOR_H
#define T8GPU_HIP_MEMORY_SHARED_DEVICE_VECTOR_H

#include <hip/hip_runtime.h>
#include <t8gpu/utils/cuda.h>

#include <array>
#include <cassert>
#include <cstddef>
#include <utility>
#include <vector>

#if __has_include(<sc.h>)
#include <sc.h>
#else
// libsc is not installed: single-process build. These two names are what user code spells in the
// constructor calls (`sc_MPI_Comm comm = sc_MPI_COMM_WORLD`).
struct t8gpu_serial_comm {
  int rank = 0, size = 1;
};
using sc_MPI_Comm = t8gpu_serial_comm;
inline constexpr t8gpu_serial_comm sc_MPI_COMM_WORLD{};
#define T8GPU_SERIAL_COMM 1
#endif

namespace t8gpu {

  namespace detail {
    inline void comm_layout(sc_MPI_Comm comm, int& rank, int& nb_ranks) {
#ifdef T8GPU_SERIAL_COMM
      rank     = comm.rank;
      nb_ranks = comm.size;
#else
      sc_MPI_Comm_rank(comm, &rank);
      sc_MPI_Comm_size(comm, &nb_ranks);
#endif
    }

    // host copy if `p` is an ordinary pointer, device copy if it is a device_ptr-like handle with get()
    template<typename P>
    auto raw(P p, int) -> decltype(p.get()) {
      return p.get();
    }
    template<typename P>
    P raw(P p, long) {
      return p;
    }
    template<typename P, typename = void>
    struct is_device_handle : std::false_type {};
    template<typename P>
    struct is_device_handle<P, std::void_t<decltype(std::declval<P>().get())>> : std::true_type {};

    template<typename T, typename Container>
    void copy_in(T* dst, Container const& src, size_t count) {
      using ptr_t = decltype(src.data());
      constexpr bool on_device = is_device_handle<ptr_t>::value;
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(dst, raw(src.data(), 0), sizeof(T) * count,
                                       on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    }
  }  // namespace detail

  /// One device array per rank (reference: shared_device_vector.h:42-163).
  template<typename T>
  class SharedDeviceVector {
   public:
    explicit SharedDeviceVector(size_t size = 0, sc_MPI_Comm comm = sc_MPI_COMM_WORLD) : m_size{size}, m_capacity{size} {
      detail::comm_layout(comm, m_rank, m_nb_ranks);
      if (m_capacity > 0) T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_data, sizeof(T) * m_capacity));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_table, sizeof(T*) * m_nb_ranks));
      publish();
    }
    ~SharedDeviceVector() { release(); }
    SharedDeviceVector(SharedDeviceVector const&)            = delete;
    SharedDeviceVector& operator=(SharedDeviceVector const&) = delete;
    SharedDeviceVector(SharedDeviceVector&& o) noexcept { steal(o); }
    SharedDeviceVector& operator=(SharedDeviceVector&& o) noexcept {
      if (this != &o) {
        release();
        steal(o);
      }
      return *this;
    }

    /// Local (not collective). Contents are discarded when the allocation grows.
    void resize(size_t new_size) {
      if (new_size > m_capacity) {
        if (m_data) T8GPU_CUDA_CHECK_ERROR(hipFree(m_data));
        m_capacity = new_size + new_size / 2;
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_data, sizeof(T) * m_capacity));
        publish();
      }
      m_size = new_size;
    }

    /// Assignment from a host or device container (thrust::host_vector, thrust::device_vector, std::vector).
    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    SharedDeviceVector const& operator=(Container const& other) {
      resize(other.size());
      if (m_size) detail::copy_in(m_data, other, m_size);
      return *this;
    }

    [[nodiscard]] size_t   size() const { return m_size; }
    void                   clear() { m_size = 0; }
    [[nodiscard]] T*       get_own() { return m_data; }
    [[nodiscard]] T const* get_own() const { return m_data; }
    [[nodiscard]] T**      get_all() { return m_table; }
    [[nodiscard]] T const* const* get_all() const { return m_table; }

   private:
    int    m_rank = 0, m_nb_ranks = 1;
    size_t m_size = 0, m_capacity = 0;
    T*     m_data  = nullptr;
    T**    m_table = nullptr;  // device: [nb_ranks], every entry = m_data

    void publish() {
      std::vector<T*> host(m_nb_ranks, m_data);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_table, host.data(), sizeof(T*) * m_nb_ranks, hipMemcpyHostToDevice));
    }
    void release() {
      if (m_data) (void)hipFree(m_data);
      if (m_table) (void)hipFree(m_table);
      m_data  = nullptr;
      m_table = nullptr;
    }
    void steal(SharedDeviceVector& o) {
      m_rank = o.m_rank; m_nb_ranks = o.m_nb_ranks; m_size = o.m_size; m_capacity = o.m_capacity;
      m_data = std::exchange(o.m_data, nullptr);
      m_table = std::exchange(o.m_table, nullptr);
    }
  };

  /// N planes in ONE allocation, plane i at base + i * capacity (reference: shared_device_vector.h:178-337,
  /// layout shared_device_vector.inl:193-197). This is the storage behind MemoryManager.
  template<typename T, size_t N>
  class SharedDeviceVector<std::array<T, N>> {
   public:
    explicit SharedDeviceVector(size_t size = 0, sc_MPI_Comm comm = sc_MPI_COMM_WORLD) : m_size{size}, m_capacity{size} {
      detail::comm_layout(comm, m_rank, m_nb_ranks);
      if (m_capacity > 0) T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_data, sizeof(T) * m_capacity * N));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_table, sizeof(T*) * N * m_nb_ranks));
      publish();
    }
    ~SharedDeviceVector() { release(); }
    SharedDeviceVector(SharedDeviceVector const&)            = delete;
    SharedDeviceVector& operator=(SharedDeviceVector const&) = delete;
    SharedDeviceVector(SharedDeviceVector&& o) noexcept { steal(o); }
    SharedDeviceVector& operator=(SharedDeviceVector&& o) noexcept {
      if (this != &o) {
        release();
        steal(o);
      }
      return *this;
    }

    void resize(size_t new_size) {
      if (new_size > m_capacity) {
        if (m_data) T8GPU_CUDA_CHECK_ERROR(hipFree(m_data));
        m_capacity = new_size + new_size / 2;
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_data, sizeof(T) * m_capacity * N));
        publish();
      }
      m_size = new_size;
    }

    /// copy a whole host/device container into plane `index`
    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    void copy(size_t index, Container const& vector) {
      assert(vector.size() <= m_capacity);
      detail::copy_in(plane(index), vector, vector.size());
    }
    /// copy `num_elements` values from a DEVICE buffer into plane `index`
    void copy(size_t index, T const* buffer, size_t num_elements) {
      assert(num_elements <= m_capacity);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(plane(index), buffer, sizeof(T) * num_elements, hipMemcpyDeviceToDevice));
    }

    [[nodiscard]] size_t   size() const { return m_size; }
    [[nodiscard]] size_t   capacity() const { return m_capacity; }
    void                   clear() { m_size = 0; }
    [[nodiscard]] T*       get_own(int index) { return plane(index); }
    [[nodiscard]] T const* get_own(int index) const { return plane(index); }
    [[nodiscard]] T**      get_all(int index) { return m_table + static_cast<size_t>(index) * m_nb_ranks; }
    [[nodiscard]] T const* const* get_all(int index) const { return m_table + static_cast<size_t>(index) * m_nb_ranks; }
    /// base of plane 0 (the pointer the C-ABI step driver takes together with capacity())
    [[nodiscard]] T* base() { return m_data; }

   private:
    int    m_rank = 0, m_nb_ranks = 1;
    size_t m_size = 0, m_capacity = 0;
    T*     m_data  = nullptr;
    T**    m_table = nullptr;  // device: [N][nb_ranks], entry (i, r) = plane i of this rank

    T* plane(size_t i) const { return m_data + i * m_capacity; }
    void publish() {
      std::vector<T*> host(N * m_nb_ranks);
      for (size_t i = 0; i < N; i++)
        for (int r = 0; r < m_nb_ranks; r++) host[i * m_nb_ranks + r] = plane(i);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_table, host.data(), sizeof(T*) * host.size(), hipMemcpyHostToDevice));
    }
    void release() {
      if (m_data) (void)hipFree(m_data);
      if (m_table) (void)hipFree(m_table);
      m_data  = nullptr;
      m_table = nullptr;
    }
    void steal(SharedDeviceVector& o) {
      m_rank = o.m_rank; m_nb_ranks = o.m_nb_ranks; m_size = o.m_size; m_capacity = o.m_capacity;
      m_data = std::exchange(o.m_data, nullptr);
      m_table = std::exchange(o.m_table, nullptr);
    }
  };

}  // namespace t8gpu

#endif  // T8GPU_HIP_MEMORY_SHARED_DEVICE_VECTOR_H
