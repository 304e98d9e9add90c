// t8gpu/memory/subgrid_memory_manager.h (MI355X backend)
//
// Subgrid<extents...>, its Accessor, SubgridMemoryAccessorOwn / All and SubgridMemoryManager of the
// reference (t8gpu/memory/subgrid_memory_manager.h:35-135, 178-276, 310-411, 424-555), HIP-backed.
// Data layout is the reference's: column-major inside a block (first index fastest, :53-64) and
// element-major across blocks, data[e * size + flat_index(i, j, k)] (:88-90); the volume is a separate
// per-block vector (:553-554). On gfx950 a Subgrid<4,4,4> block is exactly one 64-lane wavefront.
#ifndef T8GPU_HIP_MEMORY_SUBGRID_MEMORY_MANAGER_H
#define T8GPU_HIP_MEMORY_SUBGRID_MEMORY_MANAGER_H

#include <t8gpu/memory/memory_manager.h>

#include <array>
#include <type_traits>
#include <utility>

namespace t8gpu {

  template<typename VariableType, typename SubgridType>
  class SubgridMemoryAccessorOwn;
  template<typename VariableType, typename SubgridType>
  class SubgridMemoryAccessorAll;
  template<typename VariableType, typename StepType, typename SubgridType>
  class SubgridMemoryManager;
  template<typename VariableType, typename StepType, typename SubgridType>
  class SubgridMeshManager;

  template<int... extents>
  struct Subgrid {
    static constexpr int rank = sizeof...(extents);
    static constexpr int size = (extents * ...);

    template<int dim>
    static constexpr int extent = meta::argpack_at_v<dim, extents...>;
    /// column-major: stride<0> = 1, stride<1> = extent<0>, ...
    template<int i>
    static constexpr int stride = meta::argpack_mul_to_v<i, extents...>;

    template<typename... Ts>
    __host__ __device__ static constexpr inline int flat_index(Ts... is) {
      static_assert(sizeof...(Ts) == rank, "flat_index needs one index per dimension");
      return flat_impl(std::index_sequence_for<Ts...>{}, is...);
    }

    /// launch shape of kernels that map one thread to one subcell
    static constexpr dim3 block_size = {extents...};

    /// view of one variable over all blocks of a rank
    template<typename float_type>
    class Accessor {
     public:
      Accessor(Accessor const&)            = default;
      Accessor& operator=(Accessor const&) = default;

      template<typename... Ts>
      [[nodiscard]] inline __device__ std::enable_if_t<(sizeof...(Ts) == rank) && std::conjunction_v<std::is_integral<Ts>...>, float_type&>
      operator()(size_t e_idx, Ts... is) {
        return m_data[e_idx * size + flat_index(is...)];
      }
      template<typename... Ts>
      [[nodiscard]] inline __device__ std::enable_if_t<(sizeof...(Ts) == rank) && std::conjunction_v<std::is_integral<Ts>...>, float_type const&>
      operator()(size_t e_idx, Ts... is) const {
        return m_data[e_idx * size + flat_index(is...)];
      }
      __host__ __device__ explicit operator float_type*() { return m_data; }
      __host__ __device__ explicit operator float_type const*() const { return m_data; }

     private:
      __host__ __device__ Accessor(float_type const* data) : m_data{const_cast<float_type*>(data)} {}
      float_type* m_data;

      template<typename VariableType, typename SubgridType>
      friend class SubgridMemoryAccessorOwn;
      template<typename VariableType, typename SubgridType>
      friend class SubgridMemoryAccessorAll;
      template<typename VariableType, typename StepType, typename SubgridType>
      friend class SubgridMemoryManager;
    };
    template<typename float_type>
    using accessor_type = Accessor<float_type>;

   private:
    template<size_t... I, typename... Ts>
    __host__ __device__ static constexpr inline int flat_impl(std::index_sequence<I...>, Ts... is) {
      return ((stride<static_cast<int>(I)> * static_cast<int>(is)) + ...);
    }
  };

  /// reference subgrid_memory_manager.h:178-276
  template<typename VariableType, typename SubgridType>
  class SubgridMemoryAccessorOwn {
    template<typename VT, typename ST, typename SG>
    friend class SubgridMemoryManager;
    template<typename VT, typename ST, typename SG>
    friend class SubgridMeshManager;

   public:
    using variable_index_type            = typename variable_traits<VariableType>::index_type;
    using float_type                     = typename variable_traits<VariableType>::float_type;
    using subgrid_type                   = SubgridType;
    using view_type                      = typename SubgridType::template accessor_type<float_type>;
    static constexpr size_t nb_variables = variable_traits<VariableType>::nb_variables;

    SubgridMemoryAccessorOwn(SubgridMemoryAccessorOwn const&)            = default;
    SubgridMemoryAccessorOwn& operator=(SubgridMemoryAccessorOwn const&) = default;

    template<typename T>
    [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T, variable_index_type>, view_type> get(T i) const {
      return view_type{m_pointers[static_cast<variable_index_type>(i)]};
    }
    template<typename T0, typename T1, typename... Ts>
    [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T0, variable_index_type> && meta::all_same_v<T0, T1, Ts...>,
                                                              std::array<view_type, 2 + sizeof...(Ts)>>
    get(T0 i0, T1 i1, Ts... is) const {
      return {get(static_cast<variable_index_type>(i0)), get(static_cast<variable_index_type>(i1)),
              get(static_cast<variable_index_type>(is))...};
    }
    /// raw plane pointer (what the C-ABI's T8gpuVars_* holds)
    [[nodiscard]] __host__ __device__ float_type* data(variable_index_type i) const { return m_pointers[i]; }

   private:
    std::array<float_type*, nb_variables> m_pointers;
    explicit SubgridMemoryAccessorOwn(std::array<float_type*, nb_variables> const& array) : m_pointers(array) {}
  };

  /// reference subgrid_memory_manager.h:310-411: get(rank, var)
  template<typename VariableType, typename SubgridType>
  class SubgridMemoryAccessorAll {
    template<typename VT, typename ST, typename SG>
    friend class SubgridMemoryManager;
    template<typename VT, typename ST, typename SG>
    friend class SubgridMeshManager;

   public:
    using variable_index_type            = typename variable_traits<VariableType>::index_type;
    using float_type                     = typename variable_traits<VariableType>::float_type;
    using view_type                      = typename SubgridType::template accessor_type<float_type>;
    static constexpr size_t nb_variables = variable_traits<VariableType>::nb_variables;

    SubgridMemoryAccessorAll(SubgridMemoryAccessorAll const&)            = default;
    SubgridMemoryAccessorAll& operator=(SubgridMemoryAccessorAll const&) = default;

    template<typename T>
    [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T, variable_index_type>, view_type> get(int rank, T i) const {
      return view_type{m_pointers[static_cast<variable_index_type>(i)][rank]};
    }
    template<typename T0, typename T1, typename... Ts>
    [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T0, variable_index_type> && meta::all_same_v<T0, T1, Ts...>,
                                                              std::array<view_type, 2 + sizeof...(Ts)>>
    get(int rank, T0 i0, T1 i1, Ts... is) const {
      return {get(rank, static_cast<variable_index_type>(i0)), get(rank, static_cast<variable_index_type>(i1)),
              get(rank, static_cast<variable_index_type>(is))...};
    }

   private:
    std::array<float_type* const*, nb_variables> m_pointers;
    explicit SubgridMemoryAccessorAll(std::array<float_type* const*, nb_variables> const& array) : m_pointers(array) {}
  };

  /// reference subgrid_memory_manager.h:424-555
  template<typename VariableType, typename StepType, typename SubgridType>
  class SubgridMemoryManager {
   public:
    using float_type                     = typename variable_traits<VariableType>::float_type;
    using variable_index_type            = typename variable_traits<VariableType>::index_type;
    static constexpr size_t nb_variables = variable_traits<VariableType>::nb_variables;
    using step_index_type                = typename step_traits<StepType>::index_type;
    static constexpr size_t nb_steps     = step_traits<StepType>::nb_steps;
    using view_type                      = typename SubgridType::template accessor_type<float_type>;

    explicit SubgridMemoryManager(size_t nb_elements = 0, sc_MPI_Comm comm = sc_MPI_COMM_WORLD)
        : m_device_buffer(nb_elements * SubgridType::size, comm), m_device_volume(nb_elements, comm) {}
    ~SubgridMemoryManager() = default;

    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    void set_variable(step_index_type step, variable_index_type variable, Container const& buffer) {
      m_device_buffer.copy(plane_of(step, variable), buffer);
    }
    void set_variable(step_index_type step, variable_index_type variable, float_type* buffer) {
      m_device_buffer.copy(plane_of(step, variable), buffer, m_device_buffer.size());
    }
    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    void set_volume(Container const& buffer) {
      m_device_volume = buffer;
    }

    [[nodiscard]] float_type*              get_own_volume() { return m_device_volume.get_own(); }
    [[nodiscard]] float_type const*        get_own_volume() const { return m_device_volume.get_own(); }
    [[nodiscard]] float_type* const*       get_all_volume() { return m_device_volume.get_all(); }
    [[nodiscard]] float_type const* const* get_all_volume() const { return m_device_volume.get_all(); }

    [[nodiscard]] SubgridMemoryAccessorOwn<VariableType, SubgridType> get_own_variables(step_index_type step) {
      std::array<float_type*, nb_variables> a{};
      for (size_t k = 0; k < nb_variables; k++) a[k] = m_device_buffer.get_own(static_cast<int>(step * nb_variables + k));
      return SubgridMemoryAccessorOwn<VariableType, SubgridType>{a};
    }
    [[nodiscard]] SubgridMemoryAccessorAll<VariableType, SubgridType> get_all_variables(step_index_type step) {
      std::array<float_type* const*, nb_variables> a{};
      for (size_t k = 0; k < nb_variables; k++) a[k] = m_device_buffer.get_all(static_cast<int>(step * nb_variables + k));
      return SubgridMemoryAccessorAll<VariableType, SubgridType>{a};
    }
    [[nodiscard]] view_type get_own_variable(step_index_type step, variable_index_type variable) {
      return view_type{m_device_buffer.get_own(plane_of(step, variable))};
    }
    [[nodiscard]] view_type const get_own_variable(step_index_type step, variable_index_type variable) const {
      return view_type{m_device_buffer.get_own(plane_of(step, variable))};
    }

    /// number of BLOCKS; resizes the variable planes and (unlike the reference, quirk Q9) the volumes
    inline void resize(size_t new_size) {
      m_device_buffer.resize(new_size * SubgridType::size);
      m_device_volume.resize(new_size);
    }

   private:
    static int plane_of(step_index_type step, variable_index_type variable) {
      return static_cast<int>(step) * static_cast<int>(nb_variables) + static_cast<int>(variable);
    }
    SharedDeviceVector<std::array<float_type, nb_variables * nb_steps>> m_device_buffer;
    SharedDeviceVector<float_type>                                      m_device_volume;
  };

}  // namespace t8gpu

#endif  // T8GPU_HIP_MEMORY_SUBGRID_MEMORY_MANAGER_H
