// t8gpu/memory/memory_manager.h (MI355X backend)
//
// The variable / step traits, MemoryAccessorOwn / MemoryAccessorAll and MemoryManager of the reference
// (t8gpu/memory/memory_manager.h:24-42, 88-186, 217-313, 327-461), HIP-backed:
//   * float_type is a build-time switch (-DT8GPU_FLOAT_TYPE=double); the reference hard-wires float
//     (memory_manager.h:29,39) while anticipating double everywhere.
//   * same plane layout: plane(step, var) = base + (step * nb_variables + var) * capacity, plus one
//     volume plane at the end (memory_manager.h:460, memory_manager.inl:73-80);
//   * capacity may include ghost mirror slots: MemoryManager(nb_elements + nb_ghosts, ...).
// Accessors are the same trivially-copyable PODs passed to kernels by value; structured bindings on
// get(a, b, ...) work exactly as in the reference's kernels (kernels.cu:160).
#ifndef T8GPU_HIP_MEMORY_MEMORY_MANAGER_H
#define T8GPU_HIP_MEMORY_MEMORY_MANAGER_H

#include <t8gpu/memory/shared_device_vector.h>
#include <t8gpu/utils/meta.h>

#include <array>
#include <tuple>
#include <type_traits>

#ifndef T8GPU_FLOAT_TYPE
#define T8GPU_FLOAT_TYPE float
#endif

namespace t8gpu {

  template<class VariableList, typename = void>
  struct variable_traits {};
  template<class VariableType>
  struct variable_traits<VariableType, std::enable_if_t<std::is_enum_v<VariableType>>> {
    using float_type                     = T8GPU_FLOAT_TYPE;
    using index_type                     = VariableType;
    static constexpr size_t nb_variables = VariableType::nb_variables;
  };

  template<class StepList, typename = void>
  struct step_traits {};
  template<class StepType>
  struct step_traits<StepType, std::enable_if_t<std::is_enum_v<StepType>>> {
    using float_type                 = T8GPU_FLOAT_TYPE;
    using index_type                 = StepType;
    static constexpr size_t nb_steps = StepType::nb_steps;
  };

  template<typename VariableType, typename StepType>
  class MemoryManager;
  template<typename VariableType, typename StepType, typename SubgridType>
  class SubgridMemoryManager;
  template<typename VariableType, typename StepType, size_t dim>
  class MeshManager;

  namespace detail {
    /// Shared implementation of both accessors: N handles indexed by the variable enum.
    /// Handle = float_type* (Own) or float_type* const* (All: table of per-rank pointers).
    template<typename VariableType, typename Handle, typename ConstHandle>
    class VariablePack {
     public:
      using variable_index_type            = typename variable_traits<VariableType>::index_type;
      using float_type                     = typename variable_traits<VariableType>::float_type;
      static constexpr size_t nb_variables = variable_traits<VariableType>::nb_variables;

      template<typename T>
      [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T, variable_index_type>, Handle>
      get(T i) {
        return m_handles[static_cast<variable_index_type>(i)];
      }
      template<typename T>
      [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T, variable_index_type>, ConstHandle>
      get(T i) const {
        return m_handles[static_cast<variable_index_type>(i)];
      }
      /// several variables at once, for `auto [rho, rho_v1] = acc.get(Rho, Rho_v1);`
      template<typename T0, typename T1, typename... Ts>
      [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T0, variable_index_type> && meta::all_same_v<T0, T1, Ts...>,
                                                                std::array<Handle, 2 + sizeof...(Ts)>>
      get(T0 i0, T1 i1, Ts... is) {
        return {get(static_cast<variable_index_type>(i0)), get(static_cast<variable_index_type>(i1)),
                get(static_cast<variable_index_type>(is))...};
      }
      template<typename T0, typename T1, typename... Ts>
      [[nodiscard]] __host__ __device__ inline std::enable_if_t<meta::is_explicitly_convertible_to_v<T0, variable_index_type> && meta::all_same_v<T0, T1, Ts...>,
                                                                std::array<ConstHandle, 2 + sizeof...(Ts)>>
      get(T0 i0, T1 i1, Ts... is) const {
        return {get(static_cast<variable_index_type>(i0)), get(static_cast<variable_index_type>(i1)),
                get(static_cast<variable_index_type>(is))...};
      }

     protected:
      std::array<Handle, nb_variables> m_handles;
      VariablePack() = default;
      explicit VariablePack(std::array<Handle, nb_variables> const& h) : m_handles(h) {}
    };
  }  // namespace detail

  /// Variables of the elements this rank stores (owned + ghost mirror slots): one device pointer per
  /// variable (reference memory_manager.h:88-186).
  template<typename VariableType>
  class MemoryAccessorOwn
      : public detail::VariablePack<VariableType, typename variable_traits<VariableType>::float_type*,
                                    typename variable_traits<VariableType>::float_type const*> {
    using base = detail::VariablePack<VariableType, typename variable_traits<VariableType>::float_type*,
                                      typename variable_traits<VariableType>::float_type const*>;
    template<typename VT, typename ST>
    friend class MemoryManager;
    template<typename VT, typename ST, size_t dim_>
    friend class MeshManager;
    template<typename VT, typename ST, typename SubgridType>
    friend class SubgridMeshManager;

   public:
    MemoryAccessorOwn(MemoryAccessorOwn const&)            = default;
    MemoryAccessorOwn& operator=(MemoryAccessorOwn const&) = default;

   private:
    // only the managers build accessors; user code receives and copies them
    explicit MemoryAccessorOwn(std::array<typename base::float_type*, base::nb_variables> const& array) : base(array) {}
  };

  /// `var[rank][index]` view (reference memory_manager.h:217-313). Every rank entry resolves to this
  /// rank's own planes; ghosts are local mirror slots.
  template<typename VariableType>
  class MemoryAccessorAll
      : public detail::VariablePack<VariableType, typename variable_traits<VariableType>::float_type* const*,
                                    typename variable_traits<VariableType>::float_type const* const*> {
    using base = detail::VariablePack<VariableType, typename variable_traits<VariableType>::float_type* const*,
                                      typename variable_traits<VariableType>::float_type const* const*>;
    template<typename VT, typename ST>
    friend class MemoryManager;
    template<typename VT, typename ST, size_t dim_>
    friend class MeshManager;

   public:
    MemoryAccessorAll(MemoryAccessorAll const&)            = default;
    MemoryAccessorAll& operator=(MemoryAccessorAll const&) = default;

   private:
    explicit MemoryAccessorAll(std::array<typename base::float_type* const*, base::nb_variables> const& array) : base(array) {}
  };

  /// Device storage of all (step, variable) planes + the volume plane (reference memory_manager.h:327-461).
  template<typename VariableType, typename StepType>
  class MemoryManager {
   public:
    using float_type                     = typename variable_traits<VariableType>::float_type;
    using variable_index_type            = typename variable_traits<VariableType>::index_type;
    static constexpr size_t nb_variables = variable_traits<VariableType>::nb_variables;
    using step_index_type                = typename step_traits<StepType>::index_type;
    static constexpr size_t nb_steps     = step_traits<StepType>::nb_steps;
    static constexpr size_t nb_planes    = nb_variables * nb_steps + 1;

    explicit MemoryManager(size_t nb_elements = 0, sc_MPI_Comm comm = sc_MPI_COMM_WORLD) : m_device_buffer(nb_elements, comm) {}
    ~MemoryManager() = default;

    /// host / device container (thrust::host_vector, thrust::device_vector, std::vector)
    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    void set_variable(step_index_type step, variable_index_type variable, Container const& buffer) {
      m_device_buffer.copy(plane_of(step, variable), buffer);
    }
    /// raw DEVICE pointer holding size() values
    void set_variable(step_index_type step, variable_index_type variable, float_type* buffer) {
      m_device_buffer.copy(plane_of(step, variable), buffer, m_device_buffer.size());
    }
    template<typename Container, typename = decltype(std::declval<Container const&>().data())>
    void set_volume(Container const& buffer) {
      m_device_buffer.copy(nb_steps * nb_variables, buffer);
    }
    void set_volume(float_type* buffer) { m_device_buffer.copy(nb_steps * nb_variables, buffer, m_device_buffer.size()); }

    [[nodiscard]] float_type*              get_own_volume() { return m_device_buffer.get_own(nb_steps * nb_variables); }
    [[nodiscard]] float_type const*        get_own_volume() const { return m_device_buffer.get_own(nb_steps * nb_variables); }
    [[nodiscard]] float_type* const*       get_all_volume() { return m_device_buffer.get_all(nb_steps * nb_variables); }
    [[nodiscard]] float_type const* const* get_all_volume() const { return m_device_buffer.get_all(nb_steps * nb_variables); }

    [[nodiscard]] MemoryAccessorOwn<VariableType> get_own_variables(step_index_type step) {
      std::array<float_type*, nb_variables> a{};
      for (size_t k = 0; k < nb_variables; k++) a[k] = m_device_buffer.get_own(static_cast<int>(step * nb_variables + k));
      return MemoryAccessorOwn<VariableType>{a};
    }
    [[nodiscard]] MemoryAccessorAll<VariableType> get_all_variables(step_index_type step) {
      std::array<float_type* const*, nb_variables> a{};
      for (size_t k = 0; k < nb_variables; k++) a[k] = m_device_buffer.get_all(static_cast<int>(step * nb_variables + k));
      return MemoryAccessorAll<VariableType>{a};
    }
    [[nodiscard]] float_type* get_own_variable(step_index_type step, variable_index_type variable) {
      return m_device_buffer.get_own(plane_of(step, variable));
    }
    [[nodiscard]] float_type const* get_own_variable(step_index_type step, variable_index_type variable) const {
      return m_device_buffer.get_own(plane_of(step, variable));
    }

    /// Local operation (the reference's is collective); contents are discarded on growth.
    inline void resize(size_t new_size) { m_device_buffer.resize(new_size); }

    // -- additions for the C-ABI step driver (t8gpu_hip_plain_stepper_iterate_*) --
    [[nodiscard]] float_type* planes_base() { return m_device_buffer.base(); }
    [[nodiscard]] size_t      plane_stride() const { return m_device_buffer.capacity(); }
    [[nodiscard]] size_t      size() const { return m_device_buffer.size(); }

   private:
    static int plane_of(step_index_type step, variable_index_type variable) {
      return static_cast<int>(step) * static_cast<int>(nb_variables) + static_cast<int>(variable);
    }
    SharedDeviceVector<std::array<float_type, nb_planes>> m_device_buffer;
  };

}  // namespace t8gpu

#endif  // T8GPU_HIP_MEMORY_MEMORY_MANAGER_H
