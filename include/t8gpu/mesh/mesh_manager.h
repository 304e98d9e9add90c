// t8gpu/mesh/mesh_manager.h (MI355X backend)
//
// MeshConnectivityAccessor<float_type, dim>: the device-side view of the face lists, with the getters
// of the reference (t8gpu/mesh/mesh_manager.h:30-182) over the same arrays:
//   ranks[N+G], indices[N+G], face_neighbors[2F + B], face_normals[dim * (F + B)], face_surfaces[F + B].
// MeshManager<V, S, dim>: the class of the reference (t8gpu/mesh/mesh_manager.h:232-465) with the same public
// members -- constructor (comm, scheme, cmesh, forest), initialize_variables(Func), adapt(criteria, step),
// partition(step), compute_connectivity_information(), save_variable(s)_to_vtk, HostVariableInfo,
// get_host_{scalar,vector}_variable, get_connectivity_information(), get_num_local_{elements,faces,
// boundary_faces}(), get_num_ghost_elements(), min_level / max_level -- on top of MemoryManager. Where the mesh
// comes from is a provider behind the class:
//   * a t8code forest: the (comm, scheme, cmesh, forest) constructor. It is DECLARED here and defined only in a
//     build that has t8code (it fills a T8gpuForestQuery from t8code calls); this image
//     has no t8code, so example TUs compile against it and link once that adapter is compiled in.
//   * the t8code-free synthetic forest of include/t8gpu_host.h, or plain host arrays (HostMeshArrays): the two
//     extra constructors below. `SyntheticMeshManager<V,S,dim>` is an alias of MeshManager kept for code written
//     against the earlier name.
// Read-back / VTK members (mesh_manager.inl:516-623): device half in csrc/hip/kernels_readback.hip, the file is
// written by t8gpu_host_write_vtu where the reference calls t8_forest_write_vtk_ext.
#ifndef T8GPU_HIP_MESH_MESH_MANAGER_H
#define T8GPU_HIP_MESH_MESH_MANAGER_H

#include <t8gpu/backend/transport.h>
#include <t8gpu/memory/memory_manager.h>

#include <t8gpu_hip.h>
#include <t8gpu_host.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include <thrust/host_vector.h>

#if __has_include(<t8.h>)
#include <t8.h>
#else
// no t8code on the include path: the handle types of the constructor's signature, opaque
using t8_locidx_t = int32_t;
typedef struct t8_forest* t8_forest_t;
typedef struct t8_cmesh*  t8_cmesh_t;
struct t8_scheme_cxx;
typedef struct t8_scheme_cxx t8_scheme_cxx_t;
struct t8_element;
typedef struct t8_element t8_element_t;
#endif

namespace t8gpu {

  template<typename float_type, size_t dim>
  class MeshConnectivityAccessor {
    template<typename VT, typename ST, size_t dim_>
    friend class MeshManager;

   public:
    MeshConnectivityAccessor(MeshConnectivityAccessor const&)            = default;
    MeshConnectivityAccessor& operator=(MeshConnectivityAccessor const&) = default;

    [[nodiscard]] __host__ __device__ inline t8_locidx_t get_num_local_faces() const { return m_num_local_faces; }
    [[nodiscard]] __host__ __device__ inline t8_locidx_t get_num_local_boundary_faces() const { return m_num_local_boundary_faces; }

    [[nodiscard]] __device__ inline float_type get_face_surface(int f) const { return m_face_surfaces[f]; }
    [[nodiscard]] __device__ inline float_type get_boundary_face_surface(int f) const { return m_face_surfaces[m_num_local_faces + f]; }

    [[nodiscard]] __device__ inline std::array<float_type, dim> get_face_normal(int f) const { return normal_at(f); }
    [[nodiscard]] __device__ inline std::array<float_type, dim> get_boundary_face_normal(int f) const { return normal_at(m_num_local_faces + f); }

    /// (left, right) LOCAL element indices; >= num_local_elements means a ghost (mirror slot)
    [[nodiscard]] __device__ inline std::array<t8_locidx_t, 2> get_face_neighbor_indices(int f) const {
      return {m_face_neighbors[2 * f], m_face_neighbors[2 * f + 1]};
    }
    [[nodiscard]] __device__ inline t8_locidx_t get_boundary_face_neighbor_index(int f) const {
      return m_face_neighbors[2 * m_num_local_faces + f];
    }
    /// owner rank / slot of a local element index. In this backend every element (ghosts included)
    /// resolves to a slot of THIS rank's planes, so `var[rank][index]` never leaves the device.
    [[nodiscard]] __device__ inline t8_locidx_t get_element_owner_rank(int e) const { return m_ranks[e]; }
    [[nodiscard]] __device__ inline t8_locidx_t get_element_owner_remote_index(int e) const { return m_indices[e]; }

    // raw arrays, for the C-ABI (t8gpu_hip_flux_faces_* takes exactly these)
    [[nodiscard]] __host__ __device__ t8_locidx_t const* face_neighbors() const { return m_face_neighbors; }
    [[nodiscard]] __host__ __device__ t8_locidx_t const* indices() const { return m_indices; }
    [[nodiscard]] __host__ __device__ float_type const*  face_normals() const { return m_face_normals; }
    [[nodiscard]] __host__ __device__ float_type const*  face_surfaces() const { return m_face_surfaces; }

   private:
    int const*         m_ranks;
    t8_locidx_t const* m_indices;
    t8_locidx_t const* m_face_neighbors;
    float_type const*  m_face_normals;
    float_type const*  m_face_surfaces;
    t8_locidx_t        m_num_local_faces;
    t8_locidx_t        m_num_local_boundary_faces;

    __device__ inline std::array<float_type, dim> normal_at(int slot) const {
      std::array<float_type, dim> n{};
      for (size_t k = 0; k < dim; k++) n[k] = m_face_normals[dim * slot + k];
      return n;
    }
    MeshConnectivityAccessor(int const* ranks, t8_locidx_t const* indices, t8_locidx_t const* fn, float_type const* normals,
                             float_type const* surfaces, t8_locidx_t F, t8_locidx_t B)
        : m_ranks{ranks}, m_indices{indices}, m_face_neighbors{fn}, m_face_normals{normals}, m_face_surfaces{surfaces},
          m_num_local_faces{F}, m_num_local_boundary_faces{B} {}
  };

  /// Host description of one rank's mesh in the reference's array formats (doubles are converted to
  /// float_type the way the reference casts t8code's doubles, mesh_manager.inl:400-407).
  struct HostMeshArrays {
    int32_t num_local_elements = 0, num_ghost_elements = 0, num_local_faces = 0, num_local_boundary_faces = 0;
    int     rank = 0;
    std::vector<int32_t> face_neighbors;  // [2F + B]
    std::vector<double>  face_normals;    // [dim * (F + B)]
    std::vector<double>  face_surfaces;   // [F + B]
    std::vector<double>  volumes;         // [N + G]
    // only needed by save_variables_to_vtk: geometry of the owned leaves on the unit domain
    int                  mesh_dim = 2;
    int64_t              first_global_element = 0;
    std::vector<double>  centres;         // [N][3]
    std::vector<int32_t> levels;          // [N]
  };

  /// What the ghost layer needs besides HostMeshArrays (peers ascending; offsets have n_peers + 1 entries; send_idx = owned
  /// elements mirrored on a peer; the ghosts of peer j are the mirror slots N + [recv_off[j], recv_off[j + 1])).
  struct HostHaloArrays {
    std::vector<int32_t> peers, recv_off, send_off, send_idx;
  };

  /// T8_VTK_SCALAR / T8_VTK_VECTOR of t8code's t8_vtk_data_field_t: values per cell
  enum : int { T8GPU_VTK_SCALAR = 1, T8GPU_VTK_VECTOR = 3 };

  /// What initialize_variables hands to the user function as `t8_element_t const*` when the forest is the synthetic
  /// provider's (there is no t8code element behind it): centre, level and volume of the leaf on the unit domain.
  /// `t8gpu::synthetic_element(element)` turns the opaque pointer back into this record.
  struct SyntheticElement {
    double  centre[3];
    int32_t level;
    double  volume;
  };
  [[nodiscard]] inline SyntheticElement const& synthetic_element(t8_element_t const* element) {
    return *reinterpret_cast<SyntheticElement const*>(element);
  }

  template<typename VariableType, typename StepType, size_t dim>
  class MeshManager : public MemoryManager<VariableType, StepType> {
   public:
    using float_type                  = typename variable_traits<VariableType>::float_type;
    using variable_index_type         = typename variable_traits<VariableType>::index_type;
    static constexpr int nb_variables = variable_traits<VariableType>::nb_variables;

    using step_index_type            = typename step_traits<StepType>::index_type;
    static constexpr size_t nb_steps = step_traits<StepType>::nb_steps;

    static constexpr t8_locidx_t min_level = 1;   // mesh_manager.h:240-241 (adapt() of a t8code forest uses them)
    static constexpr t8_locidx_t max_level = 4;

    /// mesh_manager.h:253 / mesh_manager.inl:20-64: takes ownership of cmesh and forest. Declared for every build;
    /// DEFINED by the t8code adapter only (fills a T8gpuForestQuery from t8_forest_leaf_face_neighbors & co.,
    /// INTEGRATION.md section 4) -- a build without t8code cannot call it and gets a link error if it tries.
    MeshManager(sc_MPI_Comm comm, t8_scheme_cxx_t* scheme, t8_cmesh_t cmesh, t8_forest_t forest);

    /// From host arrays in the reference's formats (one rank's share; ghosts resolved to mirror slots).
    explicit MeshManager(HostMeshArrays const& m, sc_MPI_Comm comm = sc_MPI_COMM_WORLD)
        : MemoryManager<VariableType, StepType>(static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements, comm) {
      detail::comm_layout(comm, m_comm_rank, m_nb_ranks);
      rebuild_connectivity(m);
      this->set_volume(std::vector<float_type>(m.volumes.begin(), m.volumes.end()));
    }
    /// From a synthetic forest (t8gpu_synth_mesh_create; the manager takes ownership): stands for
    /// MeshManager(comm, scheme, cmesh, forest) (mesh_manager.inl:3-44). The connectivity comes through the
    /// forest-query adapter (csrc/host/connectivity.cpp), i.e. the way a t8code build would provide it.
    /// `lowest_level` / `highest_level` bound adapt() (the class constants min_level / max_level by default).
    /// On several ranks (comm.size > 1) every rank passes ITS OWN handle of the same forest (the description is replicated,
    /// as in t8gpu_amd/amr.py) and owns the contiguous share [n r / size, n (r + 1) / size) of the space-filling curve;
    /// set_transport() must follow before adapt() / partition() / refresh_ghost_layer() are called.
    explicit MeshManager(void* synth_mesh, int lowest_level = min_level, int highest_level = max_level,
                         sc_MPI_Comm comm = sc_MPI_COMM_WORLD)
        : MeshManager(arrays_of(synth_mesh, comm_rank_of(comm), comm_size_of(comm), nullptr), comm) {
      m_forest    = synth_mesh;
      m_min_level = lowest_level;
      m_max_level = highest_level;
      if (m_nb_ranks > 1) rebuild_connectivity(arrays_of(m_forest, m_comm_rank, m_nb_ranks, &m_halo_host));   // (+ the halo lists)
    }
    /// The channel adapt() / partition() / refresh_ghost_layer() use on several ranks (not owned). See backend/transport.h.
    void set_transport(Transport* transport) { m_transport = transport; }

    /// mesh_manager.inl:76-122: `func(accessor, forest, tree_idx, element, e_idx)` fills the variables of element
    /// e_idx in a HOST accessor; all 26 planes are zeroed, Step 0 and the volume uploaded. With the synthetic provider
    /// `forest` is its handle, tree_idx 0 and `element` a SyntheticElement (see synthetic_element()).
    template<typename Func>
    void initialize_variables(Func func) {
      const size_t n = static_cast<size_t>(m_num_local_elements);
      std::array<std::vector<float_type>, nb_variables> host_variables{};
      std::array<float_type*, nb_variables>             array{};
      for (size_t k = 0; k < static_cast<size_t>(nb_variables); k++) {
        host_variables[k].resize(n);
        array[k] = host_variables[k].data();
      }
      MemoryAccessorOwn<VariableType> host_variable_memory{array};
      std::vector<float_type>         element_volume(n + static_cast<size_t>(m_num_ghost_elements), float_type(1));
      for (size_t e = 0; e < n; e++) {
        SyntheticElement el{{m_centres[3 * e], m_centres[3 * e + 1], m_centres[3 * e + 2]}, m_levels[e], m_host.volumes[e]};
        element_volume[e] = static_cast<float_type>(el.volume);
        func(host_variable_memory, reinterpret_cast<t8_forest_t>(m_forest), t8_locidx_t{0},
             reinterpret_cast<t8_element_t const*>(&el), static_cast<t8_locidx_t>(e));
      }
      for (size_t g = n; g < element_volume.size(); g++) element_volume[g] = static_cast<float_type>(m_host.volumes[g]);
      // every plane of every step zeroed one by one (the reference memsets 26*N values from plane 0, valid only while
      // capacity == size: SURVEY quirk Q10)
      std::vector<float_type> zeros(n + static_cast<size_t>(m_num_ghost_elements), float_type(0));
      for (size_t st = 0; st < nb_steps; st++)
        for (size_t k = 0; k < static_cast<size_t>(nb_variables); k++)
          this->set_variable(static_cast<step_index_type>(st), static_cast<variable_index_type>(k), zeros);
      for (size_t k = 0; k < static_cast<size_t>(nb_variables); k++) {
        host_variables[k].resize(zeros.size(), float_type(0));
        this->set_variable(static_cast<step_index_type>(0), static_cast<variable_index_type>(k), host_variables[k]);
      }
      this->set_volume(element_volume);
    }

    /// mesh_manager.h:297 (the reference's signature); criteria above 10 refine, families below it coarsen
    void adapt(thrust::host_vector<float_type> const& refinement_criteria, step_index_type step) {
      adapt(std::vector<float_type>(refinement_criteria.begin(), refinement_criteria.end()), step);
    }

    /// mesh_manager.inl:626-723. After adapt() the elements a rank holds are no longer its equal share of the curve;
    /// partition() ships every run of adapted elements to its owner in the new equal split (t8gpu_hip_repartition_*: the
    /// old owner sends, where the reference's new owner pulls through CUDA-IPC pointers, partition_data<<<>>> :626-643),
    /// installs the new forest and rebuilds the connectivity. Only `step` and the volume are valid afterwards, as in the
    /// reference. On one rank, or when no adapt() is pending, it is the identity (t8_forest_partition moves nothing).
    void partition(step_index_type step) {
      if (!m_pending.forest) return;
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: partition() on %d ranks needs a transport (MeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      const int     R = m_nb_ranks, r = m_comm_rank;
      const int64_t n_new = t8gpu_synth_mesh_num_elements(m_pending.forest);
      auto off = [&](int q) { return n_new * q / R; };                                   // the equal split of the NEW curve
      const int64_t a = m_pending.have_off[r], b = m_pending.have_off[r + 1], lo = off(r), hi = off(r + 1);
      std::vector<int32_t> sp, sf, sc, rp, rf, rc;
      for (int q = 0; q < R; q++) {
        const int64_t s0 = std::max(a, off(q)), s1 = std::min(b, off(q + 1));
        if (s1 > s0) { sp.push_back(q); sf.push_back(static_cast<int32_t>(s0 - a)); sc.push_back(static_cast<int32_t>(s1 - s0)); }
        const int64_t r0 = std::max(m_pending.have_off[q], lo), r1 = std::min(m_pending.have_off[q + 1], hi);
        if (r1 > r0) { rp.push_back(q); rf.push_back(static_cast<int32_t>(r0 - lo)); rc.push_back(static_cast<int32_t>(r1 - r0)); }
      }
      // the new share's connectivity first: it says how many ghost slots the planes need
      HostHaloArrays halo;
      HostMeshArrays m = arrays_of(m_pending.forest, r, R, &halo);
      this->resize(static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements);
      const int32_t   nh = static_cast<int32_t>(b - a);
      float_type*     t  = m_pending.tmp;
      auto vars = [](float_type* const p[5]) {
        if constexpr (std::is_same_v<float_type, double>) { T8gpuVars_f64 v; for (int k = 0; k < 5; k++) v.p[k] = p[k]; return v; }
        else { T8gpuVars_f32 v; for (int k = 0; k < 5; k++) v.p[k] = p[k]; return v; }
      };
      float_type* src[5];
      float_type* dst[5];
      for (int k = 0; k < 5; k++) {
        src[k] = t + static_cast<size_t>(k) * std::max(nh, 1);
        dst[k] = this->get_own_variable(step, static_cast<variable_index_type>(k));
      }
      m_transport->repartition(static_cast<int>(sp.size()), sp.data(), sf.data(), sc.data(), static_cast<int>(rp.size()), rp.data(), rf.data(),
                               rc.data(), vars(src), t + 5 * static_cast<size_t>(std::max(nh, 1)), vars(dst), this->get_own_volume(), 1);
      (void)hipFree(m_pending.tmp);
      t8gpu_synth_mesh_destroy(m_forest);
      m_forest  = m_pending.forest;
      m_pending = Pending{};
      m_halo_host = std::move(halo);
      rebuild_connectivity(m);
      // the volumes of the ghost slots come with the connectivity (the owned ones arrived with the elements)
      if (m.num_ghost_elements > 0) {
        std::vector<float_type> gv(m.volumes.begin() + m.num_local_elements, m.volumes.end());
        T8GPU_CUDA_CHECK_ERROR(hipMemcpy(this->get_own_volume() + m.num_local_elements, gv.data(), sizeof(float_type) * gv.size(), hipMemcpyHostToDevice));
      }
    }

    /// Refresh the ghost mirror slots [N, N + G) of the five planes of `step` from their owners (several ranks only; a
    /// no-op on one). The reference needs no such call: a ghost is read through the owner's CUDA-IPC pointer
    /// (kernels.cu:164-168). Kernels that read ghost values outside iterate() -- estimate_gradient of the adapt criterion,
    /// solver.cu:245-263 -- call it first; the step drivers refresh what they read themselves.
    void refresh_ghost_layer(step_index_type step) {
      if (m_nb_ranks <= 1 || m_halo_host.peers.empty()) return;
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: refresh_ghost_layer() on %d ranks needs a transport (MeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      T8gpuHalo h{};
      h.num_elements = m_num_local_elements; h.num_ghosts = m_num_ghost_elements;
      h.n_peers = static_cast<int32_t>(m_halo_host.peers.size()); h.n_send = static_cast<int32_t>(m_halo_host.send_idx.size());
      h.cells_per_element = 1;
      h.peers = m_halo_host.peers.data(); h.send_off = m_halo_host.send_off.data(); h.recv_off = m_halo_host.recv_off.data();
      h.send_idx = m_d_send_idx; h.sendbuf = m_d_sendbuf; h.recvbuf = m_d_recvbuf;
      if constexpr (std::is_same_v<float_type, double>) {
        T8gpuVars_f64 v; for (int k = 0; k < 5; k++) v.p[k] = this->get_own_variable(step, static_cast<variable_index_type>(k));
        m_transport->halo_exchange(h, v);
      } else {
        T8gpuVars_f32 v; for (int k = 0; k < 5; k++) v.p[k] = this->get_own_variable(step, static_cast<variable_index_type>(k));
        m_transport->halo_exchange(h, v);
      }
    }
    [[nodiscard]] HostHaloArrays const& host_halo() const { return m_halo_host; }
    [[nodiscard]] int comm_rank() const { return m_comm_rank; }
    [[nodiscard]] int comm_size() const { return m_nb_ranks; }

    /// mesh_manager.inl:333-481: face lists, normals, areas, ghost slots of the current forest -> device arrays.
    /// adapt() already leaves them current; calling this again is harmless (the reference requires the call).
    void compute_connectivity_information() {
      if (m_forest) rebuild_connectivity(arrays_of(m_forest, m_comm_rank, m_nb_ranks, m_nb_ranks > 1 ? &m_halo_host : nullptr));
    }

    /// MeshManager::adapt (mesh_manager.inl:196-330), single rank: the reference's adapt callback on the criteria
    /// (refine above `threshold`, coarsen a family whose first four members are below it; :125-162), 2:1 balance,
    /// the data-transfer kernel adapt_variables_and_volume (:165-193) from `step` into the new buffers, new
    /// connectivity. Only `step` and the volume are valid afterwards, as in the reference.
    void adapt(std::vector<float_type> const& refinement_criteria, step_index_type step, double threshold = 10.0) {
      if (!m_forest) {
        std::fprintf(stderr, "t8gpu: adapt() needs a manager constructed from a forest\n");
        std::abort();
      }
      if (m_nb_ranks > 1) {
        adapt_partitioned(refinement_criteria, step, threshold);
        return;
      }
      const int32_t n_old = m_num_local_elements;
      std::vector<double> crit(refinement_criteria.begin(), refinement_criteria.end());
      std::vector<int8_t> marks(static_cast<size_t>(n_old));
      t8gpu_synth_mesh_marks(m_forest, crit.data(), threshold, m_min_level, m_max_level, 4, marks.data());
      void* new_forest = t8gpu_synth_mesh_adapt(m_forest, marks.data());
      if (!new_forest) {
        std::fprintf(stderr, "t8gpu: forest adaptation failed\n");
        std::abort();
      }
      const int32_t        n_new = static_cast<int32_t>(t8gpu_synth_mesh_num_elements(new_forest));
      std::vector<int32_t> adapt_data(static_cast<size_t>(n_new) + 1);
      if (t8gpu_synth_mesh_adapt_data(m_forest, new_forest, adapt_data.data()) != 0) std::abort();
      // device: transfer into temporary planes (5 variables + volume), then into the (possibly re-allocated) manager
      int32_t*    d_ad  = nullptr;
      float_type* d_tmp = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_ad, sizeof(int32_t) * adapt_data.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_ad, adapt_data.data(), sizeof(int32_t) * adapt_data.size(), hipMemcpyHostToDevice));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_tmp, sizeof(float_type) * 6 * static_cast<size_t>(std::max(n_new, 1))));
      auto old_vars = this->get_own_variables(step);
      if constexpr (std::is_same_v<float_type, double>) {
        T8gpuVars_f64 o, n;
        for (int k = 0; k < 5; k++) {
          o.p[k] = old_vars.get(k);
          n.p[k] = d_tmp + static_cast<size_t>(k) * n_new;
        }
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_adapt_variables_and_volume_f64(
            n_new, static_cast<int>(m_mesh_dim), d_ad, o, n, this->get_own_volume(), d_tmp + 5 * static_cast<size_t>(n_new), nullptr)));
      } else {
        T8gpuVars_f32 o, n;
        for (int k = 0; k < 5; k++) {
          o.p[k] = old_vars.get(k);
          n.p[k] = d_tmp + static_cast<size_t>(k) * n_new;
        }
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_adapt_variables_and_volume_f32(
            n_new, static_cast<int>(m_mesh_dim), d_ad, o, n, this->get_own_volume(), d_tmp + 5 * static_cast<size_t>(n_new), nullptr)));
      }
      T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
      this->resize(static_cast<size_t>(n_new));
      for (int k = 0; k < 5; k++)
        this->set_variable(step, static_cast<variable_index_type>(k), d_tmp + static_cast<size_t>(k) * n_new);
      this->set_volume(d_tmp + 5 * static_cast<size_t>(n_new));
      (void)hipFree(d_ad);
      (void)hipFree(d_tmp);
      t8gpu_synth_mesh_destroy(m_forest);
      m_forest = new_forest;
      rebuild_connectivity(arrays_of(m_forest, 0, 1, nullptr));
    }

   private:
    /// adapt() on several ranks (t8gpu_amd/amr.py: PartitionedAdapt, the same scheme in C++): the forest description is
    /// replicated, so the criteria of all ranks are gathered, every rank evaluates the reference's adapt callback on the whole
    /// array (families cut by a rank boundary are not coarsened: their members' data live on two ranks), adapts the forest,
    /// and transfers ITS OWN elements on the device (adapt_variables_and_volume) into temporary planes. The result is what the
    /// reference holds after its adapt(): adapted elements on their old owners. partition() then ships them.
    void adapt_partitioned(std::vector<float_type> const& refinement_criteria, step_index_type step, double threshold) {
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: adapt() on %d ranks needs a transport (MeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      if (m_pending.forest) {   // adapt() twice without partition(): drop the first
        t8gpu_synth_mesh_destroy(m_pending.forest);
        (void)hipFree(m_pending.tmp);
        m_pending = Pending{};
      }
      const int     R = m_nb_ranks, r = m_comm_rank;
      const int64_t n_glob = t8gpu_synth_mesh_num_elements(m_forest);
      std::vector<int64_t> old_off(static_cast<size_t>(R) + 1);
      for (int q = 0; q <= R; q++) old_off[q] = n_glob * q / R;
      const int64_t n_mine = old_off[r + 1] - old_off[r];
      if (static_cast<int64_t>(refinement_criteria.size()) < n_mine) std::abort();
      // 1. all criteria on every rank
      std::vector<double> mine(refinement_criteria.begin(), refinement_criteria.begin() + n_mine), all(static_cast<size_t>(n_glob));
      double *d_mine = nullptr, *d_all = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_mine, sizeof(double) * std::max<int64_t>(n_mine, 1)));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_all, sizeof(double) * std::max<int64_t>(n_glob, 1)));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_mine, mine.data(), sizeof(double) * n_mine, hipMemcpyHostToDevice));
      m_transport->allgatherv(d_mine, d_all, old_off.data());
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(all.data(), d_all, sizeof(double) * n_glob, hipMemcpyDeviceToHost));
      (void)hipFree(d_mine);
      (void)hipFree(d_all);
      // 2. the adapt callback on the whole forest, families split by a rank boundary left alone; the new forest
      std::vector<int8_t> marks(static_cast<size_t>(n_glob));
      t8gpu_synth_mesh_marks(m_forest, all.data(), threshold, m_min_level, m_max_level, 4, marks.data());
      t8gpu_synth_mesh_unmark_split_families(m_forest, marks.data(), old_off.data() + 1, R - 1);
      void* new_forest = t8gpu_synth_mesh_adapt(m_forest, marks.data());
      if (!new_forest) {
        std::fprintf(stderr, "t8gpu: forest adaptation failed\n");
        std::abort();
      }
      const int64_t        n_new = t8gpu_synth_mesh_num_elements(new_forest);
      std::vector<int32_t> adapt_data(static_cast<size_t>(n_new) + 1);
      if (t8gpu_synth_mesh_adapt_data(m_forest, new_forest, adapt_data.data()) != 0) std::abort();
      // new elements made from rank q's old elements: [have_off[q], have_off[q + 1])
      m_pending.have_off.assign(static_cast<size_t>(R) + 1, n_new);
      for (int q = 0; q < R; q++)
        m_pending.have_off[q] = std::lower_bound(adapt_data.begin(), adapt_data.begin() + n_new, static_cast<int32_t>(old_off[q])) - adapt_data.begin();
      const int64_t a = m_pending.have_off[r], b = m_pending.have_off[r + 1];
      const int32_t nh = static_cast<int32_t>(b - a);
      // 3. this rank's elements through the data-transfer kernel into 5 + 1 temporary planes of nh values
      std::vector<int32_t> local(static_cast<size_t>(nh) + 1);
      for (int32_t i = 0; i <= nh; i++) local[i] = adapt_data[a + i] - static_cast<int32_t>(old_off[r]);
      int32_t* d_ad = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_ad, sizeof(int32_t) * local.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_ad, local.data(), sizeof(int32_t) * local.size(), hipMemcpyHostToDevice));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_pending.tmp, sizeof(float_type) * 6 * static_cast<size_t>(std::max(nh, 1))));
      auto old_vars = this->get_own_variables(step);
      if (nh > 0) {
        float_type* t = m_pending.tmp;
        if constexpr (std::is_same_v<float_type, double>) {
          T8gpuVars_f64 o, n;
          for (int k = 0; k < 5; k++) { o.p[k] = old_vars.get(k); n.p[k] = t + static_cast<size_t>(k) * nh; }
          T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_adapt_variables_and_volume_f64(nh, static_cast<int>(m_mesh_dim), d_ad, o, n, this->get_own_volume(), t + 5 * static_cast<size_t>(nh), nullptr)));
        } else {
          T8gpuVars_f32 o, n;
          for (int k = 0; k < 5; k++) { o.p[k] = old_vars.get(k); n.p[k] = t + static_cast<size_t>(k) * nh; }
          T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_adapt_variables_and_volume_f32(nh, static_cast<int>(m_mesh_dim), d_ad, o, n, this->get_own_volume(), t + 5 * static_cast<size_t>(nh), nullptr)));
        }
      }
      T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
      (void)hipFree(d_ad);
      m_pending.forest = new_forest;
    }

   public:
    [[nodiscard]] void const* forest() const { return m_forest; }
    [[nodiscard]] HostMeshArrays const& host_arrays() const { return m_host; }

    ~MeshManager() {
      if (m_forest) t8gpu_synth_mesh_destroy(m_forest);
      if (m_pending.forest) t8gpu_synth_mesh_destroy(m_pending.forest);
      (void)hipFree(m_pending.tmp);
      free_connectivity();
      (void)hipFree(m_staging);
    }
    MeshManager(MeshManager const&)            = delete;
    MeshManager& operator=(MeshManager const&) = delete;

    [[nodiscard]] MeshConnectivityAccessor<float_type, dim> get_connectivity_information() const {
      return {m_ranks, m_indices, m_face_neighbors, m_face_normals, m_face_surfaces, m_num_local_faces, m_num_local_boundary_faces};
    }
    [[nodiscard]] t8_locidx_t get_num_local_elements() const { return m_num_local_elements; }
    [[nodiscard]] t8_locidx_t get_num_ghost_elements() const { return m_num_ghost_elements; }
    [[nodiscard]] t8_locidx_t get_num_local_faces() const { return m_num_local_faces; }
    [[nodiscard]] t8_locidx_t get_num_local_boundary_faces() const { return m_num_local_boundary_faces; }

    /// Named host array of doubles ready for the writer (mesh_manager.h: HostVariableInfo).
    struct HostVariableInfo {
      int                       m_type = T8GPU_VTK_SCALAR;  // T8GPU_VTK_SCALAR | T8GPU_VTK_VECTOR
      std::unique_ptr<double[]> m_data;
      std::string               m_name;
    };

    /// mesh_manager.h:326: one variable in one file
    void save_variable_to_vtk(step_index_type step, variable_index_type variable, std::string const& prefix) const {
      std::vector<HostVariableInfo> v;
      v.push_back(get_host_scalar_variable(step, variable, "variable"));
      save_variables_to_vtk(std::move(v), prefix);
    }

    /// mesh_manager.inl:516-545: one variable of one step, cast to double (on the device), on the host.
    [[nodiscard]] HostVariableInfo get_host_scalar_variable(step_index_type step, variable_index_type variable,
                                                            std::string const& name) const {
      const size_t n = static_cast<size_t>(m_num_local_elements);
      double*      d = staging(n);
      if constexpr (std::is_same_v<float_type, double>)
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_scalar_variable_f64(n, this->get_own_variable(step, variable), d, nullptr)));
      else
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_scalar_variable_f32(n, this->get_own_variable(step, variable), d, nullptr)));
      return {T8GPU_VTK_SCALAR, fetch(d, n), name};
    }
    /// mesh_manager.inl:547-586: three variables as interleaved xyz doubles.
    [[nodiscard]] HostVariableInfo get_host_vector_variable(step_index_type step, std::array<variable_index_type, 3> variables,
                                                            std::string const& name) const {
      const size_t      n = static_cast<size_t>(m_num_local_elements);
      double*           d = staging(3 * n);
      float_type const* v[3];
      for (int k = 0; k < 3; k++) v[k] = this->get_own_variable(step, variables[k]);
      if constexpr (std::is_same_v<float_type, double>)
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_vector_variable_f64(n, v[0], v[1], v[2], d, nullptr)));
      else
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_vector_variable_f32(n, v[0], v[1], v[2], d, nullptr)));
      return {T8GPU_VTK_VECTOR, fetch(d, 3 * n), name};
    }
    /// mesh_manager.inl:588-623: this rank's piece `<prefix>.vtu` (`<prefix>_RRRR.vtu` in a multi-rank run).
    void save_variables_to_vtk(std::vector<HostVariableInfo> host_variables, std::string const& prefix, int num_ranks = 1,
                               bool ascii = false) const {
      std::vector<char const*>   names;
      std::vector<int32_t>       comps;
      std::vector<double const*> data;
      for (auto const& h : host_variables) {
        names.push_back(h.m_name.c_str());
        comps.push_back(h.m_type);
        data.push_back(h.m_data.get());
      }
      char suffix[16] = "";
      if (num_ranks > 1) std::snprintf(suffix, sizeof suffix, "_%04d", m_rank);
      const std::string path = prefix + suffix + ".vtu";
      const int rc = t8gpu_host_write_vtu(path.c_str(), m_mesh_dim, m_num_local_elements, m_centres.data(), m_levels.data(), 1, m_rank,
                                          m_first_global, static_cast<int>(names.size()), names.data(), comps.data(), data.data(), ascii);
      if (rc != 0) {
        std::fprintf(stderr, "t8gpu: writing %s failed (code %d)\n", path.c_str(), rc);
        std::abort();
      }
    }

   private:
    using MemoryManager<VariableType, StepType>::resize;   // private here, as in the reference (mesh_manager.h:425)

    void*          m_forest    = nullptr;   // synthetic forest (owned) when constructed from one
    int            m_min_level = min_level, m_max_level = max_level;
    int            m_nb_ranks  = 1;
    HostMeshArrays m_host;

    /// HostMeshArrays of a synthetic forest on one rank: connectivity through the forest-query adapter, plus the
    /// leaf geometry the VTK members need
    static int comm_rank_of(sc_MPI_Comm comm) {
      int r = 0, n = 1;
      detail::comm_layout(comm, r, n);
      return r;
    }
    static int comm_size_of(sc_MPI_Comm comm) {
      int r = 0, n = 1;
      detail::comm_layout(comm, r, n);
      return n;
    }
    /// rank `rank` of `nranks`' share of the forest in the reference's array formats, through the forest-query adapter
    /// (csrc/host/connectivity.cpp) -- the way a t8code build would provide it; `halo` (nullable) receives the ghost lists
    static HostMeshArrays arrays_of(void* forest, int rank, int nranks, HostHaloArrays* halo) {
      T8gpuForestQuery* q = t8gpu_synth_query_create(forest, rank, nranks);
      void*             h = q ? t8gpu_host_connectivity_create(q) : nullptr;
      if (!h) {
        std::fprintf(stderr, "t8gpu: connectivity of the synthetic forest could not be built\n");
        std::abort();
      }
      int64_t c[6];
      t8gpu_host_connectivity_counts(h, c);
      HostMeshArrays m;
      m.num_local_elements = static_cast<int32_t>(c[0]); m.num_ghost_elements = static_cast<int32_t>(c[1]);
      m.num_local_faces = static_cast<int32_t>(c[2]); m.num_local_boundary_faces = static_cast<int32_t>(c[3]);
      m.rank = rank;
      m.face_neighbors.resize(2 * c[2] + c[3]);
      m.face_normals.resize(3 * (c[2] + c[3]));
      m.face_surfaces.resize(c[2] + c[3]);
      m.volumes.resize(c[0] + c[1]);
      HostHaloArrays hh;
      hh.peers.resize(c[4]); hh.recv_off.resize(c[4] + 1); hh.send_off.resize(c[4] + 1); hh.send_idx.resize(c[5]);
      t8gpu_host_connectivity_arrays(h, m.face_neighbors.data(), m.face_normals.data(), m.face_surfaces.data(), m.volumes.data(),
                                     hh.peers.data(), hh.recv_off.data(), hh.send_off.data(), hh.send_idx.data());
      if (halo) *halo = std::move(hh);
      t8gpu_host_connectivity_destroy(h);
      t8gpu_synth_query_destroy(q);
      if constexpr (dim != 3) {   // the adapter hands out xyz; MeshConnectivityAccessor<ft, dim> strides by `dim`
        static_assert(dim == 2, "face normals have 2 or 3 components");
        const size_t        nf = static_cast<size_t>(c[2] + c[3]);
        std::vector<double> nd(dim * nf);
        for (size_t i = 0; i < nf; i++)
          for (size_t k = 0; k < dim; k++) nd[dim * i + k] = m.face_normals[3 * i + k];
        m.face_normals.swap(nd);
      }
      void* part = t8gpu_synth_part_create(forest, rank, nranks, 0, 3);
      m.mesh_dim = t8gpu_synth_mesh_dim(forest);
      m.first_global_element = t8gpu_synth_mesh_num_elements(forest) * rank / nranks;
      m.levels.resize(c[0] + c[1]);           // (the provider lists owned + ghost elements; the manager keeps the owned ones)
      m.centres.resize(3 * (c[0] + c[1]));
      t8gpu_synth_part_elements(part, m.levels.data(), nullptr, m.centres.data());
      m.levels.resize(c[0]);
      m.centres.resize(3 * c[0]);
      t8gpu_synth_part_destroy(part);
      return m;
    }
    void free_connectivity() {
      (void)hipFree(m_d_send_idx);
      (void)hipFree(m_d_sendbuf);
      (void)hipFree(m_d_recvbuf);
      m_d_send_idx = nullptr;
      m_d_sendbuf = m_d_recvbuf = nullptr;
      (void)hipFree(m_ranks);
      (void)hipFree(m_indices);
      (void)hipFree(m_face_neighbors);
      (void)hipFree(m_face_normals);
      (void)hipFree(m_face_surfaces);
      m_ranks = nullptr; m_indices = nullptr; m_face_neighbors = nullptr; m_face_normals = nullptr; m_face_surfaces = nullptr;
    }
    /// compute_connectivity_information (mesh_manager.inl:333-481): device copies of the face arrays
    void rebuild_connectivity(HostMeshArrays const& m) {
      free_connectivity();
      m_host                     = m;
      m_num_local_elements       = m.num_local_elements;
      m_num_ghost_elements       = m.num_ghost_elements;
      m_num_local_faces          = m.num_local_faces;
      m_num_local_boundary_faces = m.num_local_boundary_faces;
      m_rank = m.rank; m_mesh_dim = m.mesh_dim; m_first_global = m.first_global_element;
      m_centres = m.centres;
      m_levels  = m.levels;
      const size_t tot = static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements;
      std::vector<int>         ranks(tot, m.rank);
      std::vector<t8_locidx_t> indices(tot);
      for (size_t i = 0; i < tot; i++) indices[i] = static_cast<t8_locidx_t>(i);
      upload(m_ranks, ranks);
      upload(m_indices, indices);
      upload(m_face_neighbors, m.face_neighbors);
      upload(m_face_normals, std::vector<float_type>(m.face_normals.begin(), m.face_normals.end()));
      upload(m_face_surfaces, std::vector<float_type>(m.face_surfaces.begin(), m.face_surfaces.end()));
      if (m_nb_ranks > 1 && !m_halo_host.peers.empty()) {   // device side of refresh_ghost_layer()
        upload(m_d_send_idx, m_halo_host.send_idx);
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_d_sendbuf, sizeof(float_type) * (5 * m_halo_host.send_idx.size() + 1)));
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_d_recvbuf, sizeof(float_type) * (5 * static_cast<size_t>(m.num_ghost_elements) + 1)));
      }
    }

    // several ranks: the channel, the ghost lists of the current share, and what adapt() leaves for partition()
    Transport*     m_transport = nullptr;
    int            m_comm_rank = 0;
    HostHaloArrays m_halo_host;
    int32_t*       m_d_send_idx = nullptr;
    float_type*    m_d_sendbuf  = nullptr;
    float_type*    m_d_recvbuf  = nullptr;
    struct Pending {
      void*                forest = nullptr;   // the adapted forest (replicated)
      float_type*          tmp    = nullptr;   // 5 + 1 planes of (have_off[r + 1] - have_off[r]) adapted elements of this rank
      std::vector<int64_t> have_off;           // new elements made from rank q's old ones: [have_off[q], have_off[q + 1])
    } m_pending;
    int                  m_rank = 0, m_mesh_dim = 2;
    int64_t              m_first_global = 0;
    std::vector<double>  m_centres;
    std::vector<int32_t> m_levels;
    mutable double*      m_staging       = nullptr;
    mutable size_t       m_staging_count = 0;

    double* staging(size_t n) const {
      if (n > m_staging_count) {
        (void)hipFree(m_staging);
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_staging, sizeof(double) * n));
        m_staging_count = n;
      }
      return m_staging;
    }
    static std::unique_ptr<double[]> fetch(double const* d, size_t n) {
      std::unique_ptr<double[]> h = std::make_unique<double[]>(n);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(h.get(), d, sizeof(double) * n, hipMemcpyDeviceToHost));
      return h;
    }

    t8_locidx_t  m_num_local_elements = 0, m_num_ghost_elements = 0, m_num_local_faces = 0, m_num_local_boundary_faces = 0;
    int*         m_ranks          = nullptr;
    t8_locidx_t* m_indices        = nullptr;
    t8_locidx_t* m_face_neighbors = nullptr;
    float_type*  m_face_normals   = nullptr;
    float_type*  m_face_surfaces  = nullptr;

    template<typename T>
    static void upload(T*& dst, std::vector<T> const& src) {
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&dst, sizeof(T) * (src.empty() ? 1 : src.size())));
      if (!src.empty()) T8GPU_CUDA_CHECK_ERROR(hipMemcpy(dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    }
  };

  /// earlier name of the class when it is built from the synthetic provider or from host arrays
  template<typename VariableType, typename StepType, size_t dim>
  using SyntheticMeshManager = MeshManager<VariableType, StepType, dim>;

}  // namespace t8gpu

#endif  // T8GPU_HIP_MESH_MESH_MANAGER_H
