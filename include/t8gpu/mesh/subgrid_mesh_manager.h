// t8gpu/mesh/subgrid_mesh_manager.h (MI355X backend)
//
// SubgridMeshConnectivityAccessor<float_type, SubgridType>: device view of the coarse-face lists of a
// Subgrid mesh with the getters of the reference (t8gpu/mesh/subgrid_mesh_manager.h:29-216): the plain
// arrays plus face_level_difference[F] (level(right) - level(left) <= 0) and face_neighbor_offset[rank*F]
// (anchor inside the right block; subgrid_mesh_manager.inl:587-680). Normals have SubgridType::rank
// components. SubgridMeshManager<V, S, Subgrid>: the class of the reference (subgrid_mesh_manager.h:266-509) with
// the same public members (constructor (comm, scheme, cmesh, forest), initialize_variables, adapt, partition,
// compute_connectivity_information, save_variable_to_vtk, save_mesh_to_vtk, HostVariableInfo, get_host_*,
// save_variables_to_vtk, get_connectivity_information, get_num_*, min_level / max_level) on top of
// SubgridMemoryManager. As for MeshManager (mesh_manager.h) the t8code constructor is declared for every build and
// defined by the t8code adapter only; the synthetic forest / host arrays constructors make the class usable here,
// and SyntheticSubgridMeshManager is an alias kept for the earlier name.
#ifndef T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H
#define T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H

#include <t8gpu/memory/subgrid_memory_manager.h>
#include <t8gpu/mesh/mesh_manager.h>

#include <algorithm>
#include <string>
#include <vector>

namespace t8gpu {

  template<typename float_type, typename SubgridType>
  class SubgridMeshConnectivityAccessor {
    template<typename VT, typename ST, typename SG>
    friend class SubgridMeshManager;
    static constexpr int dim = SubgridType::rank;

   public:
    SubgridMeshConnectivityAccessor(SubgridMeshConnectivityAccessor const&)            = default;
    SubgridMeshConnectivityAccessor& operator=(SubgridMeshConnectivityAccessor const&) = default;

    /// public constructor: the arrays come from any provider of the reference's formats
    __host__ __device__ SubgridMeshConnectivityAccessor(int const* ranks, t8_locidx_t const* indices, t8_locidx_t const* fn,
                                                        t8_locidx_t const* level_difference, t8_locidx_t const* neighbor_offset,
                                                        float_type const* normals, float_type const* surfaces, t8_locidx_t F,
                                                        t8_locidx_t B)
        : m_ranks{ranks}, m_indices{indices}, m_face_neighbors{fn}, m_face_level_difference{level_difference},
          m_face_neighbor_offset{neighbor_offset}, m_face_normals{normals}, m_face_surfaces{surfaces},
          m_num_local_faces{F}, m_num_local_boundary_faces{B} {}

    [[nodiscard]] __host__ __device__ inline t8_locidx_t get_num_local_faces() const { return m_num_local_faces; }
    [[nodiscard]] __host__ __device__ inline t8_locidx_t get_num_local_boundary_faces() const { return m_num_local_boundary_faces; }
    [[nodiscard]] __device__ inline float_type get_face_surface(int f) const { return m_face_surfaces[f]; }
    [[nodiscard]] __device__ inline float_type get_boundary_face_surface(int f) const { return m_face_surfaces[m_num_local_faces + f]; }
    [[nodiscard]] __device__ inline std::array<float_type, dim> get_face_normal(int f) const { return normal_at(f); }
    [[nodiscard]] __device__ inline std::array<float_type, dim> get_boundary_face_normal(int f) const { return normal_at(m_num_local_faces + f); }
    [[nodiscard]] __device__ inline t8_locidx_t get_face_level_difference(int f) const { return m_face_level_difference[f]; }
    [[nodiscard]] __device__ inline std::array<t8_locidx_t, SubgridType::rank> get_face_neighbor_offset(int f) const {
      std::array<t8_locidx_t, SubgridType::rank> o{};
      for (int k = 0; k < dim; k++) o[k] = m_face_neighbor_offset[dim * f + k];
      return o;
    }
    [[nodiscard]] __device__ inline std::array<t8_locidx_t, 2> get_face_neighbor_indices(int f) const {
      return {m_face_neighbors[2 * f], m_face_neighbors[2 * f + 1]};
    }
    [[nodiscard]] __device__ inline t8_locidx_t get_boundary_face_neighbor_index(int f) const { return m_face_neighbors[2 * m_num_local_faces + f]; }
    [[nodiscard]] __device__ inline t8_locidx_t get_element_owner_rank(int e) const { return m_ranks[e]; }
    [[nodiscard]] __device__ inline t8_locidx_t get_element_owner_remote_index(int e) const { return m_indices[e]; }

   private:
    int const*         m_ranks;
    t8_locidx_t const* m_indices;
    t8_locidx_t const* m_face_neighbors;
    t8_locidx_t const* m_face_level_difference;
    t8_locidx_t const* m_face_neighbor_offset;
    float_type const*  m_face_normals;
    float_type const*  m_face_surfaces;
    t8_locidx_t        m_num_local_faces;
    t8_locidx_t        m_num_local_boundary_faces;

    __device__ inline std::array<float_type, dim> normal_at(int slot) const {
      std::array<float_type, dim> n{};
      for (int k = 0; k < dim; k++) n[k] = m_face_normals[dim * slot + k];
      return n;
    }
  };

  /// One rank's Subgrid mesh in the reference's array formats (subgrid_mesh_manager.h:29-216); normals have
  /// `rank` components, level differences are level(right) - level(left) <= 0.
  struct HostSubgridMeshArrays {
    int32_t num_local_elements = 0, num_ghost_elements = 0, num_local_faces = 0, num_local_boundary_faces = 0, rank = 3;
    int     mpirank = 0;
    std::vector<int32_t> face_neighbors, face_level_difference, face_neighbor_offset;
    std::vector<double>  face_normals, face_surfaces, volumes;
    // only needed by the VTK members: geometry of the owned blocks on the unit domain
    int64_t              first_global_element = 0;
    std::vector<double>  centres;   // [N][3]
    std::vector<int32_t> levels;    // [N]
  };

  template<typename VariableType, typename StepType, typename SubgridType>
  class SubgridMeshManager : public SubgridMemoryManager<VariableType, StepType, SubgridType> {
   public:
    using float_type                  = typename variable_traits<VariableType>::float_type;
    using variable_index_type         = typename variable_traits<VariableType>::index_type;
    static constexpr int nb_variables = variable_traits<VariableType>::nb_variables;
    static constexpr int dim          = SubgridType::rank;

    using step_index_type            = typename step_traits<StepType>::index_type;
    static constexpr size_t nb_steps = step_traits<StepType>::nb_steps;

    static constexpr t8_locidx_t min_level = 1;   // subgrid_mesh_manager.h:276-277
    static constexpr t8_locidx_t max_level = 6;

    /// subgrid_mesh_manager.h:288 / .inl:20-75: takes ownership of cmesh and forest. Declared for every build, DEFINED
    /// by the t8code adapter only; see MeshManager.
    SubgridMeshManager(sc_MPI_Comm comm, t8_scheme_cxx_t* scheme, t8_cmesh_t cmesh, t8_forest_t forest);

    explicit SubgridMeshManager(HostSubgridMeshArrays const& m, sc_MPI_Comm comm = sc_MPI_COMM_WORLD)
        : SubgridMemoryManager<VariableType, StepType, SubgridType>(static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements, comm),
          m_host{m} {
      detail::comm_layout(comm, m_comm_rank, m_nb_ranks);
      rebuild_connectivity(m);
      const size_t            tot = static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements;
      std::vector<float_type> vol(m.volumes.begin(), m.volumes.end());
      vol.resize(tot, float_type(1));
      this->set_volume(vol);
    }
    /// From a synthetic forest (owned afterwards): stands for SubgridMeshManager(comm, scheme, cmesh, forest)
    /// (subgrid_mesh_manager.inl:3-60); connectivity through the forest-query adapter. On several ranks every rank passes
    /// its own handle of the same forest and owns an equal share of the curve; set_transport() must follow (see MeshManager).
    explicit SubgridMeshManager(void* synth_mesh, int lowest_level = min_level, int highest_level = max_level,
                                sc_MPI_Comm comm = sc_MPI_COMM_WORLD)
        : SubgridMeshManager(arrays_of(synth_mesh, comm_rank_of(comm), comm_size_of(comm), nullptr), comm) {
      m_forest    = synth_mesh;
      m_min_level = lowest_level;
      m_max_level = highest_level;
      if (m_nb_ranks > 1) rebuild_connectivity(arrays_of(m_forest, m_comm_rank, m_nb_ranks, &m_halo_host));   // (+ the halo lists)
    }
    /// The channel adapt() / partition() / refresh_ghost_layer() use on several ranks (not owned). See backend/transport.h.
    void set_transport(Transport* transport) { m_transport = transport; }

    /// subgrid_mesh_manager.inl:144-194: `func(accessor, forest, tree_idx, element, e_idx)` fills ONE value per
    /// variable and block in a host MemoryAccessorOwn; every subcell of the block gets that value in Step 0
    /// (copy_variables_coarse_mesh_to_fine). With the synthetic provider `element` is a SyntheticElement.
    template<typename Func>
    void initialize_variables(Func func) {
      constexpr size_t S = SubgridType::size;
      const size_t     n = static_cast<size_t>(m_host.num_local_elements), tot = n + static_cast<size_t>(m_host.num_ghost_elements);
      std::array<std::vector<float_type>, nb_variables> coarse{};
      std::array<float_type*, nb_variables>             array{};
      for (size_t k = 0; k < static_cast<size_t>(nb_variables); k++) {
        coarse[k].resize(n);
        array[k] = coarse[k].data();
      }
      MemoryAccessorOwn<VariableType> host_variable_memory{array};
      for (size_t e = 0; e < n; e++) {
        SyntheticElement el{{m_host.centres[3 * e], m_host.centres[3 * e + 1], m_host.centres[3 * e + 2]}, m_host.levels[e],
                            m_host.volumes[e]};
        func(host_variable_memory, reinterpret_cast<t8_forest_t>(m_forest), t8_locidx_t{0},
             reinterpret_cast<t8_element_t const*>(&el), static_cast<t8_locidx_t>(e));
      }
      std::vector<float_type> fine(tot * S, float_type(0));
      for (size_t k = 0; k < static_cast<size_t>(nb_variables); k++) {
        for (size_t e = 0; e < n; e++) std::fill(fine.begin() + e * S, fine.begin() + (e + 1) * S, coarse[k][e]);
        this->set_variable(static_cast<step_index_type>(0), static_cast<variable_index_type>(k), fine);
      }
    }

    /// subgrid_mesh_manager.h:327 (the reference's signature)
    void adapt(thrust::host_vector<float_type> const& refinement_criteria, step_index_type step) {
      adapt(std::vector<float_type>(refinement_criteria.begin(), refinement_criteria.end()), step);
    }

    /// subgrid_mesh_manager.inl:1217-1369. As MeshManager::partition with whole blocks for elements: every run of adapted
    /// blocks goes to its owner in the new equal split (t8gpu_hip_repartition_* with cells_per_element = 4^rank: the old owner
    /// sends where the reference's new owner pulls through CUDA-IPC pointers, partition_data<<<>>> :1217-1250), the new forest is
    /// installed and the connectivity rebuilt. Only `step` and the volumes are valid afterwards. The identity on one rank or
    /// when no adapt() is pending.
    void partition(step_index_type step) {
      if (!m_pending.forest) return;
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: partition() on %d ranks needs a transport (SubgridMeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      constexpr size_t S = SubgridType::size;
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
      HostHaloArrays        halo;
      HostSubgridMeshArrays m = arrays_of(m_pending.forest, r, R, &halo);   // first: it says how many ghost blocks the planes need
      this->resize(static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements);
      const size_t nh = static_cast<size_t>(std::max<int64_t>(b - a, 1));
      float_type*  src[5];
      float_type*  dst[5];
      for (int k = 0; k < 5; k++) {
        src[k] = m_pending.tmp + static_cast<size_t>(k) * S * nh;
        dst[k] = static_cast<float_type*>(this->get_own_variable(step, static_cast<variable_index_type>(k)));
      }
      auto vars = [](float_type* const p[5]) {
        if constexpr (std::is_same_v<float_type, double>) { T8gpuVars_f64 v; for (int k = 0; k < 5; k++) v.p[k] = p[k]; return v; }
        else { T8gpuVars_f32 v; for (int k = 0; k < 5; k++) v.p[k] = p[k]; return v; }
      };
      m_transport->repartition(static_cast<int>(sp.size()), sp.data(), sf.data(), sc.data(), static_cast<int>(rp.size()), rp.data(), rf.data(),
                               rc.data(), vars(src), m_pending.vol, vars(dst), this->get_own_volume(), static_cast<int>(S));
      (void)hipFree(m_pending.tmp);
      (void)hipFree(m_pending.vol);
      t8gpu_synth_mesh_destroy(m_forest);
      m_forest    = m_pending.forest;
      m_pending   = Pending{};
      m_halo_host = std::move(halo);
      rebuild_connectivity(m);
      if (m.num_ghost_elements > 0) {   // the ghost blocks' volumes come with the connectivity
        std::vector<float_type> gv(m.volumes.begin() + m.num_local_elements, m.volumes.end());
        T8GPU_CUDA_CHECK_ERROR(hipMemcpy(this->get_own_volume() + m.num_local_elements, gv.data(), sizeof(float_type) * gv.size(), hipMemcpyHostToDevice));
      }
      drop_scratch();
    }

    /// Refresh the ghost BLOCKS [N, N + G) of the five planes of `step` from their owners (several ranks only; see
    /// MeshManager::refresh_ghost_layer). The step drivers refresh what they read themselves.
    void refresh_ghost_layer(step_index_type step) {
      if (m_nb_ranks <= 1 || m_halo_host.peers.empty()) return;
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: refresh_ghost_layer() on %d ranks needs a transport (SubgridMeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      T8gpuHalo h{};
      h.num_elements = m_host.num_local_elements; h.num_ghosts = m_host.num_ghost_elements;
      h.n_peers = static_cast<int32_t>(m_halo_host.peers.size()); h.n_send = static_cast<int32_t>(m_halo_host.send_idx.size());
      h.cells_per_element = static_cast<int32_t>(SubgridType::size);
      h.peers = m_halo_host.peers.data(); h.send_off = m_halo_host.send_off.data(); h.recv_off = m_halo_host.recv_off.data();
      h.send_idx = m_d_send_idx; h.sendbuf = m_d_sendbuf; h.recvbuf = m_d_recvbuf;
      if constexpr (std::is_same_v<float_type, double>) {
        T8gpuVars_f64 v; for (int k = 0; k < 5; k++) v.p[k] = static_cast<double*>(this->get_own_variable(step, static_cast<variable_index_type>(k)));
        m_transport->halo_exchange(h, v);
      } else {
        T8gpuVars_f32 v; for (int k = 0; k < 5; k++) v.p[k] = static_cast<float*>(this->get_own_variable(step, static_cast<variable_index_type>(k)));
        m_transport->halo_exchange(h, v);
      }
    }
    [[nodiscard]] HostHaloArrays const& host_halo() const { return m_halo_host; }
    [[nodiscard]] int comm_rank() const { return m_comm_rank; }
    [[nodiscard]] int comm_size() const { return m_nb_ranks; }

    /// subgrid_mesh_manager.inl:560-961: coarse-face lists, level differences, neighbour offsets -> device arrays
    void compute_connectivity_information() {
      if (m_forest) rebuild_connectivity(arrays_of(m_forest, m_comm_rank, m_nb_ranks, m_nb_ranks > 1 ? &m_halo_host : nullptr));
    }

    /// SubgridMeshManager::adapt (subgrid_mesh_manager.inl:428-558), single rank: adapt callback on the per-block
    /// criteria, 2:1 balance, block-wise transfer adapt_variables + adapt_volume (:246-425) from `step`, new
    /// connectivity. Only `step` and the volumes are valid afterwards.
    void adapt(std::vector<float_type> const& refinement_criteria, step_index_type step, double threshold = 0.02) {
      if (!m_forest) {
        std::fprintf(stderr, "t8gpu: adapt() needs a manager constructed from a forest\n");
        std::abort();
      }
      if (m_nb_ranks > 1) {
        adapt_partitioned(refinement_criteria, step, threshold);
        return;
      }
      constexpr size_t     S = SubgridType::size;
      std::vector<double>  crit(refinement_criteria.begin(), refinement_criteria.end());
      std::vector<int8_t>  marks(static_cast<size_t>(m_host.num_local_elements));
      t8gpu_synth_mesh_marks(m_forest, crit.data(), threshold, m_min_level, m_max_level, 4, marks.data());
      void* new_forest = t8gpu_synth_mesh_adapt(m_forest, marks.data());
      if (!new_forest) std::abort();
      const int32_t        n_new = static_cast<int32_t>(t8gpu_synth_mesh_num_elements(new_forest));
      std::vector<int32_t> adapt_data(static_cast<size_t>(n_new) + 1);
      if (t8gpu_synth_mesh_adapt_data(m_forest, new_forest, adapt_data.data()) != 0) std::abort();
      int32_t*    d_ad  = nullptr;
      float_type *d_tmp = nullptr, *d_vol = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_ad, sizeof(int32_t) * adapt_data.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_ad, adapt_data.data(), sizeof(int32_t) * adapt_data.size(), hipMemcpyHostToDevice));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_tmp, sizeof(float_type) * 5 * S * static_cast<size_t>(std::max(n_new, 1))));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_vol, sizeof(float_type) * static_cast<size_t>(std::max(n_new, 1))));
      auto old_vars = this->get_own_variables(step);
      auto run = [&](auto o, auto n, auto fn) {
        for (int k = 0; k < 5; k++) {
          o.p[k] = old_vars.data(static_cast<variable_index_type>(k));
          n.p[k] = d_tmp + static_cast<size_t>(k) * S * n_new;
        }
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(fn(SubgridType::rank, n_new, d_ad, o, n, this->get_own_volume(), d_vol, nullptr)));
      };
      if constexpr (std::is_same_v<float_type, double>)
        run(T8gpuVars_f64{}, T8gpuVars_f64{}, t8gpu_hip_subgrid_adapt_variables_and_volume_f64);
      else
        run(T8gpuVars_f32{}, T8gpuVars_f32{}, t8gpu_hip_subgrid_adapt_variables_and_volume_f32);
      T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
      this->resize(static_cast<size_t>(n_new));
      for (int k = 0; k < 5; k++) this->set_variable(step, static_cast<variable_index_type>(k), d_tmp + static_cast<size_t>(k) * S * n_new);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(this->get_own_volume(), d_vol, sizeof(float_type) * n_new, hipMemcpyDeviceToDevice));
      (void)hipFree(d_ad);
      (void)hipFree(d_tmp);
      (void)hipFree(d_vol);
      t8gpu_synth_mesh_destroy(m_forest);
      m_forest = new_forest;
      rebuild_connectivity(arrays_of(m_forest, 0, 1, nullptr));
      drop_scratch();
    }

   private:
    /// adapt() on several ranks, the scheme of MeshManager::adapt_partitioned with blocks for elements: criteria of all ranks
    /// gathered, the reference's adapt callback evaluated on the whole (replicated) forest, families cut by a rank boundary left
    /// alone, this rank's blocks through the block-wise data-transfer kernel into temporary planes. partition() ships them.
    void adapt_partitioned(std::vector<float_type> const& refinement_criteria, step_index_type step, double threshold) {
      if (!m_transport) {
        std::fprintf(stderr, "t8gpu: adapt() on %d ranks needs a transport (SubgridMeshManager::set_transport)\n", m_nb_ranks);
        std::abort();
      }
      if (m_pending.forest) {   // adapt() twice without partition(): drop the first
        t8gpu_synth_mesh_destroy(m_pending.forest);
        (void)hipFree(m_pending.tmp);
        (void)hipFree(m_pending.vol);
        m_pending = Pending{};
      }
      constexpr size_t S = SubgridType::size;
      const int     R = m_nb_ranks, r = m_comm_rank;
      const int64_t n_glob = t8gpu_synth_mesh_num_elements(m_forest);
      std::vector<int64_t> old_off(static_cast<size_t>(R) + 1);
      for (int q = 0; q <= R; q++) old_off[q] = n_glob * q / R;
      const int64_t n_mine = old_off[r + 1] - old_off[r];
      if (static_cast<int64_t>(refinement_criteria.size()) < n_mine) std::abort();
      std::vector<double> mine(refinement_criteria.begin(), refinement_criteria.begin() + n_mine), all(static_cast<size_t>(n_glob));
      double *d_mine = nullptr, *d_all = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_mine, sizeof(double) * std::max<int64_t>(n_mine, 1)));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_all, sizeof(double) * std::max<int64_t>(n_glob, 1)));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_mine, mine.data(), sizeof(double) * n_mine, hipMemcpyHostToDevice));
      m_transport->allgatherv(d_mine, d_all, old_off.data());
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(all.data(), d_all, sizeof(double) * n_glob, hipMemcpyDeviceToHost));
      (void)hipFree(d_mine);
      (void)hipFree(d_all);
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
      m_pending.have_off.assign(static_cast<size_t>(R) + 1, n_new);
      for (int q = 0; q < R; q++)
        m_pending.have_off[q] = std::lower_bound(adapt_data.begin(), adapt_data.begin() + n_new, static_cast<int32_t>(old_off[q])) - adapt_data.begin();
      const int64_t a = m_pending.have_off[r], b = m_pending.have_off[r + 1];
      const int32_t nh = static_cast<int32_t>(b - a);
      std::vector<int32_t> local(static_cast<size_t>(nh) + 1);
      for (int32_t i = 0; i <= nh; i++) local[i] = adapt_data[a + i] - static_cast<int32_t>(old_off[r]);
      int32_t* d_ad = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d_ad, sizeof(int32_t) * local.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d_ad, local.data(), sizeof(int32_t) * local.size(), hipMemcpyHostToDevice));
      const size_t cap = static_cast<size_t>(std::max(nh, 1));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_pending.tmp, sizeof(float_type) * 5 * S * cap));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_pending.vol, sizeof(float_type) * cap));
      if (nh > 0) {
        auto old_vars = this->get_own_variables(step);
        auto run = [&](auto o, auto n, auto fn) {
          for (int k = 0; k < 5; k++) {
            o.p[k] = old_vars.data(static_cast<variable_index_type>(k));
            n.p[k] = m_pending.tmp + static_cast<size_t>(k) * S * cap;
          }
          T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(fn(SubgridType::rank, nh, d_ad, o, n, this->get_own_volume(), m_pending.vol, nullptr)));
        };
        if constexpr (std::is_same_v<float_type, double>)
          run(T8gpuVars_f64{}, T8gpuVars_f64{}, t8gpu_hip_subgrid_adapt_variables_and_volume_f64);
        else
          run(T8gpuVars_f32{}, T8gpuVars_f32{}, t8gpu_hip_subgrid_adapt_variables_and_volume_f32);
      }
      T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
      (void)hipFree(d_ad);
      m_pending.forest = new_forest;
    }
    void drop_scratch() {   // sized for the old mesh
      (void)hipFree(m_scratch);
      (void)hipFree(m_scratch64);
      m_scratch   = nullptr;
      m_scratch64 = nullptr;
    }

   public:
    [[nodiscard]] void const* forest() const { return m_forest; }

    ~SubgridMeshManager() {
      if (m_forest) t8gpu_synth_mesh_destroy(m_forest);
      if (m_pending.forest) t8gpu_synth_mesh_destroy(m_pending.forest);
      (void)hipFree(m_pending.tmp);
      (void)hipFree(m_pending.vol);
      (void)hipFree(m_d_send_idx);
      (void)hipFree(m_d_sendbuf);
      (void)hipFree(m_d_recvbuf);
      for (void* p : {static_cast<void*>(m_ranks), static_cast<void*>(m_indices), static_cast<void*>(m_face_neighbors),
                      static_cast<void*>(m_level_difference), static_cast<void*>(m_neighbor_offset), static_cast<void*>(m_face_normals),
                      static_cast<void*>(m_face_surfaces), static_cast<void*>(m_scratch), static_cast<void*>(m_scratch64)})
        (void)hipFree(p);
    }
    SubgridMeshManager(SubgridMeshManager const&)            = delete;
    SubgridMeshManager& operator=(SubgridMeshManager const&) = delete;

    [[nodiscard]] SubgridMeshConnectivityAccessor<float_type, SubgridType> get_connectivity_information() const {
      return {m_ranks, m_indices, m_face_neighbors, m_level_difference, m_neighbor_offset, m_face_normals, m_face_surfaces,
              m_host.num_local_faces, m_host.num_local_boundary_faces};
    }
    [[nodiscard]] t8_locidx_t get_num_local_elements() const { return m_host.num_local_elements; }
    [[nodiscard]] t8_locidx_t get_num_ghost_elements() const { return m_host.num_ghost_elements; }
    [[nodiscard]] t8_locidx_t get_num_local_faces() const { return m_host.num_local_faces; }
    [[nodiscard]] t8_locidx_t get_num_local_boundary_faces() const { return m_host.num_local_boundary_faces; }
    [[nodiscard]] HostSubgridMeshArrays const& host_arrays() const { return m_host; }

    /// Named host array of doubles ready for the writer (subgrid_mesh_manager.h:387-423: HostVariableInfo)
    struct HostVariableInfo {
      int                       m_type = T8GPU_VTK_SCALAR;  // T8GPU_VTK_SCALAR | T8GPU_VTK_VECTOR
      std::unique_ptr<double[]> m_data;
      std::string               m_name;
    };

    /// subgrid_mesh_manager.h:426. One variable of one step on the host: every subcell, in the z-order of the forest
    /// refined log2(extent) times, as doubles (z-order + cast on the device, one D2H copy). The reference copies the
    /// first num_local_elements values of the block-major array only (subgrid_mesh_manager.inl:1138-1157) and its
    /// save_variables_to_vtk is commented out (:1181-1206); this is the field those two were written to deliver.
    [[nodiscard]] HostVariableInfo get_host_scalar_variable(step_index_type step, variable_index_type variable,
                                                            std::string const& name) const {
      const size_t n = static_cast<size_t>(m_host.num_local_elements) * SubgridType::size;
      std::unique_ptr<double[]> h = std::make_unique<double[]>(n ? n : 1);
      z_order_doubles(step, variable);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(h.get(), m_scratch64, sizeof(double) * n, hipMemcpyDeviceToHost));
      return {T8GPU_VTK_SCALAR, std::move(h), name};
    }
    /// subgrid_mesh_manager.h:438: three variables as interleaved xyz doubles per subcell (same ordering as above)
    [[nodiscard]] HostVariableInfo get_host_vector_variable(step_index_type step, std::array<variable_index_type, 3> variables,
                                                            std::string const& name) const {
      const size_t n = static_cast<size_t>(m_host.num_local_elements) * SubgridType::size;
      std::unique_ptr<double[]> h = std::make_unique<double[]>(3 * n ? 3 * n : 1);
      std::vector<double>       one(n);
      for (int c = 0; c < 3; c++) {
        z_order_doubles(step, variables[c]);
        T8GPU_CUDA_CHECK_ERROR(hipMemcpy(one.data(), m_scratch64, sizeof(double) * n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) h[3 * i + c] = one[i];
      }
      return {T8GPU_VTK_VECTOR, std::move(h), name};
    }
    /// subgrid_mesh_manager.h:446: all fields in one file, on the forest refined down to the subcells
    void save_variables_to_vtk(std::vector<HostVariableInfo> host_variables, std::string const& prefix) const {
      std::vector<char const*>   names;
      std::vector<int32_t>       comps;
      std::vector<double const*> data;
      for (auto const& h : host_variables) {
        names.push_back(h.m_name.c_str());
        comps.push_back(h.m_type);
        data.push_back(h.m_data.get());
      }
      write(prefix, SubgridType::template extent<0>, static_cast<int>(names.size()), names.data(), comps.data(), data.data());
    }

    /// subgrid_mesh_manager.inl:1051-1138: the variable on the forest refined uniformly twice (z-order), field "variables"
    void save_variable_to_vtk(step_index_type step, variable_index_type variable, std::string const& prefix) const {
      std::vector<HostVariableInfo> v;
      v.push_back(get_host_scalar_variable(step, variable, "variables"));
      save_variables_to_vtk(std::move(v), prefix);
    }
    /// subgrid_mesh_manager.inl:1185-1206: the forest itself, no data
    void save_mesh_to_vtk(std::string const& prefix) const { write(prefix, 1, 0, nullptr, nullptr, nullptr); }

   private:
    using SubgridMemoryManager<VariableType, StepType, SubgridType>::resize;   // subgrid_mesh_manager.h:466

    void* m_forest    = nullptr;
    int   m_min_level = min_level, m_max_level = max_level;
    int   m_nb_ranks  = 1;

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
    /// rank `rank` of `nranks`' share of the forest; `halo` (nullable) receives the ghost lists
    static HostSubgridMeshArrays arrays_of(void* forest, int rank, int nranks, HostHaloArrays* halo) {
      constexpr int     R = SubgridType::rank;
      T8gpuForestQuery* q = t8gpu_synth_query_create(forest, rank, nranks);
      void*             h = q ? t8gpu_host_connectivity_create_subgrid(q, R) : nullptr;
      if (!h) {
        std::fprintf(stderr, "t8gpu: connectivity of the synthetic forest could not be built\n");
        std::abort();
      }
      int64_t c[6];
      t8gpu_host_connectivity_counts(h, c);
      HostSubgridMeshArrays m;
      m.rank = R;
      m.mpirank = rank;
      m.first_global_element = t8gpu_synth_mesh_num_elements(forest) * rank / nranks;
      m.num_local_elements = static_cast<int32_t>(c[0]); m.num_ghost_elements = static_cast<int32_t>(c[1]);
      m.num_local_faces = static_cast<int32_t>(c[2]); m.num_local_boundary_faces = static_cast<int32_t>(c[3]);
      const size_t nf = static_cast<size_t>(c[2] + c[3]);
      std::vector<double> n3(3 * nf);
      m.face_neighbors.resize(2 * c[2] + c[3]);
      m.face_surfaces.resize(nf);
      m.volumes.resize(c[0] + c[1]);
      m.face_level_difference.resize(c[2]);
      m.face_neighbor_offset.resize(static_cast<size_t>(R) * c[2]);
      HostHaloArrays hh;
      hh.peers.resize(c[4]); hh.recv_off.resize(c[4] + 1); hh.send_off.resize(c[4] + 1); hh.send_idx.resize(c[5]);
      t8gpu_host_connectivity_arrays(h, m.face_neighbors.data(), n3.data(), m.face_surfaces.data(), m.volumes.data(), hh.peers.data(),
                                     hh.recv_off.data(), hh.send_off.data(), hh.send_idx.data());
      if (halo) *halo = std::move(hh);
      t8gpu_host_connectivity_subgrid_arrays(h, m.face_level_difference.data(), m.face_neighbor_offset.data());
      t8gpu_host_connectivity_destroy(h);
      t8gpu_synth_query_destroy(q);
      m.face_normals.resize(static_cast<size_t>(R) * nf);   // the Subgrid accessors carry `rank` components
      for (size_t i = 0; i < nf; i++)
        for (int d = 0; d < R; d++) m.face_normals[R * i + d] = n3[3 * i + d];
      void* part = t8gpu_synth_part_create(forest, rank, nranks, 1, R);
      m.levels.resize(c[0] + c[1]);           // (the provider lists owned + ghost blocks; the manager keeps the owned ones)
      m.centres.resize(3 * (c[0] + c[1]));
      t8gpu_synth_part_elements(part, m.levels.data(), nullptr, m.centres.data());
      m.levels.resize(c[0]);
      m.centres.resize(3 * c[0]);
      t8gpu_synth_part_destroy(part);
      return m;
    }
    void rebuild_connectivity(HostSubgridMeshArrays const& m) {
      for (void* p : {static_cast<void*>(m_ranks), static_cast<void*>(m_indices), static_cast<void*>(m_face_neighbors),
                      static_cast<void*>(m_level_difference), static_cast<void*>(m_neighbor_offset), static_cast<void*>(m_face_normals),
                      static_cast<void*>(m_face_surfaces)})
        (void)hipFree(p);
      m_host = m;
      const size_t tot = static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements;
      std::vector<int>         ranks(tot, m.mpirank);
      std::vector<t8_locidx_t> indices(tot);
      for (size_t i = 0; i < tot; i++) indices[i] = static_cast<t8_locidx_t>(i);
      upload(m_ranks, ranks);
      upload(m_indices, indices);
      upload(m_face_neighbors, m.face_neighbors);
      upload(m_level_difference, m.face_level_difference);
      upload(m_neighbor_offset, m.face_neighbor_offset);
      upload(m_face_normals, std::vector<float_type>(m.face_normals.begin(), m.face_normals.end()));
      upload(m_face_surfaces, std::vector<float_type>(m.face_surfaces.begin(), m.face_surfaces.end()));
      (void)hipFree(m_d_send_idx);
      (void)hipFree(m_d_sendbuf);
      (void)hipFree(m_d_recvbuf);
      m_d_send_idx = nullptr;
      m_d_sendbuf = m_d_recvbuf = nullptr;
      if (m_nb_ranks > 1 && !m_halo_host.peers.empty()) {   // device side of refresh_ghost_layer(): whole blocks on the wire
        constexpr size_t S = SubgridType::size;
        upload(m_d_send_idx, m_halo_host.send_idx);
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_d_sendbuf, sizeof(float_type) * (5 * S * m_halo_host.send_idx.size() + 1)));
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_d_recvbuf, sizeof(float_type) * (5 * S * static_cast<size_t>(m.num_ghost_elements) + 1)));
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
      float_type*          tmp    = nullptr;   // 5 planes of 4^rank x (have_off[r + 1] - have_off[r]) values: this rank's adapted blocks
      float_type*          vol    = nullptr;   // their volumes
      std::vector<int64_t> have_off;           // new blocks made from rank q's old ones: [have_off[q], have_off[q + 1])
    } m_pending;

    HostSubgridMeshArrays m_host;
    int*                  m_ranks            = nullptr;
    t8_locidx_t*          m_indices          = nullptr;
    t8_locidx_t*          m_face_neighbors   = nullptr;
    t8_locidx_t*          m_level_difference = nullptr;
    t8_locidx_t*          m_neighbor_offset  = nullptr;
    float_type*           m_face_normals     = nullptr;
    float_type*           m_face_surfaces    = nullptr;
    mutable float_type*   m_scratch          = nullptr;   // z-ordered copy of one variable (float_type / double)
    mutable double*       m_scratch64        = nullptr;

    /// column_major_to_z_order (subgrid_mesh_manager.inl:1008-1049) + cast: `variable` of `step` -> m_scratch64
    void z_order_doubles(step_index_type step, variable_index_type variable) const {
      const size_t n = static_cast<size_t>(m_host.num_local_elements) * SubgridType::size;
      if (!m_scratch) T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_scratch, sizeof(float_type) * (n ? n : 1)));
      if (!m_scratch64) T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_scratch64, sizeof(double) * (n ? n : 1)));
      float_type const* src = static_cast<float_type const*>(this->get_own_variable(step, variable));
      if constexpr (std::is_same_v<float_type, double>) {
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_column_major_to_z_order_f64(SubgridType::rank, m_host.num_local_elements, src, m_scratch, nullptr)));
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_scalar_variable_f64(n, m_scratch, m_scratch64, nullptr)));
      } else {
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_column_major_to_z_order_f32(SubgridType::rank, m_host.num_local_elements, src, m_scratch, nullptr)));
        T8GPU_CUDA_CHECK_ERROR(static_cast<hipError_t>(t8gpu_hip_host_scalar_variable_f32(n, m_scratch, m_scratch64, nullptr)));
      }
      T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
    }

    void write(std::string const& prefix, int cells_per_dim, int nf, char const* const* names, int32_t const* comps,
               double const* const* data) const {
      const std::string path = prefix + ".vtu";
      const int rc = t8gpu_host_write_vtu(path.c_str(), SubgridType::rank, m_host.num_local_elements, m_host.centres.data(),
                                          m_host.levels.data(), cells_per_dim, m_host.mpirank, m_host.first_global_element, nf, names, comps,
                                          data, 0);
      if (rc != 0) {
        std::fprintf(stderr, "t8gpu: writing %s failed (code %d)\n", path.c_str(), rc);
        std::abort();
      }
    }
    template<typename T>
    static void upload(T*& dst, std::vector<T> const& src) {
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&dst, sizeof(T) * (src.empty() ? 1 : src.size())));
      if (!src.empty()) T8GPU_CUDA_CHECK_ERROR(hipMemcpy(dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    }
  };

  /// earlier name of the class when it is built from the synthetic provider or from host arrays
  template<typename VariableType, typename StepType, typename SubgridType>
  using SyntheticSubgridMeshManager = SubgridMeshManager<VariableType, StepType, SubgridType>;

}  // namespace t8gpu

#endif  // T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H
