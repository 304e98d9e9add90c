// t8gpu/backend/hip_fast.h -- C++ binding of the C-ABI (include/t8gpu_hip.h) for code written against
// the t8gpu accessor API. This is what a t8gpu maintainer adds to re-point
// CompressibleEulerSolver::iterate (examples/compressible_euler/solver.cu:75-175) at the tuned kernels:
//
//     t8gpu::hip::PlainFusedPlan plan(host_mesh);                    // at every connectivity rebuild
//     ...
//     std::swap(next, prev);                                          // solver.cu:76
//     t8gpu::hip::iterate_fused(mesh_manager, plan, prev, next, delta_t, speed_estimates);
//
// or, keeping the reference's kernel-by-kernel structure, t8gpu::hip::flux_faces / flux_boundary /
// rk3_stage in place of the three <<<>>> launches of a stage. Errors abort like T8GPU_CUDA_CHECK_ERROR.
#ifndef T8GPU_HIP_BACKEND_HIP_FAST_H
#define T8GPU_HIP_BACKEND_HIP_FAST_H

#include <t8gpu/memory/subgrid_memory_manager.h>
#include <t8gpu/mesh/mesh_manager.h>
#include <t8gpu/mesh/subgrid_mesh_manager.h>
#include <t8gpu_hip.h>
#include <t8gpu_host.h>

#include <type_traits>
#include <algorithm>
#include <array>
#include <vector>

namespace t8gpu::hip {

  template<typename ft>
  using vars_t = std::conditional_t<std::is_same_v<ft, float>, T8gpuVars_f32, T8gpuVars_f64>;

  template<typename VariableType>
  auto to_vars(MemoryAccessorOwn<VariableType> acc) {
    using ft = typename variable_traits<VariableType>::float_type;
    static_assert(variable_traits<VariableType>::nb_variables == 5, "the Euler kernels expect Rho, Rho_v1..3, Rho_e");
    vars_t<ft> v;
    for (int k = 0; k < 5; k++) v.p[k] = acc.get(k);
    return v;
  }

#define T8GPU_DISPATCH(ft, name, ...)                                \
  do {                                                               \
    if constexpr (std::is_same_v<ft, float>) {                       \
      T8GPU_HIP_CHECK_ABI(name##_f32(__VA_ARGS__));                  \
    } else {                                                         \
      T8GPU_HIP_CHECK_ABI(name##_f64(__VA_ARGS__));                  \
    }                                                                \
  } while (0)

  /// replaces kepes_compute_fluxes<<<>>> (kernels.cu:135-309)
  template<typename VariableType, size_t dim>
  void flux_faces(MeshConnectivityAccessor<typename variable_traits<VariableType>::float_type, dim> const& c,
                  MemoryAccessorOwn<VariableType> state, MemoryAccessorOwn<VariableType> fluxes,
                  typename variable_traits<VariableType>::float_type* speed, int flux_kind = T8GPU_FLUX_KEPES,
                  hipStream_t stream = nullptr) {
    using ft = typename variable_traits<VariableType>::float_type;
    T8GPU_DISPATCH(ft, t8gpu_hip_flux_faces, flux_kind, c.get_num_local_faces(), static_cast<int>(dim), c.face_neighbors(),
                   c.indices(), c.face_normals(), c.face_surfaces(), to_vars(state), to_vars(fluxes), speed, stream);
  }
  /// replaces reflective_boundary_condition<<<>>> (kernels.cu:311-469)
  template<typename VariableType, size_t dim>
  void flux_boundary(MeshConnectivityAccessor<typename variable_traits<VariableType>::float_type, dim> const& c,
                     MemoryAccessorOwn<VariableType> state, MemoryAccessorOwn<VariableType> fluxes,
                     typename variable_traits<VariableType>::float_type* speed, int flux_kind = T8GPU_FLUX_KEPES,
                     hipStream_t stream = nullptr) {
    using ft = typename variable_traits<VariableType>::float_type;
    T8GPU_DISPATCH(ft, t8gpu_hip_flux_boundary, flux_kind, c.get_num_local_faces(), c.get_num_local_boundary_faces(),
                   static_cast<int>(dim), c.face_neighbors(), c.face_normals(), c.face_surfaces(), to_vars(state),
                   to_vars(fluxes), speed, stream);
  }
  /// replaces timestepping::SSP_3RK_step{1,2,3}<<<>>> (ssp_runge_kutta.inl:30-99)
  template<typename VariableType>
  void rk3_stage(int stage, int num_elements, MemoryAccessorOwn<VariableType> prev, MemoryAccessorOwn<VariableType> mid,
                 MemoryAccessorOwn<VariableType> out, MemoryAccessorOwn<VariableType> fluxes,
                 typename variable_traits<VariableType>::float_type const* volume,
                 typename variable_traits<VariableType>::float_type delta_t, hipStream_t stream = nullptr) {
    using ft = typename variable_traits<VariableType>::float_type;
    T8GPU_DISPATCH(ft, t8gpu_hip_rk3_stage, stage, num_elements, to_vars(prev), to_vars(mid), to_vars(out), to_vars(fluxes),
                   volume, delta_t, stream);
  }

  using t8gpu::HostHaloArrays;   // (t8gpu/mesh/mesh_manager.h: what the halo exchange needs besides HostMeshArrays)

  /// the connectivity of one rank from forest-query callbacks (csrc/host/connectivity.cpp): the
  /// arrays a MeshManager's compute_connectivity_information hands to this backend.
  inline HostMeshArrays host_mesh_arrays_from_query(T8gpuForestQuery const& q, int rank = 0, HostHaloArrays* halo = nullptr) {
    void* h = t8gpu_host_connectivity_create(&q);
    if (!h) T8GPU_ABORT("t8gpu_host_connectivity_create failed (malformed forest query)");
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
    t8gpu_host_connectivity_destroy(h);
    if (halo) *halo = std::move(hh);
    return m;
  }

  /// One RCCL communicator per process (one rank per GPU). Rank 0 makes the id, the application distributes the 128
  /// bytes (MPI_Bcast in the reference's setting), every rank constructs. Stands where the reference exchanges CUDA-IPC
  /// handles with MPI_Allgather (shared_device_vector.inl:21-22,179-185).
  class Communicator {
   public:
    [[nodiscard]] static std::array<char, 128> unique_id() {
      std::array<char, 128> id{};
      T8GPU_HIP_CHECK_ABI(t8gpu_hip_comm_unique_id(id.data()));
      return id;
    }
    Communicator(std::array<char, 128> const& id, int rank, int nranks) : m_rank{rank}, m_size{nranks} {
      T8GPU_HIP_CHECK_ABI(t8gpu_hip_comm_create(id.data(), rank, nranks, &m_comm));
    }
    ~Communicator() { (void)t8gpu_hip_comm_destroy(m_comm); }
    Communicator(Communicator const&)            = delete;
    Communicator& operator=(Communicator const&) = delete;
    [[nodiscard]] void* handle() const { return m_comm; }
    [[nodiscard]] int   rank() const { return m_rank; }
    [[nodiscard]] int   size() const { return m_size; }

   private:
    void* m_comm = nullptr;
    int   m_rank, m_size;
  };

  /// Device copy of the tile plan (t8gpu_plan_plain_create) + the native step driver.
  template<typename ft>
  class PlainFusedPlan {
   public:
    /// Single rank: halo = comm = nullptr. Several ranks: the rank's halo lists (host_mesh_arrays_from_query fills
    /// them) and the communicator; iterate_fused then exchanges the ghost layer per stage (csrc/hip/stepper.hip).
    explicit PlainFusedPlan(HostMeshArrays const& m, int ndim = 3, int tmax = 256, int fcap = 512, HostHaloArrays const* halo = nullptr,
                            Communicator const* comm = nullptr) {
      // (flags 3 | 8: structured patches -- 16 x 16 quadrilateral / 8 x 8 x 4 hexahedral blocks, the irregular 3D ones included -- are cut out of the tiling and run through the patch kernels.
      //  The patch kernels address a plane by a 32-bit byte offset: a mesh whose planes reach 4 GiB is planned without patches, every element through the tile kernels -- the launchers refuse patch tiles on such planes)
      const bool  wide  = (static_cast<unsigned long long>(m.num_local_elements) + static_cast<unsigned long long>(m.num_ghost_elements)) * sizeof(ft) >= (1ull << 32);
      void* h = t8gpu_plan_plain_create_ex(m.num_local_elements, m.num_ghost_elements, m.num_local_faces,
                                           m.num_local_boundary_faces, ndim, m.face_neighbors.data(), m.face_normals.data(),
                                           m.face_surfaces.data(), tmax, fcap, (wide ? 0 : (3 | 8)) | ((halo && comm && !halo->peers.empty()) ? 32 : 0));
      // (flag 32, several ranks: interior tiles in one class -- the step driver launches them as one persistent grid per stage)
      if (!h) T8GPU_ABORT("t8gpu_plan_plain_create_ex failed");
      int64_t sz[16];
      t8gpu_plan_plain_sizes(h, sz);
      const size_t nt = sz[0], nhalo = sz[1], nfaces = sz[2], ncsr = sz[3], N = sz[8], w = sz[10], ngeo = sz[11];
      std::vector<int32_t>  elem_off(nt + 1), halo_off(nt + 1), face_off(nt + 1), halo_ids(nhalo), face_orig(nfaces),
          csr_off(N + 1), tile_order(nt);
      std::vector<uint32_t> face_lr(nfaces);
      std::vector<double>   geo(4 * nfaces), table(12 * ngeo);
      std::vector<uint16_t> csr_ent(ncsr), ell(std::max<size_t>(1, static_cast<size_t>(sz[15])) * w), geo_idx(ngeo ? nfaces : 0);
      t8gpu_plan_plain_arrays(h, elem_off.data(), halo_off.data(), face_off.data(), halo_ids.data(), face_lr.data(),
                              geo.data(), face_orig.data(), csr_off.data(), csr_ent.data(), tile_order.data());
      t8gpu_plan_plain_compressed(h, ell.data(), ngeo ? geo_idx.data() : nullptr, ngeo ? table.data() : nullptr);
      std::vector<int32_t> tile_desc(8 * std::max<size_t>(nt, 1));
      t8gpu_plan_plain_tile_desc(h, tile_desc.data());
      int32_t patch_counts[4];
      t8gpu_plan_plain_patch_counts(h, patch_counts);
      int32_t irregular_counts[3];
      t8gpu_plan_plain_irregular_counts(h, irregular_counts);
      const int32_t patch_dim = t8gpu_plan_plain_patch_dim(h);
      t8gpu_plan_plain_destroy(h);
      m_plan.elem_off   = up(elem_off);
      m_plan.halo_off   = up(halo_off);
      m_plan.face_off   = up(face_off);
      m_plan.halo_ids   = up(halo_ids);
      m_plan.face_lr    = up(face_lr);
      m_plan.face_geo   = up(std::vector<ft>(geo.begin(), geo.end()));
      m_plan.face_orig  = up(face_orig);
      m_plan.csr_off    = up(csr_off);
      m_plan.csr_ent    = up(csr_ent);
      m_plan.tile_order = up(tile_order);
      m_plan.tile_desc  = up(tile_desc);
      m_plan.ell        = up(ell);
      m_plan.geo_idx    = ngeo ? up(geo_idx) : nullptr;
      m_plan.geo_table  = ngeo ? up(std::vector<ft>(table.begin(), table.end())) : nullptr;
      if (ngeo) {   // the dictionary's tangent rows by the device's own routine (t8gpu_hip.h: same bits as per-face geometry rows)
        if constexpr (sizeof(ft) == 4)
          T8GPU_HIP_CHECK_ABI(t8gpu_hip_plain_geo_frames_f32(const_cast<void*>(static_cast<const void*>(m_plan.geo_table)), static_cast<int>(ngeo), nullptr));
        else
          T8GPU_HIP_CHECK_ABI(t8gpu_hip_plain_geo_frames_f64(const_cast<void*>(static_cast<const void*>(m_plan.geo_table)), static_cast<int>(ngeo), nullptr));
        T8GPU_CUDA_CHECK_ERROR(hipStreamSynchronize(nullptr));
      }
      m_plan.ntiles = static_cast<int32_t>(nt); m_plan.n_interior_tiles = static_cast<int32_t>(sz[7]);
      m_plan.max_elems = static_cast<int32_t>(sz[4]); m_plan.max_halo = static_cast<int32_t>(sz[5]);
      m_plan.max_faces = static_cast<int32_t>(sz[6]); m_plan.ell_width = static_cast<int32_t>(w);
      m_plan.n_geo = static_cast<int32_t>(ngeo); m_plan.max_slots = static_cast<int32_t>(sz[12]);
      m_plan.n_deep_tiles = static_cast<int32_t>(sz[13]);
      m_plan.n_slots_addressed = m.num_local_elements + m.num_ghost_elements;
      for (int c = 0; c < 3; c++) m_plan.n_patch_tiles[c] = patch_counts[c];
      for (int c = 0; c < 3; c++) m_plan.n_irregular_tiles[c] = irregular_counts[c];
      m_plan.patch_dim = patch_dim;
      T8gpuHalo  hl{};
      const bool multi = halo && comm && !halo->peers.empty();
      if (multi) {
        m_halo             = *halo;   // host index arrays must outlive the stepper
        hl.num_elements     = m.num_local_elements;
        hl.num_ghosts       = m.num_ghost_elements;
        hl.n_peers          = static_cast<int32_t>(m_halo.peers.size());
        hl.n_send           = static_cast<int32_t>(m_halo.send_idx.size());
        hl.cells_per_element = 1;
        hl.peers            = m_halo.peers.data();
        hl.send_off         = m_halo.send_off.data();
        hl.recv_off         = m_halo.recv_off.data();
        hl.send_idx         = up(m_halo.send_idx);
        hl.sendbuf          = up(std::vector<ft>(5 * m_halo.send_idx.size() + 1));
        hl.recvbuf          = up(std::vector<ft>(5 * static_cast<size_t>(m.num_ghost_elements) + 1));
        hl.comm             = comm->handle();
      }
      T8GPU_HIP_CHECK_ABI(t8gpu_hip_plain_stepper_create(&m_plan, multi ? &hl : nullptr, &m_stepper));
    }
    ~PlainFusedPlan() {
      t8gpu_hip_plain_stepper_destroy(m_stepper);
      for (void* p : m_allocs) (void)hipFree(p);
    }
    PlainFusedPlan(PlainFusedPlan const&)            = delete;
    PlainFusedPlan& operator=(PlainFusedPlan const&) = delete;
    [[nodiscard]] T8gpuPlainPlan const& view() const { return m_plan; }
    [[nodiscard]] void*                 stepper() const { return m_stepper; }

   private:
    T8gpuPlainPlan     m_plan{};
    HostHaloArrays     m_halo;
    void*              m_stepper = nullptr;
    std::vector<void*> m_allocs;
    template<typename T>
    T* up(std::vector<T> const& v) {
      T* d = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())));
      if (!v.empty()) T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
      m_allocs.push_back(d);
      return d;
    }
  };

  /// the three stages of iterate() (solver.cu:78-174) in one call; prev/next are the step ids after the swap
  template<typename VariableType, typename StepType, size_t dim>
  void iterate_fused(SyntheticMeshManager<VariableType, StepType, dim>&                                  mesh,
                     PlainFusedPlan<typename variable_traits<VariableType>::float_type> const&           plan,
                     typename step_traits<StepType>::index_type prev, typename step_traits<StepType>::index_type next,
                     typename variable_traits<VariableType>::float_type delta_t,
                     typename variable_traits<VariableType>::float_type* speed, int flux_kind = T8GPU_FLUX_KEPES,
                     hipStream_t stream = nullptr, int n_steps = 1) {
    using ft = typename variable_traits<VariableType>::float_type;
    // n_steps > 1: prev / next are the roles of the FIRST step and alternate (after an odd count the caller swaps once more)
    T8GPU_DISPATCH(ft, t8gpu_hip_plain_stepper_iterate_steps, plan.stepper(), flux_kind, mesh.planes_base(), mesh.plane_stride(),
                   static_cast<int>(prev), static_cast<int>(next), delta_t, speed, n_steps, stream);
  }
  // ---- the two scalar reductions of the reference solver, on the device ----------------------------
  /// Workspace + result slot for compute_integral / max_speed (allocate once, reuse).
  class Reducer {
   public:
    Reducer() {
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_work, t8gpu_hip_reduce_workspace_bytes()));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_result, sizeof(double)));
    }
    ~Reducer() {
      (void)hipFree(m_work);
      (void)hipFree(m_result);
    }
    Reducer(Reducer const&)            = delete;
    Reducer& operator=(Reducer const&) = delete;
    /// CompressibleEulerSolver::compute_integral (solver.cu:190-211), this rank's part (the caller all-reduces)
    template<typename ft>
    [[nodiscard]] double integral(size_t num_cells, ft const* variable, ft const* volume, int cells_per_element = 1,
                                  hipStream_t stream = nullptr) {
      T8GPU_DISPATCH(ft, t8gpu_hip_integral, num_cells, cells_per_element, variable, volume, m_work, m_result, stream);
      return fetch(stream);
    }
    /// the thrust::reduce(maximum) of compute_timestep (solver.cu:213-217); dt = cfl * 0.5^max_level / this
    template<typename ft>
    [[nodiscard]] double max_speed(size_t n, ft const* speed_estimates, hipStream_t stream = nullptr) {
      T8GPU_DISPATCH(ft, t8gpu_hip_max_speed, n, speed_estimates, m_work, m_result, stream);
      return fetch(stream);
    }

   private:
    void*   m_work   = nullptr;
    double* m_result = nullptr;
    double  fetch(hipStream_t stream) {
      double h = 0;
      T8GPU_CUDA_CHECK_ERROR(hipMemcpyAsync(&h, m_result, sizeof(double), hipMemcpyDeviceToHost, stream));
      T8GPU_CUDA_CHECK_ERROR(hipStreamSynchronize(stream));
      return h;
    }
  };

  // ---- Subgrid<4,4> / Subgrid<4,4,4> -------------------------------------------------------------------------
  template<typename VariableType, typename SubgridType>
  auto to_vars(SubgridMemoryAccessorOwn<VariableType, SubgridType> acc) {
    using ft = typename variable_traits<VariableType>::float_type;
    static_assert(variable_traits<VariableType>::nb_variables == 5, "the Euler kernels expect Rho, Rho_v1..3, Rho_e");
    vars_t<ft> v;
    for (int k = 0; k < 5; k++) v.p[k] = acc.data(static_cast<typename variable_traits<VariableType>::index_type>(k));
    return v;
  }

  using t8gpu::HostSubgridMeshArrays;

  /// Device copy of the joined per-block face records (t8gpu_plan_subgrid_create + _records) for the fused
  /// block kernel: rebuilt where compute_connectivity_information runs (subgrid_mesh_manager.inl:560-961).
  template<typename ft>
  class SubgridFusedPlan {
   public:
    explicit SubgridFusedPlan(HostSubgridMeshArrays const& m) {
      void* h = t8gpu_plan_subgrid_create(m.num_local_elements, m.num_local_faces, m.num_local_boundary_faces, m.rank,
                                          m.face_neighbors.data(), m.face_level_difference.data(), m.face_neighbor_offset.data(),
                                          m.face_normals.data());
      if (!h) T8GPU_ABORT("t8gpu_plan_subgrid_create failed (axis-aligned unit normals required, as in the reference)");
      int64_t sz[8];
      t8gpu_plan_subgrid_sizes(h, sz);
      std::vector<int32_t> block_rec(32 * static_cast<size_t>(std::max<int32_t>(1, m.num_local_elements))),
          bf_rec(4 * static_cast<size_t>(std::max<int64_t>(1, sz[0])));
      t8gpu_plan_subgrid_records(h, m.face_surfaces.data(), static_cast<int>(sizeof(ft)), block_rec.data(), bf_rec.data());
      std::vector<int32_t> fam_rec((m.rank == 3 ? 160 : 64) * static_cast<size_t>(std::max<int64_t>(1, sz[6]))), rest_rec(32 * static_cast<size_t>(std::max<int64_t>(1, sz[7])));
      if (sz[6] > 0) t8gpu_plan_subgrid_family_records(h, m.face_surfaces.data(), static_cast<int>(sizeof(ft)), fam_rec.data(), rest_rec.data());
      t8gpu_plan_subgrid_destroy(h);
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_block_rec, sizeof(int32_t) * block_rec.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_bf_rec, sizeof(int32_t) * bf_rec.size()));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_block_rec, block_rec.data(), sizeof(int32_t) * block_rec.size(), hipMemcpyHostToDevice));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_bf_rec, bf_rec.data(), sizeof(int32_t) * bf_rec.size(), hipMemcpyHostToDevice));
      if (sz[6] > 0) {
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_fam_rec, sizeof(int32_t) * fam_rec.size()));
        T8GPU_CUDA_CHECK_ERROR(hipMalloc(&m_rest_rec, sizeof(int32_t) * rest_rec.size()));
        T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_fam_rec, fam_rec.data(), sizeof(int32_t) * fam_rec.size(), hipMemcpyHostToDevice));
        T8GPU_CUDA_CHECK_ERROR(hipMemcpy(m_rest_rec, rest_rec.data(), sizeof(int32_t) * rest_rec.size(), hipMemcpyHostToDevice));
        m_plan.fam_rec    = m_fam_rec;
        m_plan.rest_rec   = m_rest_rec;
        m_plan.n_families = static_cast<int32_t>(sz[6]);
        m_plan.n_rest     = static_cast<int32_t>(sz[7]);
      }
      m_plan.block_rec = m_block_rec;
      m_plan.bf_rec    = m_bf_rec;
      m_plan.num_elements = m.num_local_elements;
      m_plan.rank         = m.rank;
      m_plan.max_faces_per_block = static_cast<int32_t>(sz[1]);
      m_plan.n_interior_blocks   = static_cast<int32_t>(sz[3]);
      m_plan.n_deep_blocks       = static_cast<int32_t>(sz[4]);
      m_plan.n_blocks_addressed  = static_cast<int32_t>(sz[5]);
    }
    ~SubgridFusedPlan() {
      (void)hipFree(m_block_rec);
      (void)hipFree(m_bf_rec);
      (void)hipFree(m_fam_rec);
      (void)hipFree(m_rest_rec);
    }
    SubgridFusedPlan(SubgridFusedPlan const&)            = delete;
    SubgridFusedPlan& operator=(SubgridFusedPlan const&) = delete;
    [[nodiscard]] T8gpuSubgridPlan const& view() const { return m_plan; }

   private:
    T8gpuSubgridPlan m_plan{};
    int32_t *        m_block_rec = nullptr, *m_bf_rec = nullptr, *m_fam_rec = nullptr, *m_rest_rec = nullptr;
  };

  /// SubgridCompressibleEulerSolver::iterate (examples/subgrid/solver.inl:152-266) after its std::swap: three fused
  /// stage launches instead of 3 x (inner + boundary + outer flux kernels, sync + barrier, RK kernel).
  template<typename VariableType, typename StepType, typename SubgridType>
  void iterate_fused(SubgridMemoryManager<VariableType, StepType, SubgridType>&                         mem,
                     SubgridFusedPlan<typename variable_traits<VariableType>::float_type> const&        plan,
                     typename step_traits<StepType>::index_type prev, typename step_traits<StepType>::index_type next,
                     typename variable_traits<VariableType>::float_type delta_t, int flux_kind = T8GPU_FLUX_KEPES,
                     hipStream_t stream = nullptr) {
    using ft   = typename variable_traits<VariableType>::float_type;
    using step = typename step_traits<StepType>::index_type;
    const step src[3] = {prev, static_cast<step>(1), static_cast<step>(2)}, dst[3] = {static_cast<step>(1), static_cast<step>(2), next};
    for (int k = 0; k < 3; k++)
      T8GPU_DISPATCH(ft, t8gpu_hip_subgrid_fused_stage, flux_kind, k + 1, &plan.view(), 0, plan.view().num_elements,
                     to_vars(mem.get_own_variables(prev)), to_vars(mem.get_own_variables(src[k])), to_vars(mem.get_own_variables(dst[k])),
                     mem.get_own_volume(), delta_t, stream);
  }
#undef T8GPU_DISPATCH

}  // namespace t8gpu::hip

#endif  // T8GPU_HIP_BACKEND_HIP_FAST_H
