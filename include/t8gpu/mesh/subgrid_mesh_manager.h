// t8gpu/mesh/subgrid_mesh_manager.h (MI355X backend)
//
// SubgridMeshConnectivityAccessor<float_type, SubgridType>: device view of the coarse-face lists of a
// Subgrid mesh with the getters of the reference (t8gpu/mesh/subgrid_mesh_manager.h:29-216): the plain
// arrays plus face_level_difference[F] (level(right) - level(left) <= 0) and face_neighbor_offset[rank*F]
// (anchor inside the right block; subgrid_mesh_manager.inl:587-680). Normals have SubgridType::rank
// components. The t8code-bound SubgridMeshManager is out of scope this round.
#ifndef T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H
#define T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H

#include <t8gpu/memory/subgrid_memory_manager.h>
#include <t8gpu/mesh/mesh_manager.h>

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

}  // namespace t8gpu

#endif  // T8GPU_HIP_MESH_SUBGRID_MESH_MANAGER_H
