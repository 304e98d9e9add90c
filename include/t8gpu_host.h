/* t8gpu_host.h -- C ABI of the host-only library (t8gpu_amd/lib/libt8gpu_host.so, no HIP dependency).
 *
 *  - t8gpu_synth_*: t8code-free provider of the hot path's input contract (the arrays
 *    MeshManager::compute_connectivity_information builds, t8gpu/mesh/mesh_manager.inl:333-481 and
 *    subgrid_mesh_manager.inl:560-961) for periodic / walled unit squares and cubes with 2:1 AMR,
 *    SFC partitions with ghost mirror slots and per-peer halo lists.
 *  - t8gpu_plan_plain_*: the tiling pre-pass of the fused kernels (see T8gpuPlainPlan in t8gpu_hip.h),
 *    run at every connectivity rebuild.
 * All pointers are HOST pointers; output arrays are caller-allocated (sizes from the *_counts/_sizes calls).
 */
#ifndef T8GPU_HOST_H
#define T8GPU_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic mesh -------------------------------------------------------------------------- */
void*   t8gpu_synth_mesh_create(int dim, int base_level, int max_level, double band, double shrink, int periodic);
void    t8gpu_synth_mesh_destroy(void* mesh);
int64_t t8gpu_synth_mesh_num_elements(const void* mesh);
int     t8gpu_synth_mesh_finest_level(const void* mesh);
int     t8gpu_synth_mesh_dim(const void* mesh);

void* t8gpu_synth_part_create(const void* mesh, int rank, int nranks, int subgrid, int normal_dim);
void  t8gpu_synth_part_destroy(void* part);
/* counts[8] = {N, G, F, B, n_peers, n_send, first_global, n_global} */
void t8gpu_synth_part_counts(const void* part, int64_t* counts);
/* face_neighbors[2F+B], normals[normal_dim*(F+B)], areas[F+B], level_diff[F], nb_offset[dim*F] (subgrid) */
void t8gpu_synth_part_connectivity(const void* part, int32_t* face_neighbors, double* normals, double* areas,
                                   int32_t* level_diff, int32_t* nb_offset);
/* the same arrays in place (no copy): ptrs[5] = {face_neighbors, normals, areas, face_level_difference, face_neighbor_offset},
 * null where empty; valid until t8gpu_synth_part_release_arrays / _destroy */
void t8gpu_synth_part_connectivity_ptrs(const void* part, const void** ptrs);
/* per owned + ghost element: level[N+G], volume[N+G], centre[(N+G)*3] */
void t8gpu_synth_part_elements(const void* part, int32_t* level, double* volume, double* centre);
void t8gpu_synth_part_halo(const void* part, int64_t* ghost_global, int32_t* ghost_owner, int32_t* peers,
                           int32_t* recv_off, int32_t* send_off, int32_t* send_idx);
/* Kelvin-Helmholtz initial state (values of examples/subgrid/solver.inl:35-56,84-103): out = 5 planes of
 * `stride` doubles; cells_per_dim = 1 (plain) or 4 (Subgrid<4,..>, index e*S + i + 4j + 16k). */
/* frees the partition's connectivity arrays (already copied out); element queries and the IC stay available */
void t8gpu_synth_part_release_arrays(void* part);
void t8gpu_synth_part_kh_ic(const void* part, int cells_per_dim, double* out, size_t stride);

/* ---- adaptation of the synthetic forest (stands where t8_forest adapt + balance run, mesh_manager.inl:196-213) --
 * marks: the reference's adapt callback (mesh_manager.inl:125-162): +1 refine, -1 coarsen (whole family), 0 keep;
 * family_members_averaged = 4 reproduces the reference, 0 = all 2^dim members.
 * adapt: refine / coarsen once + 2:1 balance; returns a new mesh (NULL on failure).
 * adapt_data[n_new+1]: first old element of every new element (mesh_manager.inl:258-281); returns 0 on success. */
void  t8gpu_synth_mesh_marks(const void* mesh, const double* criteria, double threshold, int min_level, int max_level,
                             int family_members_averaged, int8_t* marks);
/* clears the -1 marks of families cut by a partition offset (a family is only coarsened on one process) */
void  t8gpu_synth_mesh_unmark_split_families(const void* mesh, int8_t* marks, const int64_t* offsets, int n_offsets);
void* t8gpu_synth_mesh_adapt(const void* mesh, const int8_t* marks);
/* the same forest by the general procedure (leaf list and lookup grid rebuilt per balance round): the fallback of the
 * call above, exported for the tests that compare the two */
void* t8gpu_synth_mesh_adapt_by_rounds(const void* mesh, const int8_t* marks);
int   t8gpu_synth_mesh_adapt_data(const void* old_mesh, const void* new_mesh, int32_t* adapt_data);

/* ---- tile plan of the fused plain-element kernels --------------------------------------------- */
void* t8gpu_plan_plain_create(int32_t N, int32_t G, int32_t F, int32_t B, int32_t ndim, const int32_t* face_neighbors,
                              const double* normals, const double* areas, int32_t tmax, int32_t fcap);
/* flags bit 0: cut STRUCTURED PATCHES out of the tiling -- 16 x 16 same-size quadrilaterals that are 256 consecutive
 * elements in Morton order with the canonical face listing (csrc/host/tile_plan.cpp: find_patches). A patch is a tile
 * without face records: tile_desc holds {first element, 256, first halo entry, 64, id of its first own face,
 * 0x100 | flags, area (double)}, halo_ids the 64 elements across its sides ([-x | +x | -y | +y] x 16). Inside every
 * class of tile_order the patch tiles come first (t8gpu_plan_plain_patch_counts).
 * flags bit 2: the caller does not read `face_geo` when the plan has a geometry dictionary (sizes[11] > 0): the array is
 * then left empty (32 bytes per tile face less to build and copy).
 * flags bit 5 (32): no deep / near-boundary split of the interior tiles -- tile_order = [interior | ghost-reading],
 * sizes[13] (deep tiles) = sizes[7] (interior tiles), no patch tiles reported for class 1: a launch over [0, n_interior)
 * is then ONE kernel launch, which is what the two-lane step driver of the multi-rank path wants (csrc/hip/stepper.hip). */
void* t8gpu_plan_plain_create_ex(int32_t N, int32_t G, int32_t F, int32_t B, int32_t ndim, const int32_t* face_neighbors,
                                 const double* normals, const double* areas, int32_t tmax, int32_t fcap, int32_t flags);
void  t8gpu_plan_plain_destroy(void* plan);
/* counts[4] = leading patch tiles of the deep / near-boundary / ghost-reading class of tile_order, total */
void  t8gpu_plan_plain_patch_counts(const void* plan, int32_t* counts);
/* flags bit 3 of t8gpu_plan_plain_create_ex (with bit 1): IRREGULAR 3D patches too -- the same blocks next to a periodic
 * wrap, a wall or a coarser neighbour across a - side, where who lists a side face (and so the order of a cell's six face
 * ids) differs from cell to cell. tile_desc word 5 has 0x800, word 4 = first of 512 entries of face_lr / face_orig holding,
 * for cell c: face_lr[w + c] = sides the cell lists (bit per t8code face) | walls << 6 | the six sides in ascending face id
 * << 12 (3 bits each); face_orig[w + c] / face_orig[w + 256 + c] = id of the cell's first own interior / wall face (-1:
 * none). They are the LAST patch tiles of every class: counts[3] = how many (t8gpu_plan_plain_irregular_counts). */
void  t8gpu_plan_plain_irregular_counts(const void* plan, int32_t* counts);
/* flags bit 1 of t8gpu_plan_plain_create_ex: 3D patches -- 8 x 8 x 4 same-size hexahedra that are 256 consecutive elements
 * in Morton order (find_patches3); tile_desc = {first element, 256, first halo entry, 256, id of the first own face,
 * 0x300 | flags, area}, halo = [-x 32 | +x 32 | -y 32 | +y 32 | -z 64 | +z 64]. A plan holds one kind of patch.
 * 2 | 3: the kind of the plan's patch tiles, 0: none */
int32_t t8gpu_plan_plain_patch_dim(const void* plan);
/* Optional, between create and t8gpu_plan_plain_tile_desc: volumes[N] of the owned elements. A patch whose 256 elements
 * have bit for bit one volume carries it in its descriptor (flag 0x400; words 1 and 3 then hold the double instead of
 * the implied counts 256 and 64 | 256) and the patch kernels skip the per-element volume load. Returns their number. */
int32_t t8gpu_plan_plain_patch_volumes(void* plan, const double* volumes);
/* sizes[16] (12 = max over the generic tiles of own + halo elements, 13 = number of deep-interior tiles, 14 = number of patch tiles, 15 = number of ELL rows (the elements of generic tiles: ell holds sizes[15] * ell_width entries, a tile's first row is word 6 of its tile_desc record); the maxima 4-6 are over the generic tiles) = {ntiles, n_halo, n_faces, n_csr, max_elems, max_halo, max_faces, n_interior_tiles, N, F,
 *              ell_width, n_geo} */
void t8gpu_plan_plain_sizes(const void* plan, int64_t* sizes);
void t8gpu_plan_plain_arrays(const void* plan, int32_t* elem_off, int32_t* halo_off, int32_t* face_off,
                             int32_t* halo_ids, uint32_t* face_lr, double* face_geo, int32_t* face_orig,
                             int32_t* csr_off, uint16_t* csr_ent, int32_t* tile_order);
/* ell[sizes[15] * ell_width] (rows for the elements of generic tiles only, see T8gpuPlainPlan), geo_idx[n_faces],
 * geo_table[n_geo*12] = {n, area, t1, 0, t2, 0} rows */
void t8gpu_plan_plain_compressed(const void* plan, uint16_t* ell, uint16_t* geo_idx, double* geo_table);
/* the same arrays in place (no copy): ptrs[13] = {elem_off, halo_off, face_off, halo_ids, face_lr, face_geo, face_orig, csr_off,
 * csr_ent, tile_order, ell, geo_idx, geo_table}, null where empty; valid until t8gpu_plan_plain_destroy */
void t8gpu_plan_plain_array_ptrs(const void* plan, const void** ptrs);
/* tile_desc[ntiles][8] of T8gpuPlainPlan (one record per tile in tile_order order) */
void t8gpu_plan_plain_tile_desc(const void* plan, int32_t* tile_desc);

/* ---- per-block face lists of the fused Subgrid kernels (see T8gpuSubgridPlan in t8gpu_hip.h) -------- */
void* t8gpu_plan_subgrid_create(int32_t N, int32_t F, int32_t B, int32_t rank, const int32_t* face_neighbors,
                                const int32_t* face_level_difference, const int32_t* face_neighbor_offset,
                                const double* normals);
void  t8gpu_plan_subgrid_destroy(void* plan);
/* sizes[8] = {n_entries, max faces per block, F + B, n_interior_blocks, n_deep_blocks, 1 + largest block index referred to
 * (owned and ghost blocks), n_families, n_rest} */
void t8gpu_plan_subgrid_sizes(const void* plan, int64_t* sizes);
/* block_order[N]: blocks that touch no ghost block first (they can run during the halo exchange) */
void t8gpu_plan_subgrid_order(const void* plan, int32_t* block_order);
/* bf_off[N+1], bf_ent[n_entries], face_rec[(F+B)*4], plus[N*rank] (the +side face of each block, or -1) */
void t8gpu_plan_subgrid_arrays(const void* plan, int32_t* bf_off, int32_t* bf_ent, int32_t* face_rec, int32_t* plus);

/* The joined per-block records the fused Subgrid kernels read (layout: csrc/host/subgrid_plan.cpp):
 * block_rec[N][32] in block_order position order (128-byte rows, see T8gpuSubgridPlan), bf_rec[n_entries][4]; areas = face_surfaces[F + B] (doubles),
 * float_size = 4 or 8 selects how the areas are stored in the records. */
void t8gpu_plan_subgrid_records(const void* plan, const double* areas, int float_size, int32_t* block_rec, int32_t* bf_rec);
/* The 2x2x2 (RANK 3) / 2x2 (RANK 2) families of consecutive same-level deep interior blocks (sizes[6] of them) and the
 * remaining blocks (sizes[7]): fam_rec[n_families][160 or 64], rest_rec[n_rest][32] (layout: csrc/host/subgrid_plan.cpp;
 * T8gpuSubgridPlan::fam_rec / rest_rec) */
void t8gpu_plan_subgrid_family_records(const void* plan, const double* areas, int float_size, int32_t* fam_rec, int32_t* rest_rec);

/* ---- connectivity from forest queries (the t8code-independent part) -------------------------
 * What MeshManager::compute_connectivity_information (mesh_manager.inl:333-481) asks of t8code, as callbacks;
 * slots [0, num_local) are the rank's elements, [num_local, num_local + num_ghost) its ghost layer, ordered by
 * owner rank, then global index (t8code's order). csrc/host/connectivity.cpp states the listing rule. */
typedef struct T8gpuForestQuery {
  void*   ctx;
  int32_t num_local, num_ghost;
  int64_t (*global_id)(void* ctx, int32_t slot);        /* position on the space-filling curve                  */
  int32_t (*owner_rank)(void* ctx, int32_t ghost);      /* ghost = slot - num_local                             */
  int32_t (*level)(void* ctx, int32_t slot);
  int32_t (*num_faces)(void* ctx, int32_t slot);
  /* neighbours of LOCAL element e across its face f (t8_forest_leaf_face_neighbors): their slots and the face
   * numbers on their side; returns how many (0 = domain boundary, > 1 = finer neighbours), at most max_n  */
  int32_t (*face_neighbors)(void* ctx, int32_t e, int32_t f, int32_t max_n, int32_t* slots, int32_t* dual_faces);
  void    (*face_normal)(void* ctx, int32_t slot, int32_t f, double normal[3]);   /* outward, unit; ghosts too   */
  double  (*face_area)(void* ctx, int32_t slot, int32_t f);                       /* ghosts too                  */
  double  (*volume)(void* ctx, int32_t slot);                                     /* ghosts too                  */
  int32_t (*child_id)(void* ctx, int32_t slot);   /* t8_element_child_id; only for ..._create_subgrid (may be NULL otherwise) */
} T8gpuForestQuery;
void* t8gpu_host_connectivity_create(const T8gpuForestQuery* query);   /* NULL on a malformed query */
/* Subgrid meshes (subgrid_rank = 2 | 3, quad / hex forests): additionally face_level_difference[F] and
 * face_neighbor_offset[rank * F] of subgrid_mesh_manager.inl:587-680; normals keep 3 components here. */
void* t8gpu_host_connectivity_create_subgrid(const T8gpuForestQuery* query, int32_t subgrid_rank);
void  t8gpu_host_connectivity_subgrid_arrays(const void* connectivity, int32_t* face_level_difference, int32_t* face_neighbor_offset);
void  t8gpu_host_connectivity_destroy(void* connectivity);
/* counts[6] = {N, G, F, B, n_peers, n_send} */
void t8gpu_host_connectivity_counts(const void* connectivity, int64_t* counts);
/* face_neighbors[2F + B], normals[3 (F + B)], areas[F + B], volumes[N + G], peers[n_peers], recv_off[n_peers + 1],
 * send_off[n_peers + 1], send_idx[n_send]: the inputs of t8gpu_plan_plain_create and T8gpuHalo */
void t8gpu_host_connectivity_arrays(const void* connectivity, int32_t* face_neighbors, double* normals, double* areas,
                                    double* volumes, int32_t* peers, int32_t* recv_off, int32_t* send_off, int32_t* send_idx);
/* The synthetic forest behind the same callbacks (tests): destroy with t8gpu_synth_query_destroy. */
T8gpuForestQuery* t8gpu_synth_query_create(const void* mesh, int rank, int nranks);
void              t8gpu_synth_query_destroy(T8gpuForestQuery* query);

/* ---- VTK output (stands where t8_forest_write_vtk_ext is called: mesh_manager.inl:588-623,
 * subgrid_mesh_manager.inl:1051-1138,1185-1206) ------------------------------------------------------------
 * One .vtu piece: a VTK_QUAD / VTK_HEXAHEDRON per leaf with unshared corners, cell fields treeid, mpirank,
 * level, element_id and `num_fields` user fields (components 1 = scalar, 3 = xyz vector; doubles, one value
 * or triple per CELL). cells_per_dim = 4 writes every block as its 4^dim cells in z-order (level + 2), the
 * layout t8gpu_hip_column_major_to_z_order produces. ascii != 0: text arrays, else appended raw binary.
 * Returns 0, or 1 bad argument / 2 cannot open / 3 write error. */
int t8gpu_host_write_vtu(const char* path, int dim, int64_t num_elements, const double* centre, const int32_t* level,
                         int cells_per_dim, int mpirank, int64_t first_element_id, int num_fields, const char* const* names,
                         const int32_t* components, const double* const* data, int ascii);
/* The .pvtu that lists the per-rank pieces (written by rank 0). */
int t8gpu_host_write_pvtu(const char* path, int num_pieces, const char* const* piece_files, int num_fields,
                          const char* const* names, const int32_t* components);

/* ---- host utilities ------------------------------------------------------------------------------ */
/* memcpy over the planners' OpenMP threads (staging of large uploads into an application-owned pinned buffer) */
void t8gpu_host_parallel_copy(void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* T8GPU_HOST_H */
