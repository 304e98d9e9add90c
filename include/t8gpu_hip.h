/* t8gpu_hip.h -- C ABI of the MI355X (gfx950) backend for t8gpu's finite-volume hot path.
 *
 * t8gpu itself has no FFI: its boundary is the C++ template API of t8gpu/ (see include/t8gpu/ for
 * the HIP-backed mirror of those headers). This C ABI is what those headers -- or any other host
 * language -- bind to reach the hand-written HIP kernels. Every entry point names the reference
 * kernel / function it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - suffix _f32 / _f64 selects float_type (reference: t8gpu/memory/memory_manager.h:29, float only).
 *   - all pointers are DEVICE pointers unless the name says host; `stream` is a hipStream_t (NULL =
 *     default stream). Calls are asynchronous on that stream and safe to capture in a hipGraph.
 *   - T8gpuVars_*: the 5 variable planes of one step, exactly the pointers a
 *     MemoryAccessorOwn<VariableList> holds (memory_manager.h:173): Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e.
 *   - ghosts: element indices refer to local slots; ghost elements live in mirror slots [N, N+G)
 *     of the same planes. `indices` (nullable) is the element->slot map of
 *     MeshConnectivityAccessor::get_element_owner_remote_index (mesh_manager.h:155-157).
 *   - return value: 0 on success, otherwise the hipError_t (or ncclResult_t + 10000) code; the C++
 *     wrappers turn non-zero into the reference's print-and-abort (t8gpu/utils/cuda.h:7-15).
 *   - flux_kind: 0 = KEPES (the flux the reference runs), 1 = HLL (reference dead code,
 *     examples/subgrid/kernels.inl:263-332), 2 = HLLC (NOT in the reference: the project brief names it; the HLL
 *     above with the contact wave restored, same wave-speed estimates; oracle/oracle.hpp states the formulas).
 */
#ifndef T8GPU_HIP_H
#define T8GPU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T8GPU_FLUX_KEPES 0
#define T8GPU_FLUX_HLL 1
#define T8GPU_FLUX_HLLC 2

typedef struct T8gpuVars_f32 { float* p[5]; } T8gpuVars_f32;
typedef struct T8gpuVars_f64 { double* p[5]; } T8gpuVars_f64;

/* ---- library / device ------------------------------------------------------------------------ */
int         t8gpu_hip_abi_version(void);
int         t8gpu_hip_device_count(int* count);
int         t8gpu_hip_set_device(int device);
const char* t8gpu_hip_error_string(int code);
/* roctx ranges for `rocprofv3 --marker-trace`: active only with T8GPU_ROCTX=1 in the environment
 * (the roctx library is dlopen'ed then); both return 1 when a range was pushed / popped, 0 when ranges are off. The
 * step drivers mark iterate_steps and every RK stage themselves. */
int         t8gpu_hip_range_push(const char* name);
int         t8gpu_hip_range_pop(void);
/* Name of the kernel the most recent t8gpu_hip_*_fused_stage_* call launched for the bulk of its tiles / blocks, spelled as
 * rocprofv3 prints it (e.g. "k_plain_stage<double, 0, 3>"); "" before the first call. bench.py compares it with the kernels
 * named by the committed profile before it reports that profile's PMC figures. Not thread-safe (one host thread per rank). */
const char* t8gpu_hip_last_stage_kernel(void);

/* ---- plain elements, reference-dataflow kernels ("compat" tier) ------------------------------- */

/* kepes_compute_fluxes<<<ceil(F/256),256>>>, examples/compressible_euler/kernels.cu:135-309
 * (declared kernels.h:22-27). face_neighbors = [2F] (l,r) pairs, face_normals = [normal_dim*F] AoS,
 * face_surfaces = [F]; scatter-adds -F to l and +F to r; speed_estimates[F] may be NULL. */
int t8gpu_hip_flux_faces_f32(int flux_kind, int num_faces, int normal_dim, const int32_t* face_neighbors,
                             const int32_t* indices, const float* face_normals, const float* face_surfaces,
                             T8gpuVars_f32 state, T8gpuVars_f32 fluxes, float* speed_estimates, void* stream);
int t8gpu_hip_flux_faces_f64(int flux_kind, int num_faces, int normal_dim, const int32_t* face_neighbors,
                             const int32_t* indices, const double* face_normals, const double* face_surfaces,
                             T8gpuVars_f64 state, T8gpuVars_f64 fluxes, double* speed_estimates, void* stream);

/* reflective_boundary_condition<<<ceil(B/256),256>>>, kernels.cu:311-469 (kernels.h:39-44). The
 * arrays are the SAME arrays as above; boundary entries start after the F interior ones
 * (t8gpu/mesh/mesh_manager.h:68-70,92-98,132-134). speed_estimates[F + i] written if non-NULL. */
int t8gpu_hip_flux_boundary_f32(int flux_kind, int num_faces, int num_boundary_faces, int normal_dim,
                                const int32_t* face_neighbors, const float* face_normals,
                                const float* face_surfaces, T8gpuVars_f32 state, T8gpuVars_f32 fluxes,
                                float* speed_estimates, void* stream);
int t8gpu_hip_flux_boundary_f64(int flux_kind, int num_faces, int num_boundary_faces, int normal_dim,
                                const int32_t* face_neighbors, const double* face_normals,
                                const double* face_surfaces, T8gpuVars_f64 state, T8gpuVars_f64 fluxes,
                                double* speed_estimates, void* stream);

/* timestepping::SSP_3RK_step{1,2,3}<V><<<ceil(N/256),256>>>, t8gpu/timestepping/ssp_runge_kutta.inl:30-99.
 * stage 1: out = prev + dt/vol*f;  stage 2: out = .75 prev + .25 mid + .25 dt/vol*f;
 * stage 3: out = c31 prev + c32 mid + c33 dt/vol*f (truncated literals, :12-14,23-25); f := 0. */
int t8gpu_hip_rk3_stage_f32(int stage, int num_elements, T8gpuVars_f32 prev, T8gpuVars_f32 mid, T8gpuVars_f32 out,
                            T8gpuVars_f32 fluxes, const float* volume, float delta_t, void* stream);
int t8gpu_hip_rk3_stage_f64(int stage, int num_elements, T8gpuVars_f64 prev, T8gpuVars_f64 mid, T8gpuVars_f64 out,
                            T8gpuVars_f64 fluxes, const double* volume, double delta_t, void* stream);

/* ---- Subgrid<4,4> (rank 2) and Subgrid<4,4,4> (rank 3), reference-dataflow kernels ------------- */

/* compute_inner_fluxes<Subgrid><<<N, block_size>>>, examples/subgrid/kernels.inl:335-662. */
int t8gpu_hip_subgrid_inner_f32(int flux_kind, int rank, int num_elements, T8gpuVars_f32 state, T8gpuVars_f32 fluxes,
                                const float* volumes, void* stream);
int t8gpu_hip_subgrid_inner_f64(int flux_kind, int rank, int num_elements, T8gpuVars_f64 state, T8gpuVars_f64 fluxes,
                                const double* volumes, void* stream);

/* compute_outer_fluxes<Subgrid><<<F, (4,4)|(4)>>>, kernels.inl:664-911. face_level_difference[F],
 * face_neighbor_offset[rank*F] as in t8gpu/mesh/subgrid_mesh_manager.h:108-126. */
int t8gpu_hip_subgrid_outer_f32(int flux_kind, int rank, int num_faces, const int32_t* face_neighbors,
                                const int32_t* indices, const int32_t* face_level_difference,
                                const int32_t* face_neighbor_offset, const float* face_normals,
                                const float* face_surfaces, T8gpuVars_f32 state, T8gpuVars_f32 fluxes, void* stream);
int t8gpu_hip_subgrid_outer_f64(int flux_kind, int rank, int num_faces, const int32_t* face_neighbors,
                                const int32_t* indices, const int32_t* face_level_difference,
                                const int32_t* face_neighbor_offset, const double* face_normals,
                                const double* face_surfaces, T8gpuVars_f64 state, T8gpuVars_f64 fluxes, void* stream);

/* compute_boundary_fluxes<Subgrid><<<B, (4,4)|(4)>>>, kernels.inl:913-1107. */
int t8gpu_hip_subgrid_boundary_f32(int flux_kind, int rank, int num_faces, int num_boundary_faces,
                                   const int32_t* face_neighbors, const float* face_normals,
                                   const float* face_surfaces, T8gpuVars_f32 state, T8gpuVars_f32 fluxes,
                                   void* stream);
int t8gpu_hip_subgrid_boundary_f64(int flux_kind, int rank, int num_faces, int num_boundary_faces,
                                   const int32_t* face_neighbors, const double* face_normals,
                                   const double* face_surfaces, T8gpuVars_f64 state, T8gpuVars_f64 fluxes,
                                   void* stream);

/* timestepping::subgrid::SSP_3RK_step{1,2,3}<V,Subgrid><<<N, block_size>>>, ssp_runge_kutta.inl:101-221
 * (per-subcell volume = volumes[e] / Subgrid::size). */
int t8gpu_hip_subgrid_rk3_stage_f32(int stage, int rank, int num_elements, T8gpuVars_f32 prev, T8gpuVars_f32 mid,
                                    T8gpuVars_f32 out, T8gpuVars_f32 fluxes, const float* volumes, float delta_t,
                                    void* stream);
int t8gpu_hip_subgrid_rk3_stage_f64(int stage, int rank, int num_elements, T8gpuVars_f64 prev, T8gpuVars_f64 mid,
                                    T8gpuVars_f64 out, T8gpuVars_f64 fluxes, const double* volumes, double delta_t,
                                    void* stream);

/* ---- plain elements, fused tile kernels ("fast" tier) ------------------------------------------
 * One launch per RK stage replaces kepes_compute_fluxes + reflective_boundary_condition +
 * SSP_3RK_stepK of that stage (solver.cu:81-110 / 115-142 / 147-174): every tile (compact window of
 * owned elements) stages per-element primitives in LDS, evaluates each of its faces once, sums the
 * fluxes per element in a fixed order and applies the RK stage. The Fluxes planes are neither read
 * nor written (the reference leaves them zero after every stage); speed_estimates is written as the
 * reference does (may be NULL). The plan is built on the host by t8gpu_plan_plain_create()
 * (t8gpu_amd/csrc/host/tile_plan.cpp) at every connectivity rebuild and uploaded by the caller. */
typedef struct T8gpuPlainPlan {
  const int32_t*  elem_off;   /* [ntiles+1] first owned element of each tile                      */
  const int32_t*  halo_off;   /* [ntiles+1] into halo_ids                                         */
  const int32_t*  face_off;   /* [ntiles+1] into face_*                                           */
  const int32_t*  halo_ids;   /* slots of outside elements (other tiles / ghost mirrors)          */
  const uint32_t* face_lr;    /* tile-local l | r << 16 (r = 0xFFFF: reflective wall)             */
  const void*     face_geo;   /* float_type [n_faces][4] = nx, ny, nz, area                       */
  const int32_t*  face_orig;  /* original face index if this tile reports the speed, else -1      */
  const int32_t*  csr_off;    /* [N+1] into csr_ent. The two CSR arrays are read by the generic kernel only: may be NULL where `ell`
                               * and `tile_desc` are given and max_elems <= 256, max_slots <= 512, max_faces <= 1024 (the pipelined kernels) */
  const uint16_t* csr_ent;    /* tile-local face | 0x8000 when the element is the face's right side */
  const int32_t*  tile_order; /* [ntiles] deep-interior tiles, then interior tiles that read an element
                               * owned by a ghost-reading tile, then the tiles reading ghost slots */
  int32_t ntiles, n_interior_tiles, max_elems, max_halo, max_faces, ell_width;
  /* optional compressed forms (NULL = absent); with them and tiles of <= 256 elements, <= 512 own+halo
   * elements and <= 1024 faces the software-pipelined kernel variant is used (two passes of 256 faces up
   * to 512 faces per tile, up to four above) */
  const uint16_t* ell;        /* [rows][ell_width] copy of the CSR lists, 0xFFFF-padded, 16-byte rows. ABI 5: rows exist for the
                               * elements of GENERIC tiles only; a tile's first row is word 6 of its tile_desc record, element e
                               * of the tile is row (word 6) + (e - first element)                                  */
  const uint16_t* geo_idx;    /* per tile face: row of geo_table (bits 0-12) | direction code << 13: 2 * axis + (normal
                               * along +axis) for an exact axis normal, 6 otherwise. Inside every block of 256 tile
                               * faces the faces are ordered by that code                                  */
  const void*     geo_table;  /* float_type [n_geo][12]: distinct {nx,ny,nz,area, t1x,t1y,t1z,0, t2x,t2y,t2z,0} */
  int32_t n_geo;
  int32_t max_slots;          /* max over tiles of own + halo elements (0: unknown, max_elems + max_halo is used) */
  int32_t n_deep_tiles;       /* leading tiles of tile_order that read nothing a ghost-reading tile owns (0: unknown) */
  int32_t n_slots_addressed;  /* owned + ghost elements of the planes the plan indexes (ABI 5; was reserved). The patch kernels
                               * address a plane by a 32-bit byte offset and refuse plans whose planes reach 4 GiB (or that
                               * leave this 0): build the plan without patches for such meshes */
  const int32_t* tile_desc;   /* [ntiles][8], in tile_order order: {first element, elements, first halo entry, halo entries,
                               * first face, faces, first ELL row, 0} of tile_order[k] -- one 32-byte record per tile for the persistent
                               * kernel, which reads it several tiles ahead (NULL: that kernel is not used) */
  /* STRUCTURED PATCHES (ABI 4; t8gpu_plan_plain_create_ex flag 1, csrc/host/tile_plan.cpp: find_patches): tiles of 256
   * consecutive elements that form an aligned 16 x 16 block of same-size quadrilaterals with the canonical face listing.
   * They have no face records; their tile_desc is {first element, 256, first halo entry, 64, id of the first own face,
   * 0x100 | flags, area as a double}, their 64 halo entries the elements across the -x | +x | -y | +y sides. The first
   * n_patch_tiles[c] tiles of class c of tile_order (c = 0: [0, n_deep_tiles), 1: [n_deep_tiles, n_interior_tiles),
   * 2: the rest) are patch tiles; they run through kernels_fused_patch.hip. All zero: no patches.
   * Flag 0x400 in word 5 of a patch descriptor: every element of the patch has one volume, carried as a double in words
   * 1 and 3 (the planner checked the volumes it was given, t8gpu_plan_plain_patch_volumes); the kernels then do not read
   * `volume[]` for that patch -- rebuild the plan if the volumes change. */
  int32_t n_patch_tiles[3];
  int32_t patch_dim;          /* 2: the 16 x 16 patches above; 3: 8 x 8 x 4 blocks of same-size hexahedra (256 halo entries:
                               * -x 32 | +x 32 | -y 32 | +y 32 | -z 64 | +z 64; own faces fbase + 3 t + {0, 1, 2}; flags bits 0-2:
                               * -y before -x, -z before -x, -z before -y where both coordinates of the pair are 0); 0: no patches */
  int32_t n_irregular_tiles[3]; /* ABI 6: the LAST so many of the n_patch_tiles[c] patch tiles of class c are IRREGULAR 3D patches
                               * (t8gpu_host.h: t8gpu_plan_plain_create_ex flag 8): descriptor word 5 has 0x800, word 4 is the
                               * first of 512 face_lr / face_orig entries with the per-cell words. They run through the
                               * irregular instantiation of k_plain_patch3 in a launch of their own */
  /* GHOST WINDOW (ABI 7). Run-time attachments of the multi-rank step driver (csrc/hip/stepper.hip), all NULL / 0 in a plan
   * that comes from the planner -- callers of t8gpu_hip_plain_fused_stage_* leave them so unless they do what the driver does.
   * With them a launch of the ghost-reading tiles needs no pack and no unpack kernel around the RCCL exchange:
   *   ghost_buf != NULL: a slot s >= n_owned is read from ghost_buf[5 * (s - n_owned) + var] -- the receive buffer of the
   *     exchange in its wire format -- instead of the mirror slot s of the planes (which is then neither read nor written);
   *   send_map != NULL: after the RK update of owned element e the five new values also go to send_buf[5 * t + var] for every
   *     send slot t of e: send_map[e] = -1 (not sent), t >= 0 (one slot), or -(2 + i) (several: send_list[i], send_list[i+1],
   *     ... up to and including the first entry with bit 31 set; slot = entry & 0x7FFFFFFF).
   * Taken by the 2D patch kernel and the one-tile kernels (launches with either pointer set run one tile per workgroup); the
   * launchers refuse them for 3D patch tiles (hipErrorInvalidValue). */
  const void*    ghost_buf;   /* DEVICE float_type [5 * G], element-major                                   */
  const int32_t* send_map;    /* DEVICE [n_owned]                                                           */
  const int32_t* send_list;   /* DEVICE, may be NULL when no element has more than one send slot            */
  void*          send_buf;    /* DEVICE float_type [5 * n_send], element-major                              */
  int32_t        n_owned;     /* N: first ghost slot                                                        */
  int32_t        reserved7;
} T8gpuPlainPlan;

/* tile_begin/tile_count select a range of tile_order (0, ntiles = everything; [0, n_interior) can run
 * while the halo exchange of `src` is still in flight, [n_interior, ntiles) after it). mid = state
 * the fluxes are evaluated on (prev for stage 1, Step1, Step2); ghost slots of `mid` must be current.
 * Stage 1 is u1 = u0 + dt/vol f(u0) (ssp_runge_kutta.inl:30-50): prev and mid MUST be the same planes there,
 * anything else returns hipErrorInvalidValue. speed_estimates (may be NULL) gets one value per face for EVERY
 * flux_kind: |uHat| + aHat for KEPES (kernels.cu:222), max(|S_l|, |S_r|) of the wave-speed bounds for HLL / HLLC. */
int t8gpu_hip_plain_fused_stage_f32(int flux_kind, int stage, const T8gpuPlainPlan* plan, int tile_begin,
                                    int tile_count, T8gpuVars_f32 prev, T8gpuVars_f32 mid, T8gpuVars_f32 out,
                                    const float* volume, float delta_t, float* speed_estimates, void* stream);
int t8gpu_hip_plain_fused_stage_f64(int flux_kind, int stage, const T8gpuPlainPlan* plan, int tile_begin,
                                    int tile_count, T8gpuVars_f64 prev, T8gpuVars_f64 mid, T8gpuVars_f64 out,
                                    const double* volume, double delta_t, double* speed_estimates, void* stream);

/* 1 if a whole-plan launch of `tile_count` tiles (flux_kind, float_size = 4 | 8) would run the persistent, software-
 * pipelined tile kernel (kernels_fused_persistent.hip), 0 if it goes to the one-tile-per-workgroup kernels: the launcher's
 * own test, for host code that picks tile caps (t8gpu_amd/fused.py). Only the plan's integer fields and the NULL-ness of
 * tile_desc / ell / geo_idx / geo_table are looked at; no GPU is needed. */
int t8gpu_hip_plain_persistent_accepts(const T8gpuPlainPlan* plan, int flux_kind, int float_size, int tile_count);
/* 1 if a launch of this plan's generic tiles reads csr_off / csr_ent (the generic kernel), 0 if it runs the pipelined kernels
 * (ELL rows + tile descriptors), whose callers need not upload the CSR lists: the launcher's own test, from the plan's
 * integer fields and the NULL-ness of ell / tile_desc alone; no GPU is needed. */
int t8gpu_hip_plain_needs_csr(const T8gpuPlainPlan* plan);
/* The tangent rows of a geometry dictionary ALREADY ON THE DEVICE (`geo_table`: n_geo rows {n, area} {t1, .} {t2, .} of the
 * plan's float type), recomputed from its normals by the routine the per-face kernels use (flux_math.hpp: face_basis_fast --
 * the frame of kernels.cu:174-193 with one reciprocal square root instead of a square root and three divisions). Plan builders
 * call it once after the upload, so that a tile evaluated with the dictionary and the same face evaluated with per-face
 * geometry rows (a partition of the mesh, another tile cap) see the same bits; the host planner's own frames
 * (t8gpu_plan_plain_compressed) differ from these in the last bit on oblique normals. */
int t8gpu_hip_plain_geo_frames_f32(void* geo_table, int n_geo, void* stream);
int t8gpu_hip_plain_geo_frames_f64(void* geo_table, int n_geo, void* stream);

/* ---- ghost-layer exchange (device side) ----------------------------------------------------------
 * Replaces the reference's cross-rank pointer sharing (cudaIpc*, t8gpu/memory/shared_device_vector.inl:
 * 15-30,159-199) and remote atomics (kernels.cu:295-308): ghosts are mirror slots [N, N+G) of the same
 * planes, refreshed once per RK stage by  pack -> RCCL send/recv per neighbour rank -> unpack.
 * Wire format: 5 values per element, element-major (sendbuf[5*t + var]); each peer's elements are a
 * contiguous run of send_idx / of the ghost slots, so one message per peer and direction. The
 * transport (ncclSend/ncclRecv in one group, on the stream of these kernels) is issued by the host
 * side (t8gpu_amd/halo.py through torch.distributed's RCCL communicator) or natively (T8gpuHalo below).
 * cells_per_element = 1 for plain elements, Subgrid::size (16 / 64) for blocks (a ghost block mirrors all
 * its subcells; wire format (element, cell, variable)). */
int t8gpu_hip_halo_pack_f32(int n_send, int cells_per_element, const int32_t* send_idx, T8gpuVars_f32 state,
                            float* sendbuf, void* stream);
int t8gpu_hip_halo_pack_f64(int n_send, int cells_per_element, const int32_t* send_idx, T8gpuVars_f64 state,
                            double* sendbuf, void* stream);
int t8gpu_hip_halo_unpack_f32(int num_ghosts, int first_ghost_slot, int cells_per_element, const float* recvbuf,
                              T8gpuVars_f32 state, void* stream);
int t8gpu_hip_halo_unpack_f64(int num_ghosts, int first_ghost_slot, int cells_per_element, const double* recvbuf,
                              T8gpuVars_f64 state, void* stream);

/* ---- native RCCL transport + whole-step driver ----------------------------------------------------
 * One communicator per process (one rank per GPU); the 128-byte id is created on one rank and
 * distributed by the caller (MPI_Bcast, torch.distributed, ...). Replaces the MPI_Allgather of CUDA-IPC
 * handles at construction and at EVERY resize (shared_device_vector.inl:21-22,95-96,179-185,270-276). */
int t8gpu_hip_comm_unique_id(char* id128);
int t8gpu_hip_comm_create(const char* id128, int rank, int nranks, void** comm);
int t8gpu_hip_comm_destroy(void* comm);
int t8gpu_hip_comm_abort(void* comm);
int t8gpu_hip_stream_wait(void* stream, double timeout_s); /* 0 idle, 1 timed out, else hipError_t */
/* out4 = {RCCL version of the headers this library was compiled with (NCCL_VERSION_CODE), RCCL version of the library the
 * process bound (ncclGetVersion), HIP_VERSION of the headers, hipRuntimeGetVersion}. A process may bind another build than
 * the headers came from (the torch wheel's librccl / libamdhip64 against /opt/rocm's headers): t8gpu_hip_comm_create checks
 * the pairs -- same major versions, RCCL >= 2.7 (the send / recv API this library uses, unchanged since) -- and returns
 * hipErrorNotSupported otherwise; bench.py reports both pairs. */
int t8gpu_hip_runtime_versions(int out4[4]);

typedef struct T8gpuHalo {
  int32_t num_elements, num_ghosts, n_peers, n_send;
  int32_t cells_per_element, reserved; /* 1 (plain elements) or Subgrid::size                            */
  const int32_t* peers;     /* HOST [n_peers] neighbour ranks, ascending                               */
  const int32_t* send_off;  /* HOST [n_peers+1] ranges of send_idx per peer                            */
  const int32_t* recv_off;  /* HOST [n_peers+1] ranges of the ghost slots (relative to N) per peer     */
  const int32_t* send_idx;  /* DEVICE [n_send] owned elements mirrored on a peer                       */
  void* sendbuf;            /* DEVICE 5*n_send*cells_per_element float_type                            */
  void* recvbuf;            /* DEVICE 5*num_ghosts*cells_per_element float_type                        */
  void* comm;               /* from t8gpu_hip_comm_create                                              */
} T8gpuHalo;

/* pack -> ncclGroupStart, ncclRecv/ncclSend per peer, ncclGroupEnd -> unpack, all on `stream`. */
int t8gpu_hip_halo_exchange_f32(const T8gpuHalo* halo, T8gpuVars_f32 state, void* stream);
int t8gpu_hip_halo_exchange_f64(const T8gpuHalo* halo, T8gpuVars_f64 state, void* stream);

/* CompressibleEulerSolver::iterate (solver.cu:75-175) as one call: `planes` is the MemoryManager
 * allocation (26 planes of `stride`, plane = step*5+var, volume = plane 25; memory_manager.h:460), prev /
 * next the step ids AFTER the caller's std::swap (solver.cu:76). Enqueues, per stage, the exchange and the
 * ghost-reading tiles on an internal second stream beside the interior tiles on `stream` (two lanes, each waiting
 * only for the other lane's PREVIOUS stage; the comm lane is enqueued by a host thread of the stepper's own --
 * T8GPU_STEPPER_THREADS=0: by the caller's thread; csrc/hip/stepper.hip) and returns; no host sync. Work queued on
 * `stream` before the call is seen by it, work queued after the call sees its result. For plain 2D / tile plans
 * the ghost-reading tiles read the receive buffer and fill the send buffer themselves (T8gpuPlainPlan "ghost
 * window"): the ghost mirror slots [N, N+G) of the planes are then NOT refreshed by this driver -- code that reads
 * them between steps calls t8gpu_hip_halo_exchange_* (T8GPU_GHOST_WINDOW=0: pack / unpack kernels, slots refreshed).
 * iterate_steps: n_steps consecutive steps in one call, prev / next given for the FIRST step and swapped
 * from step to step (after an odd n_steps the caller's roles are swapped once more); the two streams then
 * meet only at the entry and the exit of the call instead of once per step. */
int t8gpu_hip_plain_stepper_create(const T8gpuPlainPlan* plan, const T8gpuHalo* halo_or_null, void** stepper);
int t8gpu_hip_plain_stepper_destroy(void* stepper);
int t8gpu_hip_plain_stepper_iterate_f32(void* stepper, int flux_kind, float* planes, size_t stride, int prev, int next,
                                        float delta_t, float* speed_estimates, void* stream);
int t8gpu_hip_plain_stepper_iterate_f64(void* stepper, int flux_kind, double* planes, size_t stride, int prev, int next,
                                        double delta_t, double* speed_estimates, void* stream);
int t8gpu_hip_plain_stepper_iterate_steps_f32(void* stepper, int flux_kind, float* planes, size_t stride, int prev, int next,
                                              float delta_t, float* speed_estimates, int n_steps, void* stream);
int t8gpu_hip_plain_stepper_iterate_steps_f64(void* stepper, int flux_kind, double* planes, size_t stride, int prev, int next,
                                              double delta_t, double* speed_estimates, int n_steps, void* stream);
/* optional HIP-event timing of the stage kernels (for roofline accounting): enable = 0 off, n > 0 = the stage
 * kernels of every n-th step of a call are bracketed by events (n > 1 keeps the host-side cost of the
 * events out of latency-bound multi-rank runs); elapsed() sums what has been recorded since. */
/* hipGraph replay ("hipGraph capture of the 3-stage step"): enable = 1 -> an iterate_steps() call is captured
 * once per argument set (the four most recent sets are kept: a step loop alternates between two) and replayed with ONE
 * hipGraphLaunch afterwards; enable = 0 -> direct enqueue (default); enable < 0 -> query only. counts (may be NULL)
 * receives {captures, replays}. A capture the runtime refuses returns its error code. delta_t is part of the argument set
 * (a CFL-adaptive step size re-captures per value: use the direct enqueue there).
 * A stepper WITH a halo enqueues directly whatever this switch says, unless T8GPU_GRAPH_RCCL=1 is in the environment
 * (opt-in): the RCCL groups are then captured too, on the ORIGIN stream of the capture (the deep tiles fork off instead; an
 * RCCL group on a forked stream of a capture crashes hipStreamEndCapture on this stack, HIP 7.0.51831 / RCCL 2.26.6 of the
 * torch wheel). That has only ever run on one GPU with a one-rank communicator exchanging with itself
 * (tests/test_gpu_graph.py), never across xGMI, and costs more host time than the two-lane direct enqueue. */
int t8gpu_hip_plain_stepper_graph(void* stepper, int enable, int* counts);
int t8gpu_hip_plain_stepper_timing(void* stepper, int enable);
/* Diagnostics (scripts/halo_overhead.py): with T8GPU_STEPPER_PROFILE=1 in the environment the step drivers time their own
 * host calls; ns4 / calls4 (may be NULL) receive nanoseconds and counts for {kernel launches, RCCL groups, event records,
 * stream waits} since the last reset. Returns 1 when the profile is on, 0 when off (all zeros). */
int t8gpu_hip_stepper_host_profile(int reset, double* ns4, long long* calls4);
int t8gpu_hip_plain_stepper_elapsed(void* stepper, double* total_ms, int* launches);
int t8gpu_hip_plain_stepper_timed_stages(void* stepper); /* RK stages covered by elapsed() */
/* host time the multi-rank driver spent enqueueing (wall time of its iterate calls) and the steps that covers, since
 * creation or the last reset; 0 / 0 for single-rank steppers */
int t8gpu_hip_plain_stepper_host_time(void* stepper, int reset, double* total_ms, long long* steps);

/* ---- Subgrid<4,4,4> and Subgrid<4,4>, fused block kernels ("fast" tier) -----------------------------
 * One launch per RK stage replaces compute_inner_fluxes + compute_boundary_fluxes + compute_outer_fluxes
 * + subgrid::SSP_3RK_stepK of that stage (examples/subgrid/solver.inl:166-195): one 64-lane wavefront per
 * 4x4x4 block (or per four 4x4 blocks) evaluates every flux its subcells need (an outer sub-face is evaluated by both blocks that
 * share it) and applies the RK stage; the Fluxes planes are neither read nor written. The plan is the
 * per-block face list built by t8gpu_plan_subgrid_create() (csrc/host/subgrid_plan.cpp). */
typedef struct T8gpuSubgridPlan {
  /* joined records, t8gpu_plan_subgrid_records(): one dependent load level between a wavefront's position
   * and all of its far-cell loads */
  const int32_t* block_rec;     /* [num_elements][32], blocks that touch no ghost block first: {block, n generic
                                   faces, first bf_rec entry, 0, 3 x the +d face {other block (-1 wall, -2 none / in
                                   the generic list), code, area (2 words)}, 3 x the -d face likewise, 4 spare words}
                                   (128-byte rows) */
  const int32_t* bf_rec;        /* [n_entries][4] generic faces (towards finer blocks) in the same order: {other block,
                                   code (bit 12: the block is the face's RIGHT side), area (2 words)} */
  int32_t num_elements, rank, max_faces_per_block, n_interior_blocks;
  int32_t n_deep_blocks;        /* leading blocks that have no neighbour touching a ghost block (0: unknown) */
  int32_t n_blocks_addressed;   /* 1 + the largest block index any record refers to (owned and ghost blocks; sizes[5] of
                                   t8gpu_plan_subgrid_sizes): lets the kernel use 32-bit byte offsets into the state planes
                                   when a plane is shorter than 4 GiB. 0 = unknown (64-bit addressing) */
  /* optional (t8gpu_plan_subgrid_family_records()): 2x2x2 cubes (RANK 3) / 2x2 squares (RANK 2) of consecutive same-level
   * deep interior blocks take the family kernels (one workgroup of 8 wavefronts per cube / one wavefront per square) when a
   * launch covers the whole plan or exactly its first class [0, n_deep_blocks); the blocks outside every family then run
   * through rest_rec (launches inside the later classes read rest_rec too). n_families = 0: every block through block_rec. */
  const int32_t* fam_rec;       /* RANK 3: [n_families][160] = {first block, 0, 0, 0, 36 rows {far, code, area (2 words)}}: the 12
                                   outward + faces, the 12 outward - faces, the 12 inner faces; RANK 2: [n_families][64] with
                                   4 + 4 + 4 rows (layout: subgrid_plan.cpp) */
  const int32_t* rest_rec;      /* [n_rest][32]: block_rec rows of the blocks outside every family, in block_order order */
  int32_t n_families, n_rest;
} T8gpuSubgridPlan;

/* block_begin/block_count select a range of block_order (0, num_elements = everything; [0, n_interior_blocks)
 * can run while the halo exchange of `mid` is in flight). */
int t8gpu_hip_subgrid_fused_stage_f32(int flux_kind, int stage, const T8gpuSubgridPlan* plan, int block_begin,
                                      int block_count, T8gpuVars_f32 prev, T8gpuVars_f32 mid, T8gpuVars_f32 out,
                                      const float* volumes, float delta_t, void* stream);
int t8gpu_hip_subgrid_fused_stage_f64(int flux_kind, int stage, const T8gpuSubgridPlan* plan, int block_begin,
                                      int block_count, T8gpuVars_f64 prev, T8gpuVars_f64 mid, T8gpuVars_f64 out,
                                      const double* volumes, double delta_t, void* stream);

/* Native step driver for Subgrid blocks: SubgridCompressibleEulerSolver::iterate (examples/subgrid/solver.inl:152-266)
 * in one host call, the pipeline of t8gpu_hip_plain_stepper_* over the block classes of the plan (deep interior /
 * near-boundary / ghost-touching blocks on three streams, one RCCL exchange of whole ghost blocks per stage;
 * halo->cells_per_element = 4^rank). `planes` = 25 variable planes of `stride` values (stride >= (N + G) * 4^rank),
 * `volumes` the per-block volume array. The handle is destroyed / timed with the t8gpu_hip_plain_stepper_destroy,
 * _timing, _elapsed and _timed_stages entry points above. */
int t8gpu_hip_subgrid_stepper_create(const T8gpuSubgridPlan* plan, const T8gpuHalo* halo_or_null, void** stepper);
int t8gpu_hip_subgrid_stepper_iterate_steps_f32(void* stepper, int flux_kind, float* planes, size_t stride, const float* volumes,
                                                int prev, int next, float delta_t, int n_steps, void* stream);
int t8gpu_hip_subgrid_stepper_iterate_steps_f64(void* stepper, int flux_kind, double* planes, size_t stride, const double* volumes,
                                                int prev, int next, double delta_t, int n_steps, void* stream);

/* ---- scalar reductions next to the hot path -------------------------------------------
 * Device-side replacements of the two host round trips of the reference solvers; results are device
 * scalars (double) written on `stream`; `workspace` = t8gpu_hip_reduce_workspace_bytes() device bytes.
 * Fixed reduction tree, no atomics: bitwise reproducible. */
size_t t8gpu_hip_reduce_workspace_bytes(void);
/* max over the per-face speed estimates: thrust::reduce(..., maximum) in compute_timestep,
 * examples/compressible_euler/solver.cu:213-217 (the caller all-reduces the scalar across ranks, :218-223,
 * and forms dt = cfl * 0.5^max_level / speed, :225-228). */
int t8gpu_hip_max_speed_f32(size_t n, const float* speed_estimates, void* workspace, double* result, void* stream);
int t8gpu_hip_max_speed_f64(size_t n, const double* speed_estimates, void* workspace, double* result, void* stream);
/* sum of volume * variable over the owned cells: compute_integral, solver.cu:190-211; with
 * cells_per_element = Subgrid::size the per-cell volume is volume[e] / size (examples/subgrid/solver.inl:295-297). */
int t8gpu_hip_integral_f32(size_t num_cells, int cells_per_element, const float* variable, const float* volume,
                           void* workspace, double* result, void* stream);
int t8gpu_hip_integral_f64(size_t num_cells, int cells_per_element, const double* variable, const double* volume,
                           void* workspace, double* result, void* stream);

/* ---- AMR indicator and data transfer for plain elements ---------------------------------
 * estimate_gradient<<<>>>: examples/compressible_euler/kernels.cu:471-501 (|rho_r - rho_l| added to both
 * neighbours; the reference accumulates into its Fluxes/Rho plane, any zeroed plane works). */
int t8gpu_hip_estimate_gradient_f32(int num_faces, const int32_t* face_neighbors, const int32_t* indices,
                                    const float* rho, float* gradient, void* stream);
int t8gpu_hip_estimate_gradient_f64(int num_faces, const int32_t* face_neighbors, const int32_t* indices,
                                    const double* rho, double* gradient, void* stream);
/* compute_refinement_criteria<<<>>>: examples/compressible_euler/solver.cu:231-241 (gradient / cbrt(volume)). */
int t8gpu_hip_refinement_criteria_f32(int num_elements, const float* gradient, const float* volume, float* criteria,
                                      void* stream);
int t8gpu_hip_refinement_criteria_f64(int num_elements, const double* gradient, const double* volume, double* criteria,
                                      void* stream);
/* adapt_variables_and_volume<<<>>>: t8gpu/mesh/mesh_manager.inl:165-193. adapt_data[n_new+1] = first old element of
 * each new element (mesh_manager.inl:258-281): equal neighbours = children of a refined element (injection, volume
 * / 2^dim), difference 2^dim = coarsened family (mean, volume * 2^dim). dim = 3 reproduces the reference's
 * hard-coded factors 0.125 / 8.0; pass 2 for quad forests. */
int t8gpu_hip_adapt_variables_and_volume_f32(int num_new_elements, int dim, const int32_t* adapt_data,
                                             T8gpuVars_f32 old_variables, T8gpuVars_f32 new_variables,
                                             const float* volume_old, float* volume_new, void* stream);
int t8gpu_hip_adapt_variables_and_volume_f64(int num_new_elements, int dim, const int32_t* adapt_data,
                                             T8gpuVars_f64 old_variables, T8gpuVars_f64 new_variables,
                                             const double* volume_old, double* volume_new, void* stream);
/* device half of partition_data<<<>>> (mesh_manager.inl:626-643): a contiguous run of elements <-> one message of
 * 6 planes (5 variables + volume) of n values. */
int t8gpu_hip_gather_elements_f32(int n, int first, T8gpuVars_f32 variables, const float* volume, float* out, void* stream);
int t8gpu_hip_gather_elements_f64(int n, int first, T8gpuVars_f64 variables, const double* volume, double* out, void* stream);
int t8gpu_hip_scatter_elements_f32(int n, int first, const float* in, T8gpuVars_f32 variables, float* volume, void* stream);
int t8gpu_hip_scatter_elements_f64(int n, int first, const double* in, T8gpuVars_f64 variables, double* volume, void* stream);

/* MeshManager::partition / SubgridMeshManager::partition, device half, over RCCL (mesh_manager.inl:626-723,
 * subgrid_mesh_manager.inl:1217-1369): the reference's new owner PULLS an element's variables through CUDA-IPC pointers
 * (partition_data<<<>>>), here the old owner SENDS. Elements move in contiguous runs of the space-filling curve; run j of the
 * send list = elements [send_first[j], + send_count[j]) of the five `src` planes and of `src_volume` (cells_per_element
 * values per element and variable, one volume per element) to rank send_peer[j]; the receive list likewise into `dst` /
 * `dst_volume`. Six messages per run straight between the planes, one RCCL group per call; runs whose peer is `my_rank` are
 * device copies (the i-th such send pairs with the i-th such receive) and `comm` may be NULL if there are no others.
 * All index arrays on the HOST. Collective over the ranks that exchange runs. */
int t8gpu_hip_repartition_f32(void* comm, int my_rank, int n_send, const int32_t* send_peer, const int32_t* send_first,
                              const int32_t* send_count, int n_recv, const int32_t* recv_peer, const int32_t* recv_first,
                              const int32_t* recv_count, T8gpuVars_f32 src, const float* src_volume, T8gpuVars_f32 dst,
                              float* dst_volume, int cells_per_element, void* stream);
int t8gpu_hip_repartition_f64(void* comm, int my_rank, int n_send, const int32_t* send_peer, const int32_t* send_first,
                              const int32_t* send_count, int n_recv, const int32_t* recv_peer, const int32_t* recv_first,
                              const int32_t* recv_count, T8gpuVars_f64 src, const double* src_volume, T8gpuVars_f64 dst,
                              double* dst_volume, int cells_per_element, void* stream);
/* every rank's chunk mine[offsets[rank + 1] - offsets[rank]] of a distributed array of doubles to every rank's
 * all[offsets[nranks]] (device pointers, offsets on the host): how the refinement criteria of a partitioned adapt travel
 * (the reference's adapt callback reads per-element criteria of its own rank; with a replicated forest description every
 * rank evaluates it on the whole array). Collective. */
int t8gpu_hip_comm_allgatherv_f64(void* comm, int rank, int nranks, const double* mine, double* all, const int64_t* offsets,
                                  void* stream);

/* ---- AMR indicator and data transfer for Subgrid<4,4> / Subgrid<4,4,4> blocks -------------
 * compute_refinement_criteria<Subgrid><<<>>>: examples/subgrid/kernels.inl:1110-1168 (discrete H1 seminorm of the
 * density inside a block / block volume). */
int t8gpu_hip_subgrid_refinement_criteria_f32(int rank, int num_elements, const float* rho, const float* volumes,
                                              float* criteria, void* stream);
int t8gpu_hip_subgrid_refinement_criteria_f64(int rank, int num_elements, const double* rho, const double* volumes,
                                              double* criteria, void* stream);
/* adapt_variables<Subgrid><<<>>> + adapt_volume<Subgrid><<<>>>: t8gpu/mesh/subgrid_mesh_manager.inl:246-425
 * (refined block: injection from the parent's octant; coarsened block: mean of the 2^rank fine cells). */
int t8gpu_hip_subgrid_adapt_variables_and_volume_f32(int rank, int num_new_elements, const int32_t* adapt_data,
                                                     T8gpuVars_f32 old_variables, T8gpuVars_f32 new_variables,
                                                     const float* volume_old, float* volume_new, void* stream);
int t8gpu_hip_subgrid_adapt_variables_and_volume_f64(int rank, int num_new_elements, const int32_t* adapt_data,
                                                     T8gpuVars_f64 old_variables, T8gpuVars_f64 new_variables,
                                                     const double* volume_old, double* volume_new, void* stream);

/* ---- read-back for the VTK writers ---------------------------------------------------------
 * column_major_to_z_order<Subgrid><<<>>>: subgrid_mesh_manager.inl:1008-1049 -- cell (i,j,k) of every block to
 * the z-order position it has once the block is refined uniformly twice (how the reference hands Subgrid
 * data to t8_forest_write_vtk_ext). from != to; both [num_elements * 4^rank]. */
int t8gpu_hip_column_major_to_z_order_f32(int rank, int num_elements, const float* from, float* to, void* stream);
int t8gpu_hip_column_major_to_z_order_f64(int rank, int num_elements, const double* from, double* to, void* stream);
/* get_host_scalar_variable / get_host_vector_variable (mesh_manager.inl:516-586, subgrid_mesh_manager.inl:
 * 1140-1183): the device-side half -- cast to double, vectors interleaved xyz -- so that the host needs one
 * copy of `out` (device memory, n resp. 3 n doubles). */
int t8gpu_hip_host_scalar_variable_f32(size_t n, const float* variable, double* out, void* stream);
int t8gpu_hip_host_scalar_variable_f64(size_t n, const double* variable, double* out, void* stream);
int t8gpu_hip_host_vector_variable_f32(size_t n, const float* v0, const float* v1, const float* v2, double* out, void* stream);
int t8gpu_hip_host_vector_variable_f64(size_t n, const double* v0, const double* v1, const double* v2, double* out, void* stream);

/* ---- diagnostics -----------------------------------------------------------------------------------------
 * Element-wise evaluation of the fast-tier scalar helpers (reciprocal / division / sqrt / log without the
 * IEEE special-case handling, and the logarithmic mean built on them: kernels.cu:24-36) so that their
 * accuracy can be pinned against a host libm. out[i] = op(a[i], b[i]); b may be null for unary ops. */
enum {
  T8GPU_PROBE_RCP = 0,
  T8GPU_PROBE_DIV = 1,
  T8GPU_PROBE_SQRT = 2,
  T8GPU_PROBE_LOG = 3,
  T8GPU_PROBE_LN_MEAN = 4,    /* fast tier: from the two values and the difference of their logs */
  T8GPU_PROBE_LN_MEAN_REF = 5, /* compat tier: the reference formula, IEEE division and library log */
  T8GPU_PROBE_LOG_TAB = 6,     /* the table-driven fp64 log of the plain tile kernels (fp32: same as LOG) */
  T8GPU_PROBE_SQRT_RATIO = 7,  /* sqrt(a / b) through one reciprocal square root (the sound speed of the KEPES flux) */
  T8GPU_PROBE_DIV_SHARED = 8   /* a / b from a reciprocal that several quotients share */
};
int t8gpu_hip_math_probe_f32(int op, int n, const float* a, const float* b, float* out, void* stream);
int t8gpu_hip_math_probe_f64(int op, int n, const double* a, const double* b, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T8GPU_HIP_H */
