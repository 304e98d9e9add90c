// kernels_compat.hip -- reference-dataflow kernels (face -> atomic scatter, separate RK pass).
//
// Same data flow as the reference's CUDA kernels (one lane per face / sub-face, float atomics into
// the flux planes, a streaming RK pass that also zeroes the flux planes) written directly for gfx950:
// 64-lane waves, flat local slots instead of the [rank][index] pointer-table double indirection,
// several small faces/blocks packed per wavefront. These are the kernels the accessor-level C++ API
// (include/t8gpu/) launches, and the baseline the fused tile kernels are measured against.
#include <hip/hip_runtime.h>

// Keep the reference's rounding sequence in this tier: no mul+add contraction.
#pragma clang fp contract(off)

#include "flux_math.hpp"
#include "stage_kernel_note.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
struct Vars {
  T* p[5];
};

template <class T>
T8_DEV void load5(const Vars<T>& v, size_t i, T s[5]) {
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = v.p[k][i];
}

// float/double atomic add to global memory; hipcc lowers this to global_atomic_add_f32 /
// global_atomic_add_f64 on gfx950 (no CAS loop; checked in the .s).
template <class T>
T8_DEV void gadd(T* addr, T v) {
  unsafeAtomicAdd(addr, v);
}

// ---- a4: interior faces, kernels.cu:135-309 ----------------------------------------------------
template <class T, int KIND>
__global__ __launch_bounds__(256) void k_flux_faces(int F, int ndim, const int32_t* __restrict__ fn,
                                                    const int32_t* __restrict__ idx, const T* __restrict__ normals,
                                                    const T* __restrict__ areas, Vars<T> st, Vars<T> fl,
                                                    T* __restrict__ speed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F) return;
  const T area = areas[i];
  int     l = fn[2 * (size_t)i], r = fn[2 * (size_t)i + 1];
  if (idx) {
    l = idx[l];
    r = idx[r];
  }
  T n[3] = {T(0), T(0), T(0)};
  for (int k = 0; k < ndim; k++) n[k] = normals[(size_t)ndim * i + k];
  T sl[5], sr[5];
  load5(st, l, sl);
  load5(st, r, sr);
  T t1[3], t2[3], Ff[5], g[5], spd;
  face_basis<T>(n, t1, t2);
  face_frame_flux_ref<T, KIND>(n, t1, t2, sl, sr, false, Ff, spd);
  if (speed) speed[i] = spd;
#pragma unroll
  for (int k = 0; k < 5; k++) Ff[k] = area * Ff[k];
  from_face_frame<T>(n, t1, t2, Ff, g);
#pragma unroll
  for (int k = 0; k < 5; k++) {
    gadd(&fl.p[k][l], -g[k]);
    gadd(&fl.p[k][r], g[k]);
  }
}

// ---- a5: reflective wall faces, kernels.cu:311-469 ----------------------------------------------
template <class T, int KIND>
__global__ __launch_bounds__(256) void k_flux_boundary(int F, int B, int ndim, const int32_t* __restrict__ fn,
                                                       const T* __restrict__ normals, const T* __restrict__ areas,
                                                       Vars<T> st, Vars<T> fl, T* __restrict__ speed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const T   area = areas[F + i];
  const int e    = fn[2 * (size_t)F + i];
  T         n[3] = {T(0), T(0), T(0)};
  for (int k = 0; k < ndim; k++) n[k] = normals[(size_t)ndim * (F + i) + k];
  T s[5];
  load5(st, e, s);
  T t1[3], t2[3], Ff[5], g[5], spd;
  face_basis<T>(n, t1, t2);
  face_frame_flux_ref<T, KIND>(n, t1, t2, s, s, true, Ff, spd);
  if (speed) speed[F + i] = spd;
#pragma unroll
  for (int k = 0; k < 5; k++) Ff[k] = area * Ff[k];
  from_face_frame<T>(n, t1, t2, Ff, g);
#pragma unroll
  for (int k = 0; k < 5; k++) gadd(&fl.p[k][e], -g[k]);
}

// ---- a6: SSP-RK3 stages, ssp_runge_kutta.inl:30-99 (S = 1) and :101-221 (S = Subgrid::size) -------
// One lane per cell; `volume[i / S] / S` for subgrids. Grid-stride, coalesced, fluxes zeroed.
template <class T, int STAGE, int S>
__global__ __launch_bounds__(256) void k_rk_stage(size_t ncells, Vars<T> prev, Vars<T> mid, Vars<T> out, Vars<T> fl,
                                                  const T* __restrict__ volume, T dt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncells; i += (size_t)gridDim.x * blockDim.x) {
    const T vol = (S == 1) ? volume[i] : volume[i / S] / static_cast<T>(S);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const T f = fl.p[k][i];
      T       o;
      if (STAGE == 1) {
        o = prev.p[k][i] + dt / vol * f;
      } else if (STAGE == 2) {
        o = rk3c<T>::c21 * prev.p[k][i] + rk3c<T>::c22 * mid.p[k][i] + rk3c<T>::c23 * dt / vol * f;
      } else {
        o = rk3c<T>::c31 * prev.p[k][i] + rk3c<T>::c32 * mid.p[k][i] + rk3c<T>::c33 * dt / vol * f;
      }
      out.p[k][i] = o;
      fl.p[k][i]  = T(0.0);
    }
  }
}

// ---- a13: inner fluxes of a block, kernels.inl:335-662 -------------------------------------------
// One 64-lane wavefront per workgroup: a whole Subgrid<4,4,4> block, or four Subgrid<4,4> blocks.
// The wave runs in lock-step, so the reference's latent LDS race cannot occur.
template <class T, int KIND, int RANK>
__global__ __launch_bounds__(64) void k_subgrid_inner(int N, Vars<T> st, Vars<T> fl, const T* __restrict__ volumes) {
  constexpr int S   = RANK == 3 ? 64 : 16;
  constexpr int EPB = 64 / S;  // elements per workgroup
  const int     c   = threadIdx.x % S;
  const int     sub = threadIdx.x / S;
  const int     e   = blockIdx.x * EPB + sub;
  __shared__ T  sh[5][64];
  const bool    live = e < N;
  const size_t  o    = (size_t)(live ? e : 0) * S;
  T             own[5];
  load5(st, o + c, own);
  const T vol     = volumes[live ? e : 0];
  const T edge    = (RANK == 3 ? t8_cbrt(vol) : t8_sqrt(vol)) / static_cast<T>(4);
  const T surface = RANK == 3 ? edge * edge : edge;
  T       acc[5];
#pragma unroll
  for (int k = 0; k < 5; k++) acc[k] = fl.p[k][o + c];
#pragma unroll
  for (int d = 0; d < RANK; d++) {
    const int str = d == 0 ? 1 : (d == 1 ? 4 : 16);
    const int cd  = (c / str) % 4;
    T         n[3] = {T(0.0), T(0.0), T(0.0)};
    n[d]           = T(1.0);
    T t1[3], t2[3];
    face_basis<T>(n, t1, t2);
    T g[5] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)};
    if (cd < 3) {
      T nb[5], Ff[5], spd;
      load5(st, o + c + str, nb);
      face_frame_flux_ref<T, KIND>(n, t1, t2, own, nb, false, Ff, spd);
      from_face_frame<T>(n, t1, t2, Ff, g);
#pragma unroll
      for (int k = 0; k < 5; k++) g[k] = g[k] * surface;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) sh[k][threadIdx.x] = g[k];
    __syncthreads();
    if (cd < 3) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] -= g[k];
    }
    if (cd > 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] += sh[k][threadIdx.x - str];
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 5; k++) fl.p[k][o + c] = acc[k];
  }
}

// sub-face -> cell map of compute_outer_fluxes, kernels.inl:710-758 / 837-866 (exact +-1.0 compares)
template <class T, int RANK>
T8_DEV void sg_face_cells(const T n[3], const int off[3], int ds, int i, int j, int& lflat, int& rflat) {
  int al[3] = {0, 0, 0}, si[3] = {0, 0, 0}, sj[3] = {0, 0, 0};
  if (RANK == 3) {
    if (n[0] == T(1.0)) { al[0] = 3; si[1] = 1; sj[2] = 1; }
    if (n[0] == T(-1.0)) { si[1] = 1; sj[2] = 1; }
    if (n[1] == T(1.0)) { al[1] = 3; si[0] = 1; sj[2] = 1; }
    if (n[1] == T(-1.0)) { si[0] = 1; sj[2] = 1; }
    if (n[2] == T(1.0)) { al[2] = 3; si[0] = 1; sj[1] = 1; }
    if (n[2] == T(-1.0)) { si[0] = 1; sj[1] = 1; }
  } else {
    if (n[0] == T(1.0)) { al[0] = 3; si[1] = 1; }
    if (n[0] == T(-1.0)) { si[1] = 1; }
    if (n[1] == T(1.0)) { al[1] = 3; si[0] = 1; }
    if (n[1] == T(-1.0)) { si[0] = 1; }
  }
  int lc[3], rc[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    lc[d] = al[d] + i * si[d] + j * sj[d];
    rc[d] = off[d] + ds * (i * si[d] + j * sj[d]) / 2;
  }
  lflat = lc[0] + 4 * lc[1] + 16 * lc[2];
  rflat = rc[0] + 4 * rc[1] + 16 * rc[2];
}

// ---- a14 / a15: outer and wall faces of blocks, kernels.inl:664-911 / 913-1107 --------------------
// SF = sub-faces per coarse face (16 or 4); 64/SF coarse faces share one wavefront. WALL selects
// the reflective variant (boundary slices start after the F interior entries).
template <class T, int KIND, int RANK, bool WALL>
__global__ __launch_bounds__(64) void k_subgrid_faces(int F, int count, const int32_t* __restrict__ fn,
                                                      const int32_t* __restrict__ idx,
                                                      const int32_t* __restrict__ level_diff,
                                                      const int32_t* __restrict__ nb_off,
                                                      const T* __restrict__ normals, const T* __restrict__ areas,
                                                      Vars<T> st, Vars<T> fl) {
  constexpr int S  = RANK == 3 ? 64 : 16;
  constexpr int SF = RANK == 3 ? 16 : 4;
  const int     f  = blockIdx.x * (64 / SF) + threadIdx.x / SF;
  if (f >= count) return;
  const int    t    = threadIdx.x % SF;
  const int    i    = t % 4, j = t / 4;
  const size_t slot = WALL ? (size_t)F + f : (size_t)f;
  int          l, r = 0, ds = 2, off[3] = {0, 0, 0};
  if (WALL) {
    l = fn[2 * (size_t)F + f];
  } else {
    l = fn[2 * (size_t)f];
    r = fn[2 * (size_t)f + 1];
    if (idx) {
      l = idx[l];
      r = idx[r];
    }
    ds = (level_diff[f] == 0) ? 2 : 1;
    for (int d = 0; d < RANK; d++) off[d] = nb_off[(size_t)RANK * f + d];
  }
  T n[3] = {T(0), T(0), T(0.0)};
  for (int d = 0; d < RANK; d++) n[d] = normals[(size_t)RANK * slot + d];
  T t1[3], t2[3];
  face_basis<T>(n, t1, t2);
  int lflat, rflat;
  sg_face_cells<T, RANK>(n, off, ds, i, j, lflat, rflat);
  const size_t li = (size_t)l * S + lflat;
  const size_t ri = (size_t)r * S + rflat;
  T            sl[5], sr[5], Ff[5], g[5], spd;
  load5(st, li, sl);
  if (!WALL) load5(st, ri, sr);
  face_frame_flux_ref<T, KIND>(n, t1, t2, sl, WALL ? sl : sr, WALL, Ff, spd);
  from_face_frame<T>(n, t1, t2, Ff, g);
  const T surface = areas[slot] / static_cast<T>(SF);
#pragma unroll
  for (int k = 0; k < 5; k++) {
    gadd(&fl.p[k][li], -g[k] * surface);
    if (!WALL) gadd(&fl.p[k][ri], g[k] * surface);
  }
}

// ---- launchers ------------------------------------------------------------------------------------
template <class T, class V>
Vars<T> mk(const V& v) {
  Vars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

template <class T, class V>
int flux_faces(int kind, int F, int ndim, const int32_t* fn, const int32_t* idx, const T* normals, const T* areas,
               V st, V fl, T* speed, void* stream) {
  if (F <= 0) return 0;
  if (ndim < 2 || ndim > 3 || (kind < 0 || kind > 2)) return static_cast<int>(hipErrorInvalidValue);
  const dim3  grid((F + 255) / 256), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kind == 0)
    hipLaunchKernelGGL((k_flux_faces<T, 0>), grid, block, 0, s, F, ndim, fn, idx, normals, areas, mk<T>(st), mk<T>(fl), speed);
  else if (kind == 1)
    hipLaunchKernelGGL((k_flux_faces<T, 1>), grid, block, 0, s, F, ndim, fn, idx, normals, areas, mk<T>(st), mk<T>(fl), speed);
  else
    hipLaunchKernelGGL((k_flux_faces<T, 2>), grid, block, 0, s, F, ndim, fn, idx, normals, areas, mk<T>(st), mk<T>(fl), speed);
  return launch_status();
}

template <class T, class V>
int flux_boundary(int kind, int F, int B, int ndim, const int32_t* fn, const T* normals, const T* areas, V st, V fl,
                  T* speed, void* stream) {
  if (B <= 0) return 0;
  if (ndim < 2 || ndim > 3 || (kind < 0 || kind > 2)) return static_cast<int>(hipErrorInvalidValue);
  const dim3  grid((B + 255) / 256), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kind == 0)
    hipLaunchKernelGGL((k_flux_boundary<T, 0>), grid, block, 0, s, F, B, ndim, fn, normals, areas, mk<T>(st), mk<T>(fl), speed);
  else if (kind == 1)
    hipLaunchKernelGGL((k_flux_boundary<T, 1>), grid, block, 0, s, F, B, ndim, fn, normals, areas, mk<T>(st), mk<T>(fl), speed);
  else
    hipLaunchKernelGGL((k_flux_boundary<T, 2>), grid, block, 0, s, F, B, ndim, fn, normals, areas, mk<T>(st), mk<T>(fl), speed);
  return launch_status();
}

template <class T, int S, class V>
int rk_stage(int stage, size_t ncells, V prev, V mid, V out, V fl, const T* volume, T dt, void* stream) {
  if (ncells == 0) return 0;
  if (stage < 1 || stage > 3) return static_cast<int>(hipErrorInvalidValue);
  size_t blocks = (ncells + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3  grid(static_cast<unsigned>(blocks)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Vars<T>     m = stage == 1 ? mk<T>(prev) : mk<T>(mid);
  if (stage == 1)
    hipLaunchKernelGGL((k_rk_stage<T, 1, S>), grid, block, 0, s, ncells, mk<T>(prev), m, mk<T>(out), mk<T>(fl), volume, dt);
  else if (stage == 2)
    hipLaunchKernelGGL((k_rk_stage<T, 2, S>), grid, block, 0, s, ncells, mk<T>(prev), m, mk<T>(out), mk<T>(fl), volume, dt);
  else
    hipLaunchKernelGGL((k_rk_stage<T, 3, S>), grid, block, 0, s, ncells, mk<T>(prev), m, mk<T>(out), mk<T>(fl), volume, dt);
  return launch_status();
}

template <class T, class V>
int subgrid_inner(int kind, int rank, int N, V st, V fl, const T* volumes, void* stream) {
  if (N <= 0) return 0;
  if ((rank != 2 && rank != 3) || (kind < 0 || kind > 2)) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3  block(64);
  const dim3  grid(rank == 3 ? N : (N + 3) / 4);
#define T8_INNER(K, R) hipLaunchKernelGGL((k_subgrid_inner<T, K, R>), grid, block, 0, s, N, mk<T>(st), mk<T>(fl), volumes)
  if (rank == 3) {
    if (kind == 0) T8_INNER(0, 3); else if (kind == 1) T8_INNER(1, 3); else T8_INNER(2, 3);
  } else {
    if (kind == 0) T8_INNER(0, 2); else if (kind == 1) T8_INNER(1, 2); else T8_INNER(2, 2);
  }
#undef T8_INNER
  return launch_status();
}

template <class T, bool WALL, class V>
int subgrid_faces(int kind, int rank, int F, int count, const int32_t* fn, const int32_t* idx, const int32_t* ld,
                  const int32_t* off, const T* normals, const T* areas, V st, V fl, void* stream) {
  if (count <= 0) return 0;
  if ((rank != 2 && rank != 3) || (kind < 0 || kind > 2)) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t s   = static_cast<hipStream_t>(stream);
  const int   fpb = rank == 3 ? 4 : 16;
  const dim3  block(64), grid((count + fpb - 1) / fpb);
#define T8_FACES(K, R)                                                                                              \
  hipLaunchKernelGGL((k_subgrid_faces<T, K, R, WALL>), grid, block, 0, s, F, count, fn, idx, ld, off, normals, areas, \
                     mk<T>(st), mk<T>(fl))
  if (rank == 3) {
    if (kind == 0) T8_FACES(0, 3); else if (kind == 1) T8_FACES(1, 3); else T8_FACES(2, 3);
  } else {
    if (kind == 0) T8_FACES(0, 2); else if (kind == 1) T8_FACES(1, 2); else T8_FACES(2, 2);
  }
#undef T8_FACES
  return launch_status();
}

}  // namespace t8gpu_hip

namespace t8gpu_hip {
StageKernelNote& stage_kernel_note() {
  // per host thread: the step driver's two lanes enqueue from two threads (stepper.hip); the caller's thread -- the one that asks
  // t8gpu_hip_last_stage_kernel -- launches the interior tiles, the bulk of a stage
  static thread_local StageKernelNote note = {{0}, -1};
  return note;
}
}  // namespace t8gpu_hip

using namespace t8gpu_hip;

extern "C" {

const char* t8gpu_hip_last_stage_kernel(void) { return stage_kernel_note().name; }
int t8gpu_hip_abi_version(void) { return 8; }   // 2: T8gpuPlainPlan.tile_desc; 3: T8gpuSubgridPlan row format (far-cell recipes), n_blocks_addressed, family records; 4: T8gpuPlainPlan.n_patch_tiles; 5: T8gpuPlainPlan.ell holds rows for generic tiles only (tile_desc word 6), patch_dim; 6: T8gpuPlainPlan.n_irregular_tiles; 7: T8gpuPlainPlan ghost window (ghost_buf, send_map, send_list, send_buf, n_owned); 8: t8gpu_hip_plain_geo_frames_* (plan builders must call it)
int t8gpu_hip_device_count(int* count) { return static_cast<int>(hipGetDeviceCount(count)); }
int t8gpu_hip_set_device(int device) { return static_cast<int>(hipSetDevice(device)); }
const char* t8gpu_hip_error_string(int code) {
  if (code >= 10000) return "RCCL error (code - 10000 is the ncclResult_t)";
  return hipGetErrorString(static_cast<hipError_t>(code));
}

#define T8_DEFINE_COMPAT(SUF, T, V)                                                                                  \
  int t8gpu_hip_flux_faces_##SUF(int kind, int F, int ndim, const int32_t* fn, const int32_t* idx, const T* normals, \
                                 const T* areas, V st, V fl, T* speed, void* stream) {                               \
    return flux_faces<T, V>(kind, F, ndim, fn, idx, normals, areas, st, fl, speed, stream);                          \
  }                                                                                                                  \
  int t8gpu_hip_flux_boundary_##SUF(int kind, int F, int B, int ndim, const int32_t* fn, const T* normals,           \
                                    const T* areas, V st, V fl, T* speed, void* stream) {                            \
    return flux_boundary<T, V>(kind, F, B, ndim, fn, normals, areas, st, fl, speed, stream);                         \
  }                                                                                                                  \
  int t8gpu_hip_rk3_stage_##SUF(int stage, int N, V prev, V mid, V out, V fl, const T* volume, T dt, void* stream) { \
    return rk_stage<T, 1, V>(stage, N < 0 ? 0 : (size_t)N, prev, mid, out, fl, volume, dt, stream);                  \
  }                                                                                                                  \
  int t8gpu_hip_subgrid_inner_##SUF(int kind, int rank, int N, V st, V fl, const T* volumes, void* stream) {         \
    return subgrid_inner<T, V>(kind, rank, N, st, fl, volumes, stream);                                              \
  }                                                                                                                  \
  int t8gpu_hip_subgrid_outer_##SUF(int kind, int rank, int F, const int32_t* fn, const int32_t* idx,                \
                                    const int32_t* ld, const int32_t* off, const T* normals, const T* areas, V st,   \
                                    V fl, void* stream) {                                                            \
    return subgrid_faces<T, false, V>(kind, rank, F, F, fn, idx, ld, off, normals, areas, st, fl, stream);           \
  }                                                                                                                  \
  int t8gpu_hip_subgrid_boundary_##SUF(int kind, int rank, int F, int B, const int32_t* fn, const T* normals,        \
                                       const T* areas, V st, V fl, void* stream) {                                   \
    return subgrid_faces<T, true, V>(kind, rank, F, B, fn, nullptr, nullptr, nullptr, normals, areas, st, fl,        \
                                     stream);                                                                        \
  }                                                                                                                  \
  int t8gpu_hip_subgrid_rk3_stage_##SUF(int stage, int rank, int N, V prev, V mid, V out, V fl, const T* volumes,    \
                                        T dt, void* stream) {                                                        \
    if (rank == 3) return rk_stage<T, 64, V>(stage, (size_t)(N < 0 ? 0 : N) * 64, prev, mid, out, fl, volumes, dt, stream); \
    if (rank == 2) return rk_stage<T, 16, V>(stage, (size_t)(N < 0 ? 0 : N) * 16, prev, mid, out, fl, volumes, dt, stream); \
    return static_cast<int>(hipErrorInvalidValue);                                                                   \
  }

T8_DEFINE_COMPAT(f32, float, T8gpuVars_f32)
T8_DEFINE_COMPAT(f64, double, T8gpuVars_f64)

}  // extern "C"
