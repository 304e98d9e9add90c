// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" entry points over oracle.hpp for ctypes (tests/, smoke(),
// bench.py cpu_baseline). Built by oracle/Makefile into oracle/liboracle.so
// (canonical, -ffp-contract=off, serial) and oracle/liboracle_omp.so (OpenMP,
// timing only). See oracle.hpp for the reference citations and pinning status.
#include "oracle.hpp"

#if defined(_OPENMP)
#include <omp.h>
#endif

#include <cstddef>
using namespace oracle;

extern "C" void oracle_set_num_threads(int n) {
#if defined(_OPENMP)
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// Zero-fills `bytes` bytes from the OpenMP team with the static schedule of the oracle's own loops, so that every page of a
// state array is FIRST TOUCHED by (a thread on the socket of) the thread that will work on it -- numpy's allocator would leave
// all pages on the allocating thread's NUMA node (bench.py: cpu_baseline).
extern "C" void oracle_first_touch(void* p, size_t bytes) {
  char* c = static_cast<char*>(p);
  const long n = static_cast<long>(bytes / 4096) + 1;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
  for (long i = 0; i < n; i++) {
    const size_t a = static_cast<size_t>(i) * 4096, b = a + 4096 < bytes ? a + 4096 : bytes;
    for (size_t j = a; j < b; j++) c[j] = 0;
  }
}

extern "C" int oracle_num_threads() {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define ORACLE_DEFINE(SUF, T)                                                                                         \
  extern "C" void oracle_ln_mean_##SUF(int n, const T* aL, const T* aR, T* out) {                                     \
    for (int i = 0; i < n; i++) out[i] = ln_mean<T>(aL[i], aR[i]);                                                    \
  }                                                                                                                   \
  /* batched face-frame flux: uL,uR,flux are [n][5] row-major; speed [n] or null */                                   \
  extern "C" void oracle_face_frame_flux_##SUF(int kind, int n, const T* uL, const T* uR, T* flux, T* speed) {        \
    for (int i = 0; i < n; i++) {                                                                                     \
      T s = T(0);                                                                                                     \
      face_frame_flux<T>(kind, uL + 5 * (size_t)i, uR + 5 * (size_t)i, flux + 5 * (size_t)i, &s);                     \
      if (speed) speed[i] = s;                                                                                        \
    }                                                                                                                 \
  }                                                                                                                   \
  /* batched full face flux in xyz frame: rotate, flux, rotate back (no area) */                                      \
  extern "C" void oracle_xyz_face_flux_##SUF(int kind, int n, const T* normals3, const T* sL, const T* sR, T* flux,   \
                                             int mirror) {                                                            \
    for (int i = 0; i < n; i++) {                                                                                     \
      const T* nn = normals3 + 3 * (size_t)i;                                                                         \
      T        t1[3], t2[3], a[5], b[5], Ff[5];                                                                       \
      face_basis<T>(nn, t1, t2);                                                                                      \
      to_face_frame<T>(nn, t1, t2, sL + 5 * (size_t)i, a, false);                                                     \
      to_face_frame<T>(nn, t1, t2, (mirror ? sL : sR) + 5 * (size_t)i, b, mirror != 0);                               \
      face_frame_flux<T>(kind, a, b, Ff, nullptr);                                                                    \
      T* o = flux + 5 * (size_t)i;                                                                                    \
      o[0] = Ff[0];                                                                                                   \
      o[1] = Ff[1] * nn[0] + Ff[2] * t1[0] + Ff[3] * t2[0];                                                           \
      o[2] = Ff[1] * nn[1] + Ff[2] * t1[1] + Ff[3] * t2[1];                                                           \
      o[3] = Ff[1] * nn[2] + Ff[2] * t1[2] + Ff[3] * t2[2];                                                           \
      o[4] = Ff[4];                                                                                                   \
    }                                                                                                                 \
  }                                                                                                                   \
  extern "C" void oracle_plain_interior_faces_##SUF(int kind, int F, int dim, const int32_t* fn, const int32_t* idx,  \
                                                    const T* normals, const T* areas, const T* planes_state,          \
                                                    T* planes_flux, size_t stride, T* speed) {                        \
    const T* st[5];                                                                                                   \
    T*       fl[5];                                                                                                   \
    for (int k = 0; k < 5; k++) {                                                                                     \
      st[k] = planes_state + k * stride;                                                                              \
      fl[k] = planes_flux + k * stride;                                                                               \
    }                                                                                                                 \
    plain_interior_faces<T>(kind, F, dim, fn, idx, normals, areas, st, fl, speed);                                    \
  }                                                                                                                   \
  extern "C" void oracle_plain_boundary_faces_##SUF(int kind, int F, int B, int dim, const int32_t* fn,               \
                                                    const T* normals, const T* areas, const T* planes_state,          \
                                                    T* planes_flux, size_t stride, T* speed) {                        \
    const T* st[5];                                                                                                   \
    T*       fl[5];                                                                                                   \
    for (int k = 0; k < 5; k++) {                                                                                     \
      st[k] = planes_state + k * stride;                                                                              \
      fl[k] = planes_flux + k * stride;                                                                               \
    }                                                                                                                 \
    plain_boundary_faces<T>(kind, F, B, dim, fn, normals, areas, st, fl, speed);                                      \
  }                                                                                                                   \
  extern "C" void oracle_plain_rk_stage_##SUF(int stage, int N, const T* prev, const T* mid, T* out, T* flux,         \
                                              size_t stride, const T* volume, T dt) {                                 \
    const T *pv[5], *md[5];                                                                                           \
    T *      ot[5], *fl[5];                                                                                           \
    for (int k = 0; k < 5; k++) {                                                                                     \
      pv[k] = prev + k * stride;                                                                                      \
      md[k] = mid ? mid + k * stride : nullptr;                                                                       \
      ot[k] = out + k * stride;                                                                                       \
      fl[k] = flux + k * stride;                                                                                      \
    }                                                                                                                 \
    plain_rk_stage<T>(stage, N, pv, md, ot, fl, volume, dt);                                                          \
  }                                                                                                                   \
  /* planes: 26 planes of `stride` values: (step*5+var), volume = plane 25 */                                         \
  extern "C" void oracle_plain_iterate_##SUF(int kind, int N, int F, int B, int dim, const int32_t* fn,               \
                                             const int32_t* idx, const T* normals, const T* areas, T* planes,         \
                                             size_t stride, int prev, int next, T dt, T* speed) {                     \
    plain_iterate<T>(kind, N, F, B, dim, fn, idx, normals, areas, Planes<T>{planes, stride}, prev, next, dt, speed);  \
  }                                                                                                                   \
  extern "C" void oracle_subgrid_inner_##SUF(int kind, int rank, int N, const T* planes_state, T* planes_flux,        \
                                             size_t stride, const T* volumes) {                                       \
    const T* st[5];                                                                                                   \
    T*       fl[5];                                                                                                   \
    for (int k = 0; k < 5; k++) {                                                                                     \
      st[k] = planes_state + k * stride;                                                                              \
      fl[k] = planes_flux + k * stride;                                                                               \
    }                                                                                                                 \
    subgrid_inner<T>(kind, rank, N, st, fl, volumes);                                                                 \
  }                                                                                                                   \
  extern "C" void oracle_subgrid_outer_##SUF(int kind, int rank, int F, const int32_t* fn, const int32_t* idx,        \
                                             const int32_t* level_diff, const int32_t* nb_offset, const T* normals,   \
                                             const T* areas, const T* planes_state, T* planes_flux, size_t stride) {  \
    const T* st[5];                                                                                                   \
    T*       fl[5];                                                                                                   \
    for (int k = 0; k < 5; k++) {                                                                                     \
      st[k] = planes_state + k * stride;                                                                              \
      fl[k] = planes_flux + k * stride;                                                                               \
    }                                                                                                                 \
    subgrid_outer<T>(kind, rank, F, fn, idx, level_diff, nb_offset, normals, areas, st, fl);                          \
  }                                                                                                                   \
  extern "C" void oracle_subgrid_boundary_##SUF(int kind, int rank, int F, int B, const int32_t* fn,                  \
                                                const T* normals, const T* areas, const T* planes_state,              \
                                                T* planes_flux, size_t stride) {                                      \
    const T* st[5];                                                                                                   \
    T*       fl[5];                                                                                                   \
    for (int k = 0; k < 5; k++) {                                                                                     \
      st[k] = planes_state + k * stride;                                                                              \
      fl[k] = planes_flux + k * stride;                                                                               \
    }                                                                                                                 \
    subgrid_boundary<T>(kind, rank, F, B, fn, normals, areas, st, fl);                                                \
  }                                                                                                                   \
  extern "C" void oracle_subgrid_rk_stage_##SUF(int stage, int rank, int N, const T* prev, const T* mid, T* out,      \
                                                T* flux, size_t stride, const T* volumes, T dt) {                     \
    const T *pv[5], *md[5];                                                                                           \
    T *      ot[5], *fl[5];                                                                                           \
    for (int k = 0; k < 5; k++) {                                                                                     \
      pv[k] = prev + k * stride;                                                                                      \
      md[k] = mid ? mid + k * stride : nullptr;                                                                       \
      ot[k] = out + k * stride;                                                                                       \
      fl[k] = flux + k * stride;                                                                                      \
    }                                                                                                                 \
    subgrid_rk_stage<T>(stage, rank, N, pv, md, ot, fl, volumes, dt);                                                 \
  }                                                                                                                   \
  /* planes: 25 planes of `stride` SUBCELLS (step*5+var); volumes per block */                                        \
  extern "C" void oracle_subgrid_iterate_##SUF(int kind, int rank, int N, int F, int B, const int32_t* fn,            \
                                               const int32_t* idx, const int32_t* level_diff,                         \
                                               const int32_t* nb_offset, const T* normals, const T* areas, T* planes, \
                                               size_t stride, const T* volumes, int prev, int next, T dt) {           \
    subgrid_iterate<T>(kind, rank, N, F, B, fn, idx, level_diff, nb_offset, normals, areas,                           \
                       Planes<T>{planes, stride}, volumes, prev, next, dt);                                           \
  }

#define ORACLE_DEFINE_AMR(SUF, T)                                                                                  \
  extern "C" void oracle_estimate_gradient_##SUF(int F, const int32_t* fn, const int32_t* idx, const T* rho,        \
                                                 T* gradient) {                                                     \
    estimate_gradient<T>(F, fn, idx, rho, gradient);                                                                \
  }                                                                                                                 \
  extern "C" void oracle_refinement_criteria_##SUF(int N, const T* gradient, const T* volume, T* criteria) {        \
    refinement_criteria<T>(N, gradient, volume, criteria);                                                          \
  }                                                                                                                 \
  extern "C" void oracle_adapt_variables_and_volume_##SUF(int n_new, int dim, const int32_t* adapt_data,            \
                                                          const T* old_planes, size_t old_stride, T* new_planes,    \
                                                          size_t new_stride, const T* vol_old, T* vol_new) {        \
    const T* ov[5];                                                                                                 \
    T*       nv[5];                                                                                                 \
    for (int k = 0; k < 5; k++) {                                                                                   \
      ov[k] = old_planes + k * old_stride;                                                                          \
      nv[k] = new_planes + k * new_stride;                                                                          \
    }                                                                                                               \
    adapt_variables_and_volume<T>(n_new, dim, adapt_data, ov, nv, vol_old, vol_new);                                \
  }
#define ORACLE_DEFINE_AMR_SUBGRID(SUF, T)                                                                               \
  extern "C" void oracle_subgrid_refinement_criteria_##SUF(int rank, int N, const T* rho, const T* volumes, T* criteria) { \
    subgrid_refinement_criteria<T>(rank, N, rho, volumes, criteria);                                                     \
  }                                                                                                                      \
  extern "C" void oracle_subgrid_adapt_variables_and_volume_##SUF(int rank, int n_new, const int32_t* adapt_data,        \
                                                                  const T* old_planes, size_t old_stride, T* new_planes, \
                                                                  size_t new_stride, const T* vol_old, T* vol_new) {     \
    const T* ov[5];                                                                                                      \
    T*       nv[5];                                                                                                      \
    for (int k = 0; k < 5; k++) {                                                                                        \
      ov[k] = old_planes + k * old_stride;                                                                               \
      nv[k] = new_planes + k * new_stride;                                                                               \
    }                                                                                                                    \
    subgrid_adapt_variables_and_volume<T>(rank, n_new, adapt_data, ov, nv, vol_old, vol_new);                            \
  }
ORACLE_DEFINE_AMR_SUBGRID(f32, float)
ORACLE_DEFINE_AMR_SUBGRID(f64, double)
ORACLE_DEFINE_AMR(f32, float)
ORACLE_DEFINE_AMR(f64, double)

ORACLE_DEFINE(f32, float)
ORACLE_DEFINE(f64, double)
