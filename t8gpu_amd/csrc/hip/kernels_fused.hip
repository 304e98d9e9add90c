// kernels_fused.hip -- fused tile kernels for plain elements (flux + RK stage in one launch).
//
// Workgroup = 256 lanes = one tile of the plan (tile_plan.cpp). Three phases, two barriers:
//   1. lanes = tile elements + halo: gather the 5 conserved values (own range coalesced, halo through
//      the sorted id list), turn them into per-element primitives, keep them in LDS;
//   2. lanes = tile faces: packed (l, r), {n, area} streamed from the tile-ordered arrays (coalesced),
//      primitives of both sides from LDS, one KEPES/HLL evaluation, area-scaled xyz flux to LDS;
//   3. lanes = owned elements: sum the element's faces from LDS in CSR order (deterministic, no
//      atomics), apply the SSP-RK3 stage, store the new state coalesced.
// HBM sees: state in, state out, previous-step state, volume and the plan arrays -- the flux planes
// never leave the chip. blockIdx -> tile is XCD-aware: blocks b, b+8, ... share an XCD (and its L2),
// so each XCD gets one contiguous run of tiles and neighbouring tiles' halos hit the same L2.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fused_common.hpp"
#include "stage_kernel_note.hpp"

namespace t8gpu_hip {

template <class T, int KIND, int STAGE>
__global__ __launch_bounds__(256) void k_plain_fused(T8gpuPlainPlan P, int tile_begin, FVars<T> prev, FVars<T> src,
                                                     FVars<T> out, const T* __restrict__ vol, T dt,
                                                     T* __restrict__ speed) {
  extern __shared__ double lds_raw[];
  T* const      lds = reinterpret_cast<T*>(lds_raw);
  constexpr int NW  = KIND == 0 ? kPrimWords : 5;  // words per element kept in LDS
  const int     LE  = P.max_slots > 0 ? P.max_slots : P.max_elems + P.max_halo;    // element slots per LDS plane
  const int     LF  = P.max_faces;
  T* const      pe  = lds;                         // [NW][LE]
  T* const      ff  = lds + (size_t)NW * LE;       // [5][LF]
  constexpr bool kTab = sizeof(T) == 8 && KIND == 0;   // fp64 KEPES: table-driven logarithms (flux_math.hpp: t8_log_tab)
  double* const lt  = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ff + (size_t)5 * LF) + 15) & ~uintptr_t(15));   // 16-byte rows
  if (kTab) {
    lt[threadIdx.x] = kLogTab[threadIdx.x];
    __syncthreads();
  }

  const int tile = P.tile_order[tile_begin + xcd_position(blockIdx.x, gridDim.x)];
  const int e0 = P.elem_off[tile], ne = P.elem_off[tile + 1] - e0;
  const int h0 = P.halo_off[tile], nh = P.halo_off[tile + 1] - h0;
  const int f0 = P.face_off[tile], nf = P.face_off[tile + 1] - f0;
  const int tid = threadIdx.x;

  // ---- phase 1: elements -> LDS --------------------------------------------------------------
  for (int i = tid; i < ne + nh; i += 256) {
    const int slot = i < ne ? e0 + i : P.halo_ids[h0 + (i - ne)];
    T         s[5];
    if (P.ghost_buf) {
#pragma unroll
      for (int k = 0; k < 5; k++) s[k] = ghost_window_load<T>(P, src, slot, k);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) s[k] = src.p[k][slot];
    }
    if (KIND == 0) {
      const Prim<T> q = prim_from_state<T, kTab>(s, lt);
      pe[0 * LE + i]  = q.rho;
      pe[1 * LE + i]  = q.vx;
      pe[2 * LE + i]  = q.vy;
      pe[3 * LE + i]  = q.vz;
      pe[4 * LE + i]  = q.p;
      pe[5 * LE + i]  = q.beta;
      pe[6 * LE + i]  = q.lrho;
      pe[7 * LE + i]  = q.lbeta;
      pe[8 * LE + i]  = q.v0;
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) pe[k * LE + i] = s[k];
    }
  }
  __syncthreads();

  // ---- phase 2: faces -----------------------------------------------------------------------
  using V4 = typename vec4<T>::type;
  const V4* __restrict__ geo = reinterpret_cast<const V4*>(P.face_geo) + f0;
  for (int f = tid; f < nf; f += 256) {
    const uint32_t lr = P.face_lr[f0 + f];
    const V4       gm = geo[f];
    const int      l = lr & 0xFFFFu, r16 = lr >> 16;
    const bool     wall = r16 == 0xFFFFu;
    const int      r = wall ? l : r16;
    const T        n[3] = {gm.x, gm.y, gm.z};
    T              t1[3], t2[3], g[5], spd = T(0);
    face_basis_fast<T>(n, t1, t2);
    if (KIND == 0) {
      Prim<T> L, R;
      L.rho = pe[0 * LE + l]; L.vx = pe[1 * LE + l]; L.vy = pe[2 * LE + l]; L.vz = pe[3 * LE + l]; L.p = pe[4 * LE + l];
      L.beta = pe[5 * LE + l]; L.lrho = pe[6 * LE + l]; L.lbeta = pe[7 * LE + l]; L.v0 = pe[8 * LE + l];
      R.rho = pe[0 * LE + r]; R.vx = pe[1 * LE + r]; R.vy = pe[2 * LE + r]; R.vz = pe[3 * LE + r]; R.p = pe[4 * LE + r];
      R.beta = pe[5 * LE + r]; R.lrho = pe[6 * LE + r]; R.lbeta = pe[7 * LE + r]; R.v0 = pe[8 * LE + r];
      kepes_prim<T>(L, R, wall, n, t1, t2, gm.w, g, spd);
    } else {
      T sl[5], sr[5];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        sl[k] = pe[k * LE + l];
        sr[k] = pe[k * LE + r];
      }
      hll_face<T>(sl, sr, wall, n, t1, t2, gm.w, g, spd, KIND == 2);
    }
    if (speed) {
      const int orig = P.face_orig[f0 + f];
      if (orig >= 0) speed[orig] = spd;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) ff[k * LF + f] = g[k];
  }
  __syncthreads();

  // ---- phase 3: per-element sum + RK stage (ssp_runge_kutta.inl:30-99) ---------------------------
  for (int i = tid; i < ne; i += 256) {
    const int e  = e0 + i;
    const int c0 = P.csr_off[e], c1 = P.csr_off[e + 1];
    T         acc[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int c = c0; c < c1; c++) {
      const unsigned ent = P.csr_ent[c];
      const int      f   = ent & 0x7FFFu;
      if (ent & 0x8000u) {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += ff[k * LF + f];
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] -= ff[k * LF + f];
      }
    }
    const T scale = dt / vol[e];
    T       res[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      // (stage 1: prev is the stage's source state, plain_fused_stage() checks it)
      res[k] = rk_stage_update<T, STAGE>(prev.p[k][e], STAGE == 1 ? prev.p[k][e] : src.p[k][e], scale, acc[k]);
      out.p[k][e] = res[k];
    }
    if (P.send_map) ghost_window_send<T>(P, e, res);
  }
}

}  // namespace t8gpu_hip

#include "fused_tile_body.hpp"  // k_plain_fused_p: the one-tile-per-workgroup kernel, shared with kernels_fused_patch.hip

namespace t8gpu_hip {


// Do the generic tiles of this plan run the pipelined one-tile kernels (ELL rows + tile descriptors: no CSR lists read), or the
// generic kernel, which walks csr_off / csr_ent? One definition for the launcher below and for t8gpu_hip_plain_needs_csr,
// which the host asks before it decides what to upload (t8gpu_amd/fused.py used to repeat a part of this test).
inline bool plain_tiles_pipelined(const T8gpuPlainPlan* plan) {
  const int slots = plan->max_slots > 0 ? plan->max_slots : plan->max_elems + plan->max_halo;
  return plan->ell && plan->tile_desc && plan->ell_width >= 8 && plan->ell_width % 8 == 0 && plan->max_elems <= 256 && slots <= 512 &&
         plan->max_faces <= 1024;
}

// tiles [tile_begin, tile_begin + tile_count) of tile_order, none of them a patch tile. whole_plan: the caller's launch
// covers the whole plan (what the persistent kernel is for).
template <class T, class V>
int plain_generic_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, V prev, V mid,
                        V out, const T* volume, T dt, T* speed, bool whole_plan, void* stream) {
  if (tile_count == 0) return 0;
  const int   nw = kind == 0 ? kPrimWords : 5;
  hipStream_t s  = static_cast<hipStream_t>(stream);
  const dim3  grid(tile_count), block(256);
  const int   slots = plan->max_slots > 0 ? plan->max_slots : plan->max_elems + plan->max_halo;
  const bool  pipelined = plain_tiles_pipelined(plan);
  const bool  four = plan->max_faces > 512;
  // (the generic kernel walks the CSR lists; callers that know their plan stays inside the pipelined kernels' limits need not
  //  upload them -- t8gpu_amd/fused.py does not)
  if (!pipelined && (!plan->csr_off || !plan->csr_ent)) return static_cast<int>(hipErrorInvalidValue);
  static const bool scatter = std::getenv("T8GPU_LDS_SCATTER") && std::getenv("T8GPU_LDS_SCATTER")[0] == '1';   // measured alternative
  // The persistent, software-pipelined kernel (kernels_fused_persistent.hip) for launches that cover the whole plan.
  // A multi-rank stage is split into tile classes on three streams beside the pack / RCCL / unpack kernels
  // (stepper.hip): persistent workgroups would hold every register file and LDS slot of the chip until their class is
  // done and keep those small kernels -- the exchange the split exists to overlap -- from starting, so partial
  // ranges use the one-tile-per-workgroup kernels, whose slots free up continuously. Both give the same bits.
  static const bool persistent_always = std::getenv("T8GPU_PERSISTENT") && std::getenv("T8GPU_PERSISTENT")[0] == '2';
  // (the persistent kernel knows no ghost window -- t8gpu_hip.h: such launches run one tile per workgroup)
  if (!scatter && (persistent_always || whole_plan) && !plan->ghost_buf && !plan->send_map) {
    const int rc = plain_persistent_stage<T>(kind, stage, plan, tile_begin, tile_count, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume,
                                             dt, speed, s);
    if (rc >= 0) return rc;
  }
  const size_t lds_table = (sizeof(T) == 8 && kind == 0) ? 2 * kLogTabEntries * sizeof(double) + 16 : 0;
  size_t       lds       = sizeof(T) * ((size_t)nw * slots + (size_t)5 * (pipelined ? 256 : plan->max_faces)) + lds_table;
#ifdef T8GPU_EXP_LDS_PAD   // experiment builds only: fewer workgroups per CU, to measure how much the kernel leans on occupancy
  if (const char* pad = std::getenv("T8GPU_EXP_LDS_PAD")) lds += static_cast<size_t>(std::atoi(pad));
#endif
  if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
  const bool  dict = pipelined && plan->geo_idx && plan->geo_table && plan->n_geo > 0;
  static const int dense_env = std::getenv("T8GPU_DENSE") ? std::atoi(std::getenv("T8GPU_DENSE")) : -1;   // (measurements)
  const bool  dense = pipelined && !scatter && kind == 0 && sizeof(T) == 8 && plan->geo_idx && plan->geo_table && plan->n_geo > 0 &&
                     plan->max_faces <= 512 && (dense_env >= 0 ? dense_env != 0 : lds - lds_table <= static_cast<size_t>(36) * 1024);
  if (dense) lds -= lds_table;   // (the DENSE kernel reads the logarithm table from global memory)
#define T8_LAUNCH(KERNEL, NAME)                                                                              \
  do {                                                                                                       \
    note_stage_kernel(tile_count, NAME, static_cast<int>(sizeof(T)), kind, stage);                           \
    if (lds > 64 * 1024) {                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
      if (e != hipSuccess) return static_cast<int>(e);                                                       \
    }                                                                                                        \
    hipLaunchKernelGGL(KERNEL, grid, block, lds, s, *plan, tile_begin, fmk<T>(prev), fmk<T>(mid),            \
                       fmk<T>(out), volume, dt, speed);                                                      \
  } while (0)
#define T8_FUSED(K, S)                                              \
  do {                                                              \
    if (scatter && dict && !four)                                   \
      T8_LAUNCH((k_plain_fused_p<T, K, S, true, 2, true>), "k_plain_fused_p<T, K, S, true, 2, true, false>");         \
    else if (scatter && pipelined && !four)                         \
      T8_LAUNCH((k_plain_fused_p<T, K, S, false, 2, true>), "k_plain_fused_p<T, K, S, false, 2, true, false>");        \
    else if (dict && !four && dense)                                \
      T8_LAUNCH((k_plain_fused_p<T, K, S, true, 2, false, true>), "k_plain_fused_p<T, K, S, true, 2, false, true>");  \
    else if (dict && !four)                                         \
      T8_LAUNCH((k_plain_fused_p<T, K, S, true, 2>), "k_plain_fused_p<T, K, S, true, 2, false, false>");               \
    else if (dict)                                                  \
      T8_LAUNCH((k_plain_fused_p<T, K, S, true, 4>), "k_plain_fused_p<T, K, S, true, 4, false, false>");               \
    else if (pipelined && !four)                                    \
      T8_LAUNCH((k_plain_fused_p<T, K, S, false, 2>), "k_plain_fused_p<T, K, S, false, 2, false, false>");              \
    else if (pipelined)                                             \
      T8_LAUNCH((k_plain_fused_p<T, K, S, false, 4>), "k_plain_fused_p<T, K, S, false, 4, false, false>");              \
    else                                                            \
      T8_LAUNCH((k_plain_fused<T, K, S>), "k_plain_fused<T, K, S>");                          \
  } while (0)
  if (kind == 0) {
    if (stage == 1) T8_FUSED(0, 1); else if (stage == 2) T8_FUSED(0, 2); else T8_FUSED(0, 3);
  } else if (kind == 1) {
    if (stage == 1) T8_FUSED(1, 1); else if (stage == 2) T8_FUSED(1, 2); else T8_FUSED(1, 3);
  } else {
    if (stage == 1) T8_FUSED(2, 1); else if (stage == 2) T8_FUSED(2, 2); else T8_FUSED(2, 3);
  }
#undef T8_FUSED
#undef T8_LAUNCH
  return static_cast<int>(hipGetLastError());
}

// The C-ABI entry: splits the range of tile_order into its patch tiles (kernels_fused_patch.hip) and its generic tiles
// (the kernels above / the persistent kernel). Inside every class the patch tiles come first (T8gpuPlainPlan).
template <class T, class V>
int plain_fused_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, V prev, V mid,
                      V out, const T* volume, T dt, T* speed, void* stream) {
  if (!plan || kind < 0 || kind > 2 || stage < 1 || stage > 3) return static_cast<int>(hipErrorInvalidValue);
  if (tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > plan->ntiles) return static_cast<int>(hipErrorInvalidValue);
  if (plan->max_elems > 256 * 4) return static_cast<int>(hipErrorInvalidValue);
  // stage 1 is u1 = u0 + dt/vol f(u0) (ssp_runge_kutta.inl:30-50: `prev` is both the flux source and the summand):
  // the pipelined kernel keeps the source state in registers and never reads `prev` at stage 1, the generic kernel
  // does -- so a caller passing prev != mid at stage 1 would get variant-dependent results. Refused instead.
  if (stage == 1)
    for (int k = 0; k < 5; k++)
      if (prev.p[k] != mid.p[k]) return static_cast<int>(hipErrorInvalidValue);
  if (tile_count == 0) return 0;
  if ((plan->ghost_buf || plan->send_map) && (plan->n_owned <= 0 || (plan->send_map && !plan->send_buf))) return static_cast<int>(hipErrorInvalidValue);
  stage_kernel_note_reset();
  // The interior launch of a multi-rank stage -- exactly [0, n_interior) -- is a persistent grid like a whole-plan launch. In
  // the three-stream pipeline of rounds 1-3 resident workgroups that never leave kept the exchange kernels from starting
  // (profiles/r03_halo_overhead.md); the two-lane driver queues the RCCL kernel and the ghost-reading tiles a stage ahead of
  // their deadline and they slip in at the drain between two interior launches: rank 3 of the 8-way c4 split 0.168 -> 0.153
  // ms/step (profiles/r04_halo_overhead.md). T8GPU_INTERIOR_PERSISTENT=0: one tile per workgroup (measurements).
  static const bool interior_persistent = !(std::getenv("T8GPU_INTERIOR_PERSISTENT") && std::getenv("T8GPU_INTERIOR_PERSISTENT")[0] == '0');
  const bool whole = (tile_begin == 0 && tile_count == plan->ntiles) ||
                     (interior_persistent && tile_begin == 0 && tile_count == plan->n_interior_tiles && tile_count < plan->ntiles);
  const int  np_total = plan->n_patch_tiles[0] + plan->n_patch_tiles[1] + plan->n_patch_tiles[2];
  if (np_total == 0) return plain_generic_stage<T, V>(kind, stage, plan, tile_begin, tile_count, prev, mid, out, volume, dt, speed, whole, stream);
  if (!plan->tile_desc) return static_cast<int>(hipErrorInvalidValue);
  const int nd = plan->n_deep_tiles > 0 && plan->n_deep_tiles <= plan->n_interior_tiles ? plan->n_deep_tiles : 0;
  // class segments of tile_order: [0, nd) deep, [nd, n_interior) near the boundary, [n_interior, ntiles) ghost-reading.
  // (a plan whose n_deep_tiles is 0 = "unknown" has classes 0 and 1 merged: the planner then reports no class-1 patches)
  const int seg[4] = {0, nd, plan->n_interior_tiles, plan->ntiles};
  const int b = tile_begin, e = tile_begin + tile_count;
  static const bool persistent_always = std::getenv("T8GPU_PERSISTENT") && std::getenv("T8GPU_PERSISTENT")[0] == '2';
  // generic sub-ranges that touch are launched together (single rank: one patch launch + one generic launch)
  int gb = -1, ge = -1;
  auto flush = [&]() -> int {
    if (gb < 0 || ge <= gb) return 0;
    const int rc = plain_generic_stage<T, V>(kind, stage, plan, gb, ge - gb, prev, mid, out, volume, dt, speed, whole, stream);
    gb = ge = -1;
    return rc;
  };
  static const bool scatter = std::getenv("T8GPU_LDS_SCATTER") && std::getenv("T8GPU_LDS_SCATTER")[0] == '1';
  for (int c = 0; c < 3; c++) {
    const int s0 = seg[c], s1 = seg[c + 1], p1 = s0 + plan->n_patch_tiles[c];
    if (p1 > s1) return static_cast<int>(hipErrorInvalidValue);
    const int pb = b > s0 ? b : s0, pe = e < p1 ? e : p1;   // patch tiles of this class inside the range
    int       qb = b > p1 ? b : p1, qe = e < s1 ? e : s1;   // its generic tiles
    if (pe > pb && plan->patch_dim == 3) {   // 8 x 8 x 4 hexahedral patches: their own kernel, the generic tiles apart
      if (plan->ghost_buf || plan->send_map) return static_cast<int>(hipErrorInvalidValue);   // (no ghost window in k_plain_patch3)
      if (int rc = flush()) return rc;
      // (Measured and dropped: the generic tiles on a side stream forked from / joined to the caller's stream by events, so
      //  that they run BESIDE the patches -- c5 5 235 -> 5 078, c5u 5 070 -> 4 716 M/s: the fork / join events cost more than
      //  the overlap returns, as with the Subgrid leftover blocks of round 2.)
      // (the class's patch tiles are [regular | irregular]: one launch each)
      const int ni = plan->n_irregular_tiles[c];
      if (ni < 0 || ni > plan->n_patch_tiles[c]) return static_cast<int>(hipErrorInvalidValue);
      const int i0 = p1 - ni;   // first irregular patch of the class
      const int rb = pb, re = pe < i0 ? pe : i0, ib = pb > i0 ? pb : i0, ie = pe;
      int both = -1;
      if (re > rb && ie > ib && (whole || persistent_always))   // both kinds in one launch (kernels_fused_patch3.hip)
        both = plain_patch3_both_stage<T>(kind, stage, plan, rb, re - rb, ib, ie - ib, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume, dt, speed,
                                          static_cast<hipStream_t>(stream));
      if (both > 0) return both;
      if (both == 0) {
        // done
      } else {
      if (re > rb)
        if (int rc = plain_patch3_stage<T>(kind, stage, plan, rb, re - rb, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume, dt, speed,
                                           whole || persistent_always, false, static_cast<hipStream_t>(stream)))
          return rc;
      if (ie > ib)
        if (int rc = plain_patch3_stage<T>(kind, stage, plan, ib, ie - ib, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume, dt, speed,
                                           whole || persistent_always, true, static_cast<hipStream_t>(stream)))
          return rc;
      }
    } else if (pe > pb) {
      if (int rc = flush()) return rc;
      // patches and generic tiles of the class in ONE launch where the patches carry most of it (the generic tiles then
      // run the one-tile body behind the persistent patch workgroups); otherwise the generic tiles keep their own launch
      // (the persistent tile kernel where the range covers the plan)
      const bool mixed = !scatter && qe > qb && static_cast<long long>(pe - pb) * 256 >= static_cast<long long>(qe - qb) * 128;
      int rc = plain_patch_stage<T>(kind, stage, plan, pb, pe - pb, qb, mixed ? qe - qb : 0, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume,
                                    dt, speed, whole || persistent_always, static_cast<hipStream_t>(stream));
      if (rc == -1)   // (the mixed kernel does not take this plan's generic tiles)
        rc = plain_patch_stage<T>(kind, stage, plan, pb, pe - pb, qb, 0, fmk<T>(prev), fmk<T>(mid), fmk<T>(out), volume, dt, speed,
                                  whole || persistent_always, static_cast<hipStream_t>(stream));
      else if (rc == 0 && mixed)
        qe = qb;   // done
      if (rc != 0) return rc;
    }
    if (qe > qb) {
      if (gb >= 0 && ge == qb) {
        ge = qe;
      } else {
        if (int rc = flush()) return rc;
        gb = qb;
        ge = qe;
      }
    }
  }
  return flush();
}

}  // namespace t8gpu_hip

namespace t8gpu_hip {
template <class T>
__global__ __launch_bounds__(128) void k_geo_frames(T* __restrict__ table, int n_geo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_geo) return;
  T* const row  = table + 12 * static_cast<size_t>(i);
  const T  n[3] = {row[0], row[1], row[2]};
  T        t1[3], t2[3];
  face_basis_fast<T>(n, t1, t2);
  for (int k = 0; k < 3; k++) {
    row[4 + k] = t1[k];
    row[8 + k] = t2[k];
  }
}
template <class T>
int geo_frames(void* table, int n_geo, void* stream) {
  if (n_geo < 0 || (n_geo > 0 && !table)) return static_cast<int>(hipErrorInvalidValue);
  if (n_geo == 0) return 0;
  hipLaunchKernelGGL((k_geo_frames<T>), dim3((n_geo + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), static_cast<T*>(table), n_geo);
  return static_cast<int>(hipGetLastError());
}
}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_plain_geo_frames_f32(void* geo_table, int n_geo, void* stream) { return t8gpu_hip::geo_frames<float>(geo_table, n_geo, stream); }
int t8gpu_hip_plain_geo_frames_f64(void* geo_table, int n_geo, void* stream) { return t8gpu_hip::geo_frames<double>(geo_table, n_geo, stream); }
int t8gpu_hip_plain_needs_csr(const T8gpuPlainPlan* plan) { return plan && t8gpu_hip::plain_tiles_pipelined(plan) ? 0 : 1; }
int t8gpu_hip_plain_fused_stage_f32(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count,
                                    T8gpuVars_f32 prev, T8gpuVars_f32 mid, T8gpuVars_f32 out, const float* volume,
                                    float dt, float* speed, void* stream) {
  return t8gpu_hip::plain_fused_stage<float>(kind, stage, plan, tile_begin, tile_count, prev, mid, out, volume, dt,
                                             speed, stream);
}
int t8gpu_hip_plain_fused_stage_f64(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count,
                                    T8gpuVars_f64 prev, T8gpuVars_f64 mid, T8gpuVars_f64 out, const double* volume,
                                    double dt, double* speed, void* stream) {
  return t8gpu_hip::plain_fused_stage<double>(kind, stage, plan, tile_begin, tile_count, prev, mid, out, volume, dt,
                                              speed, stream);
}
}
