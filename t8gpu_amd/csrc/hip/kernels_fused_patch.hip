// kernels_fused_patch.hip -- fused RK stage over STRUCTURED PATCHES of a plain-element mesh (round 3).
//
// Reference: examples/compressible_euler/kernels.cu:135-309 (kepes_compute_fluxes) + ssp_runge_kutta.inl:30-99, the same
// stage the tile kernels fuse (kernels_fused.hip, kernels_fused_persistent.hip).
//
// Adaptive quadrilateral meshes are locally structured: on the benchmark mesh (c4) 97 % of the elements lie in aligned
// 16 x 16 blocks of same-size squares, which are 256 CONSECUTIVE elements in Morton order. The host plan finds such
// patches from the reference-format arrays alone (csrc/host/tile_plan.cpp: find_patches) and hands them over as tiles
// WITHOUT face records. For those tiles everything the generic tile kernels fetch per face -- packed (l, r) indices,
// geometry index, original face id, the per-element face list -- follows from the lane index:
//
//   lane t = element e0 + t = cell (i, j) of the patch (t = Morton interleave of i and j, x in bit 0)
//   phase 1   primitives of the own cell -> registers and an LDS record; wave 3 also takes the 64 cells across the
//             patch's four sides (the plan lists them per patch: [-x | +x | -y | +y] x 16)
//   phase 2   the lane's OWN faces: +x against cell (i + 1, j), +y against (i, j + 1) (normals exactly +e_x, +e_y: no
//             rotation, no dictionary; left operand in registers); lanes 0-31 of wave 0 evaluate the 32 faces of the
//             -x / -y sides, which belong to the cells across (same orientation, same values as in their own tile)
//   phase 3   a cell adds its four fluxes in ascending original face id, which for a patch is: the -x and -y faces in
//             the order of the owning neighbours' indices (a function of (i, j), checked by the planner), then +x, +y;
//             RK stage
//
// Per 256 elements that is 5 + 9 wave-rounds of primitives / fluxes (the tile kernels: 8 + 8 for 234) and a quarter of
// the index traffic; speed estimates go to fbase + 2 t (+1) without an id list. Same flux functions, same operand order,
// same summation order as the tile kernels: results are bitwise theirs (tests/test_gpu_patch.py), and a face that a patch
// and a generic tile both evaluate gets the same value in both.
//
// Persistent and software-pipelined like k_plain_persistent: a workgroup walks patches j, j + n, ... of its XCD's
// contiguous share with the next patch's loads in flight; two barriers per patch.
#include <cstdlib>

#include "fused_common.hpp"
#include "fused_tile_body.hpp"
#include "patch_common.hpp"
#include "stage_kernel_note.hpp"

namespace t8gpu_hip {

// workgroup `wg` of the `nwg` that share the patch tiles [tile_begin, tile_begin + tile_count) of tile_order
// chunk = 0: persistent walk -- the XCD's workgroups stride through its share together (tiles j, j + n, ...); chunk = c > 0:
// workgroup j of the XCD takes the c CONSECUTIVE tiles [j c, j c + c) of the share and leaves (a grid of ~count / c
// workgroups that the hardware hands out as slots free up: what a launch beside other kernels wants, see plain_patch_stage)
template <class T, int KIND, int STAGE, bool NT>
T8_DEV void plain_patch_body(const T8gpuPlainPlan& P, int tile_begin, int tile_count, int wg, int nwg, int chunk, const FVars<T>& prev,
                             const FVars<T>& src, const FVars<T>& out, const T* __restrict__ vol, T dt, T* __restrict__ speed) {
  constexpr int NW  = KIND == 0 ? kPrimWords : 5;
  constexpr int REC = rec_words<T, NW>();
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  T* const  ff = reinterpret_cast<T*>(lds_raw);               // [5][kPatchFF] the patch's face fluxes
  T* const  pe = ff + 5 * kPatchFF;                           // [320][REC] primitives (or states): 256 own, 64 across the sides
  constexpr bool kTab = sizeof(T) == 8 && KIND == 0;          // fp64 KEPES: the logarithm table, behind the records
  double* const lt = reinterpret_cast<double*>(pe + REC * 320);
  const int tid = threadIdx.x;

  // this workgroup's patches (same walk as k_plain_persistent: XCD x takes one contiguous eighth of the range, its
  // workgroups walk it together)
  const int G = nwg, xcd = wg & 7, jw = wg >> 3;
  const int nxcd = G < 8 ? G : 8, nx = (G - xcd + 7) >> 3;
  const int per = tile_count / nxcd, rem = tile_count % nxcd;
  const int x0   = tile_begin + xcd * per + (xcd < rem ? xcd : rem);
  int       tend = x0 + per + (xcd < rem ? 1 : 0);
  int       t    = x0 + (chunk > 0 ? jw * chunk : jw);
  if (chunk > 0 && t + chunk < tend) tend = t + chunk;
  if (xcd >= nxcd || t >= tend) return;
  if (kTab) {
    lt[tid] = kLogTab[tid];
    __syncthreads();
  }

  // ---- lane constants -------------------------------------------------------------------------------------------------
  int ci = 0, cj = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    ci |= ((tid >> (2 * b)) & 1) << b;
    cj |= ((tid >> (2 * b + 1)) & 1) << b;
  }
  // records of the right operands of the lane's faces; flux slots of its - faces (the + faces of the cells across)
  const int  rx   = ci < 15 ? patch_morton(ci + 1, cj) : 272 + cj;
  const int  ry   = cj < 15 ? patch_morton(ci, cj + 1) : 304 + ci;
  const int  a_mx = ci > 0 ? patch_morton(ci - 1, cj) : 512 + cj;
  const int  a_my = cj > 0 ? 256 + patch_morton(ci, cj - 1) : 528 + ci;
  const bool yfirst_lane = patch_ctz4(cj) >= patch_ctz4(ci);   // (cell (0, 0): per patch, from the descriptor)
  // The two extra jobs of a patch -- the 64 cells across the sides (one round of primitives) and the 32 faces of the - sides
  // (one half-filled flux round) -- ROTATE over the four wavefronts from patch to patch: wavefront w has role (w + it) & 3 at
  // the workgroup's it-th patch, role 3 takes the side cells, role 0 the - faces. Fixed to wavefronts 3 and 0 (round 3) the
  // wavefronts of a workgroup carry 730 / 520 / 520 / 620 VALU instructions per patch, and a workgroup's wavefront i runs on
  // SIMD i of its CU: one SIMD of four does 22 % more than the average while the others wait at the barriers.
  // Measured (same box, scripts/ab_variants.sh): c4 fp64 11 640 -> 11 727 M/s, fp32 and c2 unchanged -- the dispatcher does not pin
  // wavefront i to SIMD i as strictly as feared. T8GPU_EXP_FIXED_ROLES: experiment builds keep the fixed assignment.
  const int  wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
#ifdef T8GPU_EXP_FIXED_ROLES
  auto role = [&](int) { return wv == 3 ? 3 : (wv == 0 ? 0 : 1); };
#else
  auto role = [&](int it) { return (wv + it) & 3; };
#endif
  const int  hl      = ln;                                     // side-cell lane: cell hl of [-x | +x | -y | +y] x 16
  const bool minus_y = ln >= 16;                               // - face lanes 0-31 of the role-0 wavefront: -x side, then -y side
  const int  m_l = 256 + (minus_y ? 32 : 0) + (ln & 15);       // left operand: the cell across; right: the patch's cell
  const int  m_r = minus_y ? patch_morton(ln & 15, 0) : patch_morton(0, ln & 15);

  typedef int int8v __attribute__((ext_vector_type(8)));
  struct Desc {
    int    e0, h0, fbase, flags;
    double area, vol;   // vol: the patch's uniform element volume where flags has 0x400 (tile_plan.cpp), else unused
  };
  auto load_desc = [&](int tt) {   // scalar load through the constant address space (see k_plain_persistent)
#ifdef T8GPU_EXP_TILEMOD   // experiment builds only: every patch is one of the first few, all traffic stays in the caches
    const size_t k = static_cast<size_t>(tile_begin + (tt < tend ? tt : tend - 1) % T8GPU_EXP_TILEMOD);
#else
    const size_t k = static_cast<size_t>(tt < tend ? tt : tend - 1);
#endif
    const int8v  r = *reinterpret_cast<const __attribute__((address_space(4))) int8v*>(
        reinterpret_cast<const __attribute__((address_space(4))) char*>(reinterpret_cast<uintptr_t>(P.tile_desc)) + 32 * k);
    Desc d;
    d.e0 = r[0]; d.h0 = r[2]; d.fbase = r[4]; d.flags = r[5];
    d.area = __hiloint2double(r[7], r[6]);
    d.vol  = __hiloint2double(r[3], r[1]);
    return d;
  };
  struct Pre {
    T s0[5], sh[5];
  };
  // first: the prologue's request. A launch with a ghost window (t8gpu_hip.h; the multi-rank driver's ghost-reading class)
  // runs one patch per workgroup -- plain_patch_stage() sees to it -- so only that request can meet a ghost slot.
  auto prefetch = [&](const Desc& d, int hslot, bool first, bool halo_wave) {
    Pre p;
#pragma unroll
    for (int k = 0; k < 5; k++) p.s0[k] = at32<T>(src.p[k], static_cast<unsigned>(d.e0 + tid));
    if (halo_wave && first && P.ghost_buf) {
#pragma unroll
      for (int k = 0; k < 5; k++) p.sh[k] = ghost_window_load<T>(P, src, hslot, k);
    } else if (halo_wave) {
#pragma unroll
      for (int k = 0; k < 5; k++) p.sh[k] = at32<T>(src.p[k], static_cast<unsigned>(hslot));
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) p.sh[k] = T(0);
    }
    return p;
  };

  const int stride = chunk > 0 ? 1 : nx;
  Desc      d0 = load_desc(t), d1 = load_desc(t + stride);
  int       hs_a = role(0) == 3 ? P.halo_ids[d0.h0 + hl] : 0, hs_b = role(1) == 3 ? P.halo_ids[d1.h0 + hl] : 0;
  Pre       cur = prefetch(d0, hs_a, true, role(0) == 3);
  int       it  = 0;
  T         res[5] = {T(0), T(0), T(0), T(0), T(0)};
  int       res_e  = -1;
  __builtin_amdgcn_s_waitcnt(0);   // (see k_plain_persistent: the prologue's loads must not become a wait inside the loop)

  for (; t < tend; t += stride, it++) {
    const bool halo_wave = role(it) == 3, minus_lane = role(it) == 0 && ln < 32;
    const Desc d2   = load_desc(t + 2 * stride);
    const int  hs_c = role(it + 2) == 3 ? P.halo_ids[d2.h0 + hl] : 0;
    Pre        nxt;
    if (t + stride < tend) nxt = prefetch(d1, hs_b, false, role(it + 1) == 3);
    const int e = d0.e0 + tid;
    T         pv[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (STAGE > 1) {
#pragma unroll
      for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at32<T>(prev.p[k], static_cast<unsigned>(e)));
    }
    // (patches of uniform volume carry it in their descriptor: 8 of ~130 bytes per element and stage less to load)
    const T volume = (d0.flags & 0x400) ? static_cast<T>(d0.vol) : at32<T>(vol, static_cast<unsigned>(e));
    const T area   = static_cast<T>(d0.area);

    // ---- phase 1: records of the own cell and (wave 3) of the cells across the sides ----------------------------------
    T mine[NW];
    if (KIND == 0) {
      prim_words<T>(cur.s0, mine, lt);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) mine[k] = cur.s0[k];
    }
    rec_store<T, NW>(pe + tid * REC, mine);
    if (halo_wave) {
      T w[NW];
      if (KIND == 0) {
        prim_words<T>(cur.sh, w, lt);
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = cur.sh[k];
      }
      rec_store<T, NW>(pe + (256 + hl) * REC, w);
    }
    if (res_e >= 0) {   // results of the previous patch: behind this iteration's first wait (vmcnt retires in order)
#pragma unroll
      for (int k = 0; k < 5; k++) stream_store<NT>(&at32<T>(out.p[k], static_cast<unsigned>(res_e)), res[k]);
    }
    __syncthreads();

    // ---- phase 2: the lane's +x and +y faces; the - sides ------------------------------------------------------------
    T gy[5];
    {
      T wr[NW], g[5], sx, sy;
      rec_load<T, NW>(pe + rx * REC, wr);
      patch_face<T, KIND, NW>(false, mine, wr, area, g, sx);
#pragma unroll
      for (int k = 0; k < 5; k++) ff[k * kPatchFF + tid] = g[k];
      rec_load<T, NW>(pe + ry * REC, wr);
      patch_face<T, KIND, NW>(true, mine, wr, area, gy, sy);
#pragma unroll
      for (int k = 0; k < 5; k++) ff[k * kPatchFF + 256 + tid] = gy[k];
      if (speed) {   // the patch's own faces: ids fbase + 2 t (+x) and fbase + 2 t + 1 (+y): one 16-byte store where aligned
        T* const sp = speed + d0.fbase + 2 * tid;
        if ((d0.fbase & 1) == 0 || sizeof(T) == 4) {   // (wave-uniform; fp32: an 8-byte store is aligned for every fbase)
          using V2 = typename vec2<T>::type;
          V2 v;
          v.x = sx;
          v.y = sy;
          *reinterpret_cast<V2*>(sp) = v;
        } else {
          sp[0] = sx;
          sp[1] = sy;
        }
      }
    }
    if (minus_lane) {
      T wl[NW], wr[NW], g[5], sm;
      rec_load<T, NW>(pe + m_l * REC, wl);
      rec_load<T, NW>(pe + m_r * REC, wr);
      patch_face<T, KIND, NW>(minus_y, wl, wr, area, g, sm);
#pragma unroll
      for (int k = 0; k < 5; k++) ff[k * kPatchFF + 512 + ln] = g[k];
    }
    __syncthreads();

    // ---- phase 3: the four fluxes in ascending face id, RK stage -------------------------------------------------------
    const bool yfirst = tid == 0 ? (d0.flags & 1) != 0 : yfirst_lane;
    const int  a1 = yfirst ? a_my : a_mx, a2 = yfirst ? a_mx : a_my;
    T          acc[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      acc[k] = __builtin_fma(T(1), ff[k * kPatchFF + a1], T(0));
      acc[k] = __builtin_fma(T(1), ff[k * kPatchFF + a2], acc[k]);
      acc[k] = __builtin_fma(T(-1), ff[k * kPatchFF + tid], acc[k]);
      acc[k] = __builtin_fma(T(-1), gy[k], acc[k]);
    }
    // (a patch of uniform volume: its operands are wave-uniform -- the power-of-two test runs on scalar registers)
    const T scale = (d0.flags & 0x400) ? rk_scale(dt, static_cast<T>(d0.vol)) : dt / volume;
#pragma unroll
    for (int k = 0; k < 5; k++) res[k] = rk_stage_update<T, STAGE>(pv[k], cur.s0[k], scale, acc[k]);
    res_e = e;

    cur  = nxt;
    d0   = d1;
    d1   = d2;
    hs_b = hs_c;
  }
  if (res_e >= 0) {
#pragma unroll
    for (int k = 0; k < 5; k++) stream_store<NT>(&at32<T>(out.p[k], static_cast<unsigned>(res_e)), res[k]);
    if (P.send_map) ghost_window_send<T>(P, res_e, res);   // (one patch per workgroup there: this is its only store)
  }
}

// (second launch bound = waves per SIMD the register allocation must allow: 3 workgroups per CU in fp64, 5 in fp32)
template <class T, int KIND, int STAGE, bool NT>
__global__ __launch_bounds__(256, sizeof(T) == 8 ? 3 : 5) void k_plain_patch(T8gpuPlainPlan P, int tile_begin, int tile_count, int chunk, FVars<T> prev,
                                                                                FVars<T> src, FVars<T> out, const T* __restrict__ vol, T dt,
                                                                                T* __restrict__ speed) {
  plain_patch_body<T, KIND, STAGE, NT>(P, tile_begin, tile_count, blockIdx.x, gridDim.x, chunk, prev, src, out, vol, dt, speed);
}

// ONE launch per stage for a range of tile_order that holds patch tiles AND generic tiles: the first `patch_wgs`
// workgroups walk the patch tiles, every further workgroup takes one generic tile (fused_tile_body.hpp). As two launches
// the generic tiles of the benchmark mesh -- 3 % of its elements -- cost 9 % of the stage: a launch of their own, started
// when the patch launch has drained. Here they start as the persistent patch workgroups finish and fill the ragged end.
template <class T, int KIND, int STAGE, bool NT>
__global__ __launch_bounds__(256, sizeof(T) == 8 ? 3 : 5) void k_plain_stage(T8gpuPlainPlan P, int patch_begin, int patch_count, int patch_wgs, int chunk,
                                                                                int tile_begin, int tile_count, FVars<T> prev, FVars<T> src,
                                                                                FVars<T> out, const T* __restrict__ vol, T dt,
                                                                                T* __restrict__ speed) {
  const int b = blockIdx.x;
  if (b < patch_wgs) {
    plain_patch_body<T, KIND, STAGE, NT>(P, patch_begin, patch_count, b, patch_wgs, chunk, prev, src, out, vol, dt, speed);
  } else {
#ifdef T8GPU_EXP_TILEMOD
    const int pos = tile_begin + xcd_position(b - patch_wgs, tile_count) % T8GPU_EXP_TILEMOD;
#else
    const int pos = tile_begin + xcd_position(b - patch_wgs, tile_count);
#endif
    plain_tile_body<T, KIND, STAGE, true, 2>(P, pos, prev, src, out, vol, dt, speed);
  }
}

// Patch tiles [patch_begin, patch_begin + patch_count) of tile_order, and -- in the same launch -- the generic tiles
// [tile_begin, tile_begin + tile_count) (tile_count = 0: none). persistent = false: one patch per workgroup (class-split
// multi-rank launches: slots free up continuously, see plain_fused_stage). Returns -1 when generic tiles were handed in
// that the mixed kernel does not take (the caller then launches the two parts separately), else 0 or a hipError_t.
template <class T>
int plain_patch_stage(int kind, int stage, const T8gpuPlainPlan* plan, int patch_begin, int patch_count, int tile_begin, int tile_count,
                      FVars<T> prev, FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, bool persistent, hipStream_t stream) {
  if (patch_count <= 0) return tile_count > 0 ? -1 : 0;
  // (does this launch run beside another lane's kernels? a plan with ghost-reading tiles, launched in part)
  const bool shared_gpu = plan->n_interior_tiles < plan->ntiles && patch_count + (tile_count > 0 ? tile_count : 0) < plan->ntiles;
  if (!plan->tile_desc) return static_cast<int>(hipErrorInvalidValue);
  // (the kernel addresses a plane by a 32-bit byte offset, patch_common.hpp: at32)
  if (plan->n_slots_addressed <= 0 || static_cast<unsigned long long>(plan->n_slots_addressed) * sizeof(T) >= (1ull << 32))
    return static_cast<int>(hipErrorInvalidValue);
  const int    nw  = kind == 0 ? kPrimWords : 5;
  const int    rec = sizeof(T) == 8 ? (nw > 5 ? 10 : 6) : 12;
  const size_t tab = (sizeof(T) == 8 && kind == 0) ? 2 * kLogTabEntries * sizeof(double) : 0;
  size_t       lds = sizeof(T) * (static_cast<size_t>(5) * kPatchFF + static_cast<size_t>(rec) * 320) + tab;
  if (tile_count > 0) {
    // what plain_tile_body<T, K, S, true, 2> takes (kernels_fused.hip: the pipelined kernel with a geometry dictionary, two
    // passes of 256 faces), and its LDS window
    const int  slots = plan->max_slots > 0 ? plan->max_slots : plan->max_elems + plan->max_halo;
    const bool ok = plan->ell && plan->tile_desc && plan->ell_width >= 8 && plan->ell_width % 8 == 0 && plan->max_elems <= 256 && slots <= 512 &&
                    plan->max_faces <= 512 && plan->geo_idx && plan->geo_table && plan->n_geo > 0;
    static const bool off = std::getenv("T8GPU_PATCH_MIXED") && std::getenv("T8GPU_PATCH_MIXED")[0] == '0';   // (measurements)
    if (!ok || off) return -1;
    const size_t lds_tile = sizeof(T) * (static_cast<size_t>(nw) * slots + static_cast<size_t>(5) * 256) + (tab ? tab + 16 : 0);
    if (lds_tile > lds) lds = lds_tile;
    if (sizeof(T) == 8 && 3 * lds > static_cast<size_t>(156) * 1024) return -1;   // (the kernel lives on three workgroups per CU)
  }
  const int        cus        = device_cu_count();
  static const int per_cu_env = env_per_cu("T8GPU_PATCH_WGS");
  if (cus == 0) return static_cast<int>(hipErrorInvalidDevice);
  // persistent = true: the launch covers the whole plan -- as many patch workgroups as stay resident (3 per CU in fp64).
  // persistent = false: a class of a multi-rank stage, launched beside the pack / RCCL / unpack kernels of the exchange: one
  // patch per workgroup, so that slots free up continuously. That costs the patch kernel its software pipeline (13 % at c2
  // size, 19 % at c4 size: T8GPU_PATCH_PERSISTENT=0 on one rank), but a "polite" persistent grid of two workgroups per CU
  // for the class launches -- measured in round 3 -- is worse where it matters: rank 3 of the 8-way c4 split with an RCCL
  // self-exchange 0.173 -> 0.205 ms/step (profiles/r03_halo_overhead.md); the exchange kernels wait behind resident
  // workgroups that never leave.
  static const bool never_persistent = std::getenv("T8GPU_PATCH_PERSISTENT") && std::getenv("T8GPU_PATCH_PERSISTENT")[0] == '0';   // (measurements)
  if (never_persistent) persistent = false;
  // a ghost window (t8gpu_hip.h) is honoured by a workgroup's FIRST patch only (plain_patch_body): one patch per workgroup
  if (plan->ghost_buf || plan->send_map) persistent = false;
  const int  per_cu    = per_cu_env > 0 ? per_cu_env : (sizeof(T) == 8 ? 3 : 5);
  const int  resident  = cus * per_cu;
  // A persistent launch that shares the GPU with the other lane's kernels (the interior launch of a multi-rank stage: the plan
  // has ghost-reading tiles and the range is not the whole plan) must not count on all its workgroups being resident from
  // the start: whatever slots the RCCL kernel and the ghost-reading tiles hold when it is dispatched, that many of its
  // workgroups start when the FIRST of the others retire and then walk their whole share -- the launch takes up to twice
  // as long (rank 0 of the 2-way c4 split: interior kernel 180 instead of 150 us, profiles/r04_halo_overhead.md). Such
  // launches hand out CHUNKS instead: workgroup j takes `chunk` consecutive patches, with the software pipeline inside the
  // chunk, and the hardware dispatches the chunks as slots free up.
  static const int chunk_env = std::getenv("T8GPU_PATCH_CHUNK") ? std::atoi(std::getenv("T8GPU_PATCH_CHUNK")) : -1;
  int chunk = 0;
  // (measured, rank of the c4 mesh split 2 / 4 / 8 ways with a self-exchange, ms per step: persistent walk 0.549 / 0.313 / 0.151,
  //  chunks of 2: 0.497 / 0.280 / 0.153, of 3: 0.483 / 0.268 / 0.151, of 4: 0.481 / 0.264 / 0.156, of 6: 0.469 / 0.288 / 0.157)
  if (persistent && shared_gpu && patch_count >= resident) chunk = chunk_env >= 0 ? chunk_env : 3;
  int patch_wgs = (!persistent || patch_count < resident) ? patch_count : resident;
  if (chunk > 0) {   // 8 XCD shares of ceil(count / 8) patches, ceil(share / chunk) workgroups each
    const int share = (patch_count + 7) / 8;
    patch_wgs       = 8 * ((share + chunk - 1) / chunk);
  }
  const dim3 grid(patch_wgs + (tile_count > 0 ? tile_count : 0)), block(256);
  // non-temporal stage results / previous-state loads where the stage's planes are a stream for the caches (flux_math.hpp)
  const bool nt = stream_hint(plan->n_slots_addressed, sizeof(T));
  note_stage_kernel(patch_count + (tile_count > 0 ? tile_count : 0),
                    tile_count > 0 ? (nt ? "k_plain_stage<T, K, S, true>" : "k_plain_stage<T, K, S, false>") : (nt ? "k_plain_patch<T, K, S, true>" : "k_plain_patch<T, K, S, false>"),
                    static_cast<int>(sizeof(T)), kind, stage);
#define T8_PAN(K, S, N)                                                                                                           \
  do {                                                                                                                            \
    if (tile_count > 0)                                                                                                           \
      hipLaunchKernelGGL((k_plain_stage<T, K, S, N>), grid, block, lds, stream, *plan, patch_begin, patch_count, patch_wgs, chunk, tile_begin, \
                         tile_count, prev, mid, out, volume, dt, speed);                                                          \
    else                                                                                                                          \
      hipLaunchKernelGGL((k_plain_patch<T, K, S, N>), grid, block, lds, stream, *plan, patch_begin, patch_count, chunk, prev, mid, out, volume, dt, \
                         speed);                                                                                                  \
  } while (0)
#define T8_PA(K, S)        \
  do {                     \
    if (nt)                \
      T8_PAN(K, S, true);  \
    else                   \
      T8_PAN(K, S, false); \
  } while (0)
#define T8_PAS(K)          \
  do {                     \
    if (stage == 1)        \
      T8_PA(K, 1);         \
    else if (stage == 2)   \
      T8_PA(K, 2);         \
    else                   \
      T8_PA(K, 3);         \
  } while (0)
  if (kind == 0)
    T8_PAS(0);
  else if (kind == 1)
    T8_PAS(1);
  else
    T8_PAS(2);
#undef T8_PAS
#undef T8_PA
#undef T8_PAN
  return static_cast<int>(hipGetLastError());
}

template int plain_patch_stage<float>(int, int, const T8gpuPlainPlan*, int, int, int, int, FVars<float>, FVars<float>, FVars<float>,
                                      const float*, float, float*, bool, hipStream_t);
template int plain_patch_stage<double>(int, int, const T8gpuPlainPlan*, int, int, int, int, FVars<double>, FVars<double>, FVars<double>,
                                       const double*, double, double*, bool, hipStream_t);

}  // namespace t8gpu_hip
