// kernels_fused_persistent.hip -- persistent, software-pipelined form of the fused tile kernel (plain elements).
//
// Why: measured on the c4 mesh (gpurun_out/ab_r2_*.txt) the one-tile-per-workgroup kernel is
// bound by a CHAIN of memory round trips per tile, not by bandwidth or arithmetic: alone on a CU a workgroup needs
// 4.9 us per tile with all arithmetic removed and 6.1 us with it (tile offsets -> halo ids -> states -> geometry
// rows -> previous state -> store acknowledgements), i.e. 1.2 us of instructions behind ~5 us of latency that only
// the other three workgroups of the CU can hide.
//
// Here a workgroup is persistent: it walks a contiguous run of tiles (XCD-aware: neighbouring runs share an L2) and
// keeps the NEXT tile's loads in flight while it evaluates the CURRENT tile's faces:
//
//   iteration t:  wait for what iteration t-1 requested
//                 [halo ids of the workgroup's tile after next, states + face records + ELL row of its next tile,
//                  previous-step state and volume of tile t]            requested; in flight during everything below
//                 phase 1: primitives of tile t -> LDS records;  results of tile t-1 stored;  barrier
//                 phase 2: the lane's two faces -> LDS;  barrier;  phase 3: per-element gather in list order, RK stage
//                 -> results kept in registers (two barriers per tile; the one-tile kernels need four)
//
// No load issued after the prefetch is consumed before the next iteration (vmcnt retires in order, so one late
// load would wait for the whole prefetch): the geometry dictionary lives in LDS (copied once per workgroup), the
// previous-step state is requested BEFORE the prefetch, and the stores of a tile are issued after the next tile's
// first wait. LDS: primitives as padded per-slot records (one address per gather, conflict-free for consecutive
// slots) + the tile's 512 face fluxes + the dictionary (~50 KB in fp64: three workgroups per CU, which is also what the
// register file allows).
//
// Same arithmetic, same summation order (ascending original face id through the ELL rows) as k_plain_fused_p:
// results are bitwise those of the one-tile-per-workgroup kernels, independent of tiling and partition.
#include <cstdlib>

#include "fused_common.hpp"
#include "stage_kernel_note.hpp"

namespace t8gpu_hip {

// One ELL chunk (8 entries of an element's face list, ascending original face id): entry = tile-local face (< 512) |
// 0x8000 when the element is the face's right side; 0xFFFF pads the row. All of the tile's fluxes are in LDS (one
// buffer of 512), so the list is walked once and the sum runs in list order -- the order of the two-pass kernels.
template <class T>
T8_DEV void ell_gather(uint4 w, const T* __restrict__ ff, T acc[5]) {
  const unsigned ent[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if ((ent[j] & 0x7E00u) == 0u) {   // a face index below 512: not padding
      const T* p   = ff + (ent[j] & 511u);
      const T  wgt = (ent[j] & 0x8000u) ? T(1) : T(-1);
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] = __builtin_fma(wgt, p[k * 512], acc[k]);
    }
  }
}

struct TileDesc {
  int e0, ne, h0, nh, f0, nf, el0;   // el0: the tile's first row in P.ell (rows exist for generic tiles' elements only)
};

// (second launch bound = waves per SIMD the register allocation must allow: 3 workgroups per CU in fp64, 5 in fp32)
// ELLC = 8-entry chunks per ELL row: 1 (2D meshes, uniform 3D meshes) or up to 3 (3D AMR: up to 24 faces per element)
template <class T, int KIND, int STAGE, int ELLC>
__global__ __launch_bounds__(256, sizeof(T) == 8 ? 3 : 5) void k_plain_persistent(T8gpuPlainPlan P, int tile_begin, int tile_count, FVars<T> prev,
                                                          FVars<T> src, FVars<T> out, const T* __restrict__ vol, T dt,
                                                          T* __restrict__ speed) {
  constexpr int NW  = KIND == 0 ? kPrimWords : 5;
  constexpr int REC = rec_words<T, NW>();
  using V4          = typename vec4<T>::type;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  T* const  ff = reinterpret_cast<T*>(lds_raw);                       // [5][512] the tile's face fluxes
  V4* const gt = reinterpret_cast<V4*>(ff + 5 * 512);                 // [n_geo][3] {n, area} {t1, .} {t2, .}
  T* const  pe = reinterpret_cast<T*>(gt + 3 * P.n_geo);              // [max_slots][REC] primitives (or states) per slot
  constexpr bool kTab = sizeof(T) == 8 && KIND == 0;                  // fp64 KEPES: the logarithm table, behind the records
  double* const lt = reinterpret_cast<double*>(pe + static_cast<size_t>(REC) * P.max_slots);
  const int tid = threadIdx.x;

  // This workgroup's tiles. Workgroups b, b + 8, ... share an XCD (and its L2): XCD x gets one contiguous eighth of
  // the tile range and its workgroups walk it TOGETHER -- workgroup j of the XCD's nx takes tiles j, j + nx, j + 2 nx,
  // ... -- so at any time an XCD works on ~nx neighbouring tiles: their halo reads hit the L2 lines the neighbours
  // just fetched, and in every plane the chip streams through 8 compact windows instead of hundreds of scattered
  // runs (DRAM row locality).
  const int G = gridDim.x, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int nxcd = G < 8 ? G : 8, nx = (G - xcd + 7) >> 3;           // XCDs in use, workgroups on this one
  const int per = tile_count / nxcd, rem = tile_count % nxcd;
  const int x0   = tile_begin + xcd * per + (xcd < rem ? xcd : rem);  // this XCD's tiles: [x0, tend)
  const int tend = x0 + per + (xcd < rem ? 1 : 0);
  int       t    = x0 + j;
  if (xcd >= nxcd || t >= tend) return;

  for (int i = tid; i < 3 * P.n_geo; i += 256) gt[i] = reinterpret_cast<const V4*>(P.geo_table)[i];   // visible after the first barrier
  if (kTab) {
    lt[tid] = kLogTab[tid];   // 2 x 128 doubles, one per lane; phase 1 of the first tile reads it: barrier below
    __syncthreads();
  }

  // Tile descriptors: one 32-byte record per tile (plan.tile_desc, in execution order), read two tiles ahead. The
  // plan is read-only for the kernel's lifetime, so the record is read through the constant address space: a scalar
  // load into SGPRs (a plain load could not be scalar -- the kernel stores to memory the compiler must assume aliases
  // it -- and as a vector load it would cost 8 VGPRs per tile in flight and sit in the in-order vmcnt queue).
  typedef int int8v __attribute__((ext_vector_type(8)));
  auto load_desc = [&](int tt) {
    const int last = tend - 1;
#ifdef T8GPU_EXP_TILEMOD   // experiment builds only: every tile is one of the first few, all traffic stays in the caches
    const size_t k = static_cast<size_t>((tt < tend ? tt : last) % T8GPU_EXP_TILEMOD);
#else
    const size_t k = static_cast<size_t>(tt < tend ? tt : last);   // (past the end: a valid tile, never used)
#endif
    const int8v r = *reinterpret_cast<const __attribute__((address_space(4))) int8v*>(
        reinterpret_cast<const __attribute__((address_space(4))) char*>(reinterpret_cast<uintptr_t>(P.tile_desc)) + 32 * k);
    TileDesc d;
    d.e0 = r[0]; d.ne = r[1]; d.h0 = r[2]; d.nh = r[3]; d.f0 = r[4]; d.nf = r[5]; d.el0 = r[6];
    return d;
  };
  // slots of the lane's two elements: lane < ne owns element e0 + lane, the remaining own + halo slots are halo elements
  auto halo_slots = [&](const TileDesc& d, int& slot0, int& slot1) {
    const int  i1  = tid + 256;
    const bool own = tid < d.ne, a0 = tid < d.ne + d.nh, a1 = i1 < d.ne + d.nh;
    slot0 = own ? d.e0 + tid : (a0 ? P.halo_ids[d.h0 + (tid - d.ne)] : d.e0);
    slot1 = a1 ? P.halo_ids[d.h0 + (i1 - d.ne)] : d.e0;
  };
  struct Pre {   // everything phase 1-3 of a tile read from global memory, except previous state / volume
    T        s0[5], s1[5];
    uint32_t lr0, lr1;
    uint16_t gi0, gi1;   // kept as loaded: any arithmetic on a prefetched value would wait for the whole prefetch
    int      orig0, orig1;
    uint4    ell;
  };
  auto prefetch = [&](const TileDesc& d, int slot0, int slot1) {
    Pre p;
#pragma unroll
    for (int k = 0; k < 5; k++) p.s0[k] = src.p[k][slot0];
#pragma unroll
    for (int k = 0; k < 5; k++) p.s1[k] = src.p[k][slot1];
    const int j0 = d.f0 + (tid < d.nf ? tid : 0), j1 = d.f0 + (tid + 256 < d.nf ? tid + 256 : 0);
    p.lr0   = P.face_lr[j0];
    p.lr1   = P.face_lr[j1];
    p.gi0   = P.geo_idx[j0];   // (no arithmetic on prefetched values here: a use would wait for the whole prefetch)
    p.gi1   = P.geo_idx[j1];
    p.orig0 = speed ? P.face_orig[j0] : -1;
    p.orig1 = speed ? P.face_orig[j1] : -1;
    p.ell   = *reinterpret_cast<const uint4*>(P.ell + static_cast<size_t>(d.el0 + (tid < d.ne ? tid : 0)) * P.ell_width);
    return p;
  };

  // ---- prologue: tile t fully, slots of the next tile ---------------------------------------------------------------
  const int stride = nx;
  TileDesc  d0 = load_desc(t), d1 = load_desc(t + stride);
  int       a_slot0, a_slot1, b_slot0, b_slot1;
  halo_slots(d0, a_slot0, a_slot1);
  halo_slots(d1, b_slot0, b_slot1);
  Pre cur = prefetch(d0, a_slot0, a_slot1);

  T   res[5] = {T(0), T(0), T(0), T(0), T(0)};   // RK results of the previous tile, stored one iteration late
  int res_e  = -1;
  // the prologue's loads are complete before the loop is entered: otherwise every use of `cur` inside the loop would
  // carry a wait that only the first iteration needs (the compiler merges the loop entry's pending loads into the body)
  __builtin_amdgcn_s_waitcnt(0);

  for (; t < tend; t += stride) {
    // Everything requested during the previous iteration is complete from here on (the first use below waits for it;
    // it has had a whole tile's faces to arrive): slots of the next tile, `cur`.
    const TileDesc d2 = load_desc(t + 2 * stride);
    int            c_slot0, c_slot1;
    halo_slots(d2, c_slot0, c_slot1);

    const bool own = tid < d0.ne, a0 = tid < d0.ne + d0.nh, a1 = tid + 256 < d0.ne + d0.nh;
    // ---- next tile's loads, and this tile's previous-step state / volume: in flight during the whole iteration -------
    Pre nxt;   // (left unset behind the last tile: never read)
    if (t + stride < tend) nxt = prefetch(d1, b_slot0, b_slot1);
    const int e = d0.e0 + (own ? tid : 0);
    T         pv[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (STAGE > 1) {
#pragma unroll
      for (int k = 0; k < 5; k++) pv[k] = prev.p[k][e];
    }
    const T volume = vol[e];
    // second chunk of this tile's face lists (3D AMR): requested with the previous state, needed just before it
    uint4 ell1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    const size_t erow = static_cast<size_t>(d0.el0 + (own ? tid : 0)) * P.ell_width;
    if (ELLC > 1) ell1 = *reinterpret_cast<const uint4*>(P.ell + erow + 8);
    // ---- phase 1: the tile's own + halo elements -> LDS records ---------------------------------------------
    if (a0) {
      T w[NW];
      if (KIND == 0) {
        prim_words<T>(cur.s0, w, lt);
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = cur.s0[k];
      }
      rec_store<T, NW>(pe + tid * REC, w);
    }
    if (a1) {
      T w[NW];
      if (KIND == 0) {
        prim_words<T>(cur.s1, w, lt);
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = cur.s1[k];
      }
      rec_store<T, NW>(pe + (tid + 256) * REC, w);
    }
    // results of the previous tile: stored here, behind the iteration's first wait (issued before it they would be
    // waited for together with `cur`: vmcnt retires in order)
    if (res_e >= 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) out.p[k][res_e] = res[k];
    }
    __syncthreads();   // records (and, the first time, the dictionary) visible; every wave is done with the previous
                       // tile's flux buffer (two barriers per tile: this one and the one between faces and gather)

    // ---- phase 2: the lane's two faces (faces tid and tid + 256 of the tile) -------------------------------------
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
      const bool     valid = tid + 256 * it < d0.nf;
      const uint32_t lr    = it == 0 ? cur.lr0 : cur.lr1;
      const unsigned graw  = it == 0 ? cur.gi0 : cur.gi1;            // dictionary row | direction code << 13
      const int      gi    = 3 * static_cast<int>(graw & 0x1FFFu);
      const int      orig  = it == 0 ? cur.orig0 : cur.orig1;
      if (valid) {
        const int  l = lr & 0xFFFFu, r16 = lr >> 16;
        const bool wall = r16 == 0xFFFFu;
        const int  r = wall ? l : r16;
        T          g[5], spd = T(0), wl[NW], wr[NW];
        rec_load<T, NW>(pe + l * REC, wl);
        rec_load<T, NW>(pe + r * REC, wr);
        if (KIND == 0) {
          Prim<T> L, R;
          words_prim<T>(wl, L);
          words_prim<T>(wr, R);
          // The tile's faces are ordered by direction inside each block of 256 (tile_plan.cpp), so the wavefront's
          // active lanes usually share one axis-aligned normal s * e_axis. Its frame (face_basis) consists of 0 and +-1:
          // selecting components gives the same values as the general rotation (flux_math.hpp: kepes_axis_fixed),
          // without the 27 multiply-adds. One copy of the flux itself; the arms below only route registers (the empty
          // asm statements keep them real branches -- as selects they would cost what they save).
          const int  code   = static_cast<int>(graw >> 13);
          const int  wcode  = __builtin_amdgcn_readfirstlane(code);
#ifdef T8GPU_EXP_NOAXIS   // experiment builds only: always the general rotation
          const bool shared = false;
#else
          const bool shared = wcode < 6 && __all(code == wcode);
#endif
          const T    sg     = (wcode & 1) ? T(1) : T(-1);
          T          uL, vL, wL, uR, vR, wR, area, n[3], t1[3], t2[3];
          if (shared) {
            area = reinterpret_cast<const T*>(gt + gi)[3];
            if ((wcode >> 1) == 0) {
              asm volatile("");
              uL = sg * L.vx; vL = -(sg * L.vz); wL = L.vy;
              uR = sg * R.vx; vR = -(sg * R.vz); wR = R.vy;
            } else if ((wcode >> 1) == 1) {
              asm volatile("");
              uL = sg * L.vy; vL = sg * L.vx; wL = -L.vz;
              uR = sg * R.vy; vR = sg * R.vx; wR = -R.vz;
            } else {
              asm volatile("");
              uL = sg * L.vz; vL = sg * L.vy; wL = -L.vx;
              uR = sg * R.vz; vR = sg * R.vy; wR = -R.vx;
            }
          } else {
            const V4 gm = gt[gi], b1 = gt[gi + 1], b2 = gt[gi + 2];
            n[0] = gm.x; n[1] = gm.y; n[2] = gm.z; area = gm.w;
            t1[0] = b1.x; t1[1] = b1.y; t1[2] = b1.z;
            t2[0] = b2.x; t2[1] = b2.y; t2[2] = b2.z;
            uL = dot3<T>(L.vx, L.vy, L.vz, n);      // (the rounding sequence of kepes_prim, flux_math.hpp)
            vL = dot3<T>(L.vx, L.vy, L.vz, t1);
            wL = dot3<T>(L.vx, L.vy, L.vz, t2);
            uR = dot3<T>(R.vx, R.vy, R.vz, n);
            vR = dot3<T>(R.vx, R.vy, R.vz, t1);
            wR = dot3<T>(R.vx, R.vy, R.vz, t2);
          }
          if (wall) {   // reflective wall: the right state is the mirror image of the left one (kernels.cu:371-375)
            uR = -uL;
            vR = vL;
            wR = wL;
          }
          T f[5];
#ifdef T8GPU_EXP_NOMATH
          f[0] = L.rho + R.rho + uL; f[1] = vL + wL + uR; f[2] = vR + wR + L.beta + R.beta; f[3] = L.lrho + R.lrho;
          f[4] = L.p + R.p + L.lbeta + R.lbeta + L.v0 + R.v0 + area;
          spd = f[0];
#else
          kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, spd);
#endif
          g[0] = f[0];
          g[4] = f[4];
          if (shared) {
            if ((wcode >> 1) == 0) {
              asm volatile("");
              g[1] = sg * f[1]; g[2] = f[3]; g[3] = -(sg * f[2]);
            } else if ((wcode >> 1) == 1) {
              asm volatile("");
              g[1] = sg * f[2]; g[2] = sg * f[1]; g[3] = -f[3];
            } else {
              asm volatile("");
              g[1] = -f[3]; g[2] = sg * f[2]; g[3] = sg * f[1];
            }
          } else {
            g[1] = t8_fma(f[3], t2[0], t8_fma(f[2], t1[0], f[1] * n[0]));
            g[2] = t8_fma(f[3], t2[1], t8_fma(f[2], t1[1], f[1] * n[1]));
            g[3] = t8_fma(f[3], t2[2], t8_fma(f[2], t1[2], f[1] * n[2]));
          }
        } else {
          const V4 gm = gt[gi], b1 = gt[gi + 1], b2 = gt[gi + 2];
          const T  n[3] = {gm.x, gm.y, gm.z}, t1[3] = {b1.x, b1.y, b1.z}, t2[3] = {b2.x, b2.y, b2.z};
          hll_face<T>(wl, wr, wall, n, t1, t2, gm.w, g, spd, KIND == 2);
        }
        if (orig >= 0) speed[orig] = spd;
#pragma unroll
        for (int k = 0; k < 5; k++) ff[k * 512 + tid + 256 * it] = g[k];
      }
    }
    __syncthreads();
    // ---- phase 3: each owned element sums its faces in list order -------------------------------------------------
    T acc[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (own) {
      ell_gather<T>(cur.ell, ff, acc);
      if (ELLC > 1) {
        ell_gather<T>(ell1, ff, acc);
        // a third chunk only where an element has more than 15 faces (rare: fetched here, other waves cover it)
        if (ELLC > 2 && P.ell_width > 16 && (ell1.w >> 16) != 0xFFFFu)
          ell_gather<T>(*reinterpret_cast<const uint4*>(P.ell + erow + 16), ff, acc);
      }
    }

    // ---- RK stage (ssp_runge_kutta.inl:30-99); stored at the top of the next iteration --------------------------
    const T scale = dt / volume;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      res[k] = rk_stage_update<T, STAGE>(pv[k], cur.s0[k], scale, acc[k]);
    }
    res_e = own ? e : -1;

    cur = nxt;
    d0  = d1;
    d1  = d2;
    b_slot0 = c_slot0;
    b_slot1 = c_slot1;
  }
  if (res_e >= 0) {
#pragma unroll
    for (int k = 0; k < 5; k++) out.p[k][res_e] = res[k];
  }
}

// Does the persistent kernel take a launch of `tile_count` tiles of this plan? (One definition for the launcher below and
// for t8gpu_hip_plain_persistent_accepts, which the host-side tile-cap heuristics ask -- ADVICE r2: they used to repeat a
// part of this test.) On success *lds = dynamic LDS bytes, *resident = workgroups that fill the chip.
template <class T>
bool plain_persistent_accepts(int kind, const T8gpuPlainPlan* plan, int tile_count, size_t* lds_out, int* resident_out) {
  static const bool off = std::getenv("T8GPU_PERSISTENT") && std::getenv("T8GPU_PERSISTENT")[0] == '0';
  const int slots = plan->max_slots;
  // what this kernel takes: the compressed plan with a geometry dictionary small enough for LDS, ELL rows of 8, 16 or 24
  // entries (the second chunk travels with the previous state, a third is fetched where an element has more than 15
  // faces), tiles of <= 256 elements, <= 512 own + halo slots and <= 512 faces
  if (off || !plan->tile_desc || !plan->ell || (plan->ell_width != 8 && plan->ell_width != 16 && plan->ell_width != 24) || !plan->geo_idx || !plan->geo_table || plan->n_geo <= 0 || plan->n_geo > 128 ||
      plan->max_elems > 256 || slots <= 0 || slots > 512 || plan->max_faces > 512)
    return false;
  const int    nw  = kind == 0 ? kPrimWords : 5;
  const int    rec = sizeof(T) == 8 ? (nw > 5 ? 10 : 6) : 12;
  const size_t lds = sizeof(T) * (static_cast<size_t>(5) * 512 + static_cast<size_t>(12) * plan->n_geo + static_cast<size_t>(rec) * slots) +
                     ((sizeof(T) == 8 && kind == 0) ? 2 * kLogTabEntries * sizeof(double) : 0);
  if (lds > 64 * 1024) return false;
  // fp64: the kernel lives on three workgroups per CU. 3 x 53.1 KB (a 3D tile of 376 slots) is nominally inside the 160 KB
  // and yet only two become resident (c5 on 512-face tiles: 3 950 against 4 110 M/s for the one-tile kernel); with a margin
  // the third fits (480-face tiles, 51.7 KB: 4 770). Plans above the margin go to the one-tile kernel.
  if (sizeof(T) == 8 && 3 * lds > static_cast<size_t>(156) * 1024) return false;
  // persistent grid: enough workgroups to fill the chip at the occupancy the kernel reaches, never more than there are
  // tiles. T8GPU_PERSISTENT_WGS overrides the per-CU count (tuning).
  const int        cus        = device_cu_count() > 0 ? device_cu_count() : 256;   // (no device: the build container asking through the C-ABI; MI355X has 256 CUs)
  static const int per_cu_env = env_per_cu("T8GPU_PERSISTENT_WGS");
  // resident workgroups per CU: fp64 166 VGPRs -> 3 waves per SIMD; fp32 ~100 VGPRs and half the LDS -> 5
  const int per_cu = per_cu_env > 0 ? per_cu_env : (sizeof(T) == 8 ? 3 : 5);
  // Worth it when a workgroup walks many tiles (c4: 55) or when there is at most one tile per resident workgroup anyway
  // (then this kernel is simply the cheaper one-tile kernel). In between (c2: 4.4 k tiles, < 6 per workgroup) the
  // exposed first tile of every workgroup and the ragged last round cost more than the pipelining saves: measured
  // 0.047 vs 0.044 ms per stage, so those launches go back to the one-tile kernel.
  // (fp64 KEPES: small launches too -- there the one-tile kernel has its denser register budget: c1 2 270 -> 2 400 M/s)
  const int resident = cus * per_cu;
  if (per_cu_env == 0 && tile_count < 8 * resident && (tile_count > resident || (kind == 0 && sizeof(T) == 8))) return false;
  if (lds_out) *lds_out = lds;
  if (resident_out) *resident_out = resident;
  return true;
}

template <class T>
int plain_persistent_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, FVars<T> prev,
                           FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, hipStream_t stream) {
  size_t lds      = 0;
  int    resident = 0;
  if (!plain_persistent_accepts<T>(kind, plan, tile_count, &lds, &resident)) return -1;
  const int  grid_size = tile_count < resident ? tile_count : resident;
  const dim3 grid(grid_size), block(256);
  note_stage_kernel(tile_count, plan->ell_width == 8 ? "k_plain_persistent<T, K, S, 1>" : "k_plain_persistent<T, K, S, 3>", static_cast<int>(sizeof(T)), kind,
                    stage);
#define T8_PE(K, S, C) hipLaunchKernelGGL((k_plain_persistent<T, K, S, C>), grid, block, lds, stream, *plan, tile_begin, tile_count, prev, mid, out, volume, dt, speed)
#define T8_P(K, S)          \
  do {                      \
    if (plan->ell_width == 8) \
      T8_PE(K, S, 1);       \
    else                    \
      T8_PE(K, S, 3);       \
  } while (0)
#define T8_PS(K)             \
  do {                       \
    if (stage == 1)          \
      T8_P(K, 1);            \
    else if (stage == 2)     \
      T8_P(K, 2);            \
    else                     \
      T8_P(K, 3);            \
  } while (0)
  if (kind == 0)
    T8_PS(0);
  else if (kind == 1)
    T8_PS(1);
  else
    T8_PS(2);
#undef T8_PS
#undef T8_P
#undef T8_PE
  return static_cast<int>(hipGetLastError());
}

template int plain_persistent_stage<float>(int, int, const T8gpuPlainPlan*, int, int, FVars<float>, FVars<float>, FVars<float>,
                                           const float*, float, float*, hipStream_t);
template int plain_persistent_stage<double>(int, int, const T8gpuPlainPlan*, int, int, FVars<double>, FVars<double>, FVars<double>,
                                            const double*, double, double*, hipStream_t);

}  // namespace t8gpu_hip

extern "C" int t8gpu_hip_plain_persistent_accepts(const T8gpuPlainPlan* plan, int flux_kind, int float_size, int tile_count) {
  if (!plan || flux_kind < 0 || flux_kind > 2 || (float_size != 4 && float_size != 8) || tile_count < 0) return 0;
  return float_size == 8 ? (t8gpu_hip::plain_persistent_accepts<double>(flux_kind, plan, tile_count, nullptr, nullptr) ? 1 : 0)
                         : (t8gpu_hip::plain_persistent_accepts<float>(flux_kind, plan, tile_count, nullptr, nullptr) ? 1 : 0);
}
