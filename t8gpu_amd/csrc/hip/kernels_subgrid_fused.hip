// kernels_subgrid_fused.hip -- Subgrid<4,4,4> and Subgrid<4,4>: one launch per RK stage. Three kernels:
//   k_subgrid_fused    the BLOCK kernel: one 64-lane wavefront = one 4x4x4 block (or four 4x4 blocks), described below;
//   k_subgrid_family   3D: one workgroup of eight wavefronts = a 2x2x2 cube of same-level blocks -- inner coarse faces
//                      evaluated once from LDS, the far cells of the outward faces pooled over the wavefronts;
//   k_subgrid_family2  2D: one wavefront = a 2x2 square of blocks, the same within a wavefront.
// The launcher (subgrid_fused_stage) takes the family kernels wherever the host plan found cubes / squares and runs the
// remaining blocks with the block algorithm in the same launch; all three give the same bits.
//
// Replaces, per stage, compute_inner_fluxes + compute_boundary_fluxes + compute_outer_fluxes +
// subgrid::SSP_3RK_stepK (examples/subgrid/solver.inl:166-195) and their flux-plane round trips
// (up to 6 read-modify-writes per cell and variable in the inner kernel, 10 atomics per sub-face in the
// outer one). Lane c owns subcell (i, j, k) = (c & 3, (c >> 2) & 3, c >> 4) of block blockIdx:
//   1. own state -> per-cell primitives, registers + LDS (neighbours: lane +1 / +4 / +16); the far cells
//      of the block's three + faces (same level, or the coarser side of a hanging face: one per surface
//      cell, 48 in all) are fetched by lanes 0..47 and their primitives appended to the same LDS array,
//      i.e. one round of per-cell work for all three faces;
//   2. + faces, x then y then z: EVERY lane evaluates the flux through its +d face. Lanes with coordinate
//      < 3 have their + neighbour in the block, lanes with coordinate 3 sit on the block surface and read
//      the far cell's primitives from the appended part, so these passes run with all 64 lanes busy and
//      no divergence (the wave exchanges the flux through LDS: -own, +lower);
//   3. - faces (same level, coarser neighbour or wall): ONE pass for the three of them, lane = side * 16 +
//      sub-face (lanes 0..47), far cell fetched up front like those of the + faces, fluxes to LDS, the
//      cells with coordinate 0 pick theirs up;
//   4. the faces towards finer blocks (4 sub-faces per cell), four at a time from the block's face list:
//      lane = (face slot, sub-face), far cell gathered through the 2:1 hanging map of kernels.inl:752-758,
//      sub-face fluxes to LDS, every cell picks up the ones that end on it;
//   5. RK stage on the accumulated flux, coalesced store.
// An outer sub-face is evaluated by both blocks that share it, both in the GEOMETRIC orientation (left =
// the block on the low side, normal +e_d; walls: left = the block, outward normal) rather than the one the
// face list happens to store: same arguments, same result, no operand swapping in the + passes, and the
// result does not depend on the listing rule (quirk Q4) or on the partition. There are no atomics, no flux
// planes and the result is bitwise reproducible. The wave runs in lock-step,
// so the reference's unsynchronised LDS reuse has no counterpart here.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "flux_math.hpp"
#include "stage_kernel_note.hpp"
#include "t8gpu_hip.h"

namespace t8gpu_hip {

template <class T>
struct SVars {
  T* p[5];
};

// Element `i` of a state plane. NARROW (every plane of the rank, ghost blocks included, is shorter than 4 GiB -- the
// launcher checks): the byte offset fits 32 bits, so the five planes share ONE offset register and the loads take the
// "uniform base + 32-bit lane offset" form instead of five 64-bit address computations per cell.
template <bool WIDE, class T>
T8_DEV const T& at(const T* p, size_t i) {
  if (WIDE) return p[i];
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + static_cast<uint32_t>(i * sizeof(T)));
}
template <bool WIDE, class T>
T8_DEV T& at(T* p, size_t i) {
  if (WIDE) return p[i];
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(p) + static_cast<uint32_t>(i * sizeof(T)));
}

T8_DEV int sg_xcd_position(int b, int nb) {
  const int q = nb >> 3, rem = nb & 7, x = b & 7, k = b >> 3;
  return x * q + (x < rem ? x : rem) + k;
}

// Everything below is written with shifts instead of small arrays: a runtime-indexed per-lane array
// would be placed in scratch memory.
struct FaceCode {
  int axis, positive, hanging, code;
  T8_DEV int  off(int a) const { return (code >> (4 + 2 * a)) & 3; }   // anchor inside the right block
  T8_DEV bool right() const { return (code >> 12) & 1; }               // joined records: this block is the face's RIGHT side
  T8_DEV int ta() const { return axis == 0 ? 1 : 0; }                  // tangential axes (i, j) of the sub-face grid
  T8_DEV int tb() const { return axis == 2 ? 1 : 2; }
};
T8_DEV FaceCode decode(int code) {
  FaceCode f;
  f.axis     = code & 3;
  f.positive = (code >> 2) & 1;
  f.hanging  = (code >> 3) & 1;
  f.code     = code;
  return f;
}
T8_DEV int cell_coord(int flat, int a) { return (flat >> (2 * a)) & 3; }   // flat = i + 4 j + 16 k

template <class T, int KIND>
struct CellData {  // what a flux evaluation needs from one cell: primitives (KEPES) or the raw state (HLL)
  static constexpr int words = KIND == 0 ? kPrimWords : 5;
  T v[words];
};

template <class T, int KIND>
T8_DEV CellData<T, KIND> cell_from_state(const T s[5]) {
  CellData<T, KIND> c;
#ifdef T8GPU_EXP_NOMATH   // experiment builds only: loads, LDS traffic, barriers and stores as in the product, no arithmetic
  if (KIND == 0) {
    c.v[0] = s[0]; c.v[1] = s[1]; c.v[2] = s[2]; c.v[3] = s[3]; c.v[4] = s[4]; c.v[5] = s[0]; c.v[6] = s[1]; c.v[7] = s[2]; c.v[8] = s[3];
    return c;
  }
#endif
  if (KIND == 0) {
    // fp64: table-driven logarithm, the 2 KB table read from global memory (it stays in the L1 / K$; an LDS copy per
    // one-wave workgroup would cost the kernel a wavefront per SIMD). fp32: hardware log2.
    const Prim<T> q = prim_from_state<T, true>(s, kLogTab);
    c.v[0] = q.rho; c.v[1] = q.vx; c.v[2] = q.vy; c.v[3] = q.vz; c.v[4] = q.p;
    c.v[5] = q.beta; c.v[6] = q.lrho; c.v[7] = q.lbeta; c.v[8] = q.v0;
  } else {
#pragma unroll
    for (int k = 0; k < 5; k++) c.v[k] = s[k];
  }
  return c;
}

// area-scaled xyz flux from L to R through a face with unit normal n
template <class T, int KIND>
T8_DEV void cell_flux(const CellData<T, KIND>& L, const CellData<T, KIND>& R, bool wall, int axis, bool positive, T area, T g[5]) {
#ifdef T8GPU_EXP_NOMATH
  if (KIND == 0) {
    g[0] = L.v[0] + R.v[0] + area; g[1] = L.v[1] + R.v[1] + L.v[5]; g[2] = L.v[2] + R.v[2] + R.v[6]; g[3] = L.v[3] + R.v[3] + L.v[7];
    g[4] = L.v[4] + R.v[4] + R.v[8] + T(axis + (wall ? 1 : 0) + (positive ? 2 : 0));
    return;
  }
#endif
  if (KIND == 0) {
    T spd;
    Prim<T> a, b;
    a.rho = L.v[0]; a.vx = L.v[1]; a.vy = L.v[2]; a.vz = L.v[3]; a.p = L.v[4]; a.beta = L.v[5]; a.lrho = L.v[6]; a.lbeta = L.v[7]; a.v0 = L.v[8];
    b.rho = R.v[0]; b.vx = R.v[1]; b.vy = R.v[2]; b.vz = R.v[3]; b.p = R.v[4]; b.beta = R.v[5]; b.lrho = R.v[6]; b.lbeta = R.v[7]; b.v0 = R.v[8];
    kepes_axis<T>(a, b, wall, axis, positive, area, g, spd);   // subgrid faces are axis-aligned (kernels.inl:717-750 requires it)
  } else {
    T n[3], t1[3], t2[3];
    axis_basis<T>(axis, positive, n, t1, t2);
    T spd;   // Subgrid kernels write no speed estimates (kernels.inl:204)
    hll_face<T>(L.v, R.v, wall, n, t1, t2, area, g, spd, KIND == 2);
  }
}

// What one lane needs for its (face slot, sub-face) of a generic pass; `sf` is the far cell's state.
template <class T>
struct FaceLane {
  bool active, right, wall;
  int  axis, positive, myflat;
  T    area, sf[5];
};

T8_DEV float  area_of(int lo, int, float) { return __int_as_float(lo); }
T8_DEV double area_of(int lo, int hi, double) { return __hiloint2double(hi, lo); }

// lane data of one generic face from its row {other block, code, area}; `live_row` = the row exists
template <class T, int S, bool WIDE>
T8_DEV FaceLane<T> face_lane_from_row(const SVars<T>& src, int4 rec, bool live_row, int si, int sj) {
  FaceLane<T> L;
  L.active = live_row;
  L.right = L.wall = false;
  L.axis = L.positive = L.myflat = 0;
  L.area = T(0);
#pragma unroll
  for (int k = 0; k < 5; k++) L.sf[k] = T(1);
  if (L.active) {
    const int code = rec.y;
    L.right = (code >> 12) & 1;
    L.wall  = rec.x == -1;
    L.area  = area_of(rec.z, rec.w, T(0));
    // cell(i, j) = c0 + ((i >> h) << la) + ((j >> h) << lb) on either side (subgrid_plan.cpp)
    const int la = (code >> 20) & 2, lb = 4 - ((code >> 21) & 2);
    const int hf = (code >> 19) & 1, ho = (code >> 20) & 1;
    L.myflat   = ((code >> 13) & 63) + ((si >> ho) << la) + ((sj >> ho) << lb);
    L.axis     = code & 3;
    L.positive = (code >> 2) & 1;
    if (!L.wall) {
      const size_t far = (size_t)(rec.x + ((si >> hf) << la) + ((sj >> hf) << lb));
#pragma unroll
      for (int k = 0; k < 5; k++) L.sf[k] = at<WIDE>(src.p[k], far);
    }
  }
  return L;
}
template <class T, int S, bool WIDE>
T8_DEV FaceLane<T> load_face_lane(const T8gpuSubgridPlan& P, const SVars<T>& src, int first, int nbf, int idx, int si, int sj) {
  int4 rec = make_int4(0, 0, 0, 0);
  if (idx < nbf) rec = reinterpret_cast<const int4*>(P.bf_rec)[first + idx];   // {other block, code, area}
  return face_lane_from_row<T, S, WIDE>(src, rec, idx < nbf, si, sj);
}

// The +d coarse face of a block (wave-uniform for RANK 3: these are scalar loads, which keeps the
// dependent chain face list -> face record -> far cell short).
template <class T>
struct PlusFace {
  bool on, wall;   // on: listed as ONE coarse face (faces towards finer blocks are in the generic list)
  int  far, hf;    // far cell of sub-face (0, 0); hf: two sub-faces share a far cell (the block is the fine side)
  T    area;
};
template <class T>
T8_DEV PlusFace<T> plus_face(int4 w, bool live) {   // w = the four words of the block record for this face
  PlusFace<T> f;
  f.on = f.wall = false;
  f.far = f.hf = 0;
  f.area = T(0);
  if (live && w.x != -2) {
    f.on   = true;
    f.wall = w.x == -1;
    f.far  = w.x;
    f.hf   = (w.y >> 19) & 1;
    f.area = area_of(w.z, w.w, T(0));
  }
  return f;
}
// state of the far cell behind sub-face (ti, tj) of a + / - face with tangential strides 1 << la, 1 << lb; a wall face
// reads cell `wall_cell` instead (pass -1: nothing)
template <class T, bool WIDE>
T8_DEV void load_far(const SVars<T>& src, bool on, bool wall, int far0, int hf, int la, int lb, int ti, int tj, ptrdiff_t wall_cell,
                     T sf[5]) {
#pragma unroll
  for (int k = 0; k < 5; k++) sf[k] = T(1);
  if (on && !(wall && wall_cell < 0)) {
    const size_t far = wall ? (size_t)wall_cell : (size_t)(far0 + ((ti >> hf) << la) + ((tj >> hf) << lb));
#pragma unroll
    for (int k = 0; k < 5; k++) sf[k] = at<WIDE>(src.p[k], far);
  }
}

// RANK 3: one Subgrid<4,4,4> block per wavefront. RANK 2: four Subgrid<4,4> blocks per wavefront
// (16 lanes each; every index below is relative to the lane's own block).
// Synchronisation between the rounds of ONE wavefront's LDS exchange. As a one-wavefront workgroup the block runs in
// lock-step and the barrier is free; inside the family launch (eight wavefronts per workgroup, each on its own block and
// its own LDS slice, different numbers of generic passes) it must not be a workgroup barrier: a wavefront-scope fence.
template <bool WAVE_ONLY>
T8_DEV void block_sync() {
  if (WAVE_ONLY) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// One wavefront's work on its block(s): `pos_base` = the wavefront's position in the launch (RANK 3: the block record;
// RANK 2: four records), c = lane, pe / xb = the wavefront's LDS slices ([NW][64 + BPW * PF] cells of the block(s) then
// the far cells of their + faces; [5][64] flux exchange buffer).
template <class T, int KIND, int STAGE, int RANK, bool EARLY_PREV, bool WIDE, bool WAVE_ONLY, bool NT = false>
T8_DEV void subgrid_block(const T8gpuSubgridPlan& P, int block_begin, int block_count, int pos_base, int c, const SVars<T>& prev,
                          const SVars<T>& src, const SVars<T>& out, const T* __restrict__ volumes, T dt, T* pe, T* xb) {
  // (pe / xb carry no __restrict__: other lanes write what this lane reads, and a no-alias pointer would let the
  //  compiler keep values across the synchronisation points)
  constexpr int NW  = CellData<T, KIND>::words;
  constexpr int S   = RANK == 3 ? 64 : 16;  // cells per block
  constexpr int SF  = RANK == 3 ? 16 : 4;   // sub-faces per coarse face
  constexpr int BPW = 64 / S;               // blocks per wavefront
  constexpr int PF  = RANK * SF;            // far cells of a block's + faces
  constexpr int PEL = 64 + BPW * PF;        // row length of pe
  const int    base = (c / S) * S, cl = c - base;
  const int    pos  = RANK == 3 ? pos_base : pos_base * BPW + c / S;
  const bool   live = pos < block_count;
  // ONE dependent level: the block's joined record (64 bytes; four scalar loads for RANK 3) names the block, its
  // generic face list and the far block, code and area of its three + faces
  const int4* __restrict__ brec = reinterpret_cast<const int4*>(P.block_rec) + 8 * (size_t)(block_begin + (live ? pos : 0));
  const int4   r0 = brec[0];
  const int    e  = r0.x;
  const int    cc[3] = {cl & 3, (cl >> 2) & 3, RANK == 3 ? cl >> 4 : 0};   // compile-time indices only
  const size_t o = (size_t)e * S + cl;

  T s0[5];
#pragma unroll
  for (int k = 0; k < 5; k++) s0[k] = at<WIDE>(src.p[k], o);
  const T   vol     = volumes[e];
  const int b0      = r0.z;
  const int nbf     = live ? r0.y : 0;
  const T   edge    = (RANK == 3 ? t8_cbrt(vol) : t8_sqrt(vol)) / T(4);
  const T   surface = RANK == 3 ? edge * edge : edge;
  int       npass   = nbf;  // generic passes run until the busiest block of the wave is done
#ifdef T8GPU_EXP_NOGENERIC   // experiment builds only (wrong results): no generic passes / no far cells of the + faces
  npass = 0;
#endif
  if (RANK == 2) {
    npass = max(npass, __shfl_xor(npass, 16, 64));
    npass = max(npass, __shfl_xor(npass, 32, 64));
  }

  // the far cells of the + and - faces are fetched NOW, so that their dependent loads (block record -> far cell)
  // overlap the arithmetic
  const int slot = cl / SF, sub = cl % SF, si = sub & 3, sj = RANK == 3 ? sub >> 2 : 0;
  // previous-step state: requested with everything else (it used to be fetched last to save registers, which made
  // it the fourth dependent round trip of a wavefront's life)
  T pv[5] = {T(0), T(0), T(0), T(0), T(0)};
  if (STAGE > 1 && EARLY_PREV) {
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
  }
  const PlusFace<T> fx = plus_face<T>(brec[1], live), fy = plus_face<T>(brec[2], live),
                    fz = RANK == 3 ? plus_face<T>(brec[3], live) : PlusFace<T>{false, false, 0, 0, T(0)};
  // lane cl < PF fetches far cell `cl % SF` of the block's +(cl / SF) face
  const int  pd = cl / SF, psub = cl % SF;
#ifdef T8GPU_EXP_NOFAR
  const bool p_on = false;
#else
  const bool p_on = cl < PF && (pd == 0 ? fx.on : (pd == 1 ? fy.on : fz.on));
#endif
  // (the tangential strides of side pd: 4 / 16 across x, 1 / 16 across y, 1 / 4 across z)
  const int  pla = pd == 0 ? 2 : 0, plb = pd == 2 ? 2 : 4, pti = psub & 3, ptj = RANK == 3 ? psub >> 2 : 0;
  T          pfar[5];
  load_far<T, WIDE>(src, p_on, pd == 0 ? fx.wall : (pd == 1 ? fy.wall : fz.wall), pd == 0 ? fx.far : (pd == 1 ? fy.far : fz.far),
              pd == 0 ? fx.hf : (pd == 1 ? fy.hf : fz.hf), pla, plb, pti, ptj, -1, pfar);

  // - faces: lane cl < PF owns sub-face `cl % SF` of the block's -(cl / SF) face. A wall lane fetches its OWN cell
  // again: the mirrored flux wants two copies of the same primitives, and this way no select is needed later.
  const PlusFace<T> mx = plus_face<T>(brec[4], live), my = plus_face<T>(brec[5], live),
                    mz = RANK == 3 ? plus_face<T>(brec[6], live) : PlusFace<T>{false, false, 0, 0, T(0)};
  const bool m_on   = cl < PF && (pd == 0 ? mx.on : (pd == 1 ? my.on : mz.on));
  const bool m_wall = pd == 0 ? mx.wall : (pd == 1 ? my.wall : mz.wall);
  const T    m_area = pd == 0 ? mx.area : (pd == 1 ? my.area : mz.area);
  // the cell of this block behind that sub-face: coordinate 0 along the axis, (ti, tj) across
  const int  mflat = (pti << pla) + (ptj << plb);
  T          mfar[5];
  load_far<T, WIDE>(src, m_on, m_wall, pd == 0 ? mx.far : (pd == 1 ? my.far : mz.far), pd == 0 ? mx.hf : (pd == 1 ? my.hf : mz.hf), pla, plb,
              pti, ptj, (size_t)e * S + mflat, mfar);

  const CellData<T, KIND> mine = cell_from_state<T, KIND>(s0);
#pragma unroll
  for (int w = 0; w < NW; w++) pe[(w) * PEL + (c)] = mine.v[w];
#ifdef T8GPU_EXP_NOFAR
  if (false) {
#else
  if (cl < PF) {
#endif
    const CellData<T, KIND> far = cell_from_state<T, KIND>(pfar);
#pragma unroll
    for (int w = 0; w < NW; w++) pe[(w) * PEL + (64 + (c / S) * PF + cl)] = far.v[w];
  }
  block_sync<WAVE_ONLY>();

  T acc[5] = {T(0), T(0), T(0), T(0), T(0)};

  // ---- + faces: inner (kernels.inl:364-533, 2D :554-660) and, on the block surface, the +d coarse face -
#pragma unroll
  for (int d = 0; d < RANK; d++) {
    const int           str = d == 0 ? 1 : (d == 1 ? 4 : 16);
    const PlusFace<T>& pl  = d == 0 ? fx : (d == 1 ? fy : fz);
    const bool          inner = cc[d] < 3;
    // the other cell: next lane's, or (block surface) the far cell with this lane's tangential coordinates
    const int tsub = d == 0 ? cc[1] + 4 * cc[2] : (d == 1 ? cc[0] + 4 * cc[2] : cc[0] + 4 * cc[1]);
    // (a wall mirrors this cell: kernels.inl:913-1107 / compute_boundary_fluxes)
    const bool wall = !inner && pl.wall;
    const int  oidx = inner ? c + str : (wall ? c : 64 + (c / S) * PF + d * SF + tsub);
    CellData<T, KIND> other;
#pragma unroll
    for (int w = 0; w < NW; w++) other.v[w] = pe[(w) * PEL + (oidx)];
    const T    ar   = inner ? surface : pl.area / T(SF);
    T g[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (inner || pl.on) cell_flux<T, KIND>(mine, other, wall, d, true, ar, g);   // left = this cell, normal +e_d
    block_sync<WAVE_ONLY>();
#pragma unroll
    for (int k = 0; k < 5; k++) xb[(k) * 64 + (c)] = g[k];
    block_sync<WAVE_ONLY>();
#pragma unroll
    for (int k = 0; k < 5; k++) acc[k] -= g[k];
    if (cc[d] > 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] += xb[(k) * 64 + (c - str)];
    }
  }

  // ---- - faces (same level, coarser neighbour or wall): ONE pass for the three of them, lane = side * SF + sub-face.
  //      Geometric orientation as everywhere: left = the far cell (low side), normal +e_d; walls: left = this cell,
  //      outward normal -e_d, and the flux LEAVES the cell.
  {
    T g[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (m_on) {
      CellData<T, KIND> here;
#pragma unroll
      for (int w = 0; w < NW; w++) here.v[w] = pe[(w) * PEL + (base + mflat)];
      const CellData<T, KIND> there = cell_from_state<T, KIND>(mfar);
      cell_flux<T, KIND>(there, here, m_wall, pd, !m_wall, m_area / T(SF), g);
      const T sgn = m_wall ? T(-1) : T(1);
#pragma unroll
      for (int k = 0; k < 5; k++) g[k] *= sgn;
    }
    block_sync<WAVE_ONLY>();
#pragma unroll
    for (int k = 0; k < 5; k++) xb[(k) * 64 + (c)] = g[k];
    block_sync<WAVE_ONLY>();
#pragma unroll
    for (int d = 0; d < RANK; d++) {
      const int tsub = d == 0 ? cc[1] + 4 * cc[2] : (d == 1 ? cc[0] + 4 * cc[2] : cc[0] + 4 * cc[1]);
      if (cc[d] == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += xb[(k) * 64 + (base + d * SF + tsub)];
      }
    }
  }

  // ---- remaining coarse faces (towards finer blocks: four sub-faces per surface cell; kernels.inl:664-911) ----------
  for (int p0 = 0; p0 < npass; p0 += 4) {
    T                 g[5] = {T(0), T(0), T(0), T(0), T(0)};
    const FaceLane<T> fl = load_face_lane<T, S, WIDE>(P, src, b0, nbf, p0 + slot, si, sj);
    if (fl.active) {
      CellData<T, KIND> here, there;
#pragma unroll
      for (int w = 0; w < NW; w++) here.v[w] = pe[(w) * PEL + (base + fl.myflat)];
      there = fl.wall ? here : cell_from_state<T, KIND>(fl.sf);
      // geometric orientation: the low-side cell is the left one (walls: this cell, outward normal)
      const bool low = fl.wall || (fl.right != (fl.positive != 0));
      CellData<T, KIND> L, R;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        L.v[w] = low ? here.v[w] : there.v[w];
        R.v[w] = low ? there.v[w] : here.v[w];
      }
      cell_flux<T, KIND>(L, R, fl.wall, fl.axis, fl.wall ? fl.positive != 0 : true, fl.area / T(SF), g);
      // stored with the sign it has for this block's cell: leaves the left cell, enters the right one
      const T sgn = low ? T(-1) : T(1);
#pragma unroll
      for (int k = 0; k < 5; k++) g[k] *= sgn;
    }
    block_sync<WAVE_ONLY>();
#pragma unroll
    for (int k = 0; k < 5; k++) xb[(k) * 64 + (c)] = g[k];
    block_sync<WAVE_ONLY>();
    // every cell collects the sub-face fluxes that end on it, slot by slot (list order)
    for (int s = 0; s < 4; s++) {
      if (p0 + s < nbf) {
        const FaceCode fc    = decode(P.bf_rec[4 * (size_t)(b0 + p0 + s) + 1]);
        const bool     right = fc.right();
        const int  ca = cell_coord(cl, fc.axis), ci = cell_coord(cl, fc.ta()), cj = RANK == 3 ? cell_coord(cl, fc.tb()) : 0;
        const int  q0 = base + SF * s;
        if (!right) {
          if (ca == (fc.positive ? 3 : 0)) {
            const int q = q0 + ci + 4 * cj;
#pragma unroll
            for (int k = 0; k < 5; k++) acc[k] += xb[(k) * 64 + (q)];
          }
        } else if (ca == fc.off(fc.axis)) {
          const int di = ci - fc.off(fc.ta()), dj = RANK == 3 ? cj - fc.off(fc.tb()) : 0;
          if (!fc.hanging) {
            const int q = q0 + di + 4 * dj;
#pragma unroll
            for (int k = 0; k < 5; k++) acc[k] += xb[(k) * 64 + (q)];
          } else if (di >= 0 && di < 2 && dj >= 0 && dj < 2) {
#pragma unroll
            for (int jj = 0; jj < (RANK == 3 ? 2 : 1); jj++)
#pragma unroll
              for (int ii = 0; ii < 2; ii++) {
                const int q = q0 + (2 * di + ii) + (RANK == 3 ? 4 * (2 * dj + jj) : 0);
#pragma unroll
                for (int k = 0; k < 5; k++) acc[k] += xb[(k) * 64 + (q)];
              }
          }
        }
      }
    }
  }

  // ---- RK stage (ssp_runge_kutta.inl:101-221): per-subcell volume = volumes[e] / Subgrid::size --------
  // (the previous step's state is only needed here: fetched late, it does not occupy registers during the
  //  flux passes -- fp64 stays at 128 VGPRs = 4 waves per SIMD; other waves cover the latency)
  if (live) {
    if (STAGE > 1 && !EARLY_PREV) {
#pragma unroll
      for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
    }
    const T scale = dt / (vol / T(S));
#pragma unroll
    for (int k = 0; k < 5; k++) {
      stream_store<NT>(&at<WIDE>(out.p[k], o), rk_stage_update<T, STAGE>(pv[k], s0[k], scale, acc[k]));
    }
  }
}


// RANK 3: one Subgrid<4,4,4> block per wavefront. RANK 2: four Subgrid<4,4> blocks per wavefront
// (16 lanes each; every index is relative to the lane's own block).
template <class T, int KIND, int STAGE, int RANK, bool EARLY_PREV, bool WIDE>
__global__ __launch_bounds__(64) void k_subgrid_fused(T8gpuSubgridPlan P, int block_begin, int block_count, SVars<T> prev,
                                                      SVars<T> src, SVars<T> out, const T* __restrict__ volumes, T dt) {
  constexpr int NW = CellData<T, KIND>::words;
  constexpr int PEL = 64 + (RANK == 3 ? 1 : 4) * RANK * (RANK == 3 ? 16 : 4);
  __shared__ T pe[NW * PEL];  // cells of the wave's block(s), then the far cells of their + faces
  __shared__ T xb[5 * 64];    // flux exchange buffer (+ passes: per cell; generic passes: [slot * SF + sub-face])
  // (RANK 3: the block index is wave-uniform -- blockIdx arithmetic only --, so the per-block loads (volume, face lists,
  //  face records) stay scalar loads and their branches scalar branches)
#ifdef T8GPU_EXP_TILEMOD   // experiment builds only: every wavefront works on one of the first few blocks (no HBM traffic)
  const int pos_base = sg_xcd_position(blockIdx.x, gridDim.x) % T8GPU_EXP_TILEMOD;
#else
  const int pos_base = sg_xcd_position(blockIdx.x, gridDim.x);
#endif
  subgrid_block<T, KIND, STAGE, RANK, EARLY_PREV, WIDE, false>(P, block_begin, block_count, pos_base, threadIdx.x, prev, src, out, volumes,
                                                               dt, pe, xb);
}

// ---------------------------------------------------------------------------------------------------------------
// Family kernel (RANK 3): one workgroup of eight wavefronts = a 2x2x2 cube of consecutive same-level blocks
// (subgrid_plan.cpp), wavefront w = block e0 + w at (w & 1, w >> 1 & 1, w >> 2) of the cube. Against eight runs of the
// block kernel:
//   * the 12 inner coarse faces are evaluated ONCE, by the + pass of the block on their low side, from primitives that
//     are already in LDS (no far-cell loads, no far-cell primitives); the block on the high side picks the flux up;
//   * the far cells of the 24 outward faces are POOLED over the wavefronts: the 192 behind the + faces fill waves 0-2,
//     the 192 behind the - faces waves 3-5 (which also evaluate those faces), instead of two 48-lane rounds per block.
// Per cube: 8 + 6 primitive rounds instead of 24, 24 + 3 flux rounds instead of 32, two workgroup barriers. Every flux
// has the arguments it has in the block kernel and every cell adds its faces in the same order (+x, +y, +z passes, then
// the -x, -y, -z faces), so the two kernels agree bit for bit (tests/test_gpu_subgrid_fused.py).
T8_DEV int fam_compact(int w, int d) { return d == 0 ? w >> 1 : (d == 1 ? (w & 1) | ((w >> 2) << 1) : w & 3); }   // drop bit d
T8_DEV int fam_expand(int j, int d) { return d == 0 ? j << 1 : (d == 1 ? (j & 1) | ((j >> 1) << 2) : j); }      // zero bit at d

// (second launch bound = wavefronts per SIMD the register allocation must allow: 3 workgroups per CU in fp32 (80 VGPRs),
//  2 in fp64 (128 VGPRs; its 66 KB of LDS allow no more))
template <class T, int KIND, int STAGE, bool WIDE, bool NT>
__global__ __launch_bounds__(512, sizeof(T) == 8 ? 4 : 6) void k_subgrid_family(T8gpuSubgridPlan P, SVars<T> prev, SVars<T> src, SVars<T> out,
                                                        const T* __restrict__ volumes, T dt) {
  constexpr int  NW    = CellData<T, KIND>::words;
  constexpr bool EARLY = sizeof(T) == 4;   // fp32: previous-step state requested up front; fp64: fetched last (registers)
  constexpr int FAM_WORDS = NW * 704 + 3 * 5 * 64 + 5 * 512;       // this kernel's arrays
  constexpr int BLK_WORDS = NW * 112 + 5 * 64;                      // one wavefront of the block algorithm
  constexpr int RESTB     = sizeof(T) == 8 ? 4 : 8;                 // leftover blocks per workgroup (fp64: LDS for 4 only)
#ifdef T8GPU_EXP_FAM_PAD   // experiment builds: extra LDS words, to see how many workgroups per CU the kernel really gets
  __shared__ T lds[(FAM_WORDS > RESTB * BLK_WORDS ? FAM_WORDS : RESTB * BLK_WORDS) + T8GPU_EXP_FAM_PAD];
#else
  __shared__ T lds[FAM_WORDS > RESTB * BLK_WORDS ? FAM_WORDS : RESTB * BLK_WORDS];
#endif
  const int tid = threadIdx.x, c = tid & 63;
  const int w   = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the per-block reads below stay scalar
  // The workgroups behind the cubes take the blocks outside every cube (the coarse side of 2:1 interfaces and their
  // siblings: 4 % of c3), eight per workgroup, each wavefront on its own with the block algorithm and its own LDS slice.
  // (As a second launch they cost 8 % of a stage -- ramp-up and tail of a small grid; on a side stream more: the fork /
  //  join events keep the launches from running back to back.)
  if (static_cast<int>(blockIdx.x) >= P.n_families) {
    const int pos = (static_cast<int>(blockIdx.x) - P.n_families) * RESTB + w;
    if (w < RESTB && pos < P.n_rest) {
      T8gpuSubgridPlan R = P;
      R.block_rec        = P.rest_rec;
      T* const mine_lds  = lds + w * BLK_WORDS;
      subgrid_block<T, KIND, STAGE, 3, sizeof(T) == 4, WIDE, true, NT>(R, 0, P.n_rest, pos, c, prev, src, out, volumes, dt, mine_lds,
                                                                  mine_lds + NW * 112);
    }
    return;
  }
  // primitives of the cube's cells [block * 64 + cell], then of the far cells behind the outward + faces
  // [512 + (d * 4 + j) * 16 + sub-face]; fluxes through the inner coarse faces [d][k][j * 16 + sub-face] for the block on
  // their high side; fluxes through the outward - faces [k][(d * 4 + j) * 16 + sub-face]
  T(*const pown)[704]   = reinterpret_cast<T(*)[704]>(lds);
  T(*const sfl)[5][64]  = reinterpret_cast<T(*)[5][64]>(lds + NW * 704);
  // per-wavefront flux exchange slices [8][5][64]; after its + passes a wavefront's slice is free, and waves 3-5 leave the
  // fluxes through the outward - faces in theirs ([k][lane]: slot s of the pooled round = slice 3 + s / 64, lane s % 64)
  T* const xall         = lds + NW * 704 + 3 * 5 * 64;
  T* const xw           = xall + w * 320;
  const int4* __restrict__ frec = reinterpret_cast<const int4*>(P.fam_rec) + 40 * (size_t)sg_xcd_position(blockIdx.x, P.n_families);
  const int    e0 = frec[0].x;
  const int    e  = e0 + w;
  const int    cc[3] = {c & 3, (c >> 2) & 3, c >> 4};
  const size_t o = (size_t)e * 64 + c;

  T s0[5];
#pragma unroll
  for (int k = 0; k < 5; k++) s0[k] = at<WIDE>(src.p[k], o);
  const T vol     = volumes[e];
  const T edge    = t8_cbrt(vol) / T(4);
  const T surface = edge * edge;

  // ---- pooled far cells of the outward faces: waves 0-2 the + faces, waves 3-5 the - faces ---------------------------
  const bool xp = w < 3, xm = w >= 3 && w < 6;
  const int  xs = (xp || xm) ? (xp ? w : w - 3) * 64 + c : 0;   // slot 0..191 = (d * 4 + j) * 16 + sub-face
  const int  xf = xs >> 4, xsub = xs & 15, xd = xf >> 2;  // face row, sub-face, axis
  const int  xb = fam_expand(xf & 3, xd) | (xp ? 1 << xd : 0);   // the block of the cube that owns the face
  const int  xla = xd == 0 ? 2 : 0, xlb = xd == 2 ? 2 : 4, xti = xsub & 3, xtj = xsub >> 2;
  // its cell behind the sub-face: coordinate 3 (+ faces) or 0 (- faces) along the axis
  const int  xcell = (xti << xla) + (xtj << xlb) + (xp ? 3 << (2 * xd) : 0);
  const int4 xrow  = (xp || xm) ? frec[1 + xf + (xm ? 12 : 0)] : make_int4(-2, 0, 0, 0);
  const bool x_on = xrow.x != -2, x_wall = xrow.x == -1;
  T          xst[5];
  // (a wall on the - side re-reads the cell itself, as in the block kernel; a wall on the + side needs no far cell)
  load_far<T, WIDE>(src, x_on && (xm || !x_wall), x_wall, xrow.x, (xrow.y >> 19) & 1, xla, xlb, xti, xtj,
                    (size_t)(e0 + xb) * 64 + xcell, xst);
  T pv[5] = {T(0), T(0), T(0), T(0), T(0)};
  if (STAGE > 1 && EARLY) {
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
  }

  const CellData<T, KIND> mine = cell_from_state<T, KIND>(s0);
#pragma unroll
  for (int q = 0; q < NW; q++) pown[q][tid] = mine.v[q];
  if (xp && x_on && !x_wall) {
    const CellData<T, KIND> far = cell_from_state<T, KIND>(xst);
#pragma unroll
    for (int q = 0; q < NW; q++) pown[q][512 + xs] = far.v[q];
  }
  __syncthreads();

  T acc[5] = {T(0), T(0), T(0), T(0), T(0)};
  // ---- + passes: inner faces, inner coarse faces (sibling on the high side), outward + faces ------------------------
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int  str   = d == 0 ? 1 : (d == 1 ? 4 : 16);
    const bool inner = cc[d] < 3;
    const bool sib   = !((w >> d) & 1);                       // the +d neighbour block is in the cube
    const int  j     = fam_compact(w, d);
    const int4 row   = frec[1 + (sib ? 24 : 0) + d * 4 + j];  // the block's +d coarse face (wave-uniform)
    const bool wall  = !inner && !sib && row.x == -1;
    const int  tsub  = d == 0 ? cc[1] + 4 * cc[2] : (d == 1 ? cc[0] + 4 * cc[2] : cc[0] + 4 * cc[1]);
    // the other cell: next lane's; the sibling's cell with coordinate 0; the pooled far cell; a wall mirrors this cell
    const int  oidx = inner ? tid + str : (sib ? tid + 64 * (1 << d) - 3 * str : (wall ? tid : 512 + (d * 4 + j) * 16 + tsub));
    CellData<T, KIND> other;
#pragma unroll
    for (int q = 0; q < NW; q++) other.v[q] = pown[q][oidx];
    const T ar = inner ? surface : area_of(row.z, row.w, T(0)) / T(16);
    T g[5];
    cell_flux<T, KIND>(mine, other, wall, d, true, ar, g);   // left = this cell, normal +e_d
    // the same face seen from the cell on its high side: through the wavefront's own LDS slice (wavefront-scope fences:
    // measured faster than cross-lane reads, which cost two ds_bpermute per double and their index arithmetic)
    block_sync<true>();
#pragma unroll
    for (int k = 0; k < 5; k++) xw[k * 64 + c] = g[k];
    block_sync<true>();
#pragma unroll
    for (int k = 0; k < 5; k++) acc[k] -= g[k];
    if (cc[d] > 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] += xw[k * 64 + c - str];
    }
    if (!inner && sib) {
#pragma unroll
      for (int k = 0; k < 5; k++) sfl[d][k][j * 16 + tsub] = g[k];
    }
  }
  // ---- outward - faces, pooled: left = the far cell (low side), normal +e_d; walls: left = the cell, outward normal ----
  if (xm && x_on) {
    CellData<T, KIND> here;
#pragma unroll
    for (int q = 0; q < NW; q++) here.v[q] = pown[q][xb * 64 + xcell];
    // (the far cell's primitives only now: five state words instead of nine primitives live during the + passes)
    const CellData<T, KIND> there = cell_from_state<T, KIND>(xst);
    T g[5];
    // A pooled round is 64 consecutive slots = the four faces of ONE axis (slot = (d * 4 + j) * 16 + sub-face, wave w takes
    // slots 64 (w - 3) ...): the axis is wave-uniform, w - 3. Said so, the flux is evaluated with a compile-time axis behind a
    // scalar branch instead of per-lane component selects (round 4; ~40 selects less per - wavefront, c3 within +-0.3 %).
    const T   xarea = area_of(xrow.z, xrow.w, T(0)) / T(16);
    const int xdu   = w - 3;
    if (xdu == 0)
      cell_flux<T, KIND>(there, here, x_wall, 0, !x_wall, xarea, g);
    else if (xdu == 1)
      cell_flux<T, KIND>(there, here, x_wall, 1, !x_wall, xarea, g);
    else
      cell_flux<T, KIND>(there, here, x_wall, 2, !x_wall, xarea, g);
    const T sgn = x_wall ? T(-1) : T(1);
    block_sync<true>();   // (this wavefront's last reads of its exchange slice are behind it)
#pragma unroll
    for (int k = 0; k < 5; k++) xw[k * 64 + c] = g[k] * sgn;
  }
  __syncthreads();
  // ---- every cell with coordinate 0 picks up its -d face: from the sibling's + pass or from the pooled round ----------
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if (cc[d] == 0) {
      const int tsub = d == 0 ? cc[1] + 4 * cc[2] : (d == 1 ? cc[0] + 4 * cc[2] : cc[0] + 4 * cc[1]);
      const int j    = fam_compact(w, d);
      if ((w >> d) & 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += sfl[d][k][j * 16 + tsub];
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += xall[(3 + d) * 320 + k * 64 + j * 16 + tsub];   // slot (d * 4 + j) * 16 + tsub
      }
    }
  }
  // ---- RK stage ---------------------------------------------------------------------------------------------------------
  if (STAGE > 1 && !EARLY) {
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
  }
  const T scale = dt / (vol / T(64));
#pragma unroll
  for (int k = 0; k < 5; k++) stream_store<NT>(&at<WIDE>(out.p[k], o), rk_stage_update<T, STAGE>(pv[k], s0[k], scale, acc[k]));
}

// (Round 4, measured and dropped: a PERSISTENT form of this kernel -- a resident grid of 2 (fp64) / 3 (fp32) workgroups per CU
// walking the cubes of their XCD's share, the next cube's record rows as scalar loads at the top of an iteration, its own state,
// pooled far cells and previous-step state requested after the - round, behind this cube's previous-step loads. Bitwise the
// kernel above (it ran through test_family_kernel_and_block_kernel_agree_bitwise on grids of 3, 21 and 512 workgroups), and
// much slower: c3 fp64 7 910 -> 4 610, fp32 17 200 -> 13 030 M subcell-updates/s. The loop-carried state -- lane constants
// of the pooled rounds, the next cube's rows, its 10 + 10 (+ 10) state words -- does not fit beside the 126 / 72 VGPRs of the
// flux passes: 128 / 80 VGPRs with 80 / 50 dwords of scratch spill and 100 SGPRs with ~160 lane spills, and the reloads sit
// in the in-order vmcnt queue behind the prefetch. Fetching the own state a second time for the RK update (ten registers
// less through the passes) did not remove a single spill. The block kernel's persistent form of round 2 failed the same way.
// Also measured and dropped: `chunk` consecutive cubes per workgroup with NOTHING carried between them (lane constants
// recomputed per cube from an opaque copy of the lane index, which keeps the loop free of spills: 127 / 75 VGPRs) -- chunks of
// 1 / 2 / 3 / 4 / 8 cubes: c3 fp64 7 470 / 7 300 / 7 180 / 7 360 / 7 260, fp32 15 830 / 15 950 / 15 460 / 15 680 / 15 640 M/s on a
// box where this kernel runs at ~7 750 / ~17 000: the loop form alone costs the code 4 - 7 %, and fewer dispatches buy nothing.)

// ---------------------------------------------------------------------------------------------------------------
// Family kernel, RANK 2: one wavefront = a 2x2 square of consecutive same-level Subgrid<4,4> blocks (lanes 16 w .. 16 w + 15
// = block e0 + w at (w & 1, w >> 1)). The same idea as above within one wavefront, so without workgroup barriers: the 4
// inner coarse faces are evaluated once from primitives in LDS, the far cells of the 8 outward faces are pooled (16 behind
// the + faces in lanes 0-15, 16 behind the - faces in lanes 16-31: one half-filled primitive round instead of the block
// kernel's two). Same fluxes, same summation order: bitwise equal to the block kernel. The blocks outside every square run
// behind the squares in the same launch (four per wavefront, the block algorithm).
template <class T, int KIND, int STAGE, bool WIDE, bool NT>
__global__ __launch_bounds__(64) void k_subgrid_family2(T8gpuSubgridPlan P, SVars<T> prev, SVars<T> src, SVars<T> out,
                                                        const T* __restrict__ volumes, T dt) {
  constexpr int  NW        = CellData<T, KIND>::words;
  constexpr bool EARLY     = sizeof(T) == 4;
  constexpr int  FAM_WORDS = NW * 80 + 5 * 64 + 2 * 5 * 8 + 5 * 16;
  constexpr int  BLK_WORDS = NW * 96 + 5 * 64;
  __shared__ T lds[FAM_WORDS > BLK_WORDS ? FAM_WORDS : BLK_WORDS];
  const int c = threadIdx.x;
  if (static_cast<int>(blockIdx.x) >= P.n_families) {   // the blocks outside every square: four per wavefront
    T8gpuSubgridPlan R = P;
    R.block_rec        = P.rest_rec;
    subgrid_block<T, KIND, STAGE, 2, EARLY, WIDE, false, NT>(R, 0, P.n_rest, static_cast<int>(blockIdx.x) - P.n_families, c, prev, src, out,
                                                         volumes, dt, lds, lds + NW * 96);
    return;
  }
  // primitives [64 cells of the square, then 16 far cells behind the outward + faces ((d * 2 + j) * 4 + sub-face)];
  // the wavefront's flux exchange buffer [5][64]; fluxes through the inner coarse faces [d][k][j * 4 + sub-face]; fluxes
  // through the outward - faces [k][(d * 2 + j) * 4 + sub-face]
  T(*const pown)[80] = reinterpret_cast<T(*)[80]>(lds);
  T* const xw        = lds + NW * 80;
  T(*const sfl)[5][8] = reinterpret_cast<T(*)[5][8]>(lds + NW * 80 + 320);
  T(*const mfl)[16]  = reinterpret_cast<T(*)[16]>(lds + NW * 80 + 320 + 80);
  const int4* __restrict__ frec = reinterpret_cast<const int4*>(P.fam_rec) + 16 * (size_t)sg_xcd_position(blockIdx.x, P.n_families);
  const int    e0 = frec[0].x;
  const int    w = c >> 4, cl = c & 15;
  const int    e  = e0 + w;
  const int    cc[2] = {cl & 3, cl >> 2};
  const size_t o = (size_t)e * 16 + cl;

  T s0[5];
#pragma unroll
  for (int k = 0; k < 5; k++) s0[k] = at<WIDE>(src.p[k], o);
  const T vol     = volumes[e];
  const T surface = t8_sqrt(vol) / T(4);   // edge of a subcell = length of an inner face
  // the block's two + faces (area, wall flag): outward rows 0-3 or inner rows 8-11
  const int  jx = w >> 1, jy = w & 1;      // the block's index among those with / without bit d (fam_compact for RANK 2)
  const int4 rowx = frec[1 + ((w & 1) ? 0 : 8) + jx], rowy = frec[1 + ((w & 2) ? 0 : 8) + 2 + jy];

  // ---- pooled far cells of the outward faces: lanes 0-15 the + faces, lanes 16-31 the - faces -------------------------
  const bool xp = c < 16, xm = c >= 16 && c < 32;
  const int  xs = c & 15, xf = xs >> 2, xsub = xs & 3, xd = xf >> 1;   // slot = (d * 2 + j) * 4 + sub-face
  const int  xbk = (xd == 0 ? (xf & 1) << 1 : (xf & 1)) | (xp ? 1 << xd : 0);   // the block of the square that owns the face
  const int  xla = xd == 0 ? 2 : 0;
  const int  xcell = (xsub << xla) + (xp ? 3 << (2 * xd) : 0);   // its cell behind the sub-face
  const int4 xrow  = (xp || xm) ? frec[1 + xf + (xm ? 4 : 0)] : make_int4(-2, 0, 0, 0);
  const bool x_on = xrow.x != -2, x_wall = xrow.x == -1;
  T          xst[5];
  load_far<T, WIDE>(src, x_on && (xm || !x_wall), x_wall, xrow.x, (xrow.y >> 19) & 1, xla, 0, xsub, 0, (size_t)(e0 + xbk) * 16 + xcell,
                    xst);
  T pv[5] = {T(0), T(0), T(0), T(0), T(0)};
  if (STAGE > 1 && EARLY) {
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
  }

  const CellData<T, KIND> mine = cell_from_state<T, KIND>(s0);
#pragma unroll
  for (int q = 0; q < NW; q++) pown[q][c] = mine.v[q];
  CellData<T, KIND> there;
#pragma unroll
  for (int q = 0; q < NW; q++) there.v[q] = T(1);
  if (x_on && (xm || !x_wall)) {
    there = cell_from_state<T, KIND>(xst);
    if (xp) {
#pragma unroll
      for (int q = 0; q < NW; q++) pown[q][64 + xs] = there.v[q];
    }
  }
  __syncthreads();

  T acc[5] = {T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int d = 0; d < 2; d++) {
    const int  str   = d == 0 ? 1 : 4;
    const bool inner = cc[d] < 3;
    const bool sib   = !((w >> d) & 1);
    const int  j     = d == 0 ? jx : jy;
    const int4 row   = d == 0 ? rowx : rowy;
    const bool wall  = !inner && !sib && row.x == -1;
    const int  tsub  = cc[1 - d];
    const int  oidx  = inner ? c + str : (sib ? c + 16 * (1 << d) - 3 * str : (wall ? c : 64 + (d * 2 + j) * 4 + tsub));
    CellData<T, KIND> other;
#pragma unroll
    for (int q = 0; q < NW; q++) other.v[q] = pown[q][oidx];
    const T ar = inner ? surface : area_of(row.z, row.w, T(0)) / T(4);
    T g[5];
    cell_flux<T, KIND>(mine, other, wall, d, true, ar, g);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) xw[k * 64 + c] = g[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) acc[k] -= g[k];
    if (cc[d] > 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] += xw[k * 64 + c - str];
    }
    if (!inner && sib) {
#pragma unroll
      for (int k = 0; k < 5; k++) sfl[d][k][j * 4 + tsub] = g[k];
    }
  }
  if (xm && x_on) {   // outward - faces: left = the far cell (low side), normal +e_d; walls: left = the cell, outward normal
    CellData<T, KIND> here;
#pragma unroll
    for (int q = 0; q < NW; q++) here.v[q] = pown[q][xbk * 16 + xcell];
    T g[5];
    cell_flux<T, KIND>(there, here, x_wall, xd, !x_wall, area_of(xrow.z, xrow.w, T(0)) / T(4), g);
    const T sgn = x_wall ? T(-1) : T(1);
#pragma unroll
    for (int k = 0; k < 5; k++) mfl[k][xs] = g[k] * sgn;
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 2; d++) {
    if (cc[d] == 0) {
      const int tsub = cc[1 - d], j = d == 0 ? jx : jy;
      if ((w >> d) & 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += sfl[d][k][j * 4 + tsub];
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] += mfl[k][(d * 2 + j) * 4 + tsub];
      }
    }
  }
  if (STAGE > 1 && !EARLY) {
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at<WIDE>(prev.p[k], o));
  }
  const T scale = dt / (vol / T(16));
#pragma unroll
  for (int k = 0; k < 5; k++) stream_store<NT>(&at<WIDE>(out.p[k], o), rk_stage_update<T, STAGE>(pv[k], s0[k], scale, acc[k]));
}

template <class T, class V>
SVars<T> smk(const V& v) {
  SVars<T> o;
  for (int k = 0; k < 5; k++) o.p[k] = v.p[k];
  return o;
}

template <class T, class V>
int subgrid_fused_stage(int kind, int stage, const T8gpuSubgridPlan* plan, int block_begin, int block_count, V prev, V mid,
                        V out, const T* volumes, T dt, void* stream) {
  if (!plan || kind < 0 || kind > 2 || stage < 1 || stage > 3 || (plan->rank != 2 && plan->rank != 3))
    return static_cast<int>(hipErrorInvalidValue);
  if (block_begin < 0 || block_count < 0 || block_begin + block_count > plan->num_elements) return static_cast<int>(hipErrorInvalidValue);
  if (block_count == 0) return 0;
  stage_kernel_note_reset();
  hipStream_t s = static_cast<hipStream_t>(stream);
  // fp32 requests the previous-step state up front (one round trip less per wavefront); fp64 keeps fetching it last:
  // it is bound by DP instruction issue and the 10 extra registers would cost it a wavefront per SIMD (measured both ways).
  constexpr bool early = sizeof(T) == 4;
  // 32-bit byte offsets into the state planes when every plane (owned + ghost blocks) is shorter than 4 GiB
  const int64_t cells = static_cast<int64_t>(plan->n_blocks_addressed) * (plan->rank == 3 ? 64 : 16);
  static const bool force_wide = std::getenv("T8GPU_SG_WIDE") && std::getenv("T8GPU_SG_WIDE")[0] == '1';   // (tests)
  const bool    wide  = force_wide || plan->n_blocks_addressed <= 0 || cells * static_cast<int64_t>(sizeof(T)) >= (int64_t(1) << 32);
  // one launch of the block kernel over records `pl.block_rec[begin, begin + count)`
  auto blocks = [&](const T8gpuSubgridPlan& pl, int begin, int count) {
    const dim3 grid(pl.rank == 3 ? count : (count + 3) / 4), block(64);
    {
      char pat[96];
      std::snprintf(pat, sizeof(pat), "k_subgrid_fused<T, K, S, %d, %s, %s>", pl.rank == 3 ? 3 : 2, early ? "true" : "false", wide ? "true" : "false");
      note_stage_kernel(count, pat, static_cast<int>(sizeof(T)), kind, stage);
    }
#define T8_SG(K, S, R)                                                                                                     \
  do {                                                                                                                     \
    if (wide)                                                                                                              \
      hipLaunchKernelGGL((k_subgrid_fused<T, K, S, R, early, true>), grid, block, 0, s, pl, begin, count, smk<T>(prev),   \
                         smk<T>(mid), smk<T>(out), volumes, dt);                                                           \
    else                                                                                                                   \
      hipLaunchKernelGGL((k_subgrid_fused<T, K, S, R, early, false>), grid, block, 0, s, pl, begin, count, smk<T>(prev),  \
                         smk<T>(mid), smk<T>(out), volumes, dt);                                                           \
  } while (0)
#define T8_SGR(K, S)  \
  do {                \
    if (pl.rank == 3) \
      T8_SG(K, S, 3); \
    else              \
      T8_SG(K, S, 2); \
  } while (0)
    if (kind == 0) {
      if (stage == 1) T8_SGR(0, 1); else if (stage == 2) T8_SGR(0, 2); else T8_SGR(0, 3);
    } else if (kind == 1) {
      if (stage == 1) T8_SGR(1, 1); else if (stage == 2) T8_SGR(1, 2); else T8_SGR(1, 3);
    } else {
      if (stage == 1) T8_SGR(2, 1); else if (stage == 2) T8_SGR(2, 2); else T8_SGR(2, 3);
    }
#undef T8_SGR
#undef T8_SG
  };
  // A launch that covers the whole plan of a 3D mesh: 2x2x2 cubes of same-level blocks through the family kernel, the
  // other blocks through the block kernel (T8GPU_SG_FAMILY=0: every block through the block kernel -- same bits).
  // (T8GPU_SG_FAMILY=0: every block through the block kernel -- same bits. Measured on c3: KEPES fp32 +6 %, fp64 +8 %,
  //  HLL fp32 +9 %, fp64 +3 %, HLLC fp32 +5 %.)
  static const bool fam_off = std::getenv("T8GPU_SG_FAMILY") && std::getenv("T8GPU_SG_FAMILY")[0] == '0';
  // The cubes consist of deep interior blocks only, and rest_rec lists the other blocks in block_order order, so with
  // nf = 8 * n_families: positions [0, n_deep) = the cubes + rest_rec[0, n_deep - nf), and position p >= n_deep = rest_rec[p - nf].
  // Launches that are the whole plan or exactly its first class take the family kernel; launches inside the later
  // classes read their block records from rest_rec; anything else goes through block_rec.
  const int  nf       = (plan->rank == 3 ? 8 : 4) * plan->n_families;
  const bool have_fam = !fam_off && plan->n_families > 0 && plan->fam_rec && plan->rest_rec &&
                        plan->n_rest == plan->num_elements - nf && plan->n_deep_blocks >= nf;
  const bool families = have_fam && block_begin == 0 && (block_count == plan->num_elements || block_count == plan->n_deep_blocks);
  if (!families) {
    if (have_fam && block_begin >= plan->n_deep_blocks) {
      T8gpuSubgridPlan rest = *plan;
      rest.block_rec        = plan->rest_rec;
      blocks(rest, block_begin - nf, block_count);
    } else {
      blocks(*plan, block_begin, block_count);
    }
    return static_cast<int>(hipGetLastError());
  }
  // non-temporal stage results / previous-state loads where the stage's planes are a stream for the caches (flux_math.hpp)
  const bool nt = stream_hint(cells, sizeof(T));
  const int n_rest_here = block_count - nf;   // the leading rows of rest_rec that belong to this launch
  T8gpuSubgridPlan fam = *plan;
  fam.n_rest           = n_rest_here;
  // RANK 3: workgroups of eight wavefronts, `restb` leftover blocks per workgroup behind the cubes; RANK 2: one wavefront
  // per square, four leftover blocks per wavefront
  const int  restb = plan->rank == 3 ? (sizeof(T) == 8 ? 4 : 8) : 4;   // (k_subgrid_family: RESTB)
  const dim3 grid(plan->n_families + (n_rest_here + restb - 1) / restb), block(plan->rank == 3 ? 512 : 64);
  {
    char pat[96];
    std::snprintf(pat, sizeof(pat), "%s<T, K, S, %s, %s>", plan->rank == 3 ? "k_subgrid_family" : "k_subgrid_family2", wide ? "true" : "false", nt ? "true" : "false");
    note_stage_kernel(block_count, pat, static_cast<int>(sizeof(T)), kind, stage);
  }
#define T8_FMN(K, S, N)                                                                                                         \
  do {                                                                                                                          \
    if (plan->rank == 3 && wide)                                                                                                \
      hipLaunchKernelGGL((k_subgrid_family<T, K, S, true, N>), grid, block, 0, s, fam, smk<T>(prev), smk<T>(mid), smk<T>(out),   \
                         volumes, dt);                                                                                          \
    else if (plan->rank == 3)                                                                                                   \
      hipLaunchKernelGGL((k_subgrid_family<T, K, S, false, N>), grid, block, 0, s, fam, smk<T>(prev), smk<T>(mid), smk<T>(out),  \
                         volumes, dt);                                                                                          \
    else if (wide)                                                                                                              \
      hipLaunchKernelGGL((k_subgrid_family2<T, K, S, true, N>), grid, block, 0, s, fam, smk<T>(prev), smk<T>(mid), smk<T>(out),  \
                         volumes, dt);                                                                                          \
    else                                                                                                                        \
      hipLaunchKernelGGL((k_subgrid_family2<T, K, S, false, N>), grid, block, 0, s, fam, smk<T>(prev), smk<T>(mid), smk<T>(out), \
                         volumes, dt);                                                                                          \
  } while (0)
#define T8_FM(K, S)        \
  do {                     \
    if (nt)                \
      T8_FMN(K, S, true);  \
    else                   \
      T8_FMN(K, S, false); \
  } while (0)
  if (kind == 0) {
    if (stage == 1) T8_FM(0, 1); else if (stage == 2) T8_FM(0, 2); else T8_FM(0, 3);
  } else if (kind == 1) {
    if (stage == 1) T8_FM(1, 1); else if (stage == 2) T8_FM(1, 2); else T8_FM(1, 3);
  } else {
    if (stage == 1) T8_FM(2, 1); else if (stage == 2) T8_FM(2, 2); else T8_FM(2, 3);
  }
#undef T8_FM
#undef T8_FMN
  return static_cast<int>(hipGetLastError());
}

}  // namespace t8gpu_hip

extern "C" {
int t8gpu_hip_subgrid_fused_stage_f32(int kind, int stage, const T8gpuSubgridPlan* plan, int block_begin, int block_count,
                                      T8gpuVars_f32 prev, T8gpuVars_f32 mid, T8gpuVars_f32 out, const float* volumes,
                                      float dt, void* stream) {
  return t8gpu_hip::subgrid_fused_stage<float>(kind, stage, plan, block_begin, block_count, prev, mid, out, volumes, dt, stream);
}
int t8gpu_hip_subgrid_fused_stage_f64(int kind, int stage, const T8gpuSubgridPlan* plan, int block_begin, int block_count,
                                      T8gpuVars_f64 prev, T8gpuVars_f64 mid, T8gpuVars_f64 out, const double* volumes,
                                      double dt, void* stream) {
  return t8gpu_hip::subgrid_fused_stage<double>(kind, stage, plan, block_begin, block_count, prev, mid, out, volumes, dt, stream);
}
}
