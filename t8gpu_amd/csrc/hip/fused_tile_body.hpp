// fused_tile_body.hpp -- the one-tile-per-workgroup fused stage of a GENERIC tile (software-pipelined form), as a device
// function: k_plain_fused_p (kernels_fused.hip) is a thin kernel around it, and the mixed launch of kernels_fused_patch.hip
// runs it in the workgroups behind the patch workgroups. Reference: examples/compressible_euler/kernels.cu:135-469 +
// ssp_runge_kutta.inl:30-99.
#ifndef T8GPU_HIP_FUSED_TILE_BODY_HPP
#define T8GPU_HIP_FUSED_TILE_BODY_HPP

#include "fused_common.hpp"

namespace t8gpu_hip {

// ---- software-pipelined variant --------------------------------------------------------------------
// Same three phases, but EVERY global load of the tile is issued in the prologue, before the first
// barrier: states of own + halo elements (2 per lane), the lane's two faces (packed indices, geometry
// index, original id) and the lane's element row (previous state, volume, first 8 face-list entries).
// The generic kernel above exposes three dependent memory latencies per tile (one per phase); here
// they overlap, and the own state stays in registers for the RK stage. Needs tiles of <= 256 elements,
// <= 512 own+halo elements and <= 512 faces (tile_plan.cpp guarantees it with the default caps).
// Adds the entries of one ELL chunk whose face lies in pass `pass` (faces [256*pass, 256*pass+256) of the
// tile are in the LDS flux buffer during that pass). Entries are in ascending face order, so visiting
// pass 0 then pass 1 keeps the summation order of the single-pass form.
template <class T>
T8_DEV bool ell_accumulate(uint4 w, int pass, const T* __restrict__ ff, T acc[5]) {
  const unsigned ent[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};
#pragma unroll
  for (int j = 0; j < 8; j++) {
    // face index = (pass << 8) | slot; the padding 0xFFFF has pass field 127 and matches no pass
    if (((ent[j] & 0x7FFFu) >> 8) == static_cast<unsigned>(pass)) {
      const T* p   = ff + (ent[j] & 255u);
      const T  wgt = (ent[j] & 0x8000u) ? T(1) : T(-1);
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] = __builtin_fma(wgt, p[k * 256], acc[k]);
    }
  }
  return (w.w >> 16) == 0xFFFFu;   // the row ends in this chunk
}

template <class T>
T8_DEV void store_prim(T* pe, int LE, int i, const T s[5], const double* logtab) {
#ifdef T8GPU_EXP_NOMATH    // experiment builds only: same loads, LDS traffic, barriers and stores, (almost) no arithmetic
  Prim<T> q;
  q.rho = s[0]; q.vx = s[1]; q.vy = s[2]; q.vz = s[3]; q.p = s[4]; q.beta = s[0]; q.lrho = s[1]; q.lbeta = s[2]; q.v0 = s[3];
#else
  const Prim<T> q = prim_from_state<T, sizeof(T) == 8>(s, logtab);
#endif
  pe[0 * LE + i] = q.rho;
  pe[1 * LE + i] = q.vx;
  pe[2 * LE + i] = q.vy;
  pe[3 * LE + i] = q.vz;
  pe[4 * LE + i] = q.p;
  pe[5 * LE + i] = q.beta;
  pe[6 * LE + i] = q.lrho;
  pe[7 * LE + i] = q.lbeta;
  pe[8 * LE + i] = q.v0;
}

template <class T>
T8_DEV void load_prim(const T* pe, int LE, int i, Prim<T>& q) {
  q.rho = pe[0 * LE + i]; q.vx = pe[1 * LE + i]; q.vy = pe[2 * LE + i]; q.vz = pe[3 * LE + i]; q.p = pe[4 * LE + i];
  q.beta = pe[5 * LE + i]; q.lrho = pe[6 * LE + i]; q.lbeta = pe[7 * LE + i]; q.v0 = pe[8 * LE + i];
}

// MAXP = 2: at most 512 faces per tile, both passes' face records loaded in the prologue (2D meshes).
// MAXP = 4: up to 1024 faces per tile (3D meshes: a 256-element tile has ~3 faces per element plus its
// surface); the record of pass p + 1 is fetched at the top of pass p, so two are live at any time.
// SCATTER = true is the accumulation the project brief sketches: every face lane adds -F / +F to per-element
// accumulators in LDS with ds_add_f32 / ds_add_f64 (no ELL rows, no gather, one barrier after the last pass).
// Kept as a measured alternative (T8GPU_LDS_SCATTER=1): the order of the additions is not fixed, so the
// result is no longer bitwise reproducible, and the default gather is faster.
// DENSE: register budget for 4 (fp64) / 5 (fp32) workgroups per CU. The fp64 KEPES kernel then spills ~20 registers and
// still gains 7 % where the tiles leave the LDS room for the fourth workgroup (2D meshes: c2 7 640 -> 8 150 M/s, the
// one-tile kernel on c4 7 520 -> 8 090); where they do not (3D tiles: ~39 KB) the spills cost 9 % (c5, c5u); HLL / HLLC
// spill more and lose 24 %, fp32 neither gains nor loses. The launcher takes it for fp64 KEPES tiles of <= 36 KB (2D
// meshes: 35 KB; 3D tiles are 37 KB and lose 7 % with it even though the fourth workgroup then fits -- their wavefronts mix
// face directions and run both arms of the axis path under the tighter budget). The DENSE kernel reads the logarithm
// table from global memory instead of an LDS copy (c2: +1 %).
template <class T, int KIND, int STAGE, bool DICT, int MAXP, bool SCATTER = false, bool DENSE = false>
T8_DEV void plain_tile_body(const T8gpuPlainPlan& P, int pos, const FVars<T>& prev, const FVars<T>& src, const FVars<T>& out,
                            const T* __restrict__ vol, T dt, T* __restrict__ speed) {
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  using V4 = typename vec4<T>::type;
  T* const      lds = reinterpret_cast<T*>(lds_raw);
  constexpr int NW  = KIND == 0 ? kPrimWords : 5;
  const int     LE  = P.max_slots > 0 ? P.max_slots : P.max_elems + P.max_halo;
  T* const      pe  = lds;
  T* const      ff  = lds + (size_t)NW * LE;  // [5][256]: one pass of 256 faces at a time
  constexpr bool kTab = sizeof(T) == 8 && KIND == 0;   // fp64 KEPES: table-driven logarithms, table behind the flux buffer
  // (DENSE: the table is read from global memory -- 2 KB of LDS less is what lets a 3D tile's fourth workgroup fit)
  double* const lt_lds = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ff + 5 * 256) + 15) & ~uintptr_t(15));   // 16-byte rows
  const double* const lt = DENSE ? kLogTab : lt_lds;

  const int tile = P.tile_order[pos];   // pos: position in tile_order (execution order)
  const int e0 = P.elem_off[tile], ne = P.elem_off[tile + 1] - e0;
  const int h0 = P.halo_off[tile], nh = P.halo_off[tile + 1] - h0;
  const int f0 = P.face_off[tile], nf = P.face_off[tile + 1] - f0;
  const int tid = threadIdx.x;

  // ---- prologue: all global loads of the tile ----------------------------------------------------
  const int  i1 = tid + 256;
  const bool own = tid < ne, a0 = tid < ne + nh, a1 = i1 < ne + nh;
  const int  slot0 = own ? e0 + tid : (a0 ? P.halo_ids[h0 + (tid - ne)] : e0);
  const int  slot1 = a1 ? P.halo_ids[h0 + (i1 - ne)] : e0;
  T          s0[5], s1[5];
  if (P.ghost_buf) {   // (wave-uniform: launches of the multi-rank driver's ghost-reading class; t8gpu_hip.h "ghost window")
#pragma unroll
    for (int k = 0; k < 5; k++) s0[k] = ghost_window_load<T>(P, src, slot0, k);
#pragma unroll
    for (int k = 0; k < 5; k++) s1[k] = ghost_window_load<T>(P, src, slot1, k);
  } else {
#pragma unroll
    for (int k = 0; k < 5; k++) s0[k] = src.p[k][slot0];
#pragma unroll
    for (int k = 0; k < 5; k++) s1[k] = src.p[k][slot1];
  }
  // one pass's worth of the lane's face: packed slots, geometry (row index or the row itself), original id
  struct FaceIn {
    bool     valid;
    uint32_t lr;
    V4       gm;
    int      gi, orig, code;   // code: direction code of the normal (tile_plan.cpp: 0..5 = -x +x -y +y -z +z, 7 = oblique)
  };
  auto load_face = [&](int pass) {
    FaceIn f;
    const int i = tid + 256 * pass;
    f.valid = i < nf;
    const int j = f0 + (f.valid ? i : 0);
    f.lr = P.face_lr[j];
    f.gm = V4{};
    f.gi = 0;
    f.code = 7;
    if (DICT) {  // only the 2-byte row index travels with the face; the (cache-resident) row is read in phase 2
      const unsigned graw = P.geo_idx[j];   // dictionary row | direction code << 13
      f.gi   = 3 * static_cast<int>(graw & 0x1FFFu);
      f.code = static_cast<int>(graw >> 13);
    } else
      f.gm = reinterpret_cast<const V4*>(P.face_geo)[j];
    f.orig = speed ? P.face_orig[j] : -1;
    return f;
  };
  FaceIn fin[MAXP + 1];
  fin[0] = load_face(0);
  fin[1] = load_face(1);
  const int e = e0 + (own ? tid : 0);
  T         pv[5] = {T(0), T(0), T(0), T(0), T(0)}, volume = T(1);   // fetched behind the last flux pass (register budget)
  // (ELL rows exist for the elements of generic tiles only: the tile's first row is word 6 of its descriptor)
  const int ell_row = P.tile_desc[8 * static_cast<size_t>(pos) + 6] + (own ? tid : 0);
  const uint4* __restrict__ ellrow = reinterpret_cast<const uint4*>(P.ell + (size_t)ell_row * P.ell_width);
  const uint4 ell0 = ellrow[0];

  // ---- phase 1 -----------------------------------------------------------------------------------
  if (kTab && !DENSE) {   // (requested with the loads above; the barrier costs one per tile -- the persistent kernel pays it once)
    lt_lds[tid] = kLogTab[tid];
    __syncthreads();
  }
  if (SCATTER) {
#pragma unroll
    for (int k = 0; k < 5; k++) ff[k * 256 + tid] = T(0);
  }
  if (a0) {
    if (KIND == 0) {
      store_prim<T>(pe, LE, tid, s0, lt);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) pe[k * LE + tid] = s0[k];
    }
  }
  if (a1) {
    if (KIND == 0) {
      store_prim<T>(pe, LE, i1, s1, lt);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) pe[k * LE + i1] = s1[k];
    }
  }
  __syncthreads();

  // ---- phases 2 + 3, one pass of 256 faces at a time (halves the LDS flux buffer -> 4 workgroups/CU) ---
  T acc[5] = {T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int it = 0; it < MAXP; it++) {
    if (it > 0 && nf <= 256 * it) break;
    if (MAXP > 2 && it >= 1 && it + 1 < MAXP) fin[it + 1] = load_face(it + 1);   // next pass's record flies during this pass
    const FaceIn& fi   = fin[it];
    const bool    last = it == MAXP - 1 || nf <= 256 * (it + 1);
    if (fi.valid) {
      const int  l = fi.lr & 0xFFFFu, r16 = fi.lr >> 16;
      const bool wall = r16 == 0xFFFFu;
      const int  r = wall ? l : r16;
      T          g[5], spd = T(0);
      // The tile's faces are ordered by direction inside each block of 256 (tile_plan.cpp), so a wavefront's active lanes
      // usually share one axis-aligned normal s * e_axis: selecting components then gives the same values as the general
      // rotation (flux_math.hpp: kepes_axis_fixed) without its 27 multiply-adds -- the persistent kernel's arrangement.
      const int  wcode  = __builtin_amdgcn_readfirstlane(fi.code);
      const bool shared = DICT && KIND == 0 && wcode < 6 && __all(fi.code == wcode);
      if (KIND == 0 && shared) {
        const T sg   = (wcode & 1) ? T(1) : T(-1);
        const T area = reinterpret_cast<const T*>(reinterpret_cast<const V4*>(P.geo_table) + fi.gi)[3];
        Prim<T> L, R;
        load_prim<T>(pe, LE, l, L);
        load_prim<T>(pe, LE, r, R);
        T uL, vL, wL, uR, vR, wR;
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
        if (wall) {   // reflective wall: the right state is the mirror image of the left one (kernels.cu:371-375)
          uR = -uL;
          vR = vL;
          wR = wL;
        }
        T f[5];
        kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, spd);
        g[0] = f[0];
        g[4] = f[4];
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
      const V4 gm = DICT ? reinterpret_cast<const V4*>(P.geo_table)[fi.gi] : fi.gm;
      const T    n[3] = {gm.x, gm.y, gm.z};
      T          t1[3], t2[3];
      if (DICT) {  // frame precomputed per distinct normal (table rows are L1/L2 resident)
        const V4* __restrict__ tab = reinterpret_cast<const V4*>(P.geo_table);
        const V4 b1 = tab[fi.gi + 1], b2 = tab[fi.gi + 2];
        t1[0] = b1.x; t1[1] = b1.y; t1[2] = b1.z;
        t2[0] = b2.x; t2[1] = b2.y; t2[2] = b2.z;
      } else {
        face_basis_fast<T>(n, t1, t2);
      }
      if (KIND == 0) {
        Prim<T> L, R;
        load_prim<T>(pe, LE, l, L);
        load_prim<T>(pe, LE, r, R);
#ifdef T8GPU_EXP_NOMATH
        g[0] = L.rho + R.rho + n[0] + t1[0]; g[1] = L.vx + R.vx + t2[0]; g[2] = L.vy + R.vy + L.beta + R.beta; g[3] = L.vz + R.vz + L.lrho + R.lrho;
        g[4] = L.p + R.p + L.lbeta + R.lbeta + L.v0 + R.v0 + gm.w;
        spd = g[0];
#else
        kepes_prim<T>(L, R, wall, n, t1, t2, gm.w, g, spd);
#endif
      } else {
        T sl[5], sr[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
          sl[k] = pe[k * LE + l];
          sr[k] = pe[k * LE + r];
        }
        hll_face<T>(sl, sr, wall, n, t1, t2, gm.w, g, spd, KIND == 2);
      }
      }
      if (fi.orig >= 0) speed[fi.orig] = spd;
      if (SCATTER) {
        if (l < ne) {
#pragma unroll
          for (int k = 0; k < 5; k++) atomicAdd(&ff[k * 256 + l], -g[k]);
        }
        if (!wall && r < ne) {
#pragma unroll
          for (int k = 0; k < 5; k++) atomicAdd(&ff[k * 256 + r], g[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) ff[k * 256 + tid] = g[k];
      }
    }
    if (MAXP == 2 && last) {  // last pass: start the RK stage's loads; they fly during the barrier + gather
      if (STAGE > 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) pv[k] = prev.p[k][e];
      }
      volume = vol[e];
    }
    if (SCATTER) continue;   // the accumulators take every pass; one barrier after the loop
    __syncthreads();
    if (own) {
      bool done = ell_accumulate<T>(ell0, it, ff, acc);
      for (int c = 1; c < P.ell_width / 8 && !done; c++) done = ell_accumulate<T>(ellrow[c], it, ff, acc);
    }
    if (!last) __syncthreads();   // the buffer is rewritten by the next pass
  }
  if (SCATTER) {
    __syncthreads();
    if (own) {
#pragma unroll
      for (int k = 0; k < 5; k++) acc[k] = ff[k * 256 + tid];
    }
  }

  if (MAXP > 2) {  // (the last pass is not known at compile time here: fetched after the loop, other workgroups cover it)
    if (STAGE > 1) {
#pragma unroll
      for (int k = 0; k < 5; k++) pv[k] = prev.p[k][e];
    }
    volume = vol[e];
  }
  // ---- RK stage (ssp_runge_kutta.inl:30-99) ---------------------------------------------------------
  if (own) {
    const T scale = dt / volume;
    T       res[5];
#pragma unroll
    for (int k = 0; k < 5; k++) res[k] = rk_stage_update<T, STAGE>(pv[k], s0[k], scale, acc[k]);
#pragma unroll
    for (int k = 0; k < 5; k++) out.p[k][e] = res[k];
    if (P.send_map) ghost_window_send<T>(P, e, res);
  }
}

// one tile per workgroup: workgroup b takes position tile_begin + xcd_position(b) of tile_order
template <class T, int KIND, int STAGE, bool DICT, int MAXP, bool SCATTER = false, bool DENSE = false>
__global__ __launch_bounds__(256, DENSE ? (sizeof(T) == 8 ? 4 : 5) : 1) void k_plain_fused_p(T8gpuPlainPlan P, int tile_begin, FVars<T> prev, FVars<T> src,
                                                       FVars<T> out, const T* __restrict__ vol, T dt,
                                                       T* __restrict__ speed) {
#ifdef T8GPU_EXP_TILEMOD   // experiment builds only (build.py variants): every workgroup works on one of the first few tiles,
                           // so all traffic stays in the caches -- what remains is the kernel's instruction time
  const int pos = tile_begin + xcd_position(blockIdx.x, gridDim.x) % T8GPU_EXP_TILEMOD;
#else
  const int pos = tile_begin + xcd_position(blockIdx.x, gridDim.x);
#endif
  plain_tile_body<T, KIND, STAGE, DICT, MAXP, SCATTER, DENSE>(P, pos, prev, src, out, vol, dt, speed);
}

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_FUSED_TILE_BODY_HPP
