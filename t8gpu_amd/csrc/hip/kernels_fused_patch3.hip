// kernels_fused_patch3.hip -- fused RK stage over 3D STRUCTURED PATCHES of a plain-element mesh (round 3).
//
// Reference: examples/compressible_euler/kernels.cu:135-309 + ssp_runge_kutta.inl:30-99 (the stage the tile kernels fuse).
//
// The 3D analogue of kernels_fused_patch.hip: a Cartesian hexahedral AMR forest is locally structured, and an aligned
// 8 x 8 x 4 block of same-size hexahedra is 256 CONSECUTIVE elements in Morton order (csrc/host/tile_plan.cpp:
// find_patches3; 72 % of the elements of the c5 benchmark mesh). On generic 3D tiles the face cap binds at ~134 elements per
// 256-lane workgroup, half the lanes idle in the per-element phases and every element walks a 16-entry face list; here
//
//   512 lanes = 8 wavefronts per patch: wavefronts 0-3 ("cell lanes") own the 256 cells, wavefronts 4-7 ("side lanes")
//   the 256 cells across the six sides ([-x 32 | +x 32 | -y 32 | +y 32 | -z 64 | +z 64]).
//   phase 1   every lane: primitives of its cell -> an LDS record (cell lanes keep theirs in registers)
//   phase 2   cell lanes: the +x and +y face of their cell (left operand in registers);
//             side lanes: the +z face of cell (tid - 256) -- both operands from LDS --, then wavefronts 6 / 7 the 128 faces of
//             the - sides (same orientation and values as in the neighbouring tile). Every wavefront has 2 flux rounds:
//             no wavefront waits at the barrier for another's third round
//   phase 3   cell lanes: the three - fluxes in ascending face id (the order of the owning neighbours' indices: pairwise
//             ctz rule, three flag bits per patch for the lanes on two low sides), then -(+x), -(+y), -(+z); RK stage
//
// Per 256 cells: 8 + 14 wave-rounds of primitives / fluxes and no face lists (generic 3D tiles: 8 + 8 rounds and a
// 16-entry list walk per 134 cells). Same flux functions, operand order and summation order as the tile kernels: bitwise
// their results (tests/test_gpu_patch.py). Persistent and software-pipelined; two barriers per patch.
#include <cstdlib>

#include "patch_common.hpp"
#include "stage_kernel_note.hpp"

namespace t8gpu_hip {

constexpr int kP3FF = 896;   // flux slots per variable: +x 256 | +y 256 | +z 256 | - sides 128 (-x 32, -y 32, -z 64)

T8_DEV int patch3_morton(int i, int j, int k) {   // x bits 0, 3, 6; y bits 1, 4, 7; z bits 2, 5
  int t = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) t |= (((i >> b) & 1) << (3 * b)) | (((j >> b) & 1) << (3 * b + 1));
#pragma unroll
  for (int b = 0; b < 2; b++) t |= ((k >> b) & 1) << (3 * b + 2);
  return t;
}
T8_DEV int patch3_ctz(int v) { return v == 0 ? 8 : __builtin_ctz(static_cast<unsigned>(v)); }
T8_DEV void patch3_ijk(int c, int& ci, int& cj, int& ck) {
  ci = cj = ck = 0;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    ci |= ((c >> (3 * b)) & 1) << b;
    cj |= ((c >> (3 * b + 1)) & 1) << b;
  }
#pragma unroll
  for (int b = 0; b < 2; b++) ck |= ((c >> (3 * b + 2)) & 1) << b;
}
// records of the right operands of cell c's + faces (inside the patch, or the slot of the cell across the side)
T8_DEV void patch3_plus_slots(int c, int& rx, int& ry, int& rz) {
  int ci, cj, ck;
  patch3_ijk(c, ci, cj, ck);
  rx = ci < 7 ? patch3_morton(ci + 1, cj, ck) : 256 + 32 + cj + 8 * ck;
  ry = cj < 7 ? patch3_morton(ci, cj + 1, ck) : 256 + 96 + ci + 8 * ck;
  rz = ck < 3 ? patch3_morton(ci, cj, ck + 1) : 256 + 192 + ci + 8 * cj;
}
// flux slots of cell c's - faces (the + faces of the cells before it, or the slots of the - sides)
T8_DEV void patch3_minus_slots(int c, int& a_mx, int& a_my, int& a_mz) {
  int ci, cj, ck;
  patch3_ijk(c, ci, cj, ck);
  a_mx = ci > 0 ? patch3_morton(ci - 1, cj, ck) : 768 + cj + 8 * ck;
  a_my = cj > 0 ? 256 + patch3_morton(ci, cj - 1, ck) : 768 + 32 + ci + 8 * ck;
  a_mz = ck > 0 ? 512 + patch3_morton(ci, cj, ck - 1) : 768 + 64 + ci + 8 * cj;
}
// An opaque copy of a lane constant: what the IRREGULAR instantiation derives from it inside the loop is computed there
// (some 100 integer instructions per patch), not held in registers across the whole loop -- it has none to spare, and a
// spill reload waits for the in-flight prefetch.
T8_DEV int patch3_fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// IRR: the launch covers IRREGULAR patches (descriptor flag 0x800, csrc/host/tile_plan.cpp) -- the same block, with every side
// face evaluated in the orientation its per-cell words give and the six fluxes of a cell added in the listed order. A separate
// instantiation: the regular one keeps its register budget.
// workgroup `wg` of the `nwg` that share the patch tiles [tile_begin, tile_begin + tile_count) of tile_order (nwg a multiple of 8
// or < 8: wg & 7 must be the workgroup's XCD)
template <class T, int KIND, int STAGE, bool IRR, bool NT>
T8_DEV void plain_patch3_body(const T8gpuPlainPlan& P, int tile_begin, int tile_count, int wg, int nwg, const FVars<T>& prev,
                              const FVars<T>& src, const FVars<T>& out, const T* __restrict__ vol, T dt, T* __restrict__ speed) {
  constexpr int NW  = KIND == 0 ? kPrimWords : 5;
  constexpr int REC = rec_words<T, NW>();
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  T* const  ff = reinterpret_cast<T*>(lds_raw);               // [5][kP3FF] the patch's face fluxes
  T* const  pe = ff + 5 * kP3FF;                              // [512][REC] primitives (or states): 256 own, 256 across the sides
  constexpr bool kTab = sizeof(T) == 8 && KIND == 0;          // fp64 KEPES: the logarithm table, behind the records
  double* const lt = reinterpret_cast<double*>(pe + REC * 512);
  const int tid = threadIdx.x;

  const int G = nwg, xcd = wg & 7, jw = wg >> 3;
  const int nxcd = G < 8 ? G : 8, nx = (G - xcd + 7) >> 3;
  const int per = tile_count / nxcd, rem = tile_count % nxcd;
  const int x0   = tile_begin + xcd * per + (xcd < rem ? xcd : rem);
  const int tend = x0 + per + (xcd < rem ? 1 : 0);
  int       t    = x0 + jw;
  if (xcd >= nxcd || t >= tend) return;
  if (kTab) {
    if (tid < 2 * kLogTabEntries) lt[tid] = kLogTab[tid];
    __syncthreads();
  }

  // ---- lane constants: c = the cell the lane works for (cell lanes: their own; side lanes: the cell whose +z face they take)
  const bool side = tid >= 256;
  const int  c    = tid & 255, hl = tid - 256;
  int        ci, cj, ck, rx, ry, rz, a_mx, a_my, a_mz;
  patch3_ijk(c, ci, cj, ck);
  // records of the right operands of the cell's + faces; flux slots of its - faces (the + faces of the cells across)
  patch3_plus_slots(c, rx, ry, rz);
  patch3_minus_slots(c, a_mx, a_my, a_mz);
  // (the IRREGULAR instantiation holds the three operand slots packed 10 bits each in one register; the flux slots of the - faces
  //  it recomputes per patch from an opaque copy of c: one more register held across the loop would spill)
  const unsigned pk_plus  = static_cast<unsigned>(rx) | static_cast<unsigned>(ry) << 10 | static_cast<unsigned>(rz) << 20;
  // pairwise order of the - faces (true: the second-named axis' face has the smaller id); where both coordinates of the pair
  // are 0 the patch's flags decide (bits 0 / 1 / 2)
  const bool r_yx = patch3_ctz(cj) >= patch3_ctz(ci), r_zx = patch3_ctz(ck) >= patch3_ctz(ci), r_zy = patch3_ctz(ck) >= patch3_ctz(cj);
  const bool f_yx = ci == 0 && cj == 0, f_zx = ci == 0 && ck == 0, f_zy = cj == 0 && ck == 0;
  // wavefronts 6 / 7: the faces of the - sides. m < 32: -x side (cell (0, j, k)); m < 64: -y side; else -z side (wavefront 7)
  const int  m       = hl - 128;
  const bool minus   = hl >= 128;
  const bool minus_y = m >= 32;                       // (wavefront 6 only: its lanes mix the x and y axes)
  const int  mq      = m < 32 ? m : (m < 64 ? m - 32 : m - 64);
  const int  m_l     = 256 + (m < 32 ? mq : (m < 64 ? 64 + mq : 128 + mq));
  const int  m_r     = m < 32 ? patch3_morton(0, mq & 7, mq >> 3) : (m < 64 ? patch3_morton(mq & 7, 0, mq >> 3) : patch3_morton(mq & 7, mq >> 3, 0));

  typedef int int8v __attribute__((ext_vector_type(8)));
  struct Desc {
    int    e0, h0, fbase, flags;
    double area, vol;   // vol: the patch's uniform element volume where flags has 0x400 (tile_plan.cpp), else unused
  };
  auto load_desc = [&](int tt) {   // scalar load through the constant address space (see k_plain_persistent)
#ifdef T8GPU_EXP_TILEMOD
    const size_t kk = static_cast<size_t>(tile_begin + (tt < tend ? tt : tend - 1) % T8GPU_EXP_TILEMOD);
#else
    const size_t kk = static_cast<size_t>(tt < tend ? tt : tend - 1);
#endif
    const int8v  r = *reinterpret_cast<const __attribute__((address_space(4))) int8v*>(
        reinterpret_cast<const __attribute__((address_space(4))) char*>(reinterpret_cast<uintptr_t>(P.tile_desc)) + 32 * kk);
    Desc d;
    d.e0 = r[0]; d.h0 = r[2]; d.fbase = r[4]; d.flags = r[5];
    d.area = __hiloint2double(r[7], r[6]);
    d.vol  = __hiloint2double(r[3], r[1]);
    return d;
  };
  auto fetch = [&](const Desc& d, int hslot, T s[5]) {   // the state of the lane's cell: its own, or the cell across a side
    const int slot = side ? hslot : d.e0 + tid;
#pragma unroll
    for (int k = 0; k < 5; k++) s[k] = at32<T>(src.p[k], static_cast<unsigned>(slot));
  };

  const int stride = nx;
  Desc      d0 = load_desc(t), d1 = load_desc(t + stride);
  int       hs_b = side ? P.halo_ids[d1.h0 + hl] : 0;
  T         cur[5];
  fetch(d0, side ? P.halo_ids[d0.h0 + hl] : 0, cur);
  // IRREGULAR patches: the per-cell words travel like the states -- requested one patch ahead (lane tid: face_lr[w + c] and
  // face_orig[w + tid], i.e. the first own interior id of cell tid or, on the side lanes, the first wall id of cell tid - 256),
  // parked in the PADDING of the lane's LDS record in phase 1 and read from there by whoever needs another cell's words. (A
  // load consumed in the iteration that issues it waits for the whole prefetch: vmcnt retires in order.)
  auto info_words = [&](const Desc& d, unsigned& lr, int& id) {
    const unsigned w = static_cast<unsigned>(d.fbase);   // (descriptor word 4 of an irregular patch: where its words start)
    lr = P.face_lr[w + c];
    id = P.face_orig[w + tid];
  };
  unsigned lr_cur = 0, lr_nxt = 0;
  int      id_cur = -1, id_nxt = -1;
  if (IRR) info_words(d0, lr_cur, id_cur);
  __builtin_amdgcn_s_waitcnt(0);   // (the prologue's loads must not become a wait inside the loop)

  for (; t < tend; t += stride) {
    const Desc d2   = load_desc(t + 2 * stride);
    const int  hs_c = side ? P.halo_ids[d2.h0 + hl] : 0;
    constexpr bool irr = IRR;
    T          nxt[5];
    if (t + stride < tend) {
      fetch(d1, hs_b, nxt);
      if (irr) info_words(d1, lr_nxt, id_nxt);
    }
    const int e = d0.e0 + c;
    T         pv[5] = {T(0), T(0), T(0), T(0), T(0)}, volume = T(1);
    if (!side) {
      if (STAGE > 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) pv[k] = stream_load<NT>(&at32<T>(prev.p[k], static_cast<unsigned>(e)));
      }
      volume = (d0.flags & 0x400) ? static_cast<T>(d0.vol) : at32<T>(vol, static_cast<unsigned>(e));   // (uniform patch volume from the descriptor)
    }
    const T area = static_cast<T>(d0.area);

    // ---- phase 1: the record of the lane's cell (slot tid: cell lanes 0..255, cells across the sides 256..511) -------------
    T mine[NW];
    if (KIND == 0) {
      prim_words<T>(cur, mine, lt);
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) mine[k] = cur[k];
    }
    rec_store<T, NW>(pe + tid * REC, mine);
    if (irr) {   // (the record's padding: word NW of REC -- 8 bytes in fp64, 12 in fp32 -- never touched by rec_store / rec_load)
      int* const spare = reinterpret_cast<int*>(pe + tid * REC + NW);
      spare[0] = static_cast<int>(lr_cur);
      spare[1] = id_cur;
    }
    __syncthreads();

    // ---- phase 2 -----------------------------------------------------------------------------------------------------------
    if (irr) {   // every side face in its listed orientation (patch_common.hpp: patch_face3g); interior faces come out the same
      T g[5], sp;
      auto spare_of = [&](int slot, int j) { return reinterpret_cast<const int*>(pe + slot * REC + NW)[j]; };
      // words of cell c (the lane's own cell, or the one whose +z face a side lane takes): sides | walls | order, first ids
      const unsigned ia    = side ? static_cast<unsigned>(spare_of(c, 0)) : lr_cur;
      const int      id_a  = side ? spare_of(c, 1) : id_cur;
      const int      idw_a = side ? id_cur : spare_of(256 + c, 1);
      const int      qx = pk_plus & 1023u, qy = (pk_plus >> 10) & 1023u, qz = pk_plus >> 20;
      if (!side) {   // the +x and the +y face of the lane's cell (compile-time axes: a run-time axis costs ~50 selects per face)
        {
          const bool own = (ia >> 1) & 1u;
          patch_face3g<T, KIND, NW>(0, true, own, (ia >> 7) & 1u, pe + c * REC, pe + qx * REC, area, g, sp);
#pragma unroll
          for (int k = 0; k < 5; k++) ff[k * kP3FF + c] = g[k];
          if (speed && own) speed[patch3_own_id(ia, id_a, idw_a, 1)] = sp;
        }
        {
          const bool own = (ia >> 3) & 1u;
          patch_face3g<T, KIND, NW>(1, true, own, (ia >> 9) & 1u, pe + c * REC, pe + qy * REC, area, g, sp);
#pragma unroll
          for (int k = 0; k < 5; k++) ff[k * kP3FF + 256 + c] = g[k];
          if (speed && own) speed[patch3_own_id(ia, id_a, idw_a, 3)] = sp;
        }
      } else {
        const bool own = (ia >> 5) & 1u;
        patch_face3g<T, KIND, NW>(2, true, own, (ia >> 11) & 1u, pe + c * REC, pe + qz * REC, area, g, sp);
#pragma unroll
        for (int k = 0; k < 5; k++) ff[k * kP3FF + 512 + c] = g[k];
        if (speed && own) speed[patch3_own_id(ia, id_a, idw_a, 5)] = sp;
        if (minus) {
          const unsigned ib    = static_cast<unsigned>(spare_of(m_r, 0));   // the boundary cell whose - face this lane takes
          const int      id_b  = spare_of(m_r, 1), idw_b = spare_of(256 + m_r, 1);
          const int      ax    = m < 32 ? 0 : (m < 64 ? 1 : 2);
          const bool     mown  = (ib >> (2 * ax)) & 1u;
          if (m < 64)   // wavefront 6: x / y per lane; wavefront 7: z
            patch_face3g<T, KIND, NW>(ax, false, mown, (ib >> (6 + 2 * ax)) & 1u, pe + m_r * REC, pe + m_l * REC, area, g, sp);
          else
            patch_face3g<T, KIND, NW>(2, false, mown, (ib >> 10) & 1u, pe + m_r * REC, pe + m_l * REC, area, g, sp);
#pragma unroll
          for (int k = 0; k < 5; k++) ff[k * kP3FF + 768 + m] = g[k];
          if (speed && mown) speed[patch3_own_id(ib, id_b, idw_b, 2 * ax)] = sp;
        }
      }
    } else if (!side) {
      T wr[NW], g[5], sx, sy;
      rec_load<T, NW>(pe + rx * REC, wr);
      patch_face3<T, KIND, NW>(0, mine, wr, area, g, sx);
#pragma unroll
      for (int k = 0; k < 5; k++) ff[k * kP3FF + c] = g[k];
      rec_load<T, NW>(pe + ry * REC, wr);
      if (sizeof(T) == 8) rec_load<T, NW>(pe + c * REC, mine);   // (fp64 register budget: the own record is read back, not held)
      patch_face3<T, KIND, NW>(1, mine, wr, area, g, sy);
#pragma unroll
      for (int k = 0; k < 5; k++) ff[k * kP3FF + 256 + c] = g[k];
      if (speed) {   // own faces: ids fbase + 3 t + {0, 1, 2} (+x, +y, +z)
        speed[d0.fbase + 3 * c]     = sx;
        speed[d0.fbase + 3 * c + 1] = sy;
      }
    } else {
      {   // the +z face of cell c = tid - 256
        T wl[NW], wr[NW], g[5], sz;
        rec_load<T, NW>(pe + c * REC, wl);
        rec_load<T, NW>(pe + rz * REC, wr);
        patch_face3<T, KIND, NW>(2, wl, wr, area, g, sz);
#pragma unroll
        for (int k = 0; k < 5; k++) ff[k * kP3FF + 512 + c] = g[k];
        if (speed) speed[d0.fbase + 3 * c + 2] = sz;
      }
      if (minus) {   // the - sides: left operand = the cell across, right = the patch's cell
        T wl[NW], wr[NW], g[5], sm;
        rec_load<T, NW>(pe + m_l * REC, wl);
        rec_load<T, NW>(pe + m_r * REC, wr);
        if (m < 64)
          patch_face<T, KIND, NW>(minus_y, wl, wr, area, g, sm);   // wavefront 6: x / y per lane
        else
          patch_face3<T, KIND, NW>(2, wl, wr, area, g, sm);        // wavefront 7: z
#pragma unroll
        for (int k = 0; k < 5; k++) ff[k * kP3FF + 768 + m] = g[k];
      }
    }
    __syncthreads();

    // ---- phase 3: six fluxes in ascending face id, RK stage ------------------------------------------------------------------
    if (!side && irr) {   // the six sides in the listed order (ascending face id), - for the faces the cell lists itself
      const unsigned ia = lr_cur, ord = ia >> 12;
      int            b_mx, b_my, b_mz;
      patch3_minus_slots(patch3_fresh(c), b_mx, b_my, b_mz);
      // (the six flux slots packed 10 bits each: one 64-bit shift per side instead of a chain of five selects)
      const unsigned long long slots = static_cast<unsigned long long>(b_mx) | static_cast<unsigned long long>(c) << 10 |
                                       static_cast<unsigned long long>(b_my) << 20 | static_cast<unsigned long long>(256 + c) << 30 |
                                       static_cast<unsigned long long>(b_mz) << 40 | static_cast<unsigned long long>(512 + c) << 50;
      T              acc[5] = {T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const int sd  = (ord >> (3 * q)) & 7u;
        const int pos = static_cast<int>((slots >> (10 * sd)) & 1023u);
        const T   sg  = (ia >> sd) & 1u ? T(-1) : T(1);
#pragma unroll
        for (int k = 0; k < 5; k++) acc[k] = __builtin_fma(sg, ff[k * kP3FF + pos], acc[k]);
      }
      const T scale = dt / volume;   // (rk_scale's test would cost this kernel registers it does not have: 14 - 21 spills)
#pragma unroll
      for (int k = 0; k < 5; k++) stream_store<NT>(&at32<T>(out.p[k], static_cast<unsigned>(e)), rk_stage_update<T, STAGE>(pv[k], cur[k], scale, acc[k]));
    } else if (!side) {
      const bool yx = f_yx ? (d0.flags & 1) != 0 : r_yx;   // -y before -x
      const bool zx = f_zx ? (d0.flags & 2) != 0 : r_zx;   // -z before -x
      const bool zy = f_zy ? (d0.flags & 4) != 0 : r_zy;   // -z before -y
      const int  px = (yx ? 1 : 0) + (zx ? 1 : 0), py = (yx ? 0 : 1) + (zy ? 1 : 0);   // positions of -x, -y (the third: -z)
      const int  a0 = px == 0 ? a_mx : (py == 0 ? a_my : a_mz);
      const int  a1 = px == 1 ? a_mx : (py == 1 ? a_my : a_mz);
      const int  a2 = px == 2 ? a_mx : (py == 2 ? a_my : a_mz);
      T          acc[5];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        acc[k] = __builtin_fma(T(1), ff[k * kP3FF + a0], T(0));
        acc[k] = __builtin_fma(T(1), ff[k * kP3FF + a1], acc[k]);
        acc[k] = __builtin_fma(T(1), ff[k * kP3FF + a2], acc[k]);
        acc[k] = __builtin_fma(T(-1), ff[k * kP3FF + c], acc[k]);
        acc[k] = __builtin_fma(T(-1), ff[k * kP3FF + 256 + c], acc[k]);
        acc[k] = __builtin_fma(T(-1), ff[k * kP3FF + 512 + c], acc[k]);
      }
      const T scale = dt / volume;   // (rk_scale's test would cost this kernel registers it does not have: 14 - 21 spills)
#pragma unroll
      for (int k = 0; k < 5; k++) stream_store<NT>(&at32<T>(out.p[k], static_cast<unsigned>(e)), rk_stage_update<T, STAGE>(pv[k], cur[k], scale, acc[k]));
    }
#pragma unroll
    for (int k = 0; k < 5; k++) cur[k] = nxt[k];
    d0   = d1;
    d1   = d2;
    hs_b = hs_c;
    if (irr) {
      lr_cur = lr_nxt;
      id_cur = id_nxt;
    }
  }
}

// (second launch bound = wavefronts per SIMD the register allocation must allow: two 8-wave workgroups per CU in fp64 --
//  77 KB of LDS each --, three in fp32)
template <class T, int KIND, int STAGE, bool IRR, bool NT>
__global__ __launch_bounds__(512, sizeof(T) == 8 ? 4 : 6) void k_plain_patch3(T8gpuPlainPlan P, int tile_begin, int tile_count, FVars<T> prev,
                                                                                 FVars<T> src, FVars<T> out, const T* __restrict__ vol, T dt,
                                                                                 T* __restrict__ speed) {
  plain_patch3_body<T, KIND, STAGE, IRR, NT>(P, tile_begin, tile_count, blockIdx.x, gridDim.x, prev, src, out, vol, dt, speed);
}

// REGULAR and IRREGULAR patches of one class in ONE launch (round 4): the first `reg_wgs` workgroups (a multiple of 8) walk the
// regular patches, the others the irregular ones -- two bodies behind a workgroup-uniform branch, each with its own registers
// (the regular body's 114, the irregular one's 128: nothing is live across the branch, so neither spills; both forms behind a
// per-PATCH branch inside one loop cost 13-30 spills, round 3). As two launches the second one starts when the first has
// drained: one tail and one launch gap per stage less.
template <class T, int KIND, int STAGE, bool NT>
__global__ __launch_bounds__(512, sizeof(T) == 8 ? 4 : 6) void k_plain_patch3_both(T8gpuPlainPlan P, int reg_begin, int reg_count, int reg_wgs,
                                                                                      int irr_begin, int irr_count, FVars<T> prev, FVars<T> src,
                                                                                      FVars<T> out, const T* __restrict__ vol, T dt,
                                                                                      T* __restrict__ speed) {
  const int b = blockIdx.x;
  if (b < reg_wgs)
    plain_patch3_body<T, KIND, STAGE, false, NT>(P, reg_begin, reg_count, b, reg_wgs, prev, src, out, vol, dt, speed);
  else
    plain_patch3_body<T, KIND, STAGE, true, NT>(P, irr_begin, irr_count, b - reg_wgs, static_cast<int>(gridDim.x) - reg_wgs, prev, src, out, vol, dt, speed);
}

// tiles [tile_begin, tile_begin + tile_count) of tile_order must all be 3D patch tiles. persistent = false: one patch per
// workgroup (class-split multi-rank launches).
template <class T>
int plain_patch3_stage(int kind, int stage, const T8gpuPlainPlan* plan, int tile_begin, int tile_count, FVars<T> prev, FVars<T> mid,
                       FVars<T> out, const T* volume, T dt, T* speed, bool persistent, bool irregular, hipStream_t stream) {
  if (tile_count <= 0) return 0;
  if (irregular && (!plan->face_lr || !plan->face_orig)) return static_cast<int>(hipErrorInvalidValue);
  if (!plan->tile_desc) return static_cast<int>(hipErrorInvalidValue);
  // (the kernel addresses a plane by a 32-bit byte offset, patch_common.hpp: at32)
  if (plan->n_slots_addressed <= 0 || static_cast<unsigned long long>(plan->n_slots_addressed) * sizeof(T) >= (1ull << 32))
    return static_cast<int>(hipErrorInvalidValue);
  const int    nw  = kind == 0 ? kPrimWords : 5;
  const int    rec = sizeof(T) == 8 ? (nw > 5 ? 10 : 6) : 12;
  const size_t lds = sizeof(T) * (static_cast<size_t>(5) * kP3FF + static_cast<size_t>(rec) * 512) +
                     ((sizeof(T) == 8 && kind == 0) ? 2 * kLogTabEntries * sizeof(double) : 0);
  const int        cus        = device_cu_count();
  static const int per_cu_env = env_per_cu("T8GPU_PATCH_WGS");
  if (cus == 0) return static_cast<int>(hipErrorInvalidDevice);
  const int  per_cu    = per_cu_env > 0 ? per_cu_env : (sizeof(T) == 8 ? 2 : 3);
  const int  resident  = cus * per_cu;
  const int  grid_size = (!persistent || tile_count < resident) ? tile_count : resident;
  const dim3 grid(grid_size), block(512);
  const bool nt = stream_hint(plan->n_slots_addressed, sizeof(T));   // (flux_math.hpp: stream_store)
  note_stage_kernel(tile_count, irregular ? (nt ? "k_plain_patch3<T, K, S, true, true>" : "k_plain_patch3<T, K, S, true, false>")
                                          : (nt ? "k_plain_patch3<T, K, S, false, true>" : "k_plain_patch3<T, K, S, false, false>"), static_cast<int>(sizeof(T)), kind,
                    stage);
#define T8_P3I(K, S, I, N)                                                                                                     \
  do {                                                                                                                       \
    if (lds > 64 * 1024) {                                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plain_patch3<T, K, S, I, N>),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                 \
      if (e != hipSuccess) return static_cast<int>(e);                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL((k_plain_patch3<T, K, S, I, N>), grid, block, lds, stream, *plan, tile_begin, tile_count, prev, mid, out, volume, \
                       dt, speed);                                                                                           \
  } while (0)
#define T8_P3(K, S)                  \
  do {                               \
    if (irregular && nt)             \
      T8_P3I(K, S, true, true);      \
    else if (irregular)              \
      T8_P3I(K, S, true, false);     \
    else if (nt)                     \
      T8_P3I(K, S, false, true);     \
    else                             \
      T8_P3I(K, S, false, false);    \
  } while (0)
#define T8_P3S(K)          \
  do {                     \
    if (stage == 1)        \
      T8_P3(K, 1);         \
    else if (stage == 2)   \
      T8_P3(K, 2);         \
    else                   \
      T8_P3(K, 3);         \
  } while (0)
  if (kind == 0)
    T8_P3S(0);
  else if (kind == 1)
    T8_P3S(1);
  else
    T8_P3S(2);
#undef T8_P3S
#undef T8_P3
#undef T8_P3I
  return static_cast<int>(hipGetLastError());
}

// Regular patches [reg_begin, +reg_count) and irregular patches [irr_begin, +irr_count) in one persistent launch. Returns -1 where
// that is not worth it or not possible (a class launch of a multi-rank stage, fewer patches than resident workgroups): the
// caller launches the two kinds one after the other.
template <class T>
int plain_patch3_both_stage(int kind, int stage, const T8gpuPlainPlan* plan, int reg_begin, int reg_count, int irr_begin, int irr_count,
                            FVars<T> prev, FVars<T> mid, FVars<T> out, const T* volume, T dt, T* speed, hipStream_t stream) {
  static const bool off = std::getenv("T8GPU_PATCH3_BOTH") && std::getenv("T8GPU_PATCH3_BOTH")[0] == '0';   // (measurements)
  if (off || reg_count <= 0 || irr_count <= 0) return -1;
  if (!plan->face_lr || !plan->face_orig || !plan->tile_desc) return static_cast<int>(hipErrorInvalidValue);
  if (plan->n_slots_addressed <= 0 || static_cast<unsigned long long>(plan->n_slots_addressed) * sizeof(T) >= (1ull << 32))
    return static_cast<int>(hipErrorInvalidValue);
  const int cus = device_cu_count();
  if (cus == 0) return static_cast<int>(hipErrorInvalidDevice);
  const int resident = cus * (sizeof(T) == 8 ? 2 : 3);
  if (reg_count + irr_count < 2 * resident) return -1;
  // the resident workgroups in proportion to the work (an irregular patch costs ~1.2 regular ones), the regular share a multiple of 8
  const double w_r = reg_count, w_i = 1.2 * irr_count;
  int reg_wgs = static_cast<int>(resident * w_r / (w_r + w_i) / 8.0 + 0.5) * 8;
  if (reg_wgs < 8) reg_wgs = 8;
  if (reg_wgs > resident - 8) reg_wgs = resident - 8;
  const int    nw  = kind == 0 ? kPrimWords : 5;
  const int    rec = sizeof(T) == 8 ? (nw > 5 ? 10 : 6) : 12;
  const size_t lds = sizeof(T) * (static_cast<size_t>(5) * kP3FF + static_cast<size_t>(rec) * 512) +
                     ((sizeof(T) == 8 && kind == 0) ? 2 * kLogTabEntries * sizeof(double) : 0);
  const dim3 grid(resident), block(512);
  const bool nt = stream_hint(plan->n_slots_addressed, sizeof(T));   // (flux_math.hpp: stream_store)
  note_stage_kernel(reg_count + irr_count, nt ? "k_plain_patch3_both<T, K, S, true>" : "k_plain_patch3_both<T, K, S, false>", static_cast<int>(sizeof(T)), kind, stage);
#define T8_P3BN(K, S, N)                                                                                                         \
  do {                                                                                                                       \
    if (lds > 64 * 1024) {                                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plain_patch3_both<T, K, S, N>),                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                 \
      if (e != hipSuccess) return static_cast<int>(e);                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL((k_plain_patch3_both<T, K, S, N>), grid, block, lds, stream, *plan, reg_begin, reg_count, reg_wgs, irr_begin, irr_count, \
                       prev, mid, out, volume, dt, speed);                                                                   \
  } while (0)
#define T8_P3B(K, S)        \
  do {                      \
    if (nt)                 \
      T8_P3BN(K, S, true);  \
    else                    \
      T8_P3BN(K, S, false); \
  } while (0)
#define T8_P3BS(K)         \
  do {                     \
    if (stage == 1)        \
      T8_P3B(K, 1);        \
    else if (stage == 2)   \
      T8_P3B(K, 2);        \
    else                   \
      T8_P3B(K, 3);        \
  } while (0)
  if (kind == 0)
    T8_P3BS(0);
  else if (kind == 1)
    T8_P3BS(1);
  else
    T8_P3BS(2);
#undef T8_P3BS
#undef T8_P3B
#undef T8_P3BN
  return static_cast<int>(hipGetLastError());
}

template int plain_patch3_both_stage<float>(int, int, const T8gpuPlainPlan*, int, int, int, int, FVars<float>, FVars<float>, FVars<float>, const float*,
                                            float, float*, hipStream_t);
template int plain_patch3_both_stage<double>(int, int, const T8gpuPlainPlan*, int, int, int, int, FVars<double>, FVars<double>, FVars<double>,
                                             const double*, double, double*, hipStream_t);

template int plain_patch3_stage<float>(int, int, const T8gpuPlainPlan*, int, int, FVars<float>, FVars<float>, FVars<float>, const float*,
                                       float, float*, bool, bool, hipStream_t);
template int plain_patch3_stage<double>(int, int, const T8gpuPlainPlan*, int, int, FVars<double>, FVars<double>, FVars<double>,
                                        const double*, double, double*, bool, bool, hipStream_t);

}  // namespace t8gpu_hip
