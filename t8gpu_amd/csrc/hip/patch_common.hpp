// patch_common.hpp -- pieces shared by the structured-patch kernels (kernels_fused_patch.hip: 16 x 16 quadrilateral blocks;
// kernels_fused_patch3.hip: 8 x 8 x 4 hexahedral blocks): faces whose normal is exactly +e_axis.
#ifndef T8GPU_HIP_PATCH_COMMON_HPP
#define T8GPU_HIP_PATCH_COMMON_HPP

#include "fused_common.hpp"

namespace t8gpu_hip {

// Element `i` of a state plane through a 32-bit BYTE offset from the plane's (wave-uniform) base pointer: the compiler then
// uses the scalar-base addressing form of global_load / global_store (one VALU shift for five planes instead of a sign
// extension, a 64-bit shift and five 64-bit adds per group of loads: ~30 VALU instructions per patch). Valid while a plane is
// shorter than 4 GiB -- the plan says so (n_slots_addressed, checked by the launchers; t8gpu_amd/fused.py builds plans of
// larger meshes without patches).
template <class T>
T8_DEV const T& at32(const T* base, unsigned i) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + i * static_cast<unsigned>(sizeof(T)));
}
template <class T>
T8_DEV T& at32(T* base, unsigned i) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + i * static_cast<unsigned>(sizeof(T)));
}

constexpr int kPatchFF = 544;   // flux slots per variable: 256 +x faces, 256 +y faces, 16 -x side, 16 -y side

T8_DEV int patch_morton(int i, int j) {
  int t = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) t |= (((i >> b) & 1) << (2 * b)) | (((j >> b) & 1) << (2 * b + 1));
  return t;
}
T8_DEV int patch_ctz4(int v) { return v == 0 ? 4 : __builtin_ctz(static_cast<unsigned>(v)); }

// velocity components in the frame of a face with normal +e_AXIS (the frame face_basis() gives that normal, see
// kepes_axis_fixed in flux_math.hpp), and a frame flux back to xyz. YSEL: per lane (mixed wavefront of the - sides).
template <class T>
T8_DEV void frame_in(bool y, T vx, T vy, T vz, T& u, T& v, T& w) {
  u = y ? vy : vx;
  v = y ? vx : -vz;
  w = y ? -vz : vy;
}
template <class T>
T8_DEV void frame_out(bool y, const T f[5], T g[5]) {
  g[0] = f[0];
  g[1] = y ? f[2] : f[1];
  g[2] = y ? f[1] : f[3];
  g[3] = y ? -f[3] : -f[2];
  g[4] = f[4];
}

// one face with normal +e_x (AXIS 0) or +e_y (AXIS 1); KEPES from primitives, HLL / HLLC from the conserved states
template <class T, int KIND, int NW>
T8_DEV void patch_face(bool y, const T* wl, const T* wr, T area, T g[5], T& spd) {
  T f[5];
  if (KIND == 0) {
    Prim<T> L, R;
    words_prim<T>(wl, L);
    words_prim<T>(wr, R);
    T uL, vL, wL, uR, vR, wR;
    frame_in<T>(y, L.vx, L.vy, L.vz, uL, vL, wL);
    frame_in<T>(y, R.vx, R.vy, R.vz, uR, vR, wR);
#ifdef T8GPU_EXP_NOMATH
    f[0] = L.rho + R.rho + uL; f[1] = vL + wL + uR; f[2] = vR + wR + L.beta + R.beta; f[3] = L.lrho + R.lrho;
    f[4] = L.p + R.p + L.lbeta + R.lbeta + L.v0 + R.v0 + area;
    spd = f[0];
#else
    kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, spd);
#endif
  } else {
    T a[5], b[5];
    a[0] = wl[0]; a[4] = wl[4];
    b[0] = wr[0]; b[4] = wr[4];
    frame_in<T>(y, wl[1], wl[2], wl[3], a[1], a[2], a[3]);
    frame_in<T>(y, wr[1], wr[2], wr[3], b[1], b[2], b[3]);
    if (KIND == 2)
      hllc_fast<T>(a, b, f, spd);
    else
      hll_fast<T>(a, b, f, spd);
#pragma unroll
    for (int k = 0; k < 5; k++) f[k] = area * f[k];
  }
  frame_out<T>(y, f, g);
}


// the same for a compile-time (or wave-uniform) axis 0 / 1 / 2: the frames of kepes_axis_fixed (flux_math.hpp) with s = +1
template <class T>
T8_DEV void frame_in3(int axis, T vx, T vy, T vz, T& u, T& v, T& w) {
  u = axis == 0 ? vx : (axis == 1 ? vy : vz);
  v = axis == 0 ? -vz : (axis == 1 ? vx : vy);
  w = axis == 0 ? vy : (axis == 1 ? -vz : -vx);
}
template <class T>
T8_DEV void frame_out3(int axis, const T f[5], T g[5]) {
  g[0] = f[0];
  g[1] = axis == 0 ? f[1] : (axis == 1 ? f[2] : -f[3]);
  g[2] = axis == 0 ? f[3] : (axis == 1 ? f[1] : f[2]);
  g[3] = axis == 0 ? -f[2] : (axis == 1 ? -f[3] : f[1]);
  g[4] = f[4];
}
template <class T, int KIND, int NW>
T8_DEV void patch_face3(int axis, const T* wl, const T* wr, T area, T g[5], T& spd) {
  T f[5];
  if (KIND == 0) {
    Prim<T> L, R;
    words_prim<T>(wl, L);
    words_prim<T>(wr, R);
    T uL, vL, wL, uR, vR, wR;
    frame_in3<T>(axis, L.vx, L.vy, L.vz, uL, vL, wL);
    frame_in3<T>(axis, R.vx, R.vy, R.vz, uR, vR, wR);
    kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, spd);
  } else {
    T a[5], b[5];
    a[0] = wl[0]; a[4] = wl[4];
    b[0] = wr[0]; b[4] = wr[4];
    frame_in3<T>(axis, wl[1], wl[2], wl[3], a[1], a[2], a[3]);
    frame_in3<T>(axis, wr[1], wr[2], wr[3], b[1], b[2], b[3]);
    if (KIND == 2)
      hllc_fast<T>(a, b, f, spd);
    else
      hll_fast<T>(a, b, f, spd);
#pragma unroll
    for (int k = 0; k < 5; k++) f[k] = area * f[k];
  }
  frame_out3<T>(axis, f, g);
}


// ---- IRREGULAR 3D patches (tile_plan.cpp: flag 0x800) --------------------------------------------------------------------------
// A side face of the block whose listing is not the regular one. `cell` / `other`: the LDS records of the patch's cell and of the
// element behind the face; plus: the face is on the cell's + side of `axis`; own: the cell lists the face (it is then the
// LEFT operand and the normal points away from it); wall: a boundary face (own, right state = mirror image of the left).
// The frames are those of kepes_axis_fixed (flux_math.hpp) for the normal s e_axis, s = +-1 per lane: with s = +1 and
// own == plus this is patch_face3, value for value.
template <class T>
T8_DEV void frame_in3s(int axis, T s, T vx, T vy, T vz, T& u, T& v, T& w) {
  u = s * (axis == 0 ? vx : (axis == 1 ? vy : vz));
  v = axis == 0 ? -(s * vz) : (axis == 1 ? s * vx : s * vy);
  w = axis == 0 ? vy : (axis == 1 ? -vz : -vx);
}
template <class T>
T8_DEV void frame_out3s(int axis, T s, const T f[5], T g[5]) {
  g[0] = f[0];
  g[1] = axis == 0 ? s * f[1] : (axis == 1 ? s * f[2] : -f[3]);
  g[2] = axis == 0 ? f[3] : (axis == 1 ? s * f[1] : s * f[2]);
  g[3] = axis == 0 ? -(s * f[2]) : (axis == 1 ? -f[3] : s * f[1]);
  g[4] = f[4];
}
template <class T, int KIND, int NW>
T8_DEV void patch_face3g(int axis, bool plus, bool own, bool wall, const T* cell, const T* other, T area, T g[5], T& spd) {
  const T s = (plus == own) ? T(1) : T(-1);
  T       wl[NW], wr[NW], f[5];
  // (the operands are chosen by address, not by value: `cell` / `other` are LDS records)
  rec_load<T, NW>(own ? cell : other, wl);
  rec_load<T, NW>((own && !wall) ? other : cell, wr);   // (wall: the right record is the left one, as in the tile kernels)
  if (KIND == 0) {
    Prim<T> L, R;
    words_prim<T>(wl, L);
    words_prim<T>(wr, R);
    T uL, vL, wL, uR, vR, wR;
    frame_in3s<T>(axis, s, L.vx, L.vy, L.vz, uL, vL, wL);
    frame_in3s<T>(axis, s, R.vx, R.vy, R.vz, uR, vR, wR);
    if (wall) {   // reflective wall (kernels.cu:371-375)
      uR = -uL;
      vR = vL;
      wR = wL;
    }
    kepes_core<T>(L, R, uL, vL, wL, uR, vR, wR, area, f, spd);
  } else {
    T a[5], b[5];
    a[0] = wl[0]; a[4] = wl[4];
    b[0] = wr[0]; b[4] = wr[4];
    frame_in3s<T>(axis, s, wl[1], wl[2], wl[3], a[1], a[2], a[3]);
    frame_in3s<T>(axis, s, wr[1], wr[2], wr[3], b[1], b[2], b[3]);
    if (wall) {   // mirror image of the left state (flux_math.hpp: hll_face)
      b[1] = -a[1];
      b[2] = a[2];
      b[3] = a[3];
    }
    if (KIND == 2)
      hllc_fast<T>(a, b, f, spd);
    else
      hll_fast<T>(a, b, f, spd);
#pragma unroll
    for (int k = 0; k < 5; k++) f[k] = area * f[k];
  }
  frame_out3s<T>(axis, s, f, g);
}
// id of the own face on side sd of a cell with the per-cell words {info, first own interior face, first wall face}
T8_DEV int patch3_own_id(unsigned info, int first_id, int wall_first, int sd) {
  const unsigned own = info & 63u, wall = (info >> 6) & 63u, below = (1u << sd) - 1u;
  return (wall >> sd) & 1u ? wall_first + __builtin_popcount(wall & below) : first_id + __builtin_popcount(own & ~wall & below);
}

}  // namespace t8gpu_hip

#endif  // T8GPU_HIP_PATCH_COMMON_HPP
