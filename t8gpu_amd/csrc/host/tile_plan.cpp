// tile_plan.cpp -- the face sort / permute + element-window tiling pre-pass of the fused kernels.
//
// Runs once per connectivity rebuild (where the reference runs compute_connectivity_information,
// t8gpu/mesh/mesh_manager.inl:333-481), on the host, from the reference-format arrays
// (face_neighbors, face_normals, face_surfaces). Output: for every tile (a contiguous SFC range of
// owned elements, i.e. a compact window in space)
//   * halo ids    : slots of the outside elements (other tiles' or ghost mirrors) its faces touch,
//   * tile faces  : every face with a side in the tile, re-laid out contiguously (coalesced loads):
//                   packed tile-local (l, r) indices, {nx, ny, nz, area}, original face id,
//   * element CSR : per owned element the tile-local faces it sums, with the sign of its side.
// A face cut by a tile boundary is listed in both tiles (flux evaluated twice, applied to the own
// side only): no atomics, no flux planes in HBM, bitwise-reproducible sums (CSR order = face order).
//
// Host-only: covered by the CPU test-suite through a numpy interpreter of the plan.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_set>
#include <vector>

#include "host_threads.hpp"

namespace {

// A structured patch: kPatchSide x kPatchSide same-size quadrilaterals that are kPatchElems CONSECUTIVE elements in
// Morton order (x = bit 0 of the local index), every one with exactly four interior faces in the canonical listing:
// its +x and +y faces are its own (left = the element, normal exactly +e_x / +e_y, ids fbase + 2 t and fbase + 2 t + 1
// for local index t), its -x and -y faces are the +x / +y faces of the elements across (right = the element), all with
// one area. The kernel needs no face records for such a tile: neighbours inside the patch follow from the lane index,
// the 64 elements across its four sides are listed in `halo` ([-x side by j | +x side by j | -y side by i | +y side by
// i]), and an element adds its four fluxes in ascending face id: (-x, -y in the order of the owning neighbours' indices,
// which inside the patch is a function of (i, j) alone -- patch_y_first), then +x, +y.
//
// 3D (find_patches3): 8 x 8 x 4 same-size hexahedra = 256 consecutive elements in Morton order (x = bit 0, y = bit 1, z =
// bit 2 of every triple), six interior faces each, own faces +x / +y / +z with ids fbase + 3 t (+1, +2), 256 cells across
// the six sides ([-x 32 by j + 8 k | +x 32 | -y 32 by i + 8 k | +y 32 | -z 64 by i + 8 j | +z 64]). The three - faces are
// added in the order of the owning neighbours' indices: pairwise "-y before -x" iff ctz(j) >= ctz(i), "-z before -x" iff
// ctz(k) >= ctz(i), "-z before -y" iff ctz(k) >= ctz(j); where BOTH coordinates of a pair are 0 the patch's position in
// the forest decides -- three flag bits per patch (bit 0: y before x, bit 1: z before x, bit 2: z before y).
// IRREGULAR 3D patches (flag 0x800): the same 8 x 8 x 4 block with sides that are not listed that way -- a periodic wrap
// (the cell across has the lower index on a + side, the higher one on a - side), a coarser neighbour across a - side (the
// finer cell lists a hanging face), a wall. Every cell still has exactly one face per side, of the patch's area, with one
// element (or a wall) behind it; what varies per cell is WHO lists each side face and the order of the six ids. The
// planner writes that down per cell (Patch::info: own-side mask, wall mask, the six sides in ascending face id, the ids of
// the first own interior / wall face) and the kernel evaluates each side face in its listed orientation; the interior of
// the block is as in a regular patch. Blocks next to the domain boundary (13 % of the c5 benchmark mesh) become patches.
constexpr int kPatchSide = 16, kPatchElems = 256, kPatchHalo = 64, kPatchHalo3 = 256, kPatchInfoWords = 512;
struct Patch {
  int32_t e0 = 0, fbase = 0, flags = 0;   // 2D: flags bit 0: element 0 adds its -y face before its -x face; 3D: see above
  int32_t dim = 2, nh = kPatchHalo;
  double  area = 0;
  double  volume = 0;        // > 0: every element of the patch has exactly this volume (t8gpu_plan_plain_patch_volumes)
  int32_t halo[kPatchHalo3];
  // IRREGULAR 3D patches (flags 0x800, see find_patches3): per cell {sides | walls << 6 | summation order << 12, id of its first
  // own interior face or -1, id of its first wall face or -1}; empty for regular patches
  std::vector<int32_t> info;
};

inline int morton2(int i, int j) {
  int t = 0;
  for (int b = 0; b < 4; b++) t |= ((i >> b) & 1) << (2 * b) | ((j >> b) & 1) << (2 * b + 1);
  return t;
}
inline int ctz4(int v) { return v == 0 ? 4 : __builtin_ctz(static_cast<unsigned>(v)); }
// does element (i, j) of a patch add its -y face before its -x face? (the face of the neighbour with the lower index
// first: Morton order of (i, j-1) against (i-1, j)). Element (0, 0) has both neighbours outside: decided per patch.
inline bool patch_y_first(int i, int j) { return ctz4(j) >= ctz4(i); }

struct TilePlan {
  int32_t N = 0, G = 0, F = 0, B = 0, ndim = 3, tmax = 256, fcap = 512;
  int32_t max_halo = 0, max_faces = 0, max_elems = 0, n_interior = 0, max_slots = 0, n_deep = 0;
  std::vector<int32_t>  elem_off, halo_off, face_off;  // [ntiles + 1]
  // (uvector: sized once, then written completely by the parallel per-tile loops -- no value-initialising pass)
  uvector<int32_t>      halo_ids;                      // slots
  uvector<uint32_t>     face_lr;                       // l | r << 16 (tile-local; r = 0xFFFF: wall mirror)
  uvector<double>       face_geo;                      // [nfaces][4] = nx, ny, nz, area
  uvector<int32_t>      face_orig;                     // original face id if this tile reports its speed, else -1
  std::vector<int32_t>  csr_off;                       // [N + 1]
  uvector<uint16_t>     csr_ent;                       // tile-local face | 0x8000 if the element is the RIGHT side
  std::vector<int32_t>  tile_order;                    // interior tiles first, then tiles that read ghost slots
  // compressed forms used by the pipelined kernel
  int32_t lecap = 512;                                 // max own + halo elements per tile
  int32_t ell_width = 0;                               // padded per-element face-list width (multiple of 8)
  uvector<uint16_t>     ell;                           // [ell rows][ell_width], 0xFFFF = padding (generic tiles' elements only)
  std::vector<int32_t>  ell_row0;                      // [ntiles + 1] first ELL row of each tile
  uvector<uint16_t>     geo_idx;                       // per tile face: row of geo_table (13 bits) | direction code << 13
                                                       // (empty if more than 8191 distinct rows)
  std::vector<double>   geo_table;                     // [n_geo][12]: n, area, t1, 0, t2, 0
  // structured patches (see find_patches): tiles the patch kernel evaluates without face records
  int32_t want_patches = 0;
  bool    skip_face_geo = false;                       // leave face_geo empty when the plan has a geometry dictionary
  bool    two_classes = false;                         // no deep / near-boundary split of the interior tiles (flag 32)
  std::vector<Patch>   patches;                        // in element order
  std::vector<int32_t> tile_patch;                     // [ntiles] index into patches, or -1 (generic tile)
  int32_t n_patch_class[3] = {0, 0, 0};                // leading patch tiles of the deep / near / ghost-reading class
  int32_t n_irregular_class[3] = {0, 0, 0};            // ... the last so many of which are irregular patches
};

// Direction code of a unit normal: 2 * axis + (1 if it points along +axis) for an EXACT axis normal (one component
// +-1, the others +-0), 6 otherwise. Faces of Cartesian meshes all have codes < 6; the kernels evaluate such a face
// without the rotation into the face frame when a whole wavefront shares the code.
inline int direction_code(const double* n, int ndim) {
  int axis = -1;
  for (int k = 0; k < ndim; k++) {
    if (n[k] == 0.0) continue;
    if ((n[k] != 1.0 && n[k] != -1.0) || axis >= 0) return 6;
    axis = k;
  }
  return axis < 0 ? 6 : 2 * axis + (n[axis] > 0.0 ? 1 : 0);
}

// The patches of the mesh, found from the reference-format arrays alone. deg / ef: the faces of every owned element in
// ascending face id. Anything unexpected (a hanging face, a wall, a periodic wrap that turns a face round, another
// face numbering) fails a check and leaves the elements to the generic tiles.
void find_patches(TilePlan& P, const int32_t* fn, const double* normals, const double* areas, const std::vector<int32_t>& deg,
                  const std::vector<int32_t>& ef) {
  const int32_t N = P.N, F = P.F, nd = P.ndim;
  int li[kPatchElems], lj[kPatchElems];
  for (int t = 0; t < kPatchElems; t++) {
    li[t] = lj[t] = 0;
    for (int b = 0; b < 4; b++) {
      li[t] |= ((t >> (2 * b)) & 1) << b;
      lj[t] |= ((t >> (2 * b + 1)) & 1) << b;
    }
  }
  auto axis_of = [&](int32_t f) -> int {   // 0: exactly +e_x, 1: exactly +e_y, -1: anything else
    const double* n = normals + static_cast<size_t>(nd) * f;
    if (nd == 3 && n[2] != 0.0) return -1;
    if (n[0] == 1.0 && n[1] == 0.0) return 0;
    if (n[0] == 0.0 && n[1] == 1.0) return 1;
    return -1;
  };
  // Every element is tested as a patch START on its own, in parallel: two patches cannot overlap (the checks pin a start
  // to the origin of an aligned block -- element e0 + 1 must be its +x neighbour, e0 + 2 the +y neighbour, and so on through
  // the Morton pattern), so there is no scan order to respect. Almost every candidate fails at its first element.
  const int32_t ncand = N >= kPatchElems ? N - kPatchElems + 1 : 0;
  std::vector<std::vector<Patch>> found(static_cast<size_t>(host_threads()));
#pragma omp parallel num_threads(host_threads())
  {
    std::vector<Patch>& mine = found[static_cast<size_t>(omp_get_thread_num())];
#pragma omp for schedule(static)
    for (int32_t e0 = 0; e0 < ncand; e0++) {
      Patch pt;
      bool  ok = true;
      for (int t = 0; t < kPatchElems && ok; t++) {
        const int32_t e = e0 + t;
        ok = deg[e + 1] - deg[e] == 4;
        if (!ok) break;
        const int32_t* fl = &ef[deg[e]];
        int32_t        own[2] = {-1, -1}, far[2] = {-1, -1};   // the element's +x / +y faces, its -x / -y faces
        for (int q = 0; q < 4 && ok; q++) {
          const int32_t f = fl[q];
          const int     ax = f < F ? axis_of(f) : -1;
          if (ax < 0) { ok = false; break; }
          const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
          if (l == r) ok = false;
          else if (l == e && own[ax] < 0) own[ax] = f;
          else if (r == e && far[ax] < 0) far[ax] = f;
          else ok = false;
        }
        if (!ok || own[0] < 0 || own[1] < 0 || far[0] < 0 || far[1] < 0) { ok = false; break; }
        if (t == 0) {
          pt.e0    = e0;
          pt.fbase = own[0];
          pt.area  = areas[own[0]];
        }
        for (int q = 0; q < 4; q++) ok = ok && areas[fl[q]] == pt.area;
        ok = ok && own[0] == pt.fbase + 2 * t && own[1] == pt.fbase + 2 * t + 1 && fl[2] == own[0] && fl[3] == own[1];
        if (!ok) break;
        const int  i = li[t], j = lj[t];
        const bool yfirst = fl[0] == far[1];
        if (t == 0) pt.flags = yfirst ? 1 : 0;
        else ok = yfirst == patch_y_first(i, j);
        const int32_t px = fn[2 * static_cast<size_t>(own[0]) + 1], py = fn[2 * static_cast<size_t>(own[1]) + 1];
        const int32_t mx = fn[2 * static_cast<size_t>(far[0])], my = fn[2 * static_cast<size_t>(far[1])];
        auto outside = [&](int32_t s) { return s < e0 || s >= e0 + kPatchElems; };
        // (a - side face whose left element is a ghost is reported -- speed estimate -- by the tile of its right element,
        // which a patch cannot do: such blocks stay generic tiles. Ghosts across the + sides are fine.)
        auto owned_outside = [&](int32_t s) { return s < N && (s < e0 || s >= e0 + kPatchElems); };
        if (i < kPatchSide - 1) ok = ok && px == e0 + morton2(i + 1, j); else { ok = ok && outside(px); pt.halo[16 + j] = px; }
        if (i > 0)              ok = ok && mx == e0 + morton2(i - 1, j); else { ok = ok && owned_outside(mx); pt.halo[j] = mx; }
        if (j < kPatchSide - 1) ok = ok && py == e0 + morton2(i, j + 1); else { ok = ok && outside(py); pt.halo[48 + i] = py; }
        if (j > 0)              ok = ok && my == e0 + morton2(i, j - 1); else { ok = ok && owned_outside(my); pt.halo[32 + i] = my; }
      }
      if (ok) mine.push_back(pt);
    }
  }
  for (auto& v : found)   // (static schedule: ascending e0 overall; the guard is belt and braces -- see above)
    for (const Patch& q : v)
      if (P.patches.empty() || q.e0 >= P.patches.back().e0 + kPatchElems) P.patches.push_back(q);
}

inline int morton3(int i, int j, int k) {   // 8 x 8 x 4: x bits 0, 3, 6; y bits 1, 4, 7; z bits 2, 5
  int t = 0;
  for (int b = 0; b < 3; b++) t |= ((i >> b) & 1) << (3 * b) | ((j >> b) & 1) << (3 * b + 1);
  for (int b = 0; b < 2; b++) t |= ((k >> b) & 1) << (3 * b + 2);
  return t;
}
inline int ctz_or(int v, int big) { return v == 0 ? big : __builtin_ctz(static_cast<unsigned>(v)); }

// 3D structured patches (see the comment at struct Patch). Same policy as find_patches: every expectation is checked per
// element against the arrays, anything else leaves the block to the generic tiles.
void find_patches3(TilePlan& P, const int32_t* fn, const double* normals, const double* areas, const std::vector<int32_t>& deg,
                   const std::vector<int32_t>& ef) {
  const int32_t N = P.N, F = P.F;
  if (P.ndim != 3) return;
  int li[kPatchElems], lj[kPatchElems], lk[kPatchElems];
  for (int t = 0; t < kPatchElems; t++) {
    li[t] = lj[t] = lk[t] = 0;
    for (int b = 0; b < 3; b++) {
      li[t] |= ((t >> (3 * b)) & 1) << b;
      lj[t] |= ((t >> (3 * b + 1)) & 1) << b;
    }
    for (int b = 0; b < 2; b++) lk[t] |= ((t >> (3 * b + 2)) & 1) << b;
  }
  auto axis_of = [&](int32_t f) -> int {   // 0 / 1 / 2: exactly +e_x / +e_y / +e_z, -1: anything else
    const double* n = normals + static_cast<size_t>(3) * f;
    const int nz = (n[0] != 0.0) + (n[1] != 0.0) + (n[2] != 0.0);
    if (nz != 1) return -1;
    for (int a = 0; a < 3; a++)
      if (n[a] == 1.0) return a;
    return -1;
  };
  // The irregular form (see the top of the file): tried where the regular checks fail. Sides are numbered like t8code
  // faces (0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z), which is also the order in which an element lists its own faces.
  auto irregular = [&](int32_t e0, Patch& pt) -> bool {
    pt.dim    = 3;
    pt.nh     = kPatchHalo3;
    pt.e0     = e0;
    pt.fbase  = 0;
    pt.flags  = 0x800;
    pt.volume = 0;
    pt.info.clear();
    int32_t info[3 * kPatchElems];   // (nearly every candidate fails at its first cell: nothing is allocated before it passes)
    const int ext[3] = {8, 8, 4};
    for (int t = 0; t < kPatchElems; t++) {
      const int32_t e = e0 + t;
      if (deg[e + 1] - deg[e] != 6) return false;
      const int32_t* fl      = &ef[deg[e]];
      const int      ijk[3]  = {li[t], lj[t], lk[t]};
      uint32_t       own = 0, wall = 0, order = 0, seen = 0;
      int32_t        first_id = -1, wall_first = -1, id_of[6] = {-1, -1, -1, -1, -1, -1};
      for (int q = 0; q < 6; q++) {
        const int32_t f       = fl[q];
        const bool    is_wall = f >= F;
        const int32_t l = is_wall ? fn[2 * static_cast<size_t>(F) + (f - F)] : fn[2 * static_cast<size_t>(f)];
        const int32_t r = is_wall ? -1 : fn[2 * static_cast<size_t>(f) + 1];
        const double* n = normals + static_cast<size_t>(3) * f;
        int           axis = -1;
        for (int a = 0; a < 3; a++) {
          if (n[a] == 0.0) continue;
          if ((n[a] != 1.0 && n[a] != -1.0) || axis >= 0) return false;
          axis = a;
        }
        if (axis < 0) return false;
        if (t == 0 && q == 0) pt.area = areas[f];
        if (areas[f] != pt.area) return false;
        bool mine_;
        if (l == e && r != e) mine_ = true;
        else if (r == e && l != e) mine_ = false;
        else return false;
        const bool plus = mine_ ? n[axis] > 0.0 : n[axis] < 0.0;   // (the normal points away from the listing element)
        const int  sd   = 2 * axis + (plus ? 1 : 0);
        if (seen & (1u << sd)) return false;
        seen |= 1u << sd;
        order |= static_cast<uint32_t>(sd) << (3 * q);
        id_of[sd] = f;
        if (mine_) {
          own |= 1u << sd;
          if (is_wall) {
            wall |= 1u << sd;
            if (wall_first < 0) wall_first = f;
          } else if (first_id < 0) {
            first_id = f;
          }
        }
        const int32_t nb     = is_wall ? e : (mine_ ? r : l);
        const bool    inside = plus ? ijk[axis] < ext[axis] - 1 : ijk[axis] > 0;
        if (inside) {   // the interior of the block is as in a regular patch
          int nijk[3] = {ijk[0], ijk[1], ijk[2]};
          nijk[axis] += plus ? 1 : -1;
          if (is_wall || mine_ != plus || nb != e0 + morton3(nijk[0], nijk[1], nijk[2])) return false;
        } else {
          if (!is_wall && nb >= e0 && nb < e0 + kPatchElems) return false;
          if (!mine_ && nb >= N) return false;   // (a face listed by a ghost is reported by its right element's tile: not a patch)
          const int u = axis == 0 ? ijk[1] : ijk[0], v = axis == 2 ? ijk[1] : ijk[2];
          const int base = axis == 0 ? (plus ? 32 : 0) : (axis == 1 ? (plus ? 96 : 64) : (plus ? 192 : 128));
          pt.halo[base + u + 8 * v] = nb;
        }
      }
      // an element lists its own faces in side order, interior faces and walls each with consecutive ids
      const uint32_t own_int = own & ~wall;
      for (int sd = 0; sd < 6; sd++) {
        if (own_int & (1u << sd)) {
          if (id_of[sd] != first_id + __builtin_popcount(own_int & ((1u << sd) - 1u))) return false;
        } else if (wall & (1u << sd)) {
          if (id_of[sd] != wall_first + __builtin_popcount(wall & ((1u << sd) - 1u))) return false;
        }
      }
      info[3 * t]     = static_cast<int32_t>(own | (wall << 6) | (order << 12));
      info[3 * t + 1] = first_id;
      info[3 * t + 2] = wall_first;
    }
    pt.info.assign(info, info + 3 * kPatchElems);
    return true;
  };
  // Every element is tested as a patch START on its own, in parallel: two patches cannot overlap (the checks pin a start
  // to the origin of an aligned block -- element e0 + 1 must be its +x neighbour, e0 + 2 the +y neighbour, and so on through
  // the Morton pattern), so there is no scan order to respect. Almost every candidate fails at its first element.
  const int32_t ncand = N >= kPatchElems ? N - kPatchElems + 1 : 0;
  // What both forms demand of a start before anything else: six faces, and the + neighbours of cell (0, 0, 0) are the block's
  // cells (1, 0, 0), (0, 1, 0), (0, 0, 1) = e0 + 1, e0 + 2, e0 + 4, across faces e0 lists itself. Seven of eight elements of a
  // uniform region fail this, from the face -> element pairs alone (8 bytes per face, against 40 once normals and areas are read).
  auto may_start = [&](int32_t e0) -> bool {
    if (deg[e0 + 1] - deg[e0] != 6) return false;
    const int32_t* fl = &ef[deg[e0]];
    unsigned       got = 0;
    for (int q = 0; q < 6; q++) {
      const int32_t f = fl[q];
      if (f >= F || fn[2 * static_cast<size_t>(f)] != e0) continue;
      const int32_t d = fn[2 * static_cast<size_t>(f) + 1] - e0;
      if (d == 1) got |= 1u;
      else if (d == 2) got |= 2u;
      else if (d == 4) got |= 4u;
    }
    return got == 7u;
  };
  std::vector<std::vector<Patch>> found(static_cast<size_t>(host_threads()));
#pragma omp parallel num_threads(host_threads())
  {
    std::vector<Patch>& mine = found[static_cast<size_t>(omp_get_thread_num())];
#pragma omp for schedule(static)
    for (int32_t e0 = 0; e0 < ncand; e0++) {
      if (!may_start(e0)) continue;
      Patch pt;
      pt.dim = 3;
      pt.nh  = kPatchHalo3;
      bool ok = !(P.want_patches & 16);   // (bit 4: every patch in the irregular form -- one kernel, one launch)
      for (int t = 0; t < kPatchElems && ok; t++) {
        const int32_t e = e0 + t;
        ok = deg[e + 1] - deg[e] == 6;
        if (!ok) break;
        const int32_t* fl = &ef[deg[e]];
        int32_t        own[3] = {-1, -1, -1}, far[3] = {-1, -1, -1};
        for (int q = 0; q < 6 && ok; q++) {
          const int32_t f = fl[q];
          const int     ax = f < F ? axis_of(f) : -1;
          if (ax < 0) { ok = false; break; }
          const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
          if (l == r) ok = false;
          else if (l == e && own[ax] < 0) own[ax] = f;
          else if (r == e && far[ax] < 0) far[ax] = f;
          else ok = false;
        }
        for (int a = 0; a < 3; a++) ok = ok && own[a] >= 0 && far[a] >= 0;
        if (!ok) break;
        if (t == 0) {
          pt.e0    = e0;
          pt.fbase = own[0];
          pt.area  = areas[own[0]];
        }
        for (int q = 0; q < 6; q++) ok = ok && areas[fl[q]] == pt.area;
        for (int a = 0; a < 3; a++) ok = ok && own[a] == pt.fbase + 3 * t + a && fl[3 + a] == own[a];
        if (!ok) break;
        const int i = li[t], j = lj[t], k = lk[t];
        // position of every - face among the three (ascending face id = ascending index of the owning neighbour)
        int pos[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++)
          for (int q = 0; q < 3; q++)
            if (fl[q] == far[a]) pos[a] = q;
        const bool yx = pos[1] < pos[0], zx = pos[2] < pos[0], zy = pos[2] < pos[1];
        if (t == 0) pt.flags = (yx ? 1 : 0) | (zx ? 2 : 0) | (zy ? 4 : 0);
        // the rule, with the patch's flags where both coordinates of a pair are 0 (their ctz is then the forest's business)
        const bool ryx = (i == 0 && j == 0) ? (pt.flags & 1) != 0 : ctz_or(j, 8) >= ctz_or(i, 8);
        const bool rzx = (i == 0 && k == 0) ? (pt.flags & 2) != 0 : ctz_or(k, 8) >= ctz_or(i, 8);
        const bool rzy = (j == 0 && k == 0) ? (pt.flags & 4) != 0 : ctz_or(k, 8) >= ctz_or(j, 8);
        ok = ok && yx == ryx && zx == rzx && zy == rzy;
        const int32_t pl[3] = {fn[2 * static_cast<size_t>(own[0]) + 1], fn[2 * static_cast<size_t>(own[1]) + 1], fn[2 * static_cast<size_t>(own[2]) + 1]};
        const int32_t mi[3] = {fn[2 * static_cast<size_t>(far[0])], fn[2 * static_cast<size_t>(far[1])], fn[2 * static_cast<size_t>(far[2])]};
        auto outside       = [&](int32_t s) { return s < e0 || s >= e0 + kPatchElems; };
        auto owned_outside = [&](int32_t s) { return s < N && (s < e0 || s >= e0 + kPatchElems); };   // (see find_patches)
        if (i < 7) ok = ok && pl[0] == e0 + morton3(i + 1, j, k); else { ok = ok && outside(pl[0]); pt.halo[32 + j + 8 * k] = pl[0]; }
        if (i > 0) ok = ok && mi[0] == e0 + morton3(i - 1, j, k); else { ok = ok && owned_outside(mi[0]); pt.halo[j + 8 * k] = mi[0]; }
        if (j < 7) ok = ok && pl[1] == e0 + morton3(i, j + 1, k); else { ok = ok && outside(pl[1]); pt.halo[96 + i + 8 * k] = pl[1]; }
        if (j > 0) ok = ok && mi[1] == e0 + morton3(i, j - 1, k); else { ok = ok && owned_outside(mi[1]); pt.halo[64 + i + 8 * k] = mi[1]; }
        if (k < 3) ok = ok && pl[2] == e0 + morton3(i, j, k + 1); else { ok = ok && outside(pl[2]); pt.halo[192 + i + 8 * j] = pl[2]; }
        if (k > 0) ok = ok && mi[2] == e0 + morton3(i, j, k - 1); else { ok = ok && owned_outside(mi[2]); pt.halo[128 + i + 8 * j] = mi[2]; }
      }
      if (!ok && (P.want_patches & 8)) ok = irregular(e0, pt);
      if (ok) mine.push_back(pt);
    }
  }
  for (auto& v : found)   // (static schedule: ascending e0 overall; the guard is belt and braces -- see above)
    for (const Patch& q : v)
      if (P.patches.empty() || q.e0 >= P.patches.back().e0 + kPatchElems) P.patches.push_back(q);
}

// Small open-addressing hash set / map of int32 keys, emptied in O(1) by moving to the next generation: the faces / halo
// elements of the tile under construction (greedy tiling) and the face -> position, element -> slot maps of a tile's lists.
struct StampSet {
  std::vector<int32_t> key, gen, val;
  int32_t              cur = 0;
  uint32_t             mask;
  int                  shift;
  explicit StampSet(int log2cap)
      : key(size_t(1) << log2cap), gen(size_t(1) << log2cap, -1), val(size_t(1) << log2cap), mask((1u << log2cap) - 1u), shift(32 - log2cap) {}
  void clear() { cur++; }
  uint32_t slot(int32_t k) const {   // where k is, or the free slot where it would go
    uint32_t h = (static_cast<uint32_t>(k) * 2654435761u) >> shift;
    while (gen[h] == cur && key[h] != k) h = (h + 1) & mask;
    return h;
  }
  bool contains(int32_t k) const { return gen[slot(k)] == cur; }
  bool insert(int32_t k, int32_t v = 0) {   // true: was not there
    const uint32_t h = slot(k);
    if (gen[h] == cur) return false;
    gen[h] = cur;
    key[h] = k;
    val[h] = v;
    return true;
  }
  void    set(int32_t k, int32_t v) { val[slot(k)] = v; }   // (k must be there)
  int32_t at(int32_t k) const { return val[slot(k)]; }       // (k must be there)
};

void build(TilePlan& P, const int32_t* fn, const double* normals, const double* areas) {
  const int32_t N = P.N, F = P.F, B = P.B;
  PhaseTimer timer("tile_plan");
  auto       lap = [&](const char* what) { timer.lap(what); };
  // faces of each owned element, in original face order (interior faces first, then walls): counted and placed in
  // parallel over the faces (atomic cursors), then every element's short list is sorted back into ascending face id
  std::vector<int32_t> deg(static_cast<size_t>(N) + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int32_t f = 0; f < F; f++) {
    const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
    if (l < N) __atomic_fetch_add(&deg[l + 1], 1, __ATOMIC_RELAXED);
    if (r < N && r != l) __atomic_fetch_add(&deg[r + 1], 1, __ATOMIC_RELAXED);
  }
  for (int32_t b = 0; b < B; b++) deg[fn[2 * static_cast<size_t>(F) + b] + 1]++;
  for (int32_t e = 0; e < N; e++) deg[e + 1] += deg[e];
  std::vector<int32_t> ef(deg[N]);
  {
    std::vector<int32_t> cur(deg.begin(), deg.end() - 1);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t f = 0; f < F; f++) {
      const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
      if (l < N) ef[__atomic_fetch_add(&cur[l], 1, __ATOMIC_RELAXED)] = f;
      if (r < N && r != l) ef[__atomic_fetch_add(&cur[r], 1, __ATOMIC_RELAXED)] = f;
    }
    for (int32_t b = 0; b < B; b++) ef[cur[fn[2 * static_cast<size_t>(F) + b]]++] = F + b;
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int32_t e = 0; e < N; e++) std::sort(ef.begin() + deg[e], ef.begin() + deg[e + 1]);
  }
  auto side = [&](int32_t f, int which) -> int32_t {
    if (f >= F) return which == 0 ? fn[2 * static_cast<size_t>(F) + (f - F)] : -1;
    return fn[2 * static_cast<size_t>(f) + which];
  };

  lap("element -> faces");
  if (P.want_patches & 1) find_patches(P, fn, normals, areas, deg, ef);
  if ((P.want_patches & 2) && P.patches.empty()) find_patches3(P, fn, normals, areas, deg, ef);
  std::vector<int32_t> patch_at(static_cast<size_t>(N) + 1, -1);   // patch that starts at an element
  for (size_t k = 0; k < P.patches.size(); k++) patch_at[P.patches[k].e0] = static_cast<int32_t>(k);
  lap("patches");
  int32_t most = 0;   // faces of one element
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(max : most)
  for (int32_t e = 0; e < N; e++) most = std::max(most, deg[e + 1] - deg[e]);
  // capacity of the per-thread hash tables: >= 4 x the entries a tile can hold (a tile ends at fcap faces / lecap slots, plus one
  // element's worth; a tile of tmax elements has at most tmax * most faces and twice as many halo elements)
  int log2cap = 12;
  while (log2cap < 30 && (int64_t(1) << log2cap) < 4 * (int64_t(std::max(P.fcap, P.lecap)) + 2 * int64_t(most) + 64)) log2cap++;
  // greedy tiling: grow the element range while elements <= tmax, distinct faces <= fcap and own + halo
  // elements <= lecap (the kernel's LDS window). The halo count is tracked incrementally: an element that
  // joins the tile stops being halo, its neighbours outside the range become halo.
  // A patch is a tile of its own, so the stretches of other elements between patches are tiled independently of each other:
  // in parallel, one run at a time. Long stretches are cut every kRunCut elements as well (a fixed rule: the tiling does not
  // depend on the number of threads); the faces / halo elements of the tile under construction sit in two small hash sets.
  {
    constexpr int32_t kRunCut = 1 << 16;
    std::vector<std::pair<int32_t, int32_t>> runs;
    for (int32_t e = 0; e < N;) {
      if (patch_at[e] >= 0) {
        e += kPatchElems;
        continue;
      }
      const int32_t start = e;
      while (e < N && patch_at[e] < 0 && e - start < kRunCut) e++;
      runs.push_back({start, e});
    }
    std::vector<std::vector<int32_t>> ends(runs.size());
#pragma omp parallel num_threads(host_threads())
    {
      StampSet faces(log2cap), halo(log2cap);
#pragma omp for schedule(dynamic, 1)
      for (int64_t r = 0; r < static_cast<int64_t>(runs.size()); r++) {
        int32_t       e = runs[r].first;
        const int32_t stop = runs[r].second;
        while (e < stop) {
          faces.clear();
          halo.clear();
          int32_t nf = 0, nh = 0;
          const int32_t start = e;
          while (e < stop && e - start < P.tmax) {
            int32_t add = 0, dh = halo.contains(e) ? -1 : 0;
            for (int32_t j = deg[e]; j < deg[e + 1]; j++) {
              if (faces.insert(ef[j])) add++;   // (inserted even if e is rejected below: the tile ends there, the sets with it)
              for (int w = 0; w < 2; w++) {
                const int32_t o = side(ef[j], w);
                if (o >= 0 && (o < start || o > e) && halo.insert(o)) dh++;
              }
            }
            if (e > start && (nf + add > P.fcap || (e - start + 1) + nh + dh > P.lecap)) break;
            nf += add;
            nh += dh;
            e++;
          }
          ends[r].push_back(e);
        }
      }
    }
    P.elem_off.assign(1, 0);
    size_t r = 0;
    for (int32_t e = 0; e < N;) {   // patches and runs alternate in element order
      if (patch_at[e] >= 0) {
        e += kPatchElems;
        P.elem_off.push_back(e);
      } else {
        P.elem_off.insert(P.elem_off.end(), ends[r].begin(), ends[r].end());
        e = runs[r++].second;
      }
    }
  }
  lap("greedy tiling");
  // A tile must fit the kernel's LDS window: own + halo elements <= lecap. The greedy loop counts the halo as it goes, so this
  // holds by construction; the sizing pass below checks it on the exact lists it builds anyway, and only if a tile should
  // ever exceed the window are the offenders halved (per greedy tile, in parallel) and the lists sized again.
  auto halve_oversized = [&]() {
    const int32_t nt0 = static_cast<int32_t>(P.elem_off.size()) - 1;
    std::vector<std::vector<int32_t>> cuts(nt0);   // extra offsets inside a greedy tile (almost always none)
#pragma omp parallel num_threads(host_threads())
    {
      std::vector<int32_t>                     out;
      std::vector<std::pair<int32_t, int32_t>> work;
#pragma omp for schedule(dynamic, 64)
      for (int32_t t = 0; t < nt0; t++) {
        if (patch_at[P.elem_off[t]] >= 0) continue;
        work.assign(1, {P.elem_off[t], P.elem_off[t + 1]});
        while (!work.empty()) {
          const auto [a, b] = work.back();
          work.pop_back();
          out.clear();
          for (int32_t e = a; e < b; e++)
            for (int32_t j = deg[e]; j < deg[e + 1]; j++)
              for (int w = 0; w < 2; w++) {
                const int32_t o = side(ef[j], w);
                if (o >= 0 && (o < a || o >= b)) out.push_back(o);
              }
          std::sort(out.begin(), out.end());
          const int32_t nh = static_cast<int32_t>(std::unique(out.begin(), out.end()) - out.begin());
          if ((b - a) + nh > P.lecap && b - a > 1) {
            const int32_t m = a + (b - a) / 2;
            work.push_back({m, b});
            work.push_back({a, m});
          } else if (b != P.elem_off[t + 1]) {
            cuts[t].push_back(b);
          }
        }
      }
    }
    std::vector<int32_t> off;
    off.reserve(P.elem_off.size());
    off.push_back(0);
    for (int32_t t = 0; t < nt0; t++) {
      off.insert(off.end(), cuts[t].begin(), cuts[t].end());
      off.push_back(P.elem_off[t + 1]);
    }
    P.elem_off.swap(off);
  };
  int32_t ntiles = 0;
  std::vector<uint8_t>              reads_ghost;
  std::vector<std::vector<int32_t>> tfs, halos;   // (the lists of pass 1 are kept for pass 2: sorting them twice was 40 % of this phase)
  for (int attempt = 0;; attempt++) {
    ntiles = static_cast<int32_t>(P.elem_off.size()) - 1;
    // Per-tile lists. Tiles are independent: pass 1 sizes them (faces = sorted distinct faces of the tile's
    // elements, halo = sorted distinct outside elements those faces touch), a prefix sum places them, pass 2
    // fills the arrays in place. Both passes run over the tiles in parallel.
    P.halo_off.assign(static_cast<size_t>(ntiles) + 1, 0);
    P.face_off.assign(static_cast<size_t>(ntiles) + 1, 0);
    P.csr_off.assign(static_cast<size_t>(N) + 1, 0);
    reads_ghost.assign(ntiles, 0);
    P.tile_patch.assign(ntiles, -1);
    for (int32_t t = 0; t < ntiles; t++) P.tile_patch[t] = patch_at[P.elem_off[t]];
    auto tile_lists = [&](int32_t t, std::vector<int32_t>& tf, std::vector<int32_t>& halo, StampSet& set) {
      const int32_t e0 = P.elem_off[t], e1 = P.elem_off[t + 1];
      if (P.tile_patch[t] >= 0) {   // no face records; the 64 elements across the sides in the patch kernel's fixed order
        tf.clear();
        halo.assign(P.patches[P.tile_patch[t]].halo, P.patches[P.tile_patch[t]].halo + P.patches[P.tile_patch[t]].nh);
        return;
      }
      // distinct faces / outside elements through a hash set, then sorted (half the entries of the raw lists are duplicates)
      tf.clear();
      set.clear();
      for (int32_t j = deg[e0]; j < deg[e1]; j++)
        if (set.insert(ef[j])) tf.push_back(ef[j]);
      std::sort(tf.begin(), tf.end());
      halo.clear();
      set.clear();
      for (int32_t f : tf)
        for (int w = 0; w < 2; w++) {
          const int32_t s = side(f, w);
          if (s >= 0 && (s < e0 || s >= e1) && set.insert(s)) halo.push_back(s);
        }
      std::sort(halo.begin(), halo.end());
    };
    tfs.assign(ntiles, {});
    halos.assign(ntiles, {});
    P.max_halo = P.max_faces = P.max_elems = P.max_slots = 0;
#pragma omp parallel num_threads(host_threads())
    {
      StampSet set(log2cap);
#pragma omp for schedule(dynamic, 64)
      for (int32_t t = 0; t < ntiles; t++) {
        std::vector<int32_t>&tf = tfs[t], &halo = halos[t];
        tile_lists(t, tf, halo, set);
        // (an irregular patch keeps its per-cell words where a generic tile keeps face records: 512 entries of face_lr / face_orig)
        const bool irregular = P.tile_patch[t] >= 0 && !P.patches[P.tile_patch[t]].info.empty();
        P.face_off[t + 1] = irregular ? kPatchInfoWords : static_cast<int32_t>(tf.size());
        P.halo_off[t + 1] = static_cast<int32_t>(halo.size());
        reads_ghost[t]    = !halo.empty() && *std::max_element(halo.begin(), halo.end()) >= N;
      }
    }
    for (int32_t t = 0; t < ntiles; t++) {
      const int32_t ne = P.elem_off[t + 1] - P.elem_off[t], nh = P.halo_off[t + 1], nf = P.face_off[t + 1];
      if (P.tile_patch[t] < 0) {   // the maxima size the generic kernels' LDS windows: patch tiles are not theirs
        P.max_halo  = std::max(P.max_halo, nh);
        P.max_faces = std::max(P.max_faces, nf);
        P.max_elems = std::max(P.max_elems, ne);
        P.max_slots = std::max(P.max_slots, ne + nh);
      }
      P.halo_off[t + 1] += P.halo_off[t];
      P.face_off[t + 1] += P.face_off[t];
    }
    if (P.max_slots <= P.lecap || attempt > 0) break;
    halve_oversized();
  }
  lap("per-tile lists (sizes)");
  // Dictionary of the distinct {nx, ny, nz, area} tuples (exact bit patterns) of the mesh's faces: Cartesian AMR meshes
  // have a few dozen, so a tile face carries a 2-byte index instead of 4 float_type values. Built over the ORIGINAL faces
  // (every thread collects the distinct tuples of its share; on a curved mesh each gives up after 8192), then every
  // original face gets its row | direction code << 13, which pass 2 below copies to the tile faces.
  uvector<uint16_t> orig_gidx;
  {
    struct Key {
      uint64_t w[4];
      bool     operator<(const Key& o) const { return std::lexicographical_compare(w, w + 4, o.w, o.w + 4); }
      bool     operator==(const Key& o) const { return std::equal(w, w + 4, o.w); }
    };
    struct KeyHash {
      size_t operator()(const Key& k) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 4; i++) h = (h ^ k.w[i]) * 0xff51afd7ed558ccdull + (h >> 29);
        return static_cast<size_t>(h);
      }
    };
    const int64_t nof = static_cast<int64_t>(F) + B;
    auto key_of = [&](int64_t f, double* g) {
      for (int k = 0; k < 3; k++) g[k] = k < P.ndim ? normals[static_cast<size_t>(P.ndim) * f + k] : 0.0;
      g[3] = areas[f];
      Key key;
      std::memcpy(key.w, g, 32);
      return key;
    };
    constexpr size_t kMaxRows = 8191;   // 13 bits of row index: the upper 3 bits of geo_idx carry the direction code
    std::vector<Key> uniq;
    bool             too_many = false;
    // (a few dozen tuples repeat millions of times: a small direct-mapped cache of recent keys answers nearly every face)
    constexpr int kCache = 256;
    auto slot_of = [](const Key& k) {
      const uint64_t h = (k.w[0] ^ (k.w[1] * 3) ^ (k.w[2] * 7) ^ (k.w[3] * 13)) * 0x9E3779B97F4A7C15ull;
      return static_cast<int>(h >> 56);
    };
#pragma omp parallel num_threads(host_threads())
    {
      std::unordered_set<Key, KeyHash> set;
      std::vector<Key>                 cache(kCache);
      std::vector<uint8_t>             full(kCache, 0);
#pragma omp for schedule(static) nowait
      for (int64_t f = 0; f < nof; f++) {
        if (set.size() > kMaxRows) continue;
        double    g[4];
        const Key key = key_of(f, g);
        const int c   = slot_of(key);
        if (full[c] && key == cache[c]) continue;
        cache[c] = key;
        full[c]  = 1;
        set.insert(key);
      }
#pragma omp critical
      {
        if (set.size() > kMaxRows) too_many = true;
        if (!too_many) uniq.insert(uniq.end(), set.begin(), set.end());
      }
    }
    if (!too_many) {
      std::sort(uniq.begin(), uniq.end());   // (sorted: the table does not depend on the order of discovery)
      uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
      too_many = uniq.size() > kMaxRows;
    }
    if (!too_many && nof > 0) {
      // table row = {nx, ny, nz, area, t1x, t1y, t1z, 0, t2x, t2y, t2z, 0}: the face frame (the reference
      // rebuilds it per face and stage, kernels.cu:174-193) is computed once per distinct normal
      P.geo_table.assign(uniq.size() * 12, 0.0);
      for (size_t i = 0; i < uniq.size(); i++) {
        double* row = &P.geo_table[12 * i];
        std::memcpy(row, uniq[i].w, 32);
        const double* n = row;
        double t1[3] = {n[1], n[2], -n[0]};
        const double dp = n[0] * t1[0] + n[1] * t1[1] + n[2] * t1[2];
        for (int k = 0; k < 3; k++) t1[k] -= dp * n[k];
        const double nrm = std::sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
        for (int k = 0; k < 3; k++) row[4 + k] = t1[k] / nrm;
        row[8]  = n[1] * row[6] - n[2] * row[5];
        row[9]  = n[2] * row[4] - n[0] * row[6];
        row[10] = n[0] * row[5] - n[1] * row[4];
      }
      orig_gidx.resize(static_cast<size_t>(nof));
#pragma omp parallel num_threads(host_threads())
      {
        std::vector<Key>      cache(kCache);
        std::vector<uint32_t> value(kCache, 0xFFFFFFFFu);   // row | code << 13 of the cached key
#pragma omp for schedule(static)
        for (int64_t f = 0; f < nof; f++) {
          double    g[4];
          const Key key = key_of(f, g);
          const int c   = slot_of(key);
          if (value[c] == 0xFFFFFFFFu || !(key == cache[c])) {
            const unsigned row  = static_cast<unsigned>(std::lower_bound(uniq.begin(), uniq.end(), key) - uniq.begin());
            const unsigned code = static_cast<unsigned>(direction_code(g, 3));
            cache[c] = key;
            value[c] = row | (code << 13);
          }
          orig_gidx[f] = static_cast<uint16_t>(value[c]);
        }
      }
    }
  }
  const bool have_dict = !orig_gidx.empty();
  const bool fill_geo  = !(have_dict && P.skip_face_geo);
  lap("geometry dictionary");
  for (int32_t e = 0; e < N; e++) P.csr_off[e + 1] = deg[e + 1];   // one entry per (element, face) incidence
  P.halo_ids.resize(P.halo_off[ntiles]);
  P.face_lr.resize(P.face_off[ntiles]);
  if (fill_geo) P.face_geo.resize(4 * static_cast<size_t>(P.face_off[ntiles]));
  if (have_dict) P.geo_idx.resize(P.face_off[ntiles]);
  P.face_orig.resize(P.face_off[ntiles]);
  P.csr_ent.resize(deg[N]);
#pragma omp parallel num_threads(host_threads())
  {
    std::vector<int32_t> order;
    std::vector<uint8_t> codes;
    StampSet             face_at(log2cap), slot_at(log2cap);   // face id -> position in the tile's face list, outside element -> halo index
#pragma omp for schedule(dynamic, 64)
    for (int32_t t = 0; t < ntiles; t++) {
      const int32_t e0 = P.elem_off[t], e1 = P.elem_off[t + 1], ne = e1 - e0;
      const std::vector<int32_t>&tf = tfs[t], &halo = halos[t];
      if (P.tile_patch[t] >= 0) {
        for (int32_t j = deg[e0]; j < deg[e1]; j++) P.csr_ent[j] = static_cast<uint16_t>(0xFFFFu);   // (never read)
        std::copy(halo.begin(), halo.end(), P.halo_ids.begin() + P.halo_off[t]);
        const std::vector<int32_t>& info = P.patches[P.tile_patch[t]].info;
        if (!info.empty()) {   // face_lr[q0 + c] = sides | walls | order, face_orig[q0 + c] / [q0 + 256 + c] = first own interior / wall id
          const size_t q0 = P.face_off[t];
          for (int c = 0; c < kPatchElems; c++) {
            P.face_lr[q0 + c]                 = static_cast<uint32_t>(info[3 * c]);
            P.face_lr[q0 + kPatchElems + c]   = 0u;
            P.face_orig[q0 + c]               = info[3 * c + 1];
            P.face_orig[q0 + kPatchElems + c] = info[3 * c + 2];
          }
          for (size_t q = q0; q < q0 + kPatchInfoWords; q++) {
            if (fill_geo)
              for (int k = 0; k < 4; k++) P.face_geo[4 * q + k] = 0.0;
            if (have_dict) P.geo_idx[q] = 0;
          }
        }
        continue;
      }
      slot_at.clear();
      for (size_t j = 0; j < halo.size(); j++) slot_at.insert(halo[j], static_cast<int32_t>(j));
      auto loc = [&](int32_t s) -> uint32_t {
        if (s >= e0 && s < e1) return static_cast<uint32_t>(s - e0);
        return static_cast<uint32_t>(ne + slot_at.at(s));
      };
      // Layout of the tile's faces: ascending original id, then inside every block of 256 (one pass of the two-pass
      // kernels = the faces one lane index sees) a stable sort by direction code, so that a wavefront's 64 faces
      // mostly share one direction. The per-element lists below keep ascending original order (only the positions
      // they point at move, and never across a block), so every kernel sums in the same order as before.
      const size_t nft = tf.size();
      order.resize(nft);
      codes.resize(nft);
      for (size_t j = 0; j < nft; j++) {
        order[j] = static_cast<int32_t>(j);
        codes[j] = static_cast<uint8_t>(direction_code(normals + static_cast<size_t>(P.ndim) * tf[j], P.ndim));
      }
      for (size_t b = 0; b < nft; b += 256) {   // stable counting sort by code (0..6) inside the block
        const size_t hi = std::min(nft, b + 256);
        size_t       at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t j = b; j < hi; j++) at[codes[j] + 1]++;
        for (int k = 1; k < 8; k++) at[k] += at[k - 1];
        for (size_t j = b; j < hi; j++) order[b + at[codes[j]]++] = static_cast<int32_t>(j);
      }
      face_at.clear();
      for (size_t j = 0; j < nft; j++) {

        face_at.insert(tf[order[j]], static_cast<int32_t>(j));
      }
      size_t q = P.face_off[t];
      for (size_t jj = 0; jj < nft; jj++) {
        const int32_t f = tf[order[jj]];
        const int32_t l = side(f, 0), r = side(f, 1);
        const uint32_t ll = loc(l), rr = r < 0 ? 0xFFFFu : loc(r);
        P.face_lr[q] = ll | (rr << 16);
        if (fill_geo) {
          for (int k = 0; k < 3; k++) P.face_geo[4 * q + k] = k < P.ndim ? normals[static_cast<size_t>(P.ndim) * f + k] : 0.0;
          P.face_geo[4 * q + 3] = areas[f];
        }
        if (have_dict) P.geo_idx[q] = orig_gidx[f];
        // the tile owning the left element reports the speed estimate (left is always owned or, for a
        // face whose left side is a ghost, the tile of the right element does)
        const int32_t reporter = (l < N) ? l : r;
        P.face_orig[q] = (reporter >= e0 && reporter < e1) ? f : -1;
        q++;
      }
      for (int32_t e = e0; e < e1; e++)
        for (int32_t j = deg[e]; j < deg[e + 1]; j++) {
          const int32_t  f   = ef[j];
          const uint16_t idx = static_cast<uint16_t>(face_at.at(f));
          const bool     right = side(f, 0) != e;
          P.csr_ent[j] = static_cast<uint16_t>(idx | (right ? 0x8000u : 0u));
        }
      std::copy(halo.begin(), halo.end(), P.halo_ids.begin() + P.halo_off[t]);
    }
  }
  lap("per-tile lists");
  // Three classes for the multi-rank step driver: A = tiles that read ghost slots; B = other tiles that read
  // an element owned by an A tile; C = the rest (deep interior). tile_order = C, B, A. A tile of class C
  // depends only on B/C tiles of the previous stage, one of class A only on A/B tiles and the ghosts.
  std::vector<uint8_t> near_boundary(ntiles, 0);
  {
    std::vector<int32_t> owner(static_cast<size_t>(N));
    for (int32_t t = 0; t < ntiles; t++)
      for (int32_t e = P.elem_off[t]; e < P.elem_off[t + 1]; e++) owner[e] = t;
    for (int32_t t = 0; t < ntiles; t++) {
      if (reads_ghost[t]) continue;
      for (int32_t j = P.halo_off[t]; j < P.halo_off[t + 1] && !near_boundary[t]; j++)
        if (reads_ghost[owner[P.halo_ids[j]]]) near_boundary[t] = 1;   // (no ghost ids here: the tile reads none)
    }
  }
  // (inside every class the patch tiles come first: a launch over a range of tile_order is a patch-kernel launch over
  // the patch tiles in it and a generic launch over the rest)
  P.tile_order.clear();
  auto append_class = [&](int cls) {   // regular patches, irregular patches, generic tiles
    for (int pass = 0; pass < 3; pass++) {
      const int32_t before = static_cast<int32_t>(P.tile_order.size());
      for (int32_t t = 0; t < ntiles; t++) {
        const int c    = reads_ghost[t] ? 2 : ((near_boundary[t] && !P.two_classes) ? 1 : 0);
        const int kind = P.tile_patch[t] < 0 ? 2 : (P.patches[P.tile_patch[t]].info.empty() ? 0 : 1);
        if (c == cls && kind == pass) P.tile_order.push_back(t);
      }
      if (pass == 1) {
        P.n_irregular_class[cls] = static_cast<int32_t>(P.tile_order.size()) - before;
        P.n_patch_class[cls]     = static_cast<int32_t>(P.tile_order.size());
      }
    }
  };
  append_class(0);
  P.n_deep = static_cast<int32_t>(P.tile_order.size());
  append_class(1);
  P.n_patch_class[1] -= P.n_deep;
  P.n_interior = static_cast<int32_t>(P.tile_order.size());
  append_class(2);
  P.n_patch_class[2] -= P.n_interior;

  lap("tile classes");
  // fixed-width (ELL) copy of the element face lists: one aligned 16-byte load per 8 entries. Rows exist for the elements of
  // GENERIC tiles only (patch tiles read no face lists: 97 % of the benchmark mesh), in tile-index order; tile t's rows
  // start at ell_row0[t] (tile_desc word 6), element e of the tile is row ell_row0[t] + (e - elem_off[t]).
  int32_t maxdeg = 0;
  for (int32_t e = 0; e < N; e++) maxdeg = std::max(maxdeg, P.csr_off[e + 1] - P.csr_off[e]);
  P.ell_width = std::max(8, (maxdeg + 7) / 8 * 8);
  P.ell_row0.assign(static_cast<size_t>(ntiles) + 1, 0);
  for (int32_t t = 0; t < ntiles; t++)
    P.ell_row0[t + 1] = P.ell_row0[t] + (P.tile_patch[t] >= 0 ? 0 : P.elem_off[t + 1] - P.elem_off[t]);
  P.ell.resize(static_cast<size_t>(P.ell_row0[ntiles]) * P.ell_width);
#pragma omp parallel for num_threads(host_threads()) schedule(dynamic, 64)
  for (int32_t t = 0; t < ntiles; t++) {
    if (P.tile_patch[t] >= 0) continue;
    for (int32_t e = P.elem_off[t]; e < P.elem_off[t + 1]; e++) {
      uint16_t*     row = &P.ell[static_cast<size_t>(P.ell_row0[t] + (e - P.elem_off[t])) * P.ell_width];
      const int32_t n   = P.csr_off[e + 1] - P.csr_off[e];
      for (int32_t c = 0; c < P.ell_width; c++) row[c] = c < n ? P.csr_ent[P.csr_off[e] + c] : static_cast<uint16_t>(0xFFFFu);
    }
  }

  lap("ELL rows");
}

}  // namespace

extern "C" {

// fn = [2F + B] reference face_neighbors (local slots), normals = [ndim * (F + B)], areas = [F + B].
// Returns null if a limit of the packed format is exceeded (tile-local index >= 0xFFFF, > 32767 faces).
// flags bit 0 / 1: cut structured 2D / 3D patches (find_patches, find_patches3) out of the tiling; bit 2: the caller does not
// read `face_geo` when the plan has a geometry dictionary (sizes[11] > 0): it is left empty then
void* t8gpu_plan_plain_create_ex(int32_t N, int32_t G, int32_t F, int32_t B, int32_t ndim, const int32_t* fn,
                                 const double* normals, const double* areas, int32_t tmax, int32_t fcap, int32_t flags) {
  if (N < 0 || F < 0 || B < 0 || ndim < 2 || ndim > 3 || tmax < 1 || tmax > 1024 || fcap < 1) return nullptr;
  TilePlan* P = new TilePlan;
  P->N = N; P->G = G; P->F = F; P->B = B; P->ndim = ndim; P->tmax = tmax; P->fcap = fcap;
  P->want_patches  = flags & 27;   // bit 0: 2D patches (16 x 16), bit 1: 3D patches (8 x 8 x 4), bit 3: irregular 3D patches too, bit 4: no regular 3D ones
  P->skip_face_geo = (flags & 4) != 0;   // bit 2: no face_geo rows if the plan has a geometry dictionary
  P->two_classes   = (flags & 32) != 0;  // bit 5: interior tiles in ONE class (n_deep_tiles = n_interior_tiles): a launch over
                                         // [0, n_interior) is then one kernel launch (the two-lane step driver, stepper.hip)
  build(*P, fn, normals, areas);
  if (P->max_elems + P->max_halo >= 0xFFFF || P->max_faces > 0x7FFE) {
    delete P;
    return nullptr;
  }
  return P;
}
void* t8gpu_plan_plain_create(int32_t N, int32_t G, int32_t F, int32_t B, int32_t ndim, const int32_t* fn,
                              const double* normals, const double* areas, int32_t tmax, int32_t fcap) {
  return t8gpu_plan_plain_create_ex(N, G, F, B, ndim, fn, normals, areas, tmax, fcap, 0);
}
void t8gpu_plan_plain_destroy(void* h) { delete static_cast<TilePlan*>(h); }

// counts[4] = leading patch tiles of the deep / near-boundary / ghost-reading class of tile_order, and their total
void t8gpu_plan_plain_patch_counts(const void* h, int32_t* counts) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  for (int c = 0; c < 3; c++) counts[c] = P->n_patch_class[c];
  counts[3] = static_cast<int32_t>(P->patches.size());
}
// counts[3] = how many of the patch tiles of each class are IRREGULAR patches (flag 0x800; the last ones among the class's patches)
void t8gpu_plan_plain_irregular_counts(const void* h, int32_t* counts) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  for (int c = 0; c < 3; c++) counts[c] = P->n_irregular_class[c];
}
// Optional, after create and before t8gpu_plan_plain_tile_desc: volumes[N] of the owned elements. A patch whose 256 elements
// all have bit for bit the same volume gets it into its descriptor (flag 0x400, words 1 and 3), and the kernels then skip
// the per-element volume load (8 of ~130 bytes per element and stage); any other patch keeps the load. Returns the number
// of patches with a uniform volume.
int32_t t8gpu_plan_plain_patch_volumes(void* h, const double* volumes) {
  TilePlan* P = static_cast<TilePlan*>(h);
  int32_t   n = 0;
  if (!volumes) return 0;
  for (Patch& pt : P->patches) {
    const double v = volumes[pt.e0];
    bool         same = v > 0.0;
    for (int t = 1; t < kPatchElems && same; t++) same = volumes[pt.e0 + t] == v;
    pt.volume = same ? v : 0.0;
    n += same ? 1 : 0;
  }
  return n;
}
// 2 or 3: the kind of the plan's patch tiles (one kind per plan); 0: none
int32_t t8gpu_plan_plain_patch_dim(const void* h) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  return P->patches.empty() ? 0 : P->patches[0].dim;
}

// sizes[16] = {ntiles, n_halo, n_faces, n_csr, max_elems, max_halo, max_faces, n_interior_tiles, N, F,
//              ell_width, n_geo (0: no dictionary), max_slots, n_deep_tiles, n_patches, n_ell_rows}; the maxima are over the
//              generic tiles only
void t8gpu_plan_plain_sizes(const void* h, int64_t* sizes) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  sizes[0] = static_cast<int64_t>(P->elem_off.size()) - 1;
  sizes[1] = static_cast<int64_t>(P->halo_ids.size());
  sizes[2] = static_cast<int64_t>(P->face_lr.size());
  sizes[3] = static_cast<int64_t>(P->csr_ent.size());
  sizes[4] = P->max_elems;
  sizes[5] = P->max_halo;
  sizes[6] = P->max_faces;
  sizes[7] = P->n_interior;
  sizes[8] = P->N;
  sizes[9] = P->F;
  sizes[10] = P->ell_width;
  sizes[11] = static_cast<int64_t>(P->geo_table.size() / 12);
  sizes[12] = P->max_slots;
  sizes[13] = P->n_deep;
  sizes[14] = static_cast<int64_t>(P->patches.size());
  sizes[15] = P->ell_width > 0 ? static_cast<int64_t>(P->ell.size() / static_cast<size_t>(P->ell_width)) : 0;   // ELL rows
}

void t8gpu_plan_plain_compressed(const void* h, uint16_t* ell, uint16_t* geo_idx, double* geo_table) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  if (ell && !P->ell.empty()) std::memcpy(ell, P->ell.data(), P->ell.size() * sizeof(uint16_t));
  if (geo_idx && !P->geo_idx.empty()) std::memcpy(geo_idx, P->geo_idx.data(), P->geo_idx.size() * sizeof(uint16_t));
  if (geo_table && !P->geo_table.empty()) std::memcpy(geo_table, P->geo_table.data(), P->geo_table.size() * sizeof(double));
}

void t8gpu_plan_plain_tile_desc(const void* h, int32_t* tile_desc) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  for (size_t k = 0; k < P->tile_order.size(); k++) {
    const int32_t t = P->tile_order[k];
    int32_t*      d = tile_desc + 8 * k;
    d[0] = P->elem_off[t]; d[1] = P->elem_off[t + 1] - P->elem_off[t];
    d[2] = P->halo_off[t]; d[3] = P->halo_off[t + 1] - P->halo_off[t];
    d[4] = P->face_off[t]; d[5] = P->face_off[t + 1] - P->face_off[t];
    d[6] = P->ell_row0.empty() ? 0 : P->ell_row0[t];   // generic tiles: first row of the tile in `ell`
    d[7] = 0;
    if (!P->tile_patch.empty() && P->tile_patch[t] >= 0) {   // patch tile: {e0, 256, first halo entry, 64 | 256, fbase, 0x100 | 0x200 (3D) | flags, area}
      const Patch& pt = P->patches[P->tile_patch[t]];
      d[4] = pt.info.empty() ? pt.fbase : P->face_off[t];   // (irregular patch: where its per-cell words start in face_lr / face_orig)
      d[5] = 0x100 | (pt.dim == 3 ? 0x200 : 0) | pt.flags;
      std::memcpy(d + 6, &pt.area, 8);
      if (pt.volume > 0.0) {   // uniform volume: flag 0x400, the double in words 1 and 3 (element / halo counts are implied)
        int32_t w[2];
        std::memcpy(w, &pt.volume, 8);
        d[5] |= 0x400;
        d[1] = w[0];
        d[3] = w[1];
      }
    }
  }
}

// The arrays of t8gpu_plan_plain_arrays / _compressed in place (no copy): ptrs[13] = {elem_off, halo_off, face_off, halo_ids,
// face_lr, face_geo, face_orig, csr_off, csr_ent, tile_order, ell, geo_idx, geo_table}, null where empty; sizes as reported by
// t8gpu_plan_plain_sizes. Valid until t8gpu_plan_plain_destroy.
void t8gpu_plan_plain_array_ptrs(const void* h, const void** ptrs) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  auto at = [](const auto& v) -> const void* { return v.empty() ? nullptr : static_cast<const void*>(v.data()); };
  ptrs[0] = at(P->elem_off); ptrs[1] = at(P->halo_off); ptrs[2] = at(P->face_off); ptrs[3] = at(P->halo_ids);
  ptrs[4] = at(P->face_lr); ptrs[5] = at(P->face_geo); ptrs[6] = at(P->face_orig); ptrs[7] = at(P->csr_off);
  ptrs[8] = at(P->csr_ent); ptrs[9] = at(P->tile_order); ptrs[10] = at(P->ell); ptrs[11] = at(P->geo_idx);
  ptrs[12] = at(P->geo_table);
}

void t8gpu_plan_plain_arrays(const void* h, int32_t* elem_off, int32_t* halo_off, int32_t* face_off,
                             int32_t* halo_ids, uint32_t* face_lr, double* face_geo, int32_t* face_orig,
                             int32_t* csr_off, uint16_t* csr_ent, int32_t* tile_order) {
  const TilePlan* P = static_cast<const TilePlan*>(h);
  auto cp = [](auto* dst, const auto& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cp(elem_off, P->elem_off);
  cp(halo_off, P->halo_off);
  cp(face_off, P->face_off);
  cp(halo_ids, P->halo_ids);
  cp(face_lr, P->face_lr);
  cp(face_geo, P->face_geo);
  cp(face_orig, P->face_orig);
  cp(csr_off, P->csr_off);
  cp(csr_ent, P->csr_ent);
  cp(tile_order, P->tile_order);
}

}  // extern "C"
