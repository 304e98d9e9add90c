// subgrid_plan.cpp -- per-block face lists for the fused Subgrid kernels.
//
// The reference walks the coarse-face list from the FACE side (one CUDA block per coarse face,
// atomicAdd into both neighbours, examples/subgrid/kernels.inl:664-911). The fused kernel works from
// the BLOCK side: one wavefront owns one 4x4x4 block, evaluates every flux its subcells need and
// applies the RK stage, so it needs, per block, the coarse faces that touch it and on which side.
// Input = the reference-format arrays (face_neighbors, face_level_difference, face_neighbor_offset,
// face_normals, face_surfaces); output:
//   plus[N][rank]          : the coarse face on the block's +x/+y/+z side when each of the block's surface
//                            cells sees exactly ONE sub-face of it (same level, wall, or the block is the
//                            fine side of a hanging face): the kernel folds these into its inner-face passes;
//                            -1 otherwise. Bit 31 = the block is the face's RIGHT side.
//   minus[N][rank]         : the same for the -x/-y/-z side: one pass of the kernel evaluates the three of them
//                            (lane = side * sub-faces + sub-face), with no face list to walk
//   bf_off[N+1], bf_ent[]  : per owned block its remaining faces, i.e. those towards FINER blocks (four sub-faces
//                            per surface cell), in original order; bit 31 set when the block is the face's RIGHT side
//   face_rec[F+B][4]       : {left slot, right slot (-1: wall), code, 0} with
//                            code = axis | positive<<2 | hanging<<3 | off0<<4 | off1<<6 | off2<<8
// Normals must be exact +-unit axis vectors -- the reference's kernels require the same
// (kernels.inl:717-750 select the face plane by comparing the normal with +-1.0).
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct SubgridPlan {
  int32_t N = 0, F = 0, B = 0, rank = 3, max_bf = 0, n_addressed = 0;
  std::vector<int32_t> bf_off, bf_ent, face_rec, plus, minus, block_order;  // block_order: interior blocks first
  std::vector<int32_t> fam_first;   // first block of every 2x2x2 family (below)
  std::vector<uint8_t> in_family;
  int32_t n_interior = 0, n_deep = 0;
};
}  // namespace

extern "C" {

void* t8gpu_plan_subgrid_create(int32_t N, int32_t F, int32_t B, int32_t rank, const int32_t* fn,
                                const int32_t* level_diff, const int32_t* nb_offset, const double* normals) {
  if (N < 0 || F < 0 || B < 0 || (rank != 2 && rank != 3)) return nullptr;
  SubgridPlan* P = new SubgridPlan;
  P->N = N; P->F = F; P->B = B; P->rank = rank;
  P->face_rec.assign(4 * (static_cast<size_t>(F) + B), 0);
  std::vector<int32_t> cnt(static_cast<size_t>(N) + 1, 0);
  for (int32_t f = 0; f < F + B; f++) {
    int axis = -1, positive = 0;
    for (int d = 0; d < rank; d++) {
      const double c = normals[static_cast<size_t>(rank) * f + d];
      if (c == 1.0 || c == -1.0) {
        if (axis >= 0) { delete P; return nullptr; }
        axis     = d;
        positive = c > 0;
      } else if (c != 0.0) {
        delete P;
        return nullptr;
      }
    }
    if (axis < 0) { delete P; return nullptr; }
    int32_t code = axis | (positive << 2);
    int32_t l, r = -1;
    if (f < F) {
      l = fn[2 * static_cast<size_t>(f)];
      r = fn[2 * static_cast<size_t>(f) + 1];
      if (level_diff[f] != 0) code |= 1 << 3;
      for (int d = 0; d < rank; d++) code |= (nb_offset[static_cast<size_t>(rank) * f + d] & 3) << (4 + 2 * d);
    } else {
      l = fn[2 * static_cast<size_t>(F) + (f - F)];
    }
    int32_t* rec = &P->face_rec[4 * static_cast<size_t>(f)];
    rec[0] = l; rec[1] = r; rec[2] = code; rec[3] = 0;
    if (l + 1 > P->n_addressed) P->n_addressed = l + 1;
    if (r + 1 > P->n_addressed) P->n_addressed = r + 1;
  }
  // pass 1: which faces fold into the +side passes / the -side pass
  P->plus.assign(static_cast<size_t>(N) * rank, -1);
  P->minus.assign(static_cast<size_t>(N) * rank, -1);
  std::vector<uint8_t> folded_l(static_cast<size_t>(F) + B, 0), folded_r(static_cast<size_t>(F) + B, 0);
  for (int32_t f = 0; f < F + B; f++) {
    const int32_t* rec = &P->face_rec[4 * static_cast<size_t>(f)];
    const int32_t  l = rec[0], r = rec[1], code = rec[2];
    const int      axis = code & 3, positive = (code >> 2) & 1, hanging = (code >> 3) & 1;
    // left block: the face is on its +axis side iff the normal (outward from left) is positive
    std::vector<int32_t>& lside = positive ? P->plus : P->minus;
    if (l < N && lside[static_cast<size_t>(l) * rank + axis] == -1) {
      lside[static_cast<size_t>(l) * rank + axis] = f;
      folded_l[f] = 1;
    }
    // right block (never a wall, never finer than left): on ITS +axis side iff the normal is negative;
    // foldable only at equal level (as the coarse side of a hanging face it sees 4 sub-faces per cell)
    std::vector<int32_t>& rside = positive ? P->minus : P->plus;
    if (r >= 0 && r < N && r != l && !hanging && rside[static_cast<size_t>(r) * rank + axis] == -1) {
      rside[static_cast<size_t>(r) * rank + axis] = f | static_cast<int32_t>(0x80000000u);
      folded_r[f] = 1;
    }
  }
  // pass 2: everything else goes to the generic per-block lists
  std::fill(cnt.begin(), cnt.end(), 0);
  for (int32_t f = 0; f < F + B; f++) {
    const int32_t* rec = &P->face_rec[4 * static_cast<size_t>(f)];
    const int32_t  l = rec[0], r = rec[1];
    if (l < N && !folded_l[f]) cnt[l + 1]++;
    if (r >= 0 && r < N && r != l && !folded_r[f]) cnt[r + 1]++;
  }
  for (int32_t e = 0; e < N; e++) {
    if (cnt[e + 1] > P->max_bf) P->max_bf = cnt[e + 1];
    cnt[e + 1] += cnt[e];
  }
  P->bf_off = cnt;
  P->bf_ent.assign(cnt[N], 0);
  std::vector<int32_t> cur(cnt.begin(), cnt.end() - 1);
  for (int32_t b = 0; b < B; b++) {                      // walls first (reference order: inner, boundary, outer)
    const int32_t l = fn[2 * static_cast<size_t>(F) + b];
    if (l < N && !folded_l[F + b]) P->bf_ent[cur[l]++] = F + b;
  }
  for (int32_t f = 0; f < F; f++) {
    const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
    if (l < N && !folded_l[f]) P->bf_ent[cur[l]++] = f;
    if (r < N && r != l && !folded_r[f]) P->bf_ent[cur[r]++] = f | static_cast<int32_t>(0x80000000u);
  }
  // blocks whose faces all stay among owned blocks can run while the halo exchange is in flight
  std::vector<uint8_t> ghosty(static_cast<size_t>(N), 0);
  for (int32_t f = 0; f < F; f++) {
    const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
    if (l >= N && r < N) ghosty[r] = 1;
    if (r >= N && l < N) ghosty[l] = 1;
  }
  // three classes for the multi-rank step driver (as for plain tiles, tile_plan.cpp): deep interior blocks (no
  // neighbour that touches a ghost block), near-boundary interior blocks, ghost-touching blocks
  std::vector<uint8_t> near(static_cast<size_t>(N), 0);
  for (int32_t f = 0; f < F; f++) {
    const int32_t l = fn[2 * static_cast<size_t>(f)], r = fn[2 * static_cast<size_t>(f) + 1];
    if (l < N && r < N) {
      if (ghosty[l] && !ghosty[r]) near[r] = 1;
      if (ghosty[r] && !ghosty[l]) near[l] = 1;
    }
  }
  // 2x2x2 families (RANK 3): eight CONSECUTIVE owned blocks e .. e + 7 that form a cube in Morton order -- the +x / +y /
  // +z neighbour of block e + w is block e + w + 1 / 2 / 4 at the same level wherever that bit of w is clear --, each of
  // whose 24 outward sides is one foldable coarse face (same level, coarser neighbour or wall) and none of which has a
  // generic face, and all of which are deep interior blocks (so that a multi-rank stage's first class = the cubes + the
// leading part of the rest list). One workgroup of the family kernel takes such a cube: the 12 inner coarse faces are evaluated once,
  // from primitives that are already in LDS, and the far cells of the outward faces are pooled over the eight wavefronts.
  // (RANK 2: the same with four blocks -- a 2x2 square, one wavefront of the 2D family kernel.)
  P->in_family.assign(static_cast<size_t>(N), 0);
  {
    const int NB = 1 << rank;
    auto other_of = [&](int32_t ent, int32_t* hanging) {
      const int32_t* rec = &P->face_rec[4 * static_cast<size_t>(ent & 0x7FFFFFFF)];
      *hanging = (rec[2] >> 3) & 1;
      return ent < 0 ? rec[0] : rec[1];
    };
    auto is_family = [&](int32_t e) {
      for (int w = 0; w < NB; w++) {
        const int32_t b = e + w;
        if (ghosty[b] || near[b]) return false;   // cubes of DEEP blocks only: they then lie inside the first class below
        if (P->bf_off[b + 1] != P->bf_off[b]) return false;
        for (int d = 0; d < rank; d++) {
          const int32_t pe = P->plus[static_cast<size_t>(b) * rank + d], me = P->minus[static_cast<size_t>(b) * rank + d];
          if (pe == -1 || me == -1) return false;
          int32_t hang = 0;
          if (!((w >> d) & 1)) {
            if (other_of(pe, &hang) != b + (1 << d) || hang) return false;
          } else {
            if (other_of(me, &hang) != b - (1 << d) || hang) return false;
          }
        }
      }
      return true;
    };
    for (int32_t e = 0; e + NB <= N;) {
      if (is_family(e)) {
        P->fam_first.push_back(e);
        for (int w = 0; w < NB; w++) P->in_family[e + w] = 1;
        e += NB;
      } else {
        e++;
      }
    }
  }
  for (int32_t e = 0; e < N; e++)
    if (!ghosty[e] && !near[e]) P->block_order.push_back(e);
  P->n_deep = static_cast<int32_t>(P->block_order.size());
  for (int32_t e = 0; e < N; e++)
    if (!ghosty[e] && near[e]) P->block_order.push_back(e);
  P->n_interior = static_cast<int32_t>(P->block_order.size());
  for (int32_t e = 0; e < N; e++)
    if (ghosty[e]) P->block_order.push_back(e);
  return P;
}

void t8gpu_plan_subgrid_destroy(void* h) { delete static_cast<SubgridPlan*>(h); }

/* sizes[8] = {n_entries, max faces per block, F + B, n_interior_blocks, n_deep_blocks, 1 + largest block index referred to,
 *             n_families, n_rest = blocks outside every family} */
void t8gpu_plan_subgrid_sizes(const void* h, int64_t* sizes) {
  const SubgridPlan* P = static_cast<const SubgridPlan*>(h);
  sizes[0] = static_cast<int64_t>(P->bf_ent.size());
  sizes[1] = P->max_bf;
  sizes[2] = static_cast<int64_t>(P->F) + P->B;
  sizes[3] = P->n_interior;
  sizes[4] = P->n_deep;
  sizes[5] = P->n_addressed > P->N ? P->n_addressed : P->N;
  sizes[6] = static_cast<int64_t>(P->fam_first.size());
  sizes[7] = static_cast<int64_t>(P->N) - (static_cast<int64_t>(1) << P->rank) * static_cast<int64_t>(P->fam_first.size());
}

void t8gpu_plan_subgrid_order(const void* h, int32_t* block_order) {
  const SubgridPlan* P = static_cast<const SubgridPlan*>(h);
  if (block_order && !P->block_order.empty()) std::memcpy(block_order, P->block_order.data(), P->block_order.size() * sizeof(int32_t));
}

void t8gpu_plan_subgrid_arrays(const void* h, int32_t* bf_off, int32_t* bf_ent, int32_t* face_rec, int32_t* plus) {
  const SubgridPlan* P = static_cast<const SubgridPlan*>(h);
  if (plus && !P->plus.empty()) std::memcpy(plus, P->plus.data(), P->plus.size() * sizeof(int32_t));
  if (bf_off) std::memcpy(bf_off, P->bf_off.data(), P->bf_off.size() * sizeof(int32_t));
  if (bf_ent && !P->bf_ent.empty()) std::memcpy(bf_ent, P->bf_ent.data(), P->bf_ent.size() * sizeof(int32_t));
  if (face_rec && !P->face_rec.empty()) std::memcpy(face_rec, P->face_rec.data(), P->face_rec.size() * sizeof(int32_t));
}


// What the kernel reads, joined per block so that a wavefront needs ONE dependent level (its record) before it
// can issue every far-cell load, instead of block_order -> face list -> face record -> far cell:
//   block_rec[N][32] in block_order position order (128-byte rows):
//     {block, n generic faces, first entry in bf_rec, 0,  then for d = 0..2 the +d face: other, code, area (2 words),
//      then for d = 0..2 the -d face likewise, then four spare words} -- a wavefront can request every far cell of
//     these six faces as soon as it knows its position
//   bf_rec[n_entries][4], the generic faces of the blocks in the same position order: far, code, area (2 words)
// far = index of the far cell of sub-face (0, 0) in the state arrays (far block * cells per block + its cell there; the
// far block is the left block if this block is the face's right side and vice versa), -1 = wall, -2 (+ / - faces only) =
// not foldable (finer neighbours: those faces are in the generic list);
// code = the face code of face_rec | 1 << 12 when this block is the face's RIGHT side | this block's cell behind
// sub-face (0, 0) << 13 | (two sub-faces per far cell) << 19 | (two sub-faces per own cell) << 20 | (stride of the first
// tangential axis is 4 instead of 1) << 21 | (stride of the second is 4 instead of 16) << 22, so that
//   cell(i, j) = c0 + ((i >> h) << la) + ((j >> h) << lb)
// on either side needs no decoding of the anchor; area = face_surfaces[f] as float (word 0) or double.
// (32-bit cell indices: a rank holds fewer than 2^31 subcells including its ghost blocks.)
static void put_row(const SubgridPlan* P, const double* areas, int float_size, int32_t* dst, int32_t ent) {
  const int32_t S = P->rank == 3 ? 64 : 16;
  {
    const int32_t  f     = ent & 0x7FFFFFFF;
    const bool     right = ent < 0;
    const int32_t* rec   = &P->face_rec[4 * static_cast<size_t>(f)];
    const int32_t  code  = rec[2];
    const int      axis = code & 3, positive = (code >> 2) & 1, hanging = (code >> 3) & 1;
    // the two cells of sub-face (i, j): cell = c0 + ((i >> h) << la) + ((j >> h) << lb), with la / lb the strides of the
    // two tangential axes. On the LEFT block it is the face plane (coordinate 3 or 0 along the axis), on the RIGHT block
    // the stored anchor, and on the coarse (right) side of a hanging face two sub-faces share a cell (kernels.inl:710-758)
    int32_t c0_anchor = 0;
    for (int a = 0; a < 3; a++) c0_anchor += ((code >> (4 + 2 * a)) & 3) << (2 * a);
    const int32_t c0_plane = (positive ? 3 : 0) << (2 * axis);
    const int32_t far_c0 = right ? c0_plane : c0_anchor, own_c0 = right ? c0_anchor : c0_plane;
    const int32_t far_h = right ? 0 : hanging, own_h = right ? hanging : 0;
    const int32_t other = right ? rec[0] : rec[1];
    dst[0] = other < 0 ? -1 : other * S + far_c0;
    dst[1] = code | (right ? 1 << 12 : 0) | own_c0 << 13 | far_h << 19 | own_h << 20 | (axis == 0 ? 1 << 21 : 0) | (axis == 2 ? 1 << 22 : 0);
    dst[2] = dst[3] = 0;
    if (float_size == 4) {
      const float a = static_cast<float>(areas[f]);
      std::memcpy(&dst[2], &a, 4);
    } else {
      std::memcpy(&dst[2], &areas[f], 8);
    }
  }
}

void t8gpu_plan_subgrid_records(const void* h, const double* areas, int float_size, int32_t* block_rec, int32_t* bf_rec) {
  const SubgridPlan* P = static_cast<const SubgridPlan*>(h);
  auto put = [&](int32_t* dst, int32_t ent) { put_row(P, areas, float_size, dst, ent); };
  int32_t first = 0;
  for (int32_t pos = 0; pos < P->N; pos++) {
    const int32_t e   = P->block_order[pos];
    int32_t*      rec = block_rec + 32 * static_cast<size_t>(pos);
    std::memset(rec, 0, 128);
    rec[0] = e;
    rec[1] = P->bf_off[e + 1] - P->bf_off[e];
    rec[2] = first;
    for (int d = 0; d < 3; d++) {
      int32_t* pd = rec + 4 + 4 * d;
      pd[0] = -2;
      if (d < P->rank) {
        const int32_t ent = P->plus[static_cast<size_t>(e) * P->rank + d];
        if (ent != -1) put(pd, ent);
      }
    }
    for (int d = 0; d < 3; d++) {
      int32_t* pd = rec + 16 + 4 * d;
      pd[0] = -2;
      if (d < P->rank) {
        const int32_t ent = P->minus[static_cast<size_t>(e) * P->rank + d];
        if (ent != -1) put(pd, ent);
      }
    }
    for (int32_t j = P->bf_off[e]; j < P->bf_off[e + 1]; j++) put(bf_rec + 4 * static_cast<size_t>(first++), P->bf_ent[j]);
  }
}


// Family records (see t8gpu_plan_subgrid_create). RANK 3: fam_rec[n_families][160] =
//   {first block, 0, 0, 0,
//    12 rows for the outward + faces: row d * 4 + j = the +d face of the j-th block (ascending) that has bit d SET,
//    12 rows for the outward - faces: row 12 + d * 4 + j = the -d face of the j-th block that has bit d CLEAR,
//    12 rows for the inner coarse faces: row 24 + d * 4 + j = the +d face of the j-th block that has bit d CLEAR (only
//    its area is read: the far cell is the sibling's, in LDS)}, rows as in block_rec: {far, code, area (2 words)};
// RANK 2: fam_rec[n_families][64] = {first block, 0, 0, 0, 4 + 4 + 4 rows likewise: row d * 2 + j, 4 + d * 2 + j, 8 + d * 2 + j};
// rest_rec[n_rest][32] = the block_rec rows of the blocks outside every family, in block_order order (same bf_rec).
void t8gpu_plan_subgrid_family_records(const void* h, const double* areas, int float_size, int32_t* fam_rec, int32_t* rest_rec) {
  const SubgridPlan* P = static_cast<const SubgridPlan*>(h);
  const int rank = P->rank, per = 1 << (rank - 1), rows = rank * per, words = rank == 3 ? 160 : 64;   // per: blocks per (axis, side)
  auto expand = [&](int j, int d) {   // a zero bit at position d
    if (rank == 2) return d == 0 ? j << 1 : j;
    return d == 0 ? j << 1 : (d == 1 ? (j & 1) | ((j >> 1) << 2) : j);
  };
  for (size_t q = 0; q < P->fam_first.size(); q++) {
    int32_t*      rec = fam_rec + static_cast<size_t>(words) * q;
    const int32_t e0  = P->fam_first[q];
    std::memset(rec, 0, words * sizeof(int32_t));
    rec[0] = e0;
    for (int d = 0; d < rank; d++)
      for (int j = 0; j < per; j++) {
        const int lo = expand(j, d), hi = lo | (1 << d);
        put_row(P, areas, float_size, rec + 4 + 4 * (d * per + j), P->plus[static_cast<size_t>(e0 + hi) * rank + d]);
        put_row(P, areas, float_size, rec + 4 + 4 * (rows + d * per + j), P->minus[static_cast<size_t>(e0 + lo) * rank + d]);
        put_row(P, areas, float_size, rec + 4 + 4 * (2 * rows + d * per + j), P->plus[static_cast<size_t>(e0 + lo) * rank + d]);
      }
  }
  // the remaining blocks keep their block records (and their rows of bf_rec: `first` counts every block's entries)
  int32_t first = 0;
  size_t  r     = 0;
  for (int32_t pos = 0; pos < P->N; pos++) {
    const int32_t e = P->block_order[pos];
    const int32_t n = P->bf_off[e + 1] - P->bf_off[e];
    if (!P->in_family[e]) {
      int32_t* rec = rest_rec + 32 * r++;
      std::memset(rec, 0, 128);
      rec[0] = e;
      rec[1] = n;
      rec[2] = first;
      for (int d = 0; d < 3; d++) {
        int32_t *pd = rec + 4 + 4 * d, *md = rec + 16 + 4 * d;
        pd[0] = md[0] = -2;
        if (d < P->rank) {
          const int32_t pe = P->plus[static_cast<size_t>(e) * P->rank + d], me = P->minus[static_cast<size_t>(e) * P->rank + d];
          if (pe != -1) put_row(P, areas, float_size, pd, pe);
          if (me != -1) put_row(P, areas, float_size, md, me);
        }
      }
    }
    first += n;
  }
}

}  // extern "C"
