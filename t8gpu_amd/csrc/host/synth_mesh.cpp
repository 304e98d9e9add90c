// synth_mesh.cpp -- t8code-free provider of the hot path's INPUT CONTRACT.
//
// The reference builds its face lists by walking a t8code forest
// (t8gpu/mesh/mesh_manager.inl:333-481, subgrid_mesh_manager.inl:560-961).
// t8code is not available here, so this file generates the same array formats
// analytically for a single-tree periodic (or walled) unit square / cube:
//   * leaves of a 2:1 face-balanced quadtree/octree in Morton (t8code SFC) order,
//   * face_neighbors / face_normals / face_surfaces (+ face_level_difference,
//     face_neighbor_offset for Subgrid meshes) with the reference's listing rule
//     (SURVEY quirk Q4): a same-level face is listed once by the lower-index
//     element, a hanging face by the finer element; normal outward from the
//     listing element; t8code face numbering 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z,
//   * an SFC-contiguous k-way partition with ghost mirror slots appended after
//     the owned elements and per-peer send/recv lists for the halo exchange.
// Cross-rank faces are listed on BOTH ranks with the single-rank orientation
// (redundant flux evaluation instead of remote atomics).
//
// Host-only, no HIP: the CPU test-suite and the multi-rank gloo tests use it too.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_threads.hpp"

#include "t8gpu_host.h"

namespace {

struct Leaf {
  int32_t  level;
  uint32_t c[3];
};

struct Mesh {
  int    dim = 2, base = 1, lmax = 1, periodic = 1;
  double band = 0, shrink = 1;
  std::vector<Leaf>    leaves;
  std::vector<int32_t> owner;  // finest-level grid -> leaf index
  // set by the one-pass adapt: the forest this one was adapted from and the first old element of every new element
  const void*          adapted_from = nullptr;
  std::vector<int32_t> adapt_data;

  size_t grid_index(const uint32_t p[3]) const {
    size_t i = dim == 3 ? p[2] : 0;
    i        = (i << lmax) + p[1];
    return (i << lmax) + p[0];
  }

  bool wants_refine(const Leaf& l) const {
    if (l.level < base) return true;
    if (l.level >= lmax) return false;
    const int    ax = dim - 1;
    const double h  = std::ldexp(1.0, -l.level);
    const double c  = (l.c[ax] + 0.5) * h;
    const double w  = band * std::pow(shrink, l.level - base);
    const double d0 = std::max(0.0, std::fabs(c - 0.25) - 0.5 * h);
    const double d1 = std::max(0.0, std::fabs(c - 0.75) - 0.5 * h);
    return std::min(d0, d1) < w;
  }

  void push_children(std::vector<Leaf>& out, const Leaf& l) const {
    for (int ch = 0; ch < (1 << dim); ch++) {
      Leaf k;
      k.level = l.level + 1;
      for (int d = 0; d < 3; d++) k.c[d] = d < dim ? 2 * l.c[d] + ((ch >> d) & 1) : 0;
      out.push_back(k);
    }
  }

  void build_rec(const Leaf& l) {
    if (wants_refine(l)) {
      for (int ch = 0; ch < (1 << dim); ch++) {
        Leaf k;
        k.level = l.level + 1;
        for (int d = 0; d < 3; d++) k.c[d] = d < dim ? 2 * l.c[d] + ((ch >> d) & 1) : 0;
        build_rec(k);
      }
    } else {
      leaves.push_back(l);
    }
  }

  void fill_owner() {
    // (the leaves tile the domain, so every entry is rewritten below: no fill; leaves own disjoint boxes of the grid)
    owner.resize(static_cast<size_t>(1) << (dim * lmax));
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int64_t e = 0; e < static_cast<int64_t>(leaves.size()); e++) {
      const Leaf&    l = leaves[e];
      const uint32_t s = 1u << (lmax - l.level);
      const uint32_t z0 = dim == 3 ? l.c[2] * s : 0, z1 = dim == 3 ? z0 + s : 1;
      for (uint32_t z = z0; z < z1; z++)
        for (uint32_t y = l.c[1] * s; y < (l.c[1] + 1) * s; y++) {
          uint32_t p[3] = {l.c[0] * s, y, z};
          int32_t* row  = owner.data() + grid_index(p);
          std::fill(row, row + s, static_cast<int32_t>(e));
        }
    }
  }

  // leaf across face f of leaf e, or -1 at a wall.
  int32_t across(size_t e, int f) const {
    const Leaf&    l   = leaves[e];
    const uint32_t s   = 1u << (lmax - l.level);
    const uint32_t ext = 1u << lmax;
    const int      d   = f / 2;
    uint32_t       p[3] = {l.c[0] * s, l.c[1] * s, dim == 3 ? l.c[2] * s : 0};
    if (f & 1) {
      if (p[d] + s >= ext) {
        if (!periodic) return -1;
        p[d] = 0;
      } else {
        p[d] += s;
      }
    } else {
      if (p[d] == 0) {
        if (!periodic) return -1;
        p[d] = ext - 1;
      } else {
        p[d] -= 1;
      }
    }
    return owner[grid_index(p)];
  }

  void build() {
    leaves.clear();
    Leaf root{0, {0, 0, 0}};
    build_rec(root);
    fill_owner();
    // 2:1 face balance: a fine leaf marks any face neighbour coarser by >= 2.
    for (;;) {
      std::vector<uint8_t> mark(leaves.size(), 0);
      int                  any = 0;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(| : any)
      for (int64_t e = 0; e < static_cast<int64_t>(leaves.size()); e++)
        for (int f = 0; f < 2 * dim; f++) {
          const int32_t nb = across(static_cast<size_t>(e), f);
          if (nb >= 0 && leaves[nb].level < leaves[e].level - 1) {
#pragma omp atomic write
            mark[nb] = 1;
            any |= 1;
          }
        }
      if (!any) break;
      std::vector<Leaf> next;
      next.reserve(leaves.size() + leaves.size() / 8);
      for (size_t e = 0; e < leaves.size(); e++) {
        if (mark[e])
          push_children(next, leaves[e]);
        else
          next.push_back(leaves[e]);
      }
      leaves.swap(next);
      fill_owner();
    }
  }
};

struct Part {
  const Mesh* m = nullptr;
  int         rank = 0, nranks = 1, subgrid = 0, ndim = 3;
  int64_t     first = 0;
  int32_t     N = 0, G = 0, F = 0, B = 0;
  uvector<int32_t> fn;       // 2F + B      (uvector: sized, then written completely by parallel loops)
  uvector<double>  normals;  // ndim * (F + B)
  uvector<double>  areas;    // F + B
  std::vector<int32_t> level_diff, nb_offset;
  std::vector<int64_t> ghost_global;
  std::vector<int32_t> ghost_owner;
  std::vector<int32_t> peers, recv_off, send_off, send_idx;  // offsets have n_peers + 1 entries
};

int owner_rank(int64_t g, int64_t n, int nranks) {
  // inverse of first(r) = floor(n * r / nranks)
  int r = static_cast<int>((static_cast<__int128>(g + 1) * nranks - 1) / n);
  while (r > 0 && (n * r) / nranks > g) r--;
  while (r + 1 < nranks && (n * (r + 1)) / nranks <= g) r++;
  return r;
}

void build_part(Part& P) {
  const Mesh&   M   = *P.m;
  const int     dim = M.dim;
  const int64_t n   = static_cast<int64_t>(M.leaves.size());
  const int64_t lo = (n * P.rank) / P.nranks, hi = (n * (P.rank + 1)) / P.nranks;
  P.first = lo;
  P.N     = static_cast<int32_t>(hi - lo);
  PhaseTimer timer("synth part");

  struct RawFace {
    int64_t l, r;
    int     f;
  };
  uvector<RawFace> faces, walls;   // (sized by the counting pass, written by the second)
  // the faces a rank lists, in element order then face order (the single-rank listing rule). Two passes over the
  // elements that can touch the rank's range -- count, prefix sum, fill -- both parallel over elements.
  // (the counting pass keeps the neighbours it looked up -- six random reads of the 512 MB lookup grid per element -- for the
  // filling pass)
  uvector<int32_t> nbs(static_cast<size_t>(n) * 2 * dim);
  auto visit = [&](int64_t e, RawFace* fo, RawFace* wo, int32_t& nf, int32_t& nw) {
    const bool mine = e >= lo && e < hi;
    nf = nw = 0;
    int32_t* const cache = nbs.data() + static_cast<size_t>(e) * 2 * dim;
    for (int f = 0; f < 2 * dim; f++) {
      const int32_t nb = fo || wo ? cache[f] : (cache[f] = M.across(static_cast<size_t>(e), f));
      if (nb < 0) {
        if (mine) {
          if (wo) wo[nw] = {e, -1, f};
          nw++;
        }
        continue;
      }
      const bool nb_mine = nb >= lo && nb < hi;
      if (!mine && !nb_mine) continue;
      const int le = M.leaves[e].level, ln = M.leaves[nb].level;
      if (ln > le) continue;  // several finer neighbours: they list the face
      if (nb > e || (nb < e && ln < le)) {
        if (fo) fo[nf] = {e, nb, f};
        nf++;
      }
    }
  };
  {
    std::vector<int32_t> cf(static_cast<size_t>(n) + 1, 0), cw(static_cast<size_t>(n) + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int64_t e = 0; e < n; e++) visit(e, nullptr, nullptr, cf[e + 1], cw[e + 1]);
    std::vector<int64_t> of(static_cast<size_t>(n) + 1, 0), ow(static_cast<size_t>(n) + 1, 0);
    for (int64_t e = 0; e < n; e++) {
      of[e + 1] = of[e] + cf[e + 1];
      ow[e + 1] = ow[e] + cw[e + 1];
    }
    faces.resize(static_cast<size_t>(of[n]));
    walls.resize(static_cast<size_t>(ow[n]));
#pragma omp parallel for num_threads(host_threads()) schedule(static)
    for (int64_t e = 0; e < n; e++) {
      if (cf[e + 1] == 0 && cw[e + 1] == 0) continue;
      int32_t a, b;
      visit(e, faces.data() + of[e], walls.data() + ow[e], a, b);
    }
  }
  timer.lap("face listing");
  // ghosts: referenced elements outside [lo, hi), sorted by global index
  std::vector<int64_t> gh;
  if (P.nranks > 1) {   // (one rank owns every element: nothing to look for)
    for (const RawFace& rf : faces) {
      if (rf.l < lo || rf.l >= hi) gh.push_back(rf.l);
      if (rf.r < lo || rf.r >= hi) gh.push_back(rf.r);
    }
  }
  std::sort(gh.begin(), gh.end());
  gh.erase(std::unique(gh.begin(), gh.end()), gh.end());
  P.ghost_global = gh;
  P.G            = static_cast<int32_t>(gh.size());
  P.ghost_owner.resize(gh.size());
  for (size_t i = 0; i < gh.size(); i++) P.ghost_owner[i] = owner_rank(gh[i], n, P.nranks);
  auto local = [&](int64_t g) -> int32_t {
    if (g >= lo && g < hi) return static_cast<int32_t>(g - lo);
    return P.N + static_cast<int32_t>(std::lower_bound(gh.begin(), gh.end(), g) - gh.begin());
  };

  timer.lap("ghost list");
  P.F = static_cast<int32_t>(faces.size());
  P.B = static_cast<int32_t>(walls.size());
  P.fn.resize(2 * static_cast<size_t>(P.F) + P.B);
  P.normals.resize(static_cast<size_t>(P.ndim) * (P.F + P.B));
  P.areas.resize(static_cast<size_t>(P.F) + P.B);
  if (P.subgrid) {
    P.level_diff.resize(P.F);
    P.nb_offset.assign(static_cast<size_t>(dim) * P.F, 0);
  }
  auto geom = [&](size_t slot, int64_t e, int f) {
    const double h = std::ldexp(1.0, -M.leaves[e].level);
    for (int d = 0; d < P.ndim; d++) P.normals[P.ndim * slot + d] = d == f / 2 ? ((f & 1) ? 1.0 : -1.0) : 0.0;
    P.areas[slot]                    = dim == 3 ? h * h : h;
  };
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int32_t i = 0; i < P.F; i++) {
    const RawFace& rf = faces[i];
    P.fn[2 * static_cast<size_t>(i)]     = local(rf.l);
    P.fn[2 * static_cast<size_t>(i) + 1] = local(rf.r);
    geom(i, rf.l, rf.f);
    if (P.subgrid) {
      // subgrid_mesh_manager.inl:587-647: anchor inside the right block.
      const Leaf& L = M.leaves[rf.l];
      const Leaf& R = M.leaves[rf.r];
      P.level_diff[i] = R.level - L.level;
      const int ax = rf.f / 2;
      for (int d = 0; d < dim; d++) {
        int o = 0;
        if (d == ax)
          o = (rf.f & 1) ? 0 : 3;
        else if (R.level < L.level)
          o = 2 * static_cast<int>(L.c[d] & 1u);
        P.nb_offset[static_cast<size_t>(dim) * i + d] = o;
      }
    }
  }
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int32_t i = 0; i < P.B; i++) {
    P.fn[2 * static_cast<size_t>(P.F) + i] = local(walls[i].l);
    geom(static_cast<size_t>(P.F) + i, walls[i].l, walls[i].f);
  }

  timer.lap("face arrays");
  // peers + recv ranges (ghost slots are grouped by owner because sorted by global id)
  P.peers.clear();
  P.recv_off.assign(1, 0);
  for (int32_t g = 0; g < P.G; g++) {
    if (P.peers.empty() || P.peers.back() != P.ghost_owner[g]) {
      if (!P.peers.empty()) P.recv_off.push_back(g);
      P.peers.push_back(P.ghost_owner[g]);
    }
  }
  if (!P.peers.empty()) P.recv_off.push_back(P.G);
  // send lists: my elements that share a listed face with an element of peer p
  std::vector<std::vector<int32_t>> send(P.peers.size());
  auto peer_slot = [&](int r) { return static_cast<size_t>(std::lower_bound(P.peers.begin(), P.peers.end(), r) - P.peers.begin()); };
  if (!P.peers.empty()) {
    for (int32_t i = 0; i < P.F; i++) {
      const int32_t l = P.fn[2 * static_cast<size_t>(i)], r = P.fn[2 * static_cast<size_t>(i) + 1];
      if (l >= P.N && r < P.N) send[peer_slot(P.ghost_owner[l - P.N])].push_back(r);
      if (r >= P.N && l < P.N) send[peer_slot(P.ghost_owner[r - P.N])].push_back(l);
    }
  }
  P.send_off.assign(1, 0);
  P.send_idx.clear();
  for (auto& s : send) {
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    P.send_idx.insert(P.send_idx.end(), s.begin(), s.end());
    P.send_off.push_back(static_cast<int32_t>(P.send_idx.size()));
  }
  timer.lap("peers + send lists");
}

// Kelvin-Helmholtz initial state at a point, SURVEY 8d (restating the values of
// examples/subgrid/solver.inl:84-103 (2D) and :35-56 (3D), quirk Q8 included:
// rho_v1 = -/+0.5 is NOT multiplied by rho).
void kh_state(int dim, const double x[3], double out[5]) {
  const double kPi = 3.14159265358979323846;
  const double sigma = 0.05 / std::sqrt(2.0);
  const double gamma = 1.4;
  const double s     = x[dim - 1];
  const bool   in    = std::fabs(s - 0.5) < 0.25;
  const double rho   = in ? 2.0 : 1.0;
  const double a = (s - 0.75) / (2 * sigma), b = (s - 0.25) / (2 * sigma);
  const double pert = rho * (0.1 * std::sin(4.0 * kPi * (x[0] - 0.5)) * (std::exp(-a * a) + std::exp(-b * b)));
  out[0] = rho;
  out[1] = in ? -0.5 : 0.5;
  out[2] = dim == 2 ? pert : 0.0;
  out[3] = dim == 3 ? pert : 0.0;
  out[4] = 2.5 / (gamma - 1.0) + 0.5 * (out[1] * out[1] + out[2] * out[2] + out[3] * out[3]) / rho;
}

}  // namespace

extern "C" {

void* t8gpu_synth_mesh_create(int dim, int base_level, int max_level, double band, double shrink, int periodic) {
  if (dim < 2 || dim > 3 || base_level < 1 || max_level < base_level || dim * max_level > 28) return nullptr;
  Mesh* m     = new Mesh;
  m->dim      = dim;
  m->base     = base_level;
  m->lmax     = max_level;
  m->band     = band;
  m->shrink   = shrink;
  m->periodic = periodic;
  m->build();
  return m;
}
void    t8gpu_synth_mesh_destroy(void* h) { delete static_cast<Mesh*>(h); }
int t8gpu_synth_mesh_dim(const void* h) { return static_cast<const Mesh*>(h)->dim; }
int64_t t8gpu_synth_mesh_num_elements(const void* h) { return static_cast<int64_t>(static_cast<const Mesh*>(h)->leaves.size()); }
int     t8gpu_synth_mesh_finest_level(const void* h) {
  int l = 0;
  for (const Leaf& k : static_cast<const Mesh*>(h)->leaves) l = std::max(l, k.level);
  return l;
}

void* t8gpu_synth_part_create(const void* mesh, int rank, int nranks, int subgrid, int normal_dim) {
  const Mesh* m = static_cast<const Mesh*>(mesh);
  if (!m || rank < 0 || rank >= nranks || normal_dim < m->dim || normal_dim > 3) return nullptr;
  Part* p    = new Part;
  p->m       = m;
  p->rank    = rank;
  p->nranks  = nranks;
  p->subgrid = subgrid;
  p->ndim    = normal_dim;
  build_part(*p);
  return p;
}
void t8gpu_synth_part_destroy(void* h) { delete static_cast<Part*>(h); }
// frees the connectivity arrays of a partition (the caller has copied them) but keeps what the element queries below need
// (first element, ghost list): the initial condition of a partition can then be evaluated when -- and if -- it is asked for
void t8gpu_synth_part_release_arrays(void* h) {
  Part* p = static_cast<Part*>(h);
  uvector<int32_t>().swap(p->fn);
  uvector<double>().swap(p->normals);
  uvector<double>().swap(p->areas);
  std::vector<int32_t>().swap(p->level_diff);
  std::vector<int32_t>().swap(p->nb_offset);
}

// counts[8] = {N, G, F, B, n_peers, n_send, first_global_lo32, first_global_hi32}
void t8gpu_synth_part_counts(const void* h, int64_t* counts) {
  const Part* p = static_cast<const Part*>(h);
  counts[0] = p->N;
  counts[1] = p->G;
  counts[2] = p->F;
  counts[3] = p->B;
  counts[4] = static_cast<int64_t>(p->peers.size());
  counts[5] = static_cast<int64_t>(p->send_idx.size());
  counts[6] = p->first;
  counts[7] = static_cast<int64_t>(p->m->leaves.size());
}

// Any output pointer may be null. normals/areas/volumes/centers are double; the
// caller converts to float_type (the reference casts t8code doubles the same way,
// mesh_manager.inl:400-407).
void t8gpu_synth_part_connectivity(const void* h, int32_t* face_neighbors, double* normals, double* areas,
                                   int32_t* level_diff, int32_t* nb_offset) {
  const Part* p = static_cast<const Part*>(h);
  if (face_neighbors) std::memcpy(face_neighbors, p->fn.data(), p->fn.size() * sizeof(int32_t));
  if (normals) std::memcpy(normals, p->normals.data(), p->normals.size() * sizeof(double));
  if (areas) std::memcpy(areas, p->areas.data(), p->areas.size() * sizeof(double));
  if (level_diff && !p->level_diff.empty()) std::memcpy(level_diff, p->level_diff.data(), p->level_diff.size() * sizeof(int32_t));
  if (nb_offset && !p->nb_offset.empty()) std::memcpy(nb_offset, p->nb_offset.data(), p->nb_offset.size() * sizeof(int32_t));
}

// The same arrays without a copy: ptrs[5] = {face_neighbors, normals, areas, face_level_difference, face_neighbor_offset}
// (null where empty), valid until t8gpu_synth_part_release_arrays / _destroy of this partition.
void t8gpu_synth_part_connectivity_ptrs(const void* h, const void** ptrs) {
  const Part* p = static_cast<const Part*>(h);
  ptrs[0] = p->fn.empty() ? nullptr : p->fn.data();
  ptrs[1] = p->normals.empty() ? nullptr : p->normals.data();
  ptrs[2] = p->areas.empty() ? nullptr : p->areas.data();
  ptrs[3] = p->level_diff.empty() ? nullptr : p->level_diff.data();
  ptrs[4] = p->nb_offset.empty() ? nullptr : p->nb_offset.data();
}

// per owned+ghost element: level, volume, centre (N + G entries; centre is [N+G][3])
void t8gpu_synth_part_elements(const void* h, int32_t* level, double* volume, double* centre) {
  const Part* p   = static_cast<const Part*>(h);
  const Mesh& M   = *p->m;
  const int   tot = p->N + p->G;
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int i = 0; i < tot; i++) {
    const int64_t g = i < p->N ? p->first + i : p->ghost_global[i - p->N];
    const Leaf&   l = M.leaves[g];
    const double  hh = std::ldexp(1.0, -l.level);
    if (level) level[i] = l.level;
    if (volume) volume[i] = M.dim == 3 ? hh * hh * hh : hh * hh;
    if (centre)
      for (int d = 0; d < 3; d++) centre[3 * static_cast<size_t>(i) + d] = d < M.dim ? (l.c[d] + 0.5) * hh : 0.0;
  }
}

// halo plan: ghost_global[G], ghost_owner[G], peers[n_peers], recv_off[n_peers+1] (ghost-slot
// ranges relative to N), send_off[n_peers+1], send_idx[n_send] (local element indices).
void t8gpu_synth_part_halo(const void* h, int64_t* ghost_global, int32_t* ghost_owner, int32_t* peers,
                           int32_t* recv_off, int32_t* send_off, int32_t* send_idx) {
  const Part* p = static_cast<const Part*>(h);
  if (ghost_global) std::copy(p->ghost_global.begin(), p->ghost_global.end(), ghost_global);
  if (ghost_owner) std::copy(p->ghost_owner.begin(), p->ghost_owner.end(), ghost_owner);
  if (peers) std::copy(p->peers.begin(), p->peers.end(), peers);
  if (recv_off) std::copy(p->recv_off.begin(), p->recv_off.end(), recv_off);
  if (send_off) std::copy(p->send_off.begin(), p->send_off.end(), send_off);
  if (send_idx) std::copy(p->send_idx.begin(), p->send_idx.end(), send_idx);
}

// Kelvin-Helmholtz initial condition for owned AND ghost elements.
// cells_per_dim = 1: one value per element (plain path), out = 5 planes of `stride` doubles;
// cells_per_dim = 4: Subgrid<4,..>, value per subcell at index e*S + i + 4j + 16k.
void t8gpu_synth_part_kh_ic(const void* h, int cells_per_dim, double* out, size_t stride) {
  const Part* p   = static_cast<const Part*>(h);
  const Mesh& M   = *p->m;
  const int   dim = M.dim;
  const int   E   = cells_per_dim;
  const int   S   = dim == 3 ? E * E * E : E * E;
  const int   tot = p->N + p->G;
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int i = 0; i < tot; i++) {
    const int64_t g  = i < p->N ? p->first + i : p->ghost_global[i - p->N];
    const Leaf&   l  = M.leaves[g];
    const double  hh = std::ldexp(1.0, -l.level);
    for (int c = 0; c < S; c++) {
      const int ci[3] = {c % E, (c / E) % E, c / (E * E)};
      double    x[3]  = {0, 0, 0};
      for (int d = 0; d < dim; d++) x[d] = (l.c[d] + (ci[d] + 0.5) / E) * hh;
      double u[5];
      kh_state(dim, x, u);
      for (int k = 0; k < 5; k++) out[k * stride + static_cast<size_t>(i) * S + c] = u[k];
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Adaptation of the synthetic forest (stands where t8_forest adapt + balance run in
// MeshManager::adapt, t8gpu/mesh/mesh_manager.inl:196-213).
//
// t8gpu_synth_mesh_marks: the reference's adapt callback (mesh_manager.inl:125-162) applied to every
// element: +1 (refine) if level < max_level and criteria > threshold; -1 (coarsen) for ALL members of a
// complete family if level > min_level and the mean criterion is < threshold. The reference averages only
// the first 4 members even for 8-child families; `family_members_averaged` = 4
// reproduces that, 0 means "all 2^dim members".
// t8gpu_synth_mesh_adapt: new forest = old forest with marked elements refined once, marked complete
// families coarsened once, then 2:1 face balance (refinement wins over coarsening, as t8code's
// adapt-then-balance does). Every element changes by at most one level.
// t8gpu_synth_mesh_adapt_data: the old->new correspondence of mesh_manager.inl:258-281:
// adapt_data[i] = first old element of new element i (n_new + 1 entries).
// ---------------------------------------------------------------------------------------------------
static bool is_family_start(const Mesh& M, size_t e) {
  const int nsub = 1 << M.dim;
  if (e + nsub > M.leaves.size()) return false;
  const Leaf& a = M.leaves[e];
  if (a.level == 0) return false;
  for (int d = 0; d < M.dim; d++)
    if (a.c[d] & 1u) return false;  // must be child 0
  for (int ch = 0; ch < nsub; ch++) {
    const Leaf& b = M.leaves[e + ch];
    if (b.level != a.level) return false;
    for (int d = 0; d < M.dim; d++)
      if (b.c[d] != a.c[d] + ((ch >> d) & 1)) return false;
  }
  return true;
}

void t8gpu_synth_mesh_marks(const void* mesh, const double* criteria, double threshold, int min_level, int max_level,
                            int family_members_averaged, int8_t* marks) {
  const Mesh&  M    = *static_cast<const Mesh*>(mesh);
  const size_t n    = M.leaves.size();
  const int    nsub = 1 << M.dim;
  const int    navg = family_members_averaged > 0 ? std::min(family_members_averaged, nsub) : nsub;
  // t8code calls the callback element by element; with is_family = 1 only when the element opens a
  // complete family. Refinement is tested first and on the element's own criterion only; -1 coarsens
  // the whole family and skips its other members. Two parallel passes give the marks of that sequential walk: every
  // element's own refinement test, then the families (disjoint: child 0 opens one) whose first member does not refine.
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int64_t e = 0; e < static_cast<int64_t>(n); e++)
    marks[e] = (M.leaves[e].level < max_level && criteria[e] > threshold) ? 1 : 0;
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int64_t e = 0; e < static_cast<int64_t>(n); e++) {
    if (__atomic_load_n(&marks[e], __ATOMIC_RELAXED) != 0 || M.leaves[e].level <= min_level || !is_family_start(M, static_cast<size_t>(e))) continue;
    double mean = 0;
    for (int ch = 0; ch < navg; ch++) mean += criteria[e + ch] / navg;
    if (mean < threshold)
      for (int ch = 0; ch < nsub; ch++) __atomic_store_n(&marks[e + ch], static_cast<int8_t>(-1), __ATOMIC_RELAXED);
  }
}

// A family whose members live on two ranks is not coarsened (t8code only coarsens families that are
// complete on one process): clear the -1 marks of families cut by a partition offset.
void t8gpu_synth_mesh_unmark_split_families(const void* mesh, int8_t* marks, const int64_t* offsets, int n_offsets) {
  const Mesh&   M    = *static_cast<const Mesh*>(mesh);
  const int64_t n    = static_cast<int64_t>(M.leaves.size());
  const int     nsub = 1 << M.dim;
  for (int k = 0; k < n_offsets; k++) {
    const int64_t b = offsets[k];
    if (b <= 0 || b >= n || marks[b] >= 0 || marks[b - 1] >= 0) continue;
    for (int64_t s = std::max<int64_t>(0, b - nsub + 1); s < b; s++)
      if (is_family_start(M, static_cast<size_t>(s)) && s + nsub > b) {
        for (int ch = 0; ch < nsub; ch++) marks[s + ch] = 0;
        break;
      }
  }
}

// The general procedure: rebuild the leaf list and the lookup grid after every balance round. Kept as the fallback of
// t8gpu_synth_mesh_adapt below (which needs the old forest to be 2:1 balanced).
static void* adapt_by_rounds(const void* mesh, const int8_t* marks) {
  const Mesh& O    = *static_cast<const Mesh*>(mesh);
  const int   nsub = 1 << O.dim;
  Mesh*       M    = new Mesh;
  M->dim = O.dim; M->base = O.base; M->band = O.band; M->shrink = O.shrink; M->periodic = O.periodic;
  // the finest-level lookup grid is sized for lmax: refining a finest leaf needs one more level
  int newmax = O.lmax;
  for (size_t e = 0; e < O.leaves.size(); e++)
    if (marks[e] > 0 && O.leaves[e].level + 1 > newmax) newmax = O.leaves[e].level + 1;
  if (O.dim * newmax > 28) {
    delete M;
    return nullptr;
  }
  M->lmax = newmax;
  PhaseTimer timer("synth adapt (rounds)");
  // 1. refinements and tentative coarsenings; `origin` remembers how each new leaf was made:
  //    0 kept, 1 refined child, 2 coarsened parent (can be undone by the balance step)
  std::vector<uint8_t> origin;
  for (size_t e = 0; e < O.leaves.size();) {
    const Leaf& l = O.leaves[e];
    if (marks[e] < 0 && is_family_start(O, e)) {
      bool all = true;
      for (int ch = 0; ch < nsub; ch++) all = all && marks[e + ch] < 0;
      if (all) {
        Leaf p;
        p.level = l.level - 1;
        for (int d = 0; d < 3; d++) p.c[d] = l.c[d] >> 1;
        M->leaves.push_back(p);
        origin.push_back(2);
        e += nsub;
        continue;
      }
    }
    if (marks[e] > 0) {
      M->push_children(M->leaves, l);
      origin.insert(origin.end(), nsub, 1);
    } else {
      M->leaves.push_back(l);
      origin.push_back(0);
    }
    e++;
  }
  timer.lap("refine / coarsen");
  M->fill_owner();
  timer.lap("lookup grid");
  // 2. balance: a leaf two or more levels coarser than a face neighbour is split. A leaf made by
  //    coarsening goes back to its children (net change 0); a kept leaf is refined once (net +1).
  for (;;) {
    std::vector<uint8_t> mark(M->leaves.size(), 0);
    int                  any = 0;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(| : any)
    for (int64_t e = 0; e < static_cast<int64_t>(M->leaves.size()); e++)
      for (int f = 0; f < 2 * M->dim; f++) {
        const int32_t nb = M->across(static_cast<size_t>(e), f);
        if (nb >= 0 && M->leaves[nb].level < M->leaves[e].level - 1) {
#pragma omp atomic write
          mark[nb] = 1;
          any |= 1;
        }
      }
    if (!any) break;
    std::vector<Leaf>    next;
    std::vector<uint8_t> norigin;
    for (size_t e = 0; e < M->leaves.size(); e++) {
      if (mark[e]) {
        M->push_children(next, M->leaves[e]);
        norigin.insert(norigin.end(), nsub, origin[e] == 2 ? 0 : 1);
      } else {
        next.push_back(M->leaves[e]);
        norigin.push_back(origin[e]);
      }
    }
    M->leaves.swap(next);
    origin.swap(norigin);
    M->fill_owner();
    timer.lap("balance round");
  }
  timer.lap("balance check");
  return M;
}

// One pass over the OLD forest instead of a rebuilt forest per balance round. The old forest is 2:1 balanced, so the new
// one (the coarsest balanced refinement of "old forest with the marks applied") never differs from it by more than one
// level anywhere: the result is a level change d[e] in {-1, 0, +1} per old leaf (-1 for all members of a family or for
// none), and balance is the least fixed point of "raise the coarser side of a face whose levels differ by two" -- iterated
// on d with the OLD lookup grid, every face looked at from its finer-or-equal old leaf (whose coarser neighbour is unique).
// The new leaves are then written in one parallel expansion and the lookup grid of the new forest is filled once
// (it used to be rebuilt after every round: 512 MB per round at level 9).
void* t8gpu_synth_mesh_adapt(const void* mesh, const int8_t* marks) {
  const Mesh&   O    = *static_cast<const Mesh*>(mesh);
  const int     nsub = 1 << O.dim;
  const int64_t n    = static_cast<int64_t>(O.leaves.size());
  PhaseTimer    timer("synth adapt");
  int           newmax = O.lmax;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(max : newmax)
  for (int64_t e = 0; e < n; e++)
    if (marks[e] > 0 && O.leaves[e].level + 1 > newmax) newmax = O.leaves[e].level + 1;
  if (O.dim * newmax > 28) return nullptr;
  std::vector<int8_t> d(static_cast<size_t>(n), 0);
  auto get = [&](int64_t e) { return __atomic_load_n(&d[e], __ATOMIC_RELAXED); };
  auto put = [&](int64_t e, int8_t v) { __atomic_store_n(&d[e], v, __ATOMIC_RELAXED); };
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int64_t e = 0; e < n; e++) {
    if (marks[e] > 0) {
      d[e] = 1;
    } else if (marks[e] < 0 && is_family_start(O, static_cast<size_t>(e))) {   // (families are disjoint: child 0 opens one)
      bool all = true;
      for (int ch = 0; ch < nsub; ch++) all = all && marks[e + ch] < 0;
      if (all)
        for (int ch = 0; ch < nsub; ch++) put(e + ch, -1);
    }
  }
  timer.lap("marks -> level changes");
  int broken = 0;
  for (;;) {
    int any = 0;
#pragma omp parallel for num_threads(host_threads()) schedule(static) reduction(| : any, broken)
    for (int64_t e = 0; e < n; e++) {
      const Leaf& le = O.leaves[e];
      for (int f = 0; f < 2 * O.dim; f++) {
        const int32_t nb = O.across(static_cast<size_t>(e), f);
        if (nb < 0) continue;
        const Leaf& ln = O.leaves[nb];
        if (ln.level > le.level) continue;   // that pair is looked at from the finer side
        const int8_t dn = get(nb);
        if (ln.level + dn >= le.level + get(e) - 1) continue;
        any |= 1;
        if (dn < 0) {   // the neighbour's family is not coarsened after all
          int ch = 0;
          for (int k = 0; k < O.dim; k++) ch |= static_cast<int>(ln.c[k] & 1u) << k;
          for (int k = 0; k < nsub; k++) put(nb - ch + k, 0);
        } else if (dn == 0) {
          put(nb, 1);
        } else {
          broken |= 1;   // a second refinement: the old forest was not balanced
        }
      }
    }
    timer.lap("balance round");
    if (!any || broken) break;
  }
  if (broken) return adapt_by_rounds(mesh, marks);
  // the new leaves: every old leaf becomes itself, its children, or (the first member of a coarsened family) its parent
  std::vector<int64_t> at(static_cast<size_t>(n) + 1, 0);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int64_t e = 0; e < n; e++) {
    bool first = true;
    if (d[e] < 0)
      for (int k = 0; k < O.dim; k++) first = first && !(O.leaves[e].c[k] & 1u);
    at[e + 1] = d[e] > 0 ? nsub : (d[e] < 0 ? (first ? 1 : 0) : 1);
  }
  for (int64_t e = 0; e < n; e++) at[e + 1] += at[e];
  Mesh* M = new Mesh;
  M->dim = O.dim; M->base = O.base; M->band = O.band; M->shrink = O.shrink; M->periodic = O.periodic;
  M->lmax = newmax;
  M->leaves.resize(static_cast<size_t>(at[n]));
  M->adapted_from = mesh;   // (t8gpu_synth_mesh_adapt_data copies the correspondence instead of walking the two forests)
  M->adapt_data.resize(static_cast<size_t>(at[n]) + 1);
  M->adapt_data[static_cast<size_t>(at[n])] = static_cast<int32_t>(n);
#pragma omp parallel for num_threads(host_threads()) schedule(static)
  for (int64_t e = 0; e < n; e++) {
    const Leaf& l = O.leaves[e];
    Leaf*       o = M->leaves.data() + at[e];
    if (at[e + 1] == at[e]) continue;
    for (int64_t i = at[e]; i < at[e + 1]; i++) M->adapt_data[static_cast<size_t>(i)] = static_cast<int32_t>(e);
    if (d[e] > 0) {
      for (int ch = 0; ch < nsub; ch++) {
        o[ch].level = l.level + 1;
        for (int k = 0; k < 3; k++) o[ch].c[k] = k < O.dim ? 2 * l.c[k] + ((ch >> k) & 1) : 0;
      }
    } else if (d[e] < 0) {
      o->level = l.level - 1;
      for (int k = 0; k < 3; k++) o->c[k] = l.c[k] >> 1;
    } else {
      *o = l;
    }
  }
  timer.lap("new leaves");
  M->fill_owner();
  timer.lap("lookup grid");
  return M;
}

// the general procedure by itself (tests compare the two)
void* t8gpu_synth_mesh_adapt_by_rounds(const void* mesh, const int8_t* marks) { return adapt_by_rounds(mesh, marks); }

int t8gpu_synth_mesh_adapt_data(const void* old_mesh, const void* new_mesh, int32_t* adapt_data) {
  const Mesh&  O = *static_cast<const Mesh*>(old_mesh);
  const Mesh&  N = *static_cast<const Mesh*>(new_mesh);
  if (N.adapted_from == old_mesh && N.adapt_data.size() == N.leaves.size() + 1 &&
      N.adapt_data.back() == static_cast<int32_t>(O.leaves.size())) {   // recorded by the one-pass adapt
    std::memcpy(adapt_data, N.adapt_data.data(), N.adapt_data.size() * sizeof(int32_t));
    return 0;
  }
  const int    nsub = 1 << O.dim;
  size_t       oi = 0, ni = 0;
  const size_t no = O.leaves.size(), nn = N.leaves.size();
  while (oi < no && ni < nn) {
    const int lo = O.leaves[oi].level, ln = N.leaves[ni].level;
    if (lo < ln) {  // refined: nsub children point at the same old element
      if (ln != lo + 1 || ni + nsub > nn) return 1;
      for (int i = 0; i < nsub; i++) adapt_data[ni + i] = static_cast<int32_t>(oi);
      oi += 1;
      ni += nsub;
    } else if (lo > ln) {  // coarsened
      if (lo != ln + 1 || oi + nsub > no) return 1;
      adapt_data[ni] = static_cast<int32_t>(oi);
      oi += nsub;
      ni += 1;
    } else {
      adapt_data[ni] = static_cast<int32_t>(oi);
      oi += 1;
      ni += 1;
    }
  }
  if (oi != no || ni != nn) return 1;
  adapt_data[nn] = static_cast<int32_t>(no);
  return 0;
}


// ---- the synthetic forest behind the T8gpuForestQuery callbacks (tests of csrc/host/connectivity.cpp) --------
namespace {
struct SynthQuery {
  T8gpuForestQuery q;   // first member: the handle handed out is &q
  const Mesh*      m;
  Part             part;   // only first / N / ghost_global / ghost_owner are used (the slot numbering)
  int64_t gid(int32_t slot) const { return slot < part.N ? part.first + slot : part.ghost_global[slot - part.N]; }
  int32_t slot_of(int64_t g) const {
    if (g >= part.first && g < part.first + part.N) return static_cast<int32_t>(g - part.first);
    return part.N + static_cast<int32_t>(std::lower_bound(part.ghost_global.begin(), part.ghost_global.end(), g) - part.ghost_global.begin());
  }
};
SynthQuery* SQ(void* c) { return static_cast<SynthQuery*>(c); }
}  // namespace

T8gpuForestQuery* t8gpu_synth_query_create(const void* mesh, int rank, int nranks) {
  const Mesh* m = static_cast<const Mesh*>(mesh);
  if (!m || rank < 0 || rank >= nranks) return nullptr;
  SynthQuery* s = new SynthQuery;
  s->m           = m;
  s->part.m      = m;
  s->part.rank   = rank;
  s->part.nranks = nranks;
  s->part.ndim   = 3;
  build_part(s->part);
  T8gpuForestQuery& q = s->q;
  q.ctx        = s;
  q.num_local  = s->part.N;
  q.num_ghost  = s->part.G;
  q.global_id  = [](void* c, int32_t slot) -> int64_t { return SQ(c)->gid(slot); };
  q.owner_rank = [](void* c, int32_t g) -> int32_t { return SQ(c)->part.ghost_owner[g]; };
  q.level      = [](void* c, int32_t slot) -> int32_t { return SQ(c)->m->leaves[SQ(c)->gid(slot)].level; };
  q.num_faces  = [](void* c, int32_t) -> int32_t { return 2 * SQ(c)->m->dim; };
  q.face_neighbors = [](void* c, int32_t e, int32_t f, int32_t max_n, int32_t* slots, int32_t* dual) -> int32_t {
    const SynthQuery* s = SQ(c);
    const Mesh&       M = *s->m;
    const int64_t     ge = s->gid(e);
    const int32_t     one = M.across(static_cast<size_t>(ge), f);
    if (one < 0) return 0;
    const Leaf& l = M.leaves[ge];
    if (M.leaves[one].level <= l.level) {
      slots[0] = s->slot_of(one);
      dual[0]  = f ^ 1;
      return 1;
    }
    // finer neighbours (2:1 balance: exactly one level finer): probe the 2^(dim-1) sub-faces
    const uint32_t sz = 1u << (M.lmax - l.level), ext = 1u << M.lmax, half = sz / 2;
    const int      d  = f / 2;
    int32_t        k  = 0;
    for (int b = 0; b < (1 << (M.dim - 1)); b++) {
      uint32_t p[3] = {l.c[0] * sz, l.c[1] * sz, M.dim == 3 ? l.c[2] * sz : 0};
      p[d] = (f & 1) ? (p[d] + sz) % ext : (p[d] + ext - 1) % ext;
      int bit = 0;
      for (int a = 0; a < M.dim; a++)
        if (a != d) p[a] += ((b >> bit++) & 1) * half;
      if (k >= max_n) return -1;
      slots[k] = s->slot_of(M.owner[M.grid_index(p)]);
      dual[k]  = f ^ 1;
      k++;
    }
    return k;
  };
  q.face_normal = [](void*, int32_t, int32_t f, double n[3]) {
    n[0] = n[1] = n[2] = 0.0;
    n[f / 2] = (f & 1) ? 1.0 : -1.0;
  };
  q.face_area = [](void* c, int32_t slot, int32_t) -> double {
    const double h = std::ldexp(1.0, -SQ(c)->m->leaves[SQ(c)->gid(slot)].level);
    return SQ(c)->m->dim == 3 ? h * h : h;
  };
  q.child_id = [](void* c, int32_t slot) -> int32_t {
    const Leaf& l = SQ(c)->m->leaves[SQ(c)->gid(slot)];
    return static_cast<int32_t>((l.c[0] & 1u) | ((l.c[1] & 1u) << 1) | ((l.c[2] & 1u) << 2));
  };
  q.volume = [](void* c, int32_t slot) -> double {
    const double h = std::ldexp(1.0, -SQ(c)->m->leaves[SQ(c)->gid(slot)].level);
    return SQ(c)->m->dim == 3 ? h * h * h : h * h;
  };
  return &s->q;
}

void t8gpu_synth_query_destroy(T8gpuForestQuery* q) { delete reinterpret_cast<SynthQuery*>(q); }

}  // extern "C"
