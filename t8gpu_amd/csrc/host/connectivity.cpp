// connectivity.cpp -- SURVEY 8f-1, the part that does not need t8code: from per-element face-neighbour
// queries (what MeshManager::compute_connectivity_information gets from t8_forest_leaf_face_neighbors,
// t8_forest_element_face_normal / _face_area, and from the ghost layer; t8gpu/mesh/mesh_manager.inl:333-481)
// to the arrays the kernels consume, in this backend's multi-rank convention:
//   * listing rule of the reference (mesh_manager.inl:411-424): a face between two elements of the same
//     level is listed once, by the lower index; between levels, by the finer element; (left, right) =
//     (listing element, neighbour), normal outward from left, area of the listing element's face;
//   * "index" is the GLOBAL index, also across a rank boundary: a cut face is listed on BOTH ranks with the
//     orientation a single-rank run would give it (the reference lists it on the lower rank only and reaches
//     the remote element through CUDA-IPC pointers, :396-409), ghosts sit in slots [N, N + G);
//   * faces are ordered by (global index of the listing element, its face number), which is the order of a
//     single-rank run, so per-element sums run in the same order on any partition;
//   * peers / recv ranges from the ghost layer (ghosts ordered by owner, then global index), send lists = my
//     elements that share a listed face with a ghost of that peer, ascending.
// A t8code build supplies the callbacks (40 lines around the calls named above); here they are exercised
// over the synthetic forest (t8gpu_synth_query_*) and must reproduce its direct builder array for array.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <tuple>
#include <vector>

#include "t8gpu_host.h"

namespace {

struct Conn {
  int32_t N = 0, G = 0, F = 0, B = 0;
  std::vector<int32_t> fn;
  std::vector<double>  normals, areas, volumes;
  std::vector<int32_t> peers, recv_off, send_off, send_idx;
  std::vector<int32_t> level_diff, nb_offset;   // Subgrid meshes only
  int32_t              rank = 0;
};

struct Raw {
  int64_t key_elem;   // global index of the listing (left) element
  int32_t key_face;   // its face number
  int32_t l, r;       // local slots
  int32_t gslot, gface;   // geometry source: slot and face number
};

}  // namespace

extern "C" {

void* t8gpu_host_connectivity_create(const T8gpuForestQuery* q) { return t8gpu_host_connectivity_create_subgrid(q, 0); }

// subgrid_rank = 2 | 3 adds what the Subgrid kernels need per face (subgrid_mesh_manager.inl:587-680):
// face_level_difference = level(right) - level(left) <= 0 and face_neighbor_offset = the anchor inside the right
// (coarser-or-equal) block: on the shared face plane along the face axis, the half-block selected by the left
// element's child id along the other axes when the right block is coarser. Needs q->child_id; quad / hex faces
// numbered -x, +x, -y, +y, -z, +z (t8code's order for these classes).
void* t8gpu_host_connectivity_create_subgrid(const T8gpuForestQuery* q, int32_t subgrid_rank) {
  if ((subgrid_rank != 0 && subgrid_rank != 2 && subgrid_rank != 3) || (subgrid_rank != 0 && (!q || !q->child_id))) return nullptr;
  if (!q || q->num_local < 0 || q->num_ghost < 0 || !q->global_id || !q->level || !q->num_faces || !q->face_neighbors ||
      !q->face_normal || !q->face_area || !q->volume || (q->num_ghost > 0 && !q->owner_rank))
    return nullptr;
  Conn* C = new Conn;
  const int32_t N = q->num_local, G = q->num_ghost;
  C->N = N;
  C->G = G;
  std::vector<Raw> faces, walls;
  int32_t nb[16], dual[16];
  for (int32_t e = 0; e < N; e++) {
    const int64_t ge = q->global_id(q->ctx, e);
    const int32_t le = q->level(q->ctx, e);
    const int32_t nf = q->num_faces(q->ctx, e);
    for (int32_t f = 0; f < nf; f++) {
      const int32_t k = q->face_neighbors(q->ctx, e, f, 16, nb, dual);
      if (k < 0 || k > 16) {
        delete C;
        return nullptr;
      }
      if (k == 0) {
        walls.push_back({ge, f, e, -1, e, f});
        continue;
      }
      for (int32_t j = 0; j < k; j++) {
        const int32_t n = nb[j];
        if (n == e) continue;   // (a periodic mesh of one element per direction)
        const int64_t gn = q->global_id(q->ctx, n);
        const int32_t ln = q->level(q->ctx, n);
        const bool    ghost = n >= N;
        const bool    neighbour_lists = ln > le || (ln == le && gn < ge);
        if (!neighbour_lists)
          faces.push_back({ge, f, e, n, e, f});
        else if (ghost)   // the listing element lives on another rank: list it here too, from ITS point of view
          faces.push_back({gn, dual[j], n, e, n, dual[j]});
      }
    }
  }
  auto by_key = [](const Raw& a, const Raw& b) { return std::tie(a.key_elem, a.key_face, a.r) < std::tie(b.key_elem, b.key_face, b.r); };
  std::sort(faces.begin(), faces.end(), by_key);
  std::sort(walls.begin(), walls.end(), by_key);
  C->F = static_cast<int32_t>(faces.size());
  C->B = static_cast<int32_t>(walls.size());
  C->fn.resize(2 * faces.size() + walls.size());
  C->normals.resize(3 * (faces.size() + walls.size()));
  C->areas.resize(faces.size() + walls.size());
  for (size_t i = 0; i < faces.size() + walls.size(); i++) {
    const Raw& r = i < faces.size() ? faces[i] : walls[i - faces.size()];
    if (i < faces.size()) {
      C->fn[2 * i]     = r.l;
      C->fn[2 * i + 1] = r.r;
    } else {
      C->fn[2 * faces.size() + (i - faces.size())] = r.l;
    }
    q->face_normal(q->ctx, r.gslot, r.gface, &C->normals[3 * i]);
    C->areas[i] = q->face_area(q->ctx, r.gslot, r.gface);
  }
  if (subgrid_rank) {
    C->rank = subgrid_rank;
    C->level_diff.resize(faces.size());
    C->nb_offset.assign(static_cast<size_t>(subgrid_rank) * faces.size(), 0);
    for (size_t i = 0; i < faces.size(); i++) {
      const Raw&    r  = faces[i];
      const int32_t ll = q->level(q->ctx, r.l), lr = q->level(q->ctx, r.r);
      C->level_diff[i] = lr - ll;
      const int     ax = r.key_face / 2;
      const int32_t child = q->child_id(q->ctx, r.l);
      for (int d = 0; d < subgrid_rank; d++) {
        int o = 0;
        if (d == ax)
          o = (r.key_face & 1) ? 0 : 3;
        else if (lr < ll)
          o = 2 * ((child >> d) & 1);
        C->nb_offset[static_cast<size_t>(subgrid_rank) * i + d] = o;
      }
    }
  }
  C->volumes.resize(static_cast<size_t>(N) + G);
  for (int32_t s = 0; s < N + G; s++) C->volumes[s] = q->volume(q->ctx, s);
  // halo lists
  C->recv_off.assign(1, 0);
  for (int32_t g = 0; g < G; g++) {
    const int32_t o = q->owner_rank(q->ctx, g);
    if (C->peers.empty() || C->peers.back() != o) {
      if (!C->peers.empty()) C->recv_off.push_back(g);
      C->peers.push_back(o);
    }
  }
  if (!C->peers.empty()) C->recv_off.push_back(G);
  std::vector<std::vector<int32_t>> send(C->peers.size());
  auto peer_slot = [&](int32_t ghost_slot) {
    const int32_t o = q->owner_rank(q->ctx, ghost_slot - N);
    return static_cast<size_t>(std::lower_bound(C->peers.begin(), C->peers.end(), o) - C->peers.begin());
  };
  for (const Raw& r : faces) {
    if (r.l >= N && r.r < N) send[peer_slot(r.l)].push_back(r.r);
    if (r.r >= N && r.l < N) send[peer_slot(r.r)].push_back(r.l);
  }
  C->send_off.assign(1, 0);
  for (auto& s : send) {
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    C->send_idx.insert(C->send_idx.end(), s.begin(), s.end());
    C->send_off.push_back(static_cast<int32_t>(C->send_idx.size()));
  }
  return C;
}

void t8gpu_host_connectivity_destroy(void* h) { delete static_cast<Conn*>(h); }

/* counts[6] = {N, G, F, B, n_peers, n_send} */
void t8gpu_host_connectivity_counts(const void* h, int64_t* counts) {
  const Conn* C = static_cast<const Conn*>(h);
  counts[0] = C->N; counts[1] = C->G; counts[2] = C->F; counts[3] = C->B;
  counts[4] = static_cast<int64_t>(C->peers.size());
  counts[5] = static_cast<int64_t>(C->send_idx.size());
}

void t8gpu_host_connectivity_arrays(const void* h, int32_t* face_neighbors, double* normals3, double* areas, double* volumes,
                                    int32_t* peers, int32_t* recv_off, int32_t* send_off, int32_t* send_idx) {
  const Conn* C = static_cast<const Conn*>(h);
  auto cp = [](auto* dst, const auto& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cp(face_neighbors, C->fn);
  cp(normals3, C->normals);
  cp(areas, C->areas);
  cp(volumes, C->volumes);
  cp(peers, C->peers);
  cp(recv_off, C->recv_off);
  cp(send_off, C->send_off);
  cp(send_idx, C->send_idx);
}


/* Subgrid meshes: face_level_difference[F], face_neighbor_offset[rank * F] */
void t8gpu_host_connectivity_subgrid_arrays(const void* h, int32_t* level_diff, int32_t* nb_offset) {
  const Conn* C = static_cast<const Conn*>(h);
  if (level_diff && !C->level_diff.empty()) std::memcpy(level_diff, C->level_diff.data(), C->level_diff.size() * sizeof(int32_t));
  if (nb_offset && !C->nb_offset.empty()) std::memcpy(nb_offset, C->nb_offset.data(), C->nb_offset.size() * sizeof(int32_t));
}

}  // extern "C"
