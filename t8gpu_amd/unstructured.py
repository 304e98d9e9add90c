"""Conforming mixed prism / hexahedron meshes with curved geometry, in the reference's array formats.

The reference's own example runs on a prismed spherical shell (examples/compressible_euler/main.cu:20-24):
five faces per element and a different normal on every face. t8code is not available here, so this
module builds meshes of that class directly -- the geometry-synthetic stand-in SURVEY 8d asks for under
C5 -- and hands them over exactly as `MeshManager::compute_connectivity_information` would
(t8gpu/mesh/mesh_manager.inl:333-481): face_neighbors[2F + B] (interior faces listed once, by the element
with the lower index, whose outward normal is stored; then one entry per boundary face), face_normals
[3 (F + B)], face_surfaces[F + B], element volumes, ghosts in slots [N, N + G).

Construction: an nx x ny x nz grid of cells on the unit cube; the cells of a "split" column are cut along
a diagonal into two prisms (5 faces), the others stay hexahedra (6 faces); every vertex is then moved by a
smooth map. Face area vectors come from the vertex loops (1/2 (v2 - v0) x (v3 - v1) for a quad), so every
cell is closed to rounding (sum of outward area vectors = 0) however curved the map is. Elements are
numbered along the Morton curve of their cells.
"""
import numpy as np


def _morton3(i, j, k, bits):
    m = np.zeros_like(i, dtype=np.int64)
    for b in range(bits):
        m |= ((i >> b) & 1) << (3 * b) | ((j >> b) & 1) << (3 * b + 1) | ((k >> b) & 1) << (3 * b + 2)
    return m


def shell_map(x, y, z):
    """Unit cube -> a thick curved shell sector: radius 0.6..1.0 along x, 100 x 100 degrees in (y, z)."""
    r = 0.6 + 0.4 * x
    th = (y - 0.5) * 1.75
    ph = (z - 0.5) * 1.75
    return np.stack([r * np.cos(th) * np.cos(ph), r * np.sin(th) * np.cos(ph), r * np.sin(ph)], axis=-1)


def wavy_map(x, y, z):
    """A smooth perturbation of the identity (keeps the cube's topology; used for periodic meshes)."""
    tp = 2 * np.pi
    return np.stack([x + 0.03 * np.sin(tp * y) * np.cos(tp * z), y + 0.03 * np.sin(tp * z) * np.cos(tp * x),
                     z + 0.03 * np.sin(tp * x) * np.cos(tp * y)], axis=-1)


class PrismHexMesh:
    """Global mesh. split: "all" | "none" | "checker" | float (fraction of columns, seeded)."""

    def __init__(self, n, split="checker", mapping=shell_map, periodic=False, seed=12345):
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        self.n, self.periodic, self.dim = (nx, ny, nz), bool(periodic), 3
        bits = int(np.ceil(np.log2(max(nx, ny, nz))))
        self.finest_level = bits
        ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        if split == "all":
            sp = np.ones((nx, ny), bool)
        elif split == "none":
            sp = np.zeros((nx, ny), bool)
        elif split == "checker":
            sp = ((ii + jj) % 2 == 0)
        else:
            sp = np.random.default_rng(seed).random((nx, ny)) < float(split)
        self.split = sp
        # element numbering: cells along the Morton curve; a split cell holds two consecutive elements
        ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
        order = np.argsort(_morton3(ci, cj, ck, bits), kind="stable")
        per_cell = np.where(sp[ci, cj], 2, 1)
        first = np.zeros(nx * ny * nz, np.int64)
        first[order] = np.concatenate([[0], np.cumsum(per_cell[order])[:-1]])
        first = first.reshape(nx, ny, nz)
        self.num_elements = int(per_cell.sum())
        # vertices
        gx, gy, gz = np.meshgrid(np.arange(nx + 1) / nx, np.arange(ny + 1) / ny, np.arange(nz + 1) / nz, indexing="ij")
        X = mapping(gx, gy, gz)

        def V(i, j, k):
            return X[i, j, k]

        I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        S = sp[I, J]
        lo = first                      # hex, or prism T0 (owns the x-, y- sides)
        hi = first + S                  # hex, or prism T1 (owns the x+, y+ sides)

        def quad(v0, v1, v2, v3):       # area vector of the loop v0 v1 v2 v3 and its centroid
            return 0.5 * np.cross(v2 - v0, v3 - v1), 0.25 * (v0 + v1 + v2 + v3)

        def tri(v0, v1, v2):
            return 0.5 * np.cross(v1 - v0, v2 - v0), (v0 + v1 + v2) / 3.0

        faces_l, faces_r, avec, cen = [], [], [], []

        def add(mask, left, right, a, c):
            faces_l.append(left[mask])
            faces_r.append(right[mask] if right is not None else np.full(int(mask.sum()), -1, np.int64))
            avec.append(a[mask])
            cen.append(c[mask])

        full = np.ones_like(S)
        # +x faces (area vector along +x), owner: hi of this cell; neighbour: lo of cell i+1
        a, c = quad(V(I + 1, J, K), V(I + 1, J + 1, K), V(I + 1, J + 1, K + 1), V(I + 1, J, K + 1))
        nb = lo[(I + 1) % nx, J, K]
        inner = full if periodic else (I + 1 < nx)
        add(inner, hi, nb, a, c)
        add(~inner, hi, None, a, c)
        # +y faces (loop chosen so that the area vector points along +y)
        a, c = quad(V(I, J + 1, K), V(I, J + 1, K + 1), V(I + 1, J + 1, K + 1), V(I + 1, J + 1, K))
        nb = lo[I, (J + 1) % ny, K]
        inner = full if periodic else (J + 1 < ny)
        add(inner, hi, nb, a, c)
        add(~inner, hi, None, a, c)
        # +z faces: one quad for a hexahedron, two triangles for a split cell
        inner = full if periodic else (K + 1 < nz)
        Kp = (K + 1) % nz
        a, c = quad(V(I, J, K + 1), V(I + 1, J, K + 1), V(I + 1, J + 1, K + 1), V(I, J + 1, K + 1))
        add(inner & ~S, lo, lo[I, J, Kp], a, c)
        add(~inner & ~S, lo, None, a, c)
        a, c = tri(V(I, J, K + 1), V(I + 1, J, K + 1), V(I, J + 1, K + 1))
        add(inner & S, lo, lo[I, J, Kp], a, c)
        add(~inner & S, lo, None, a, c)
        a, c = tri(V(I + 1, J + 1, K + 1), V(I, J + 1, K + 1), V(I + 1, J, K + 1))
        add(inner & S, hi, hi[I, J, Kp], a, c)
        add(~inner & S, hi, None, a, c)
        # the diagonal of a split cell: from T0 to T1
        a, c = quad(V(I + 1, J, K), V(I, J + 1, K), V(I, J + 1, K + 1), V(I + 1, J, K + 1))
        add(S, lo, hi, a, c)
        if not periodic:  # the three low walls; area vectors of the same loops as above, pointing outward (-)
            a, c = quad(V(I, J, K), V(I, J + 1, K), V(I, J + 1, K + 1), V(I, J, K + 1))
            add(I == 0, lo, None, -a, c)
            a, c = quad(V(I, J, K), V(I, J, K + 1), V(I + 1, J, K + 1), V(I + 1, J, K))
            add(J == 0, lo, None, -a, c)
            a, c = quad(V(I, J, K), V(I + 1, J, K), V(I + 1, J + 1, K), V(I, J + 1, K))
            add((K == 0) & ~S, lo, None, -a, c)
            a, c = tri(V(I, J, K), V(I + 1, J, K), V(I, J + 1, K))
            add((K == 0) & S, lo, None, -a, c)
            a, c = tri(V(I + 1, J + 1, K), V(I, J + 1, K), V(I + 1, J, K))
            add((K == 0) & S, hi, None, -a, c)
        L, R = np.concatenate(faces_l), np.concatenate(faces_r)
        A, C = np.concatenate(avec), np.concatenate(cen)
        # the reference lists an interior face by the lower index, with that element's outward normal
        swap = (R >= 0) & (R < L)
        L, R = np.where(swap, R, L), np.where(swap, L, R)
        A = np.where(swap[:, None], -A, A)
        interior = R >= 0
        key = np.argsort(np.where(interior, L, L + self.num_elements), kind="stable")   # interior by left element, then walls
        L, R, A, C = L[key], R[key], A[key], C[key]
        self.F, self.B = int(interior.sum()), int((~interior).sum())
        self.face_left, self.face_right = L, R
        self.area_vec, self.face_centroid = A, C
        # volumes and centres from the divergence theorem over each element's faces
        vol = np.zeros(self.num_elements)
        np.add.at(vol, L, np.einsum("ij,ij->i", C, A) / 3.0)
        np.add.at(vol, R[: self.F], -np.einsum("ij,ij->i", C[: self.F], A[: self.F]) / 3.0)
        if periodic:
            # the map is not periodic in space, so faces across the seam carry the geometry of one side only:
            # take the volume from the unmapped cell measure instead
            cellvol = 1.0 / (nx * ny * nz)
            vol = np.zeros(self.num_elements)
            vol[lo.ravel()] = np.where(S.ravel(), 0.5, 1.0) * cellvol
            vol[hi.ravel()] = np.where(S.ravel(), 0.5, 1.0) * cellvol
        assert (vol > 0).all()
        self.volumes = vol
        ctr = np.zeros((self.num_elements, 3))
        cc = V(I, J, K) * 0
        for di in (0, 1):
            for dj in (0, 1):
                for dk in (0, 1):
                    cc = cc + V(I + di, J + dj, K + dk) / 8.0
        ctr[lo.ravel()] = cc.reshape(-1, 3)
        ctr[hi.ravel()] = cc.reshape(-1, 3)
        self.centres = ctr
        self.faces_per_element = (2 * self.F + self.B) / self.num_elements

    def partition(self, rank=0, nranks=1):
        return UnstructuredPartition(self, rank, nranks)

    def initial_state(self):
        """A smooth admissible state on the element centres, (5, num_elements): a shear layer across the
        mapped y direction with a transverse perturbation, pressure 2.5 (the 3D KH set-up of
        examples/subgrid/solver.inl:35-56 evaluated on this geometry)."""
        if getattr(self, "_ic", None) is None:
            x, y, z = self.centres.T
            s = (y - y.min()) / np.ptp(y)
            rho = 1.0 + 0.5 * (1 + np.tanh(20 * (0.25 - np.abs(s - 0.5))))
            v1 = 0.5 * np.tanh(20 * (0.25 - np.abs(s - 0.5)))
            v2 = 0.1 * np.sin(4 * np.pi * x) * (np.exp(-((s - 0.75) / 0.1) ** 2) + np.exp(-((s - 0.25) / 0.1) ** 2))
            v3 = 0.05 * np.cos(2 * np.pi * z)
            e = 2.5 / 0.4 + 0.5 * rho * (v1 * v1 + v2 * v2 + v3 * v3)
            self._ic = np.stack([rho, rho * v1, rho * v2, rho * v3, e])
        return self._ic


class UnstructuredPartition:
    """One rank's share (contiguous range of the element numbering) with the attributes the solvers, the
    tile planner and the halo exchange read from `synth.Partition`. A cut face is listed on both ranks with
    the single-rank orientation; ghosts from a peer are ordered by global index, which is also the order
    in which the peer packs them."""

    subgrid = False
    normal_dim = 3
    cells_per_element = 1

    def __init__(self, mesh, rank, nranks):
        self.mesh, self.rank, self.nranks = mesh, rank, nranks
        Ng = mesh.num_elements
        bounds = [(Ng * r) // nranks for r in range(nranks + 1)]
        a, b = bounds[rank], bounds[rank + 1]
        self.first_global, self.num_global, self.N = a, Ng, b - a
        L, R, F = mesh.face_left, mesh.face_right, mesh.F
        own = lambda e: (e >= a) & (e < b)
        keep_i = own(L[:F]) | own(R[:F])
        keep_b = own(L[F:])
        li, ri = L[:F][keep_i], R[:F][keep_i]
        ends = np.concatenate([li, ri])
        ghosts = np.unique(ends[~own(ends)])
        self.G = int(ghosts.size)
        self.ghost_global = ghosts
        owner = np.searchsorted(np.asarray(bounds[1:]), ghosts, side="right").astype(np.int32)
        self.ghost_owner = owner

        def local(e):
            out = np.where(own(e), e - a, 0)
            g = ~own(e)
            out[g] = self.N + np.searchsorted(ghosts, e[g])
            return out.astype(np.int32)

        self.F, self.B = int(keep_i.sum()), int(keep_b.sum())
        fn = np.empty(2 * self.F + self.B, np.int32)
        fn[0: 2 * self.F: 2], fn[1: 2 * self.F: 2] = local(li), local(ri)
        fn[2 * self.F:] = local(L[F:][keep_b])
        self.face_neighbors = fn
        A = np.concatenate([mesh.area_vec[:F][keep_i], mesh.area_vec[F:][keep_b]])
        area = np.linalg.norm(A, axis=1)
        self.areas = area
        self.normals = (A / area[:, None]).reshape(-1)
        gl = np.concatenate([np.arange(a, b), ghosts])
        self._global_ids = gl
        self.volumes = mesh.volumes[gl]
        self.centres = mesh.centres[gl]
        self.levels = np.full(gl.size, mesh.finest_level, np.int32)
        # halo lists: what each peer sends me (= my ghosts, grouped by owner) and what I send each peer
        peers = sorted(set(owner.tolist()) | self._receivers(mesh, bounds, a, b))
        self.peers = np.array(peers, np.int32)
        self.recv_off = np.zeros(len(peers) + 1, np.int32)
        self.send_off = np.zeros(len(peers) + 1, np.int32)
        send = []
        for k, p in enumerate(peers):
            self.recv_off[k + 1] = self.recv_off[k] + int((owner == p).sum())
            pa, pb = bounds[p], bounds[p + 1]
            in_p = lambda e: (e >= pa) & (e < pb)
            mine = np.concatenate([L[:F][own(L[:F]) & in_p(R[:F])], R[:F][own(R[:F]) & in_p(L[:F])]])
            s = np.unique(mine) - a
            send.append(s)
            self.send_off[k + 1] = self.send_off[k] + s.size
        self.send_idx = (np.concatenate(send) if send else np.zeros(0)).astype(np.int32)
        tot = self.N + self.G
        self.ranks = np.full(tot, rank, np.int32)
        self.indices = np.arange(tot, dtype=np.int32)

    @staticmethod
    def _receivers(mesh, bounds, a, b):
        L, R, F = mesh.face_left[: mesh.F], mesh.face_right[: mesh.F], mesh.F
        own = lambda e: (e >= a) & (e < b)
        other = np.concatenate([R[own(L) & ~own(R)], L[own(R) & ~own(L)]])
        return set(np.searchsorted(np.asarray(bounds[1:]), np.unique(other), side="right").tolist())

    def kh_initial_state(self):
        """(5, N + G) slice of the mesh's initial state (ghosts included)."""
        return self.mesh.initial_state()[:, self._global_ids]


class TetHexMesh:
    """Mixed tetrahedron / hexahedron mesh (BASELINE config 5's mesh class), in the reference's array formats and with
    the attributes UnstructuredPartition reads. An nx x ny x nz grid of cells on the unit cube, walls all round; a cell
    is either one hexahedron (6 faces) or its Kuhn triangulation into 6 tetrahedra (4 faces each, all sharing the
    cell's main diagonal, every cell side cut along the diagonal from its lowest to its highest corner -- the same for
    both cells that share the side, so tetrahedra conform). Where a hexahedron meets a tetrahedron cell its
    quadrilateral side is listed as the two triangles of the neighbour (the finite-volume scheme only sees faces: such
    a hexahedron simply has 7-12 of them, like an element next to a finer one in an AMR forest).

    tets: "blocks" (2x2x2 blocks of cells alternate between the two kinds: many interfaces), "half" (x < 1/2),
    "all", "none", or a boolean array [nx, ny, nz]. Faces are matched through their vertex sets, area vectors come
    from the vertex loops of the mapped geometry (every element closed to rounding), volumes and the listing order as
    in PrismHexMesh. Elements are numbered along the Morton curve of their cells."""

    # Kuhn tetrahedra: one per permutation of the axes, vertices p0 = (0,0,0), p0 + e_a, p0 + e_a + e_b, (1,1,1)
    _PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]

    def __init__(self, n, tets="blocks", mapping=shell_map):
        nx, ny, nz = (n, n, n) if np.isscalar(n) else n
        self.n, self.periodic, self.dim = (nx, ny, nz), False, 3
        bits = int(np.ceil(np.log2(max(nx, ny, nz))))
        self.finest_level = bits
        I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        if isinstance(tets, str):
            T = {"blocks": ((I // 2 + J // 2 + K // 2) % 2 == 0), "half": (I < nx // 2), "all": np.ones_like(I, bool),
                 "none": np.zeros_like(I, bool)}[tets]
        else:
            T = np.asarray(tets, bool)
        self.is_tet_cell = T
        ci, cj, ck = I.ravel(), J.ravel(), K.ravel()
        order = np.argsort(_morton3(ci, cj, ck, bits), kind="stable")
        per_cell = np.where(T.ravel(), 6, 1)
        first = np.zeros(nx * ny * nz, np.int64)
        first[order] = np.concatenate([[0], np.cumsum(per_cell[order])[:-1]])
        self.num_elements = int(per_cell.sum())
        gx, gy, gz = np.meshgrid(np.arange(nx + 1) / nx, np.arange(ny + 1) / ny, np.arange(nz + 1) / nz, indexing="ij")
        X = mapping(gx, gy, gz).reshape(-1, 3)

        def vid(i, j, k):
            return (i * (ny + 1) + j) * (nz + 1) + k

        corner = lambda c, d: vid(ci[c] + d[0], cj[c] + d[1], ck[c] + d[2])
        elem, loops = [], []          # one row per (element, face): element id, 4 vertex ids (triangles: last = -1)
        elem_of_vertices = []         # (element id, its vertex ids) for the centres
        tc, hc = np.nonzero(T.ravel())[0], np.nonzero(~T.ravel())[0]
        e3 = np.eye(3, dtype=np.int64)
        for t, perm in enumerate(self._PERMS):
            p = [np.zeros(3, np.int64), e3[perm[0]], e3[perm[0]] + e3[perm[1]], np.ones(3, np.int64)]
            v = [corner(tc, d) for d in p]
            eid = first[tc] + t
            elem_of_vertices.append((eid, np.stack(v, 1)))
            for skip in range(4):
                tri = [v[q] for q in range(4) if q != skip]
                elem.append(eid)
                loops.append(np.stack(tri + [np.full(tc.size, -1, np.int64)], 1))
        # hexahedra: side d, position s (0 = low, 1 = high); a side shared with a tetrahedron cell is cut like that cell's
        cube = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.int64)
        elem_of_vertices.append((first[hc], np.stack([corner(hc, d) for d in cube], 1)))
        Tpad = np.zeros((nx + 2, ny + 2, nz + 2), bool)
        Tpad[1:-1, 1:-1, 1:-1] = T
        for d in range(3):
            a, b = (d + 1) % 3, (d + 2) % 3
            for s in (0, 1):
                base = s * e3[d]
                q = [base, base + e3[a], base + e3[a] + e3[b], base + e3[b]]      # the side's loop, lowest corner first
                v = [corner(hc, x) for x in q]
                off = e3[d] * (2 * s - 1)
                nb_tet = Tpad[ci[hc] + 1 + off[0], cj[hc] + 1 + off[1], ck[hc] + 1 + off[2]]
                whole = ~nb_tet
                elem.append(first[hc][whole])
                loops.append(np.stack([x[whole] for x in v], 1))
                for tri in ((0, 1, 2), (0, 2, 3)):                                  # cut along lowest -> highest corner
                    elem.append(first[hc][nb_tet])
                    loops.append(np.stack([v[tri[0]][nb_tet], v[tri[1]][nb_tet], v[tri[2]][nb_tet], np.full(int(nb_tet.sum()), -1, np.int64)], 1))
        E, Lp = np.concatenate(elem), np.concatenate(loops)
        # element centres (mean of the vertices) -> orientation of every face loop: outward from its element
        ctr = np.zeros((self.num_elements, 3))
        for eid, vs in elem_of_vertices:
            ctr[eid] = X[vs].mean(axis=1)
        is_tri = Lp[:, 3] < 0
        P0, P1, P2 = X[Lp[:, 0]], X[Lp[:, 1]], X[Lp[:, 2]]
        P3 = X[np.where(is_tri, Lp[:, 0], Lp[:, 3])]
        A = np.where(is_tri[:, None], 0.5 * np.cross(P1 - P0, P2 - P0), 0.5 * np.cross(P2 - P0, P3 - P1))
        C = np.where(is_tri[:, None], (P0 + P1 + P2) / 3.0, 0.25 * (P0 + P1 + P2 + P3))
        flip = np.einsum("ij,ij->i", A, C - ctr[E]) < 0
        A = np.where(flip[:, None], -A, A)
        # pair the two sides of every interior face through the sorted vertex set
        key = np.sort(np.where(Lp < 0, np.iinfo(np.int64).max, Lp), axis=1)
        order = np.lexsort(key.T[::-1])
        ks, Es, As, Cs = key[order], E[order], A[order], C[order]
        same = np.all(ks[1:] == ks[:-1], axis=1)
        first_of_pair = np.concatenate([same, [False]])
        second_of_pair = np.concatenate([[False], same])
        assert not (first_of_pair & second_of_pair).any(), "a face is shared by more than two elements"
        single = ~first_of_pair & ~second_of_pair
        i0 = np.nonzero(first_of_pair)[0]
        a, b = Es[i0], Es[i0 + 1]
        lo_first = a < b
        L = np.concatenate([np.where(lo_first, a, b), Es[single]])
        R = np.concatenate([np.where(lo_first, b, a), np.full(int(single.sum()), -1, np.int64)])
        Avec = np.concatenate([np.where(lo_first[:, None], As[i0], As[i0 + 1]), As[single]])   # outward from the listed element
        Cen = np.concatenate([Cs[i0], Cs[single]])
        self.F, self.B = int(i0.size), int(single.sum())
        srt = np.argsort(np.where(R >= 0, L, L + self.num_elements), kind="stable")           # interior by left element, then walls
        self.face_left, self.face_right, self.area_vec, self.face_centroid = L[srt], R[srt], Avec[srt], Cen[srt]
        L, R, Avec, Cen = self.face_left, self.face_right, self.area_vec, self.face_centroid
        # every wall face lies on the boundary of the unit cube (checked in index space through its vertices)
        vol = np.zeros(self.num_elements)
        np.add.at(vol, L, np.einsum("ij,ij->i", Cen, Avec) / 3.0)
        np.add.at(vol, R[: self.F], -np.einsum("ij,ij->i", Cen[: self.F], Avec[: self.F]) / 3.0)
        assert (vol > 0).all()
        self.volumes, self.centres = vol, ctr
        self.faces_per_element = (2 * self.F + self.B) / self.num_elements
        self.num_tets = int(6 * T.sum())

    partition = PrismHexMesh.partition
    initial_state = PrismHexMesh.initial_state
