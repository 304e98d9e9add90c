"""ctypes mirror of csrc/host/tile_plan.cpp: the host-side tiling pre-pass of the fused kernels."""
import ctypes as C

import numpy as np

from . import synth as _synth

_ready = False


def _lib():
    global _ready
    lib = _synth.lib()
    if not _ready:
        lib.t8gpu_plan_plain_create.restype = C.c_void_p
        lib.t8gpu_plan_plain_create.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 3 + [C.c_int32] * 2
        lib.t8gpu_plan_plain_create_ex.restype = C.c_void_p
        lib.t8gpu_plan_plain_create_ex.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 3 + [C.c_int32] * 3
        lib.t8gpu_plan_plain_patch_counts.argtypes = [C.c_void_p] * 2
        lib.t8gpu_plan_plain_irregular_counts.argtypes = [C.c_void_p] * 2
        lib.t8gpu_plan_plain_patch_dim.argtypes = [C.c_void_p]
        lib.t8gpu_plan_plain_patch_dim.restype = C.c_int32
        lib.t8gpu_plan_plain_patch_volumes.argtypes = [C.c_void_p, C.c_void_p]
        lib.t8gpu_plan_plain_patch_volumes.restype = C.c_int32
        lib.t8gpu_plan_plain_destroy.argtypes = [C.c_void_p]
        lib.t8gpu_plan_plain_sizes.argtypes = [C.c_void_p, C.c_void_p]
        lib.t8gpu_plan_plain_arrays.argtypes = [C.c_void_p] * 11
        lib.t8gpu_plan_plain_array_ptrs.argtypes = [C.c_void_p] * 2
        lib.t8gpu_plan_plain_compressed.argtypes = [C.c_void_p] * 4
        lib.t8gpu_plan_plain_tile_desc.argtypes = [C.c_void_p] * 2
        _ready = True
    return lib


class HostPlainPlan:
    """Host arrays of the plain-element tile plan (see include/t8gpu_hip.h, T8gpuPlainPlan)."""

    FIELDS = ("elem_off", "halo_off", "face_off", "halo_ids", "face_lr", "face_geo", "face_orig", "csr_off",
              "csr_ent", "tile_order")

    def __init__(self, N, G, F, B, ndim, face_neighbors, normals, areas, tmax=256, fcap=512, want_face_geo=True,
                 patches=False, volumes=None, irregular=True, two_classes=False):
        """patches=True: structured 16 x 16 patches are cut out of the tiling (tile_plan.cpp: find_patches); they are
        tiles without face records (`tile_patch[t]` = 1), first inside every class of `tile_order` (`n_patch_class`).
        want_face_geo=False: leave `face_geo` (32 bytes per tile face, only read by the kernels that have no geometry
        dictionary) empty when the plan has a dictionary -- at c4 size that is 700 MB of host copying per plan."""
        lib = _lib()
        fn = np.ascontiguousarray(face_neighbors, np.int32)
        nr = np.ascontiguousarray(normals, np.float64)
        ar = np.ascontiguousarray(areas, np.float64)
        assert fn.size == 2 * F + B and nr.size == ndim * (F + B) and ar.size == F + B
        p = _synth._p
        # patches: True = both kinds (16 x 16 quadrilateral blocks, 8 x 8 x 4 hexahedral blocks), 2 / 3 = that kind only
        pflags = {False: 0, True: 3, 2: 1, 3: 2}[patches] | (0 if want_face_geo else 4)
        if irregular and (pflags & 2):
            pflags |= 8       # 3D blocks next to a periodic wrap / wall / coarser - side neighbour become (irregular) patches too
            if irregular == "all":
                pflags |= 16  # ... and the regular blocks take the irregular form as well (one kernel, one launch per stage)
        if two_classes:
            pflags |= 32      # interior tiles in one class (no deep / near-boundary split): one launch per stage for [0, n_interior)
        h = lib.t8gpu_plan_plain_create_ex(N, G, F, B, ndim, p(fn), p(nr), p(ar), tmax, fcap, pflags)
        if not h:
            raise ValueError("tile plan exceeds the packed index format (use smaller tmax / fcap)")
        # The plan's arrays are VIEWS of the planner's own arrays (no copy: a few hundred MB per plan at 3 M elements); a view
        # keeps the plan handle alive (synth._view), which is destroyed with the last of them.
        self._owner = owner = _synth._Handle(h, lib.t8gpu_plan_plain_destroy)
        sz = np.zeros(16, np.int64)
        lib.t8gpu_plan_plain_sizes(h, p(sz))
        (self.ntiles, n_halo, n_faces, n_csr, self.max_elems, self.max_halo, self.max_faces,
         self.n_interior) = (int(x) for x in sz[:8])
        self.N, self.F, self.B, self.tmax, self.fcap = N, F, B, tmax, fcap
        self.ell_width, n_geo, self.max_slots, self.n_deep = int(sz[10]), int(sz[11]), int(sz[12]), int(sz[13])
        self.n_ell_rows = int(sz[15])                  # rows exist for the elements of generic tiles only (tile_desc word 6)
        ptrs = (C.c_void_p * 13)()
        lib.t8gpu_plan_plain_array_ptrs(h, ptrs)
        view = _synth._view
        self.elem_off = view(ptrs[0], self.ntiles + 1, np.int32, owner)
        self.halo_off = view(ptrs[1], self.ntiles + 1, np.int32, owner)
        self.face_off = view(ptrs[2], self.ntiles + 1, np.int32, owner)
        self.halo_ids = view(ptrs[3], n_halo, np.int32, owner)
        self.face_lr = view(ptrs[4], n_faces, np.uint32, owner)
        n_face_geo = 0 if (n_geo and not want_face_geo) else n_faces       # (flag 4: no rows where a dictionary exists)
        self.face_geo = view(ptrs[5], 4 * n_face_geo, np.float64, owner).reshape(-1, 4)
        self.face_orig = view(ptrs[6], n_faces, np.int32, owner)
        self.csr_off = view(ptrs[7], N + 1, np.int32, owner)
        self.csr_ent = view(ptrs[8], n_csr, np.uint16, owner)
        self.tile_order = view(ptrs[9], self.ntiles, np.int32, owner)
        self.ell = (view(ptrs[10], self.n_ell_rows * self.ell_width, np.uint16, owner).reshape(-1, self.ell_width) if self.n_ell_rows
                    else np.zeros((1, self.ell_width), np.uint16))
        self.geo_idx = view(ptrs[11], n_faces if n_geo else 0, np.uint16, owner)
        self.geo_table = view(ptrs[12], 12 * n_geo, np.float64, owner).reshape(-1, 12)
        # volumes of the owned elements (optional): patches of uniform volume carry it in their descriptor
        self.n_patches_uniform_volume = 0
        if volumes is not None and patches:
            vol = np.ascontiguousarray(np.asarray(volumes, np.float64)[:N])
            self.n_patches_uniform_volume = int(lib.t8gpu_plan_plain_patch_volumes(h, p(vol)))
        self.tile_desc = np.zeros((max(1, self.ntiles), 8), np.int32)
        if self.ntiles:
            lib.t8gpu_plan_plain_tile_desc(h, p(self.tile_desc))
        cnt = np.zeros(4, np.int32)
        lib.t8gpu_plan_plain_patch_counts(h, p(cnt))
        self.n_patch_class, self.n_patches = tuple(int(x) for x in cnt[:3]), int(cnt[3])
        lib.t8gpu_plan_plain_irregular_counts(h, p(cnt))
        self.n_irregular_class = tuple(int(x) for x in cnt[:3])      # the last patch tiles of every class (flag 0x800)
        self.patch_dim = int(lib.t8gpu_plan_plain_patch_dim(h))            # 2 | 3 | 0 (no patch tiles)
        # per tile (index, not position): is it a patch tile? (word 5 of a GENERIC tile's descriptor is its face count: the flag
        # bits mean something in patch descriptors only, so the patch tiles are taken from their positions -- the first
        # n_patch_class[c] of every class)
        self.tile_patch = np.zeros(self.ntiles, bool)
        for c, a in enumerate((0, self.n_deep, self.n_interior)):
            self.tile_patch[self.tile_order[a:a + self.n_patch_class[c]]] = True

    @classmethod
    def from_partition(cls, part, **kw):
        kw.setdefault("volumes", getattr(part, "volumes", None))
        return cls(part.N, part.G, part.F, part.B, part.normal_dim, part.face_neighbors, part.normals, part.areas, **kw)


class HostSubgridPlan:
    """Host arrays of the per-block face lists (see include/t8gpu_hip.h, T8gpuSubgridPlan)."""

    def __init__(self, part):
        assert part.subgrid
        lib = _synth.lib()
        lib.t8gpu_plan_subgrid_create.restype = C.c_void_p
        lib.t8gpu_plan_subgrid_create.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 4
        lib.t8gpu_plan_subgrid_destroy.argtypes = [C.c_void_p]
        lib.t8gpu_plan_subgrid_sizes.argtypes = [C.c_void_p, C.c_void_p]
        lib.t8gpu_plan_subgrid_arrays.argtypes = [C.c_void_p] * 5
        p = _synth._p
        rank = part.mesh.dim
        fn = np.ascontiguousarray(part.face_neighbors, np.int32)
        nr = np.ascontiguousarray(part.normals, np.float64)
        h = lib.t8gpu_plan_subgrid_create(part.N, part.F, part.B, rank, p(fn), p(part.level_diff), p(part.nb_offset), p(nr))
        if not h:
            raise ValueError("subgrid plan needs axis-aligned unit normals (as the reference's subgrid kernels do)")
        self._h = h
        sz = np.zeros(8, np.int64)
        lib.t8gpu_plan_subgrid_sizes(h, p(sz))
        self.N, self.rank, self.max_bf, self.n_interior, self.n_deep = part.N, rank, int(sz[1]), int(sz[3]), int(sz[4])
        self.n_entries = int(sz[0])
        self.n_addressed = int(sz[5])     # 1 + largest block index any face refers to (owned and ghost blocks)
        self.n_families, self.n_rest = int(sz[6]), int(sz[7])
        lib.t8gpu_plan_subgrid_order.argtypes = [C.c_void_p, C.c_void_p]
        self.block_order = np.zeros(part.N, np.int32)
        lib.t8gpu_plan_subgrid_order(h, p(self.block_order))
        self.bf_off = np.zeros(part.N + 1, np.int32)
        self.bf_ent = np.zeros(int(sz[0]), np.int32)
        self.face_rec = np.zeros((int(sz[2]), 4), np.int32)
        self.plus = np.zeros((part.N, rank), np.int32)
        lib.t8gpu_plan_subgrid_arrays(h, p(self.bf_off), p(self.bf_ent), p(self.face_rec), p(self.plus))

    def records(self, areas, float_size):
        """The joined records the kernels read: block_rec [N, 32], bf_rec [n_entries, 4] (int32 words)."""
        lib = _synth.lib()
        lib.t8gpu_plan_subgrid_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        ar = np.ascontiguousarray(areas, np.float64)
        block_rec = np.zeros((max(1, self.N), 32), np.int32)
        bf_rec = np.zeros((max(1, self.n_entries), 4), np.int32)
        lib.t8gpu_plan_subgrid_records(self._h, _synth._p(ar), int(float_size), _synth._p(block_rec), _synth._p(bf_rec))
        return block_rec, bf_rec

    def family_records(self, areas, float_size):
        """fam_rec [n_families, 160 (RANK 3) / 64 (RANK 2)], rest_rec [n_rest, 32] (see T8gpuSubgridPlan)."""
        lib = _synth.lib()
        lib.t8gpu_plan_subgrid_family_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        ar = np.ascontiguousarray(areas, np.float64)
        fam_rec = np.zeros((max(1, self.n_families), 160 if self.rank == 3 else 64), np.int32)
        rest_rec = np.zeros((max(1, self.n_rest), 32), np.int32)
        lib.t8gpu_plan_subgrid_family_records(self._h, _synth._p(ar), int(float_size), _synth._p(fam_rec), _synth._p(rest_rec))
        return fam_rec, rest_rec

    def __del__(self):
        if getattr(self, "_h", None):
            _synth.lib().t8gpu_plan_subgrid_destroy(self._h)
            self._h = None
