"""ctypes mirror of csrc/host/synth_mesh.cpp (the t8code-free mesh provider)."""
import ctypes as C
import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        import os
        _lib = C.CDLL(os.environ.get("T8GPU_HOST_LIB") or _build.build_host())   # override: sanitizer builds
        _lib.t8gpu_synth_mesh_create.restype = C.c_void_p
        _lib.t8gpu_synth_mesh_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        _lib.t8gpu_synth_mesh_destroy.argtypes = [C.c_void_p]
        _lib.t8gpu_synth_mesh_num_elements.restype = C.c_int64
        _lib.t8gpu_synth_mesh_num_elements.argtypes = [C.c_void_p]
        _lib.t8gpu_synth_mesh_finest_level.restype = C.c_int
        _lib.t8gpu_synth_mesh_finest_level.argtypes = [C.c_void_p]
        _lib.t8gpu_synth_part_create.restype = C.c_void_p
        _lib.t8gpu_synth_part_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.t8gpu_synth_part_destroy.argtypes = [C.c_void_p]
        _lib.t8gpu_synth_part_release_arrays.argtypes = [C.c_void_p]
        _lib.t8gpu_synth_part_counts.argtypes = [C.c_void_p, C.c_void_p]
        _lib.t8gpu_synth_part_connectivity.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        _lib.t8gpu_synth_part_connectivity_ptrs.argtypes = [C.c_void_p, C.c_void_p]
        _lib.t8gpu_synth_part_elements.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        _lib.t8gpu_synth_part_halo.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        _lib.t8gpu_synth_part_kh_ic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        _lib.t8gpu_synth_mesh_marks.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.t8gpu_synth_mesh_unmark_split_families.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.t8gpu_synth_mesh_adapt.restype = C.c_void_p
        _lib.t8gpu_synth_mesh_adapt.argtypes = [C.c_void_p, C.c_void_p]
        _lib.t8gpu_synth_mesh_adapt_by_rounds.restype = C.c_void_p
        _lib.t8gpu_synth_mesh_adapt_by_rounds.argtypes = [C.c_void_p, C.c_void_p]
        _lib.t8gpu_synth_mesh_adapt_data.restype = C.c_int
        _lib.t8gpu_synth_mesh_adapt_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


class _Handle:
    """Owner of a C handle that numpy views point into: destroyed when the last view (and the Python object) is gone."""

    def __init__(self, h, destroy):
        self.h, self._destroy = h, destroy

    def __del__(self):
        if self.h:
            try:
                self._destroy(self.h)
            except TypeError:                                   # interpreter shutdown: the module globals are gone
                pass
            self.h = None


def _view(addr, n, dtype, owner):
    """numpy array over n items at C address `addr` (no copy); its buffer object keeps `owner` alive."""
    if not addr or n <= 0:
        return np.empty(0, dtype)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(addr)
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype, count=int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SynthMesh:
    """Global 2:1-balanced periodic (or walled) quad/hex mesh in Morton order."""

    def __init__(self, dim, base_level, max_level, band=0.0, shrink=1.0, periodic=True):
        self.dim, self.base_level, self.max_level = dim, base_level, max_level
        self._h = lib().t8gpu_synth_mesh_create(dim, base_level, max_level, float(band), float(shrink), int(periodic))
        if not self._h:
            raise ValueError("invalid synthetic mesh parameters")
        self.num_elements = lib().t8gpu_synth_mesh_num_elements(self._h)
        self.finest_level = lib().t8gpu_synth_mesh_finest_level(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().t8gpu_synth_mesh_destroy(self._h)
            self._h = None

    def partition(self, rank=0, nranks=1, subgrid=False, normal_dim=None):
        return Partition(self, rank, nranks, subgrid, normal_dim)

    # -- adaptation (MeshManager::adapt without the data transfer, which runs on the device) --------
    def marks_from_criteria(self, criteria, threshold, min_level, max_level, family_members_averaged=4):
        """The reference's adapt callback for every element: +1 refine, -1 coarsen, 0 keep."""
        crit = np.ascontiguousarray(criteria, np.float64)
        assert crit.size == self.num_elements
        marks = np.zeros(self.num_elements, np.int8)
        lib().t8gpu_synth_mesh_marks(self._h, _p(crit), float(threshold), int(min_level), int(max_level),
                                     int(family_members_averaged), _p(marks))
        return marks

    def partition_offsets(self, nranks):
        """first global element of every rank (+ the total): the SFC-contiguous equal split used by partition()."""
        n = self.num_elements
        return np.array([(n * r) // nranks for r in range(nranks + 1)], np.int64)

    def unmark_split_families(self, marks, offsets):
        marks = np.ascontiguousarray(marks, np.int8)
        off = np.ascontiguousarray(offsets, np.int64)
        lib().t8gpu_synth_mesh_unmark_split_families(self._h, _p(marks), _p(off), int(off.size))
        return marks

    def adapt(self, marks, by_rounds=False):
        """Returns (new mesh, adapt_data[n_new + 1]); every element changes by at most one level.
        by_rounds=True: the provider's general procedure (tests compare it with the one-pass default)."""
        marks = np.ascontiguousarray(marks, np.int8)
        assert marks.size == self.num_elements
        h = (lib().t8gpu_synth_mesh_adapt_by_rounds if by_rounds else lib().t8gpu_synth_mesh_adapt)(self._h, _p(marks))
        if not h:
            raise ValueError("adaptation exceeds the finest representable level")
        new = object.__new__(SynthMesh)
        new.dim, new.base_level = self.dim, self.base_level
        new._h = h
        new.num_elements = lib().t8gpu_synth_mesh_num_elements(h)
        new.finest_level = lib().t8gpu_synth_mesh_finest_level(h)
        new.max_level = max(self.max_level, new.finest_level)
        adapt_data = np.empty(new.num_elements + 1, np.int32)
        if lib().t8gpu_synth_mesh_adapt_data(self._h, h, _p(adapt_data)) != 0:
            raise RuntimeError("old and new forests are not one refinement / coarsening step apart")
        return new, adapt_data


class Partition:
    """One rank's share: the exact arrays the reference's connectivity accessors expose
    (t8gpu/mesh/mesh_manager.h:159-166, subgrid_mesh_manager.h), ghosts in slots [N, N+G)."""

    def __init__(self, mesh, rank, nranks, subgrid, normal_dim):
        self.mesh, self.rank, self.nranks, self.subgrid = mesh, rank, nranks, bool(subgrid)
        dim = mesh.dim
        if normal_dim is None:
            normal_dim = dim if subgrid else 3  # MeshManager<..., 3> stores 3 comps
        self.normal_dim = normal_dim
        h = lib().t8gpu_synth_part_create(mesh._h, rank, nranks, int(self.subgrid), normal_dim)
        if not h:
            raise ValueError("invalid partition parameters")
        # The partition handle belongs to `owner`, which the array views below share: it is destroyed with the last of them.
        self._owner = owner = _Handle(h, lib().t8gpu_synth_part_destroy)
        self._h = h
        cnt = np.zeros(8, np.int64)
        lib().t8gpu_synth_part_counts(h, _p(cnt))
        self.N, self.G, self.F, self.B, npeer, nsend = (int(x) for x in cnt[:6])
        self.first_global, self.num_global = int(cnt[6]), int(cnt[7])
        # The connectivity arrays are VIEWS of the provider's own arrays (363 MB at 3 M elements: copying them was a third of
        # the provider's share of an adapt cycle); a view keeps the handle alive (_view), so it may outlive this object.
        ptrs = (C.c_void_p * 5)()
        lib().t8gpu_synth_part_connectivity_ptrs(h, ptrs)
        self.face_neighbors = _view(ptrs[0], 2 * self.F + self.B, np.int32, owner)
        self.normals = _view(ptrs[1], normal_dim * (self.F + self.B), np.float64, owner)
        self.areas = _view(ptrs[2], self.F + self.B, np.float64, owner)
        self.level_diff = _view(ptrs[3], self.F, np.int32, owner) if subgrid else None
        self.nb_offset = _view(ptrs[4], dim * self.F, np.int32, owner) if subgrid else None
        tot = self.N + self.G
        self.levels = np.empty(tot, np.int32)                # (np.empty: the provider overwrites every entry)
        self.volumes = np.empty(tot, np.float64)
        self.centres = np.empty((tot, 3), np.float64)
        lib().t8gpu_synth_part_elements(h, _p(self.levels), _p(self.volumes), _p(self.centres))
        self.ghost_global = np.zeros(self.G, np.int64)
        self.ghost_owner = np.zeros(self.G, np.int32)
        self.peers = np.zeros(npeer, np.int32)
        self.recv_off = np.zeros(npeer + 1, np.int32)
        self.send_off = np.zeros(npeer + 1, np.int32)
        self.send_idx = np.zeros(nsend, np.int32)
        lib().t8gpu_synth_part_halo(h, _p(self.ghost_global), _p(self.ghost_owner), _p(self.peers),
                                    _p(self.recv_off), _p(self.send_off), _p(self.send_idx))
        self._ic = {}                                       # evaluated on demand (kh_initial_state): an adapt cycle never asks
        # ranks[]/indices[] of the reference accessors (mesh_manager.h:141-157): ghosts resolve to local slots.
        self.ranks = np.full(tot, rank, np.int32)
        self.indices = np.arange(tot, dtype=np.int32)

    @property
    def cells_per_element(self):
        return 4 ** self.mesh.dim if self.subgrid else 1

    def kh_initial_state(self):
        """(5, (N+G)*S) float64 Kelvin-Helmholtz state, ghosts included."""
        cpd = 4 if self.subgrid else 1
        if cpd not in self._ic:
            tot, S = self.N + self.G, cpd ** self.mesh.dim
            out = np.zeros((5, tot * S), np.float64)
            lib().t8gpu_synth_part_kh_ic(self._h, cpd, _p(out), tot * S)
            self._ic[cpd] = out
        return self._ic[cpd]

    # (the partition handle belongs to self._owner, which the array views share: destroyed with the last of them)
