"""ctypes access to oracle/liboracle.so -- the CPU checker. Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_libs = {}


def lib(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name not in _libs:
        path = os.path.join(ROOT, "oracle", name)
        src = [os.path.join(ROOT, "oracle", f) for f in ("oracle.hpp", "oracle_capi.cpp")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _libs[name] = C.CDLL(path)
    return _libs[name]


def suf(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


def p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fs(dtype, x):
    return C.c_float(x) if np.dtype(dtype) == np.float32 else C.c_double(x)


def face_frame_flux(kind, uL, uR, want_speed=False):
    uL = np.ascontiguousarray(uL)
    uR = np.ascontiguousarray(uR, dtype=uL.dtype)
    n = uL.shape[0]
    F = np.zeros_like(uL)
    s = np.zeros(n, uL.dtype)
    getattr(lib(), "oracle_face_frame_flux_" + suf(uL.dtype))(kind, n, p(uL), p(uR), p(F), p(s))
    return (F, s) if want_speed else F


def xyz_face_flux(kind, normals, sL, sR, mirror=False):
    sL = np.ascontiguousarray(sL)
    sR = np.ascontiguousarray(sR, dtype=sL.dtype)
    normals = np.ascontiguousarray(normals, dtype=sL.dtype)
    F = np.zeros_like(sL)
    getattr(lib(), "oracle_xyz_face_flux_" + suf(sL.dtype))(kind, sL.shape[0], p(normals), p(sL), p(sR), p(F), int(mirror))
    return F


def ln_mean(aL, aR):
    aL = np.ascontiguousarray(aL)
    aR = np.ascontiguousarray(aR, dtype=aL.dtype)
    out = np.zeros_like(aL)
    getattr(lib(), "oracle_ln_mean_" + suf(aL.dtype))(aL.size, p(aL), p(aR), p(out))
    return out


class PlainCase:
    """Host copy of one rank's plain-element problem in the reference's memory layout:
    planes[26, stride] (plane = step*5 + var, volume = plane 25; memory_manager.h:460)."""

    def __init__(self, part, dtype, capacity=None, state=None, first_touch=None):
        """first_touch: an oracle library whose OpenMP team zero-fills the planes (bench.py's CPU baseline: pages placed where
        the threads that work on them run)"""
        self.part, self.dtype = part, np.dtype(dtype)
        tot = part.N + part.G
        self.stride = capacity or tot
        if first_touch is None:
            self.planes = np.zeros((26, self.stride), dtype)
        else:
            self.planes = np.empty((26, self.stride), dtype)
            first_touch.oracle_first_touch(p(self.planes), C.c_size_t(self.planes.nbytes))
        ic = part.kh_initial_state() if state is None else state
        self.planes[0:5, :tot] = ic.astype(dtype)          # Step0 = `next` before the first iterate()
        self.planes[25, :tot] = part.volumes.astype(dtype)
        self.fn = part.face_neighbors
        self.normals = part.normals.astype(dtype)
        self.areas = part.areas.astype(dtype)
        self.speed = np.zeros(part.F + part.B, dtype)
        self.next, self.prev = 0, 3                        # solver.h:100-101

    def iterate(self, dt, kind=0, omp=False):
        self.next, self.prev = self.prev, self.next        # solver.cu:76
        P = self.part
        getattr(lib(omp), "oracle_plain_iterate_" + suf(self.dtype))(
            kind, P.N, P.F, P.B, P.normal_dim, p(self.fn), p(P.indices), p(self.normals), p(self.areas),
            p(self.planes), C.c_size_t(self.stride), self.prev, self.next, fs(self.dtype, dt), p(self.speed))

    def current(self):
        return self.planes[5 * self.next:5 * self.next + 5]


class SubgridCase:
    """planes[25, stride] in SUBCELLS + separate per-block volumes (subgrid_memory_manager.h:553-554)."""

    def __init__(self, part, dtype, state=None, first_touch=None):
        self.part, self.dtype = part, np.dtype(dtype)
        self.rank = part.mesh.dim
        self.S = 4 ** self.rank
        tot = part.N + part.G
        self.stride = tot * self.S
        if first_touch is None:
            self.planes = np.zeros((25, self.stride), dtype)
        else:
            self.planes = np.empty((25, self.stride), dtype)
            first_touch.oracle_first_touch(p(self.planes), C.c_size_t(self.planes.nbytes))
        ic = part.kh_initial_state() if state is None else state
        self.planes[0:5] = ic.astype(dtype)
        self.volumes = part.volumes.astype(dtype)
        self.fn = part.face_neighbors
        self.normals = part.normals.astype(dtype)
        self.areas = part.areas.astype(dtype)
        self.next, self.prev = 0, 3

    def iterate(self, dt, kind=0, omp=False):
        self.prev, self.next = self.next, self.prev        # solver.inl:154
        P = self.part
        getattr(lib(omp), "oracle_subgrid_iterate_" + suf(self.dtype))(
            kind, self.rank, P.N, P.F, P.B, p(self.fn), p(P.indices), p(P.level_diff), p(P.nb_offset),
            p(self.normals), p(self.areas), p(self.planes), C.c_size_t(self.stride), p(self.volumes),
            self.prev, self.next, fs(self.dtype, dt))

    def current(self):
        return self.planes[5 * self.next:5 * self.next + 5]


def random_states(n, seed, dtype=np.float64, lo_p=0.5, hi_p=5.0):
    """SURVEY 8d microbench distribution: rho in U[0.5,2], v in U[-1,1]^3, p in U[0.5,5]."""
    rng = np.random.default_rng(seed)
    rho = rng.uniform(0.5, 2.0, n)
    v = rng.uniform(-1.0, 1.0, (n, 3))
    pr = rng.uniform(lo_p, hi_p, n)
    E = pr / 0.4 + 0.5 * rho * (v ** 2).sum(1)
    return np.stack([rho, rho * v[:, 0], rho * v[:, 1], rho * v[:, 2], E], 1).astype(dtype)
