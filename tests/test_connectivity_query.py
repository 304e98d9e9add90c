"""CPU: the t8code-independent half of SURVEY 8f-1 -- connectivity, ghost and send lists built from per-element
face-neighbour callbacks (csrc/host/connectivity.cpp) must equal, array for array, what the synthetic provider
builds directly, on single-rank and partitioned 2D / 3D AMR meshes, periodic and walled."""
import ctypes as C

import numpy as np
import pytest

from t8gpu_amd import synth
from t8gpu_amd.synth import SynthMesh


def build_from_query(mesh, rank, nranks):
    lib = synth.lib()
    lib.t8gpu_synth_query_create.restype = C.c_void_p
    lib.t8gpu_synth_query_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.t8gpu_synth_query_destroy.argtypes = [C.c_void_p]
    lib.t8gpu_host_connectivity_create.restype = C.c_void_p
    lib.t8gpu_host_connectivity_create.argtypes = [C.c_void_p]
    lib.t8gpu_host_connectivity_destroy.argtypes = [C.c_void_p]
    lib.t8gpu_host_connectivity_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.t8gpu_host_connectivity_arrays.argtypes = [C.c_void_p] * 9
    q = lib.t8gpu_synth_query_create(mesh._h, rank, nranks)
    assert q
    h = lib.t8gpu_host_connectivity_create(q)
    assert h
    try:
        cnt = np.zeros(6, np.int64)
        lib.t8gpu_host_connectivity_counts(h, cnt.ctypes.data)
        N, G, F, B, npeer, nsend = (int(x) for x in cnt)
        out = dict(N=N, G=G, F=F, B=B, fn=np.zeros(2 * F + B, np.int32), normals=np.zeros(3 * (F + B)), areas=np.zeros(F + B),
                   volumes=np.zeros(N + G), peers=np.zeros(npeer, np.int32), recv_off=np.zeros(npeer + 1, np.int32),
                   send_off=np.zeros(npeer + 1, np.int32), send_idx=np.zeros(nsend, np.int32))
        p = lambda a: a.ctypes.data if a.size else None
        lib.t8gpu_host_connectivity_arrays(h, p(out["fn"]), p(out["normals"]), p(out["areas"]), p(out["volumes"]), p(out["peers"]),
                                           p(out["recv_off"]), p(out["send_off"]), p(out["send_idx"]))
        return out
    finally:
        lib.t8gpu_host_connectivity_destroy(h)
        lib.t8gpu_synth_query_destroy(q)


@pytest.mark.parametrize("dim,base,lmax,band,periodic", [(2, 1, 1, 0.0, True), (2, 2, 5, 0.06, True), (2, 3, 6, 0.03, False),
                                                         (3, 1, 3, 0.1, True), (3, 2, 4, 0.08, False), (2, 4, 4, 0.0, False)])
@pytest.mark.parametrize("nranks", [1, 2, 5])
def test_query_builder_reproduces_the_direct_builder(dim, base, lmax, band, periodic, nranks):
    mesh = SynthMesh(dim, base, lmax, band=band, periodic=periodic)
    if mesh.num_elements < nranks:
        pytest.skip("fewer elements than ranks")
    for rank in range(nranks):
        want = mesh.partition(rank, nranks, normal_dim=3)
        got = build_from_query(mesh, rank, nranks)
        assert (got["N"], got["G"], got["F"], got["B"]) == (want.N, want.G, want.F, want.B)
        assert np.array_equal(got["fn"], want.face_neighbors)
        assert np.array_equal(got["normals"], want.normals) and np.array_equal(got["areas"], want.areas)
        assert np.array_equal(got["volumes"], want.volumes)
        assert np.array_equal(got["peers"], want.peers) and np.array_equal(got["recv_off"][: len(want.recv_off)], want.recv_off)
        assert np.array_equal(got["send_off"][: len(want.send_off)], want.send_off) and np.array_equal(got["send_idx"], want.send_idx)


def test_malformed_queries_are_rejected():
    lib = synth.lib()
    lib.t8gpu_host_connectivity_create.restype = C.c_void_p
    lib.t8gpu_host_connectivity_create.argtypes = [C.c_void_p]
    assert not lib.t8gpu_host_connectivity_create(None)
    zero = (C.c_byte * 128)()                       # all callbacks NULL
    assert not lib.t8gpu_host_connectivity_create(C.cast(zero, C.c_void_p))


@pytest.mark.parametrize("dim,base,lmax,band,periodic", [(2, 2, 4, 0.1, True), (2, 3, 5, 0.05, False), (3, 1, 3, 0.15, True),
                                                         (3, 2, 3, 0.1, False)])
@pytest.mark.parametrize("nranks", [1, 3])
def test_subgrid_arrays_from_the_query(dim, base, lmax, band, periodic, nranks):
    """face_level_difference / face_neighbor_offset (subgrid_mesh_manager.inl:587-680) from the callbacks + child ids."""
    lib = synth.lib()
    lib.t8gpu_synth_query_create.restype = C.c_void_p
    lib.t8gpu_synth_query_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.t8gpu_synth_query_destroy.argtypes = [C.c_void_p]
    lib.t8gpu_host_connectivity_create_subgrid.restype = C.c_void_p
    lib.t8gpu_host_connectivity_create_subgrid.argtypes = [C.c_void_p, C.c_int32]
    lib.t8gpu_host_connectivity_destroy.argtypes = [C.c_void_p]
    lib.t8gpu_host_connectivity_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.t8gpu_host_connectivity_arrays.argtypes = [C.c_void_p] * 9
    lib.t8gpu_host_connectivity_subgrid_arrays.argtypes = [C.c_void_p] * 3
    mesh = SynthMesh(dim, base, lmax, band=band, periodic=periodic)
    for rank in range(nranks):
        want = mesh.partition(rank, nranks, subgrid=True)
        q = lib.t8gpu_synth_query_create(mesh._h, rank, nranks)
        h = lib.t8gpu_host_connectivity_create_subgrid(q, dim)
        assert h
        try:
            cnt = np.zeros(6, np.int64)
            lib.t8gpu_host_connectivity_counts(h, cnt.ctypes.data)
            F, B = int(cnt[2]), int(cnt[3])
            assert (F, B) == (want.F, want.B)
            fn, nrm = np.zeros(2 * F + B, np.int32), np.zeros(3 * (F + B))
            ld, off = np.zeros(F, np.int32), np.zeros(dim * F, np.int32)
            lib.t8gpu_host_connectivity_arrays(h, fn.ctypes.data, nrm.ctypes.data, None, None, None, None, None, None)
            lib.t8gpu_host_connectivity_subgrid_arrays(h, ld.ctypes.data if F else None, off.ctypes.data if F else None)
            assert np.array_equal(fn, want.face_neighbors)
            assert np.array_equal(nrm.reshape(-1, 3)[:, :dim].reshape(-1), want.normals)
            assert np.array_equal(ld, want.level_diff) and np.array_equal(off, want.nb_offset)
            assert (ld <= 0).all()
        finally:
            lib.t8gpu_host_connectivity_destroy(h)
            lib.t8gpu_synth_query_destroy(q)
