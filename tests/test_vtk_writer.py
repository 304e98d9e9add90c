"""CPU: the host VTU / PVTU writer -- geometry, bookkeeping fields and data round trip, both
encodings, plain and Subgrid (z-order) layouts."""
import ctypes as C
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from t8gpu_amd import synth
from tests._vtu import read_vtu


def write(path, part, cells_per_dim, fields, ascii):
    lib = synth.lib()
    lib.t8gpu_host_write_vtu.restype = C.c_int
    lib.t8gpu_host_write_vtu.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    names = (C.c_char_p * len(fields))(*[n.encode() for n, _ in fields])
    comps = np.array([1 if a.ndim == 1 else 3 for _, a in fields], np.int32)
    arrs = [np.ascontiguousarray(a, np.float64) for _, a in fields]
    ptrs = (C.c_void_p * len(fields))(*[a.ctypes.data for a in arrs])
    cen, lev = np.ascontiguousarray(part.centres[: part.N]), np.ascontiguousarray(part.levels[: part.N])
    return lib.t8gpu_host_write_vtu(str(path).encode(), part.mesh.dim, part.N, cen.ctypes.data, lev.ctypes.data, cells_per_dim,
                                    part.rank, part.first_global, len(fields), names, comps.ctypes.data, ptrs, int(ascii))


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("ascii", [True, False])
def test_plain_piece_roundtrip(tmp_path, dim, ascii):
    mesh = synth.SynthMesh(dim, 2, 4, band=0.1)
    part = mesh.partition(1, 2)
    rng = np.random.default_rng(dim)
    rho, mom = rng.standard_normal(part.N), rng.standard_normal((part.N, 3))
    assert write(tmp_path / "p.vtu", part, 1, [("rho", rho), ("momentum", mom)], ascii) == 0
    v = read_vtu(tmp_path / "p.vtu")
    corners = 2 ** dim
    assert v["n_cells"] == part.N and v["n_points"] == part.N * corners
    A = v["arrays"]
    assert np.array_equal(A["rho"], rho) and np.array_equal(A["momentum"], mom)          # %.17g is lossless
    assert (A["types"] == (12 if dim == 3 else 9)).all()
    assert np.array_equal(A["offsets"], corners * np.arange(1, part.N + 1))
    assert np.array_equal(A["connectivity"], np.arange(part.N * corners))
    assert (A["mpirank"] == 1).all() and (A["treeid"] == 0).all()
    assert np.array_equal(A["level"], part.levels[: part.N])
    assert np.array_equal(A["element_id"], part.first_global + np.arange(part.N))
    # geometry: corner mean = centre, cell measure = volume, and the corner order is a valid quad / hexahedron
    P = A["Position"].reshape(part.N, corners, 3)
    assert np.allclose(P.mean(axis=1)[:, :dim], part.centres[: part.N, :dim], atol=1e-15)
    ext = P.max(axis=1) - P.min(axis=1)
    assert np.allclose(np.prod(ext[:, :dim], axis=1), part.volumes[: part.N], rtol=1e-14)
    e01 = P[:, 1] - P[:, 0]
    e03 = P[:, 3] - P[:, 0]
    assert (e01[:, 0] > 0).all() and (e03[:, 1] > 0).all() and np.allclose(e01[:, 1:], 0) and np.allclose(e03[:, [0, 2]], 0)
    if dim == 3:
        assert ((P[:, 4] - P[:, 0])[:, 2] > 0).all()


@pytest.mark.parametrize("dim", [2, 3])
def test_subgrid_piece_is_twice_refined_z_order(tmp_path, dim):
    mesh = synth.SynthMesh(dim, 1, 2, band=0.2)
    part = mesh.partition(0, 1, subgrid=True)
    S = 4 ** dim
    # per-cell value = its own centre coordinate, laid out in z-order by the restatement of
    # subgrid_mesh_manager.inl:1008-1049, must land on the cell the writer draws at that position
    h = 0.5 ** part.levels[: part.N]
    vals = np.zeros((3, part.N * S))
    for flat in range(S):
        ijk = [(flat >> (2 * a)) & 3 for a in range(3)]
        m = 0
        for lbit in range(2):
            for a in range(dim):
                m |= ((ijk[a] >> lbit) & 1) << (dim * lbit + a)
        for a in range(dim):
            vals[a, np.arange(part.N) * S + m] = part.centres[: part.N, a] - h / 2 + (ijk[a] + 0.5) * h / 4
    fields = [("cx", vals[0]), ("cy", vals[1])] + ([("cz", vals[2])] if dim == 3 else [])
    assert write(tmp_path / "s.vtu", part, 4, fields, False) == 0
    v = read_vtu(tmp_path / "s.vtu")
    assert v["n_cells"] == part.N * S
    A = v["arrays"]
    mid = A["Position"].reshape(part.N * S, 2 ** dim, 3).mean(axis=1)
    assert np.allclose(mid[:, 0], A["cx"], atol=1e-15) and np.allclose(mid[:, 1], A["cy"], atol=1e-15)
    if dim == 3:
        assert np.allclose(mid[:, 2], A["cz"], atol=1e-15)
    assert np.array_equal(A["level"], np.repeat(part.levels[: part.N] + 2, S))
    assert np.array_equal(A["element_id"], np.arange(part.N * S))


def test_pvtu_lists_pieces_and_rejects_bad_arguments(tmp_path):
    lib = synth.lib()
    lib.t8gpu_host_write_pvtu.restype = C.c_int
    lib.t8gpu_host_write_pvtu.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    files = (C.c_char_p * 2)(b"kh_0000.vtu", b"kh_0001.vtu")
    names = (C.c_char_p * 2)(b"rho", b"momentum")
    comps = np.array([1, 3], np.int32)
    assert lib.t8gpu_host_write_pvtu(str(tmp_path / "kh.pvtu").encode(), 2, files, 2, names, comps.ctypes.data) == 0
    root = ET.parse(tmp_path / "kh.pvtu").getroot()
    assert [p.get("Source") for p in root.iter("Piece")] == ["kh_0000.vtu", "kh_0001.vtu"]
    cd = {d.get("Name"): d.get("NumberOfComponents") for d in root.find("PUnstructuredGrid/PCellData")}
    assert cd["rho"] == "1" and cd["momentum"] == "3" and {"treeid", "mpirank", "level", "element_id"} <= set(cd)
    part = synth.SynthMesh(2, 1, 1).partition()
    assert write(tmp_path / "x.vtu", part, 3, [], True) == 1                      # cells_per_dim must be 1 or 4
    assert write(tmp_path / "nodir" / "x.vtu", part, 1, [], True) == 2            # cannot open
