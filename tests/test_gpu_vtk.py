"""-m gpu: SURVEY 8f-4 read-back kernels (column_major_to_z_order, host scalar / vector variables) against
numpy restatements of subgrid_mesh_manager.inl:1008-1049,1140-1183, and the VTK files end to end."""
import numpy as np
import pytest
import torch

from t8gpu_amd import vtk
from t8gpu_amd.solver import PlainSolver, SubgridSolver
from t8gpu_amd.synth import SynthMesh
from tests._vtu import read_vtu

pytestmark = pytest.mark.gpu


def z_order_permutation(dim):
    """to[morton(i,j,k)] = from[i + 4 j + 16 k] (subgrid_mesh_manager.inl:1017-1024,1043-1048)."""
    S = 4 ** dim
    perm = np.zeros(S, np.int64)
    for flat in range(S):
        c = [(flat >> (2 * a)) & 3 for a in range(3)]
        m = 0
        for lbit in range(2):
            for a in range(dim):
                m |= ((c[a] >> lbit) & 1) << (dim * lbit + a)
        perm[m] = flat
    return perm


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_z_order_and_host_variables(dim, dtype):
    part = SynthMesh(dim, 2, 3, band=0.1).partition(subgrid=True)
    s = SubgridSolver(part, dtype=dtype, mode="fused")
    S, n = 4 ** dim, part.N * 4 ** dim
    state = s.state().cpu().numpy()[:, :n]
    z = vtk.column_major_to_z_order(s, s.step_planes(s.next)[0][:n].contiguous()).cpu().numpy()
    want = state[0].reshape(part.N, S)[:, z_order_permutation(dim)].reshape(-1)
    assert np.array_equal(z, want)
    rho = vtk.get_host_scalar_variable(s, s.next, 0, "rho")
    mom = vtk.get_host_vector_variable(s, s.next, (1, 2, 3), "momentum")
    assert rho.type == vtk.SCALAR and rho.data.dtype == np.float64 and np.array_equal(rho.data, state[0].astype(np.float64))
    assert mom.type == vtk.VECTOR and np.array_equal(mom.data, state[1:4].T.astype(np.float64))


def test_plain_and_subgrid_files(tmp_path):
    part = SynthMesh(2, 3, 5, band=0.05).partition()
    s = PlainSolver(part, mode="fused")
    s.iterate(1e-4)
    f = vtk.save_variables_to_vtk(s, [vtk.get_host_scalar_variable(s, s.next, 0, "rho"),
                                      vtk.get_host_vector_variable(s, s.next, (1, 2, 3), "momentum")], str(tmp_path / "kh"))
    v = read_vtu(f)
    st = s.state().cpu().numpy()
    assert v["n_cells"] == part.N and np.array_equal(v["arrays"]["rho"], st[0, : part.N])
    assert np.array_equal(v["arrays"]["momentum"], st[1:4, : part.N].T)

    sp = SynthMesh(3, 1, 2, band=0.2).partition(subgrid=True)
    g = SubgridSolver(sp, dtype=torch.float32, mode="fused")
    f = vtk.save_variable_to_vtk(g, g.next, 0, str(tmp_path / "sub"), ascii=True)
    v = read_vtu(f)
    assert v["n_cells"] == sp.N * 64
    want = g.state().cpu().numpy()[0, : sp.N * 64].reshape(sp.N, 64)[:, z_order_permutation(3)].reshape(-1).astype(np.float64)
    assert np.array_equal(v["arrays"]["variables"], want)
    # the KH density is 2 inside the band |z - 0.5| < 0.25 and 1 outside (solver.inl:35-56): the value must sit
    # on the cell drawn at that height
    mid = v["arrays"]["Position"].reshape(-1, 8, 3).mean(axis=1)
    inside = np.abs(mid[:, 2] - 0.5) < 0.25   # 3D: the shear layer is normal to z
    assert (v["arrays"]["variables"][inside] == 2).all() and (v["arrays"]["variables"][~inside] == 1).all()
    m = vtk.save_mesh_to_vtk(g, str(tmp_path / "mesh"))
    assert read_vtu(m)["n_cells"] == sp.N
