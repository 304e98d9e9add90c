"""-m gpu: fused Subgrid<4,4,4> block kernel against the CPU oracle and the compat tier."""
import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import SubgridSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
@pytest.mark.parametrize("dim,mesh_args", [(3, dict(base_level=2, max_level=2)), (3, dict(base_level=3, max_level=4, band=0.03)),
                                           (3, dict(base_level=3, max_level=4, band=0.03, periodic=False)),
                                           (2, dict(base_level=3, max_level=3)), (2, dict(base_level=3, max_level=6, band=0.03)),
                                           (2, dict(base_level=3, max_level=5, band=0.03, periodic=False))])
def test_fused_block_kernel_vs_oracle(dtype, kind, dim, mesh_args):
    mesh = SynthMesh(dim, **mesh_args)
    part = mesh.partition(subgrid=True)
    st = perturbed_state(part, 31)
    g = SubgridSolver(part, dtype, flux_kind=kind, mode="fused", state=st)
    o = O.SubgridCase(part, NP[dtype], state=st)
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    g.iterate(dt)
    o.iterate(dt, kind=kind)
    torch.cuda.synchronize()
    n = part.N * g.S
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :n]) < TOL1[dtype]
    assert (g.planes[20:25] == 0).all()
    for _ in range(4):
        g.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :n]) < TOL10[dtype]


def test_fused_block_kernel_properties_at_c3_size():
    """BASELINE C3-sized mesh (204 800 blocks = 13.1 M subcells, fp32): conservation, reproducibility, fused == compat."""
    mesh = SynthMesh(3, 5, 6, band=0.17)
    part = mesh.partition(subgrid=True)
    a = SubgridSolver(part, torch.float32, mode="fused")
    b = SubgridSolver(part, torch.float32, mode="fused")
    c = SubgridSolver(part, torch.float32, mode="compat")
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    cellvol = torch.from_numpy(np.repeat(part.volumes / 64, 64)).cuda()
    m0 = (a.state().double() * cellvol).sum(1)
    for _ in range(2):
        a.iterate(dt)
        b.iterate(dt)
        c.iterate(dt)
    assert torch.equal(a.state(), b.state())
    m1 = (a.state().double() * cellvol).sum(1)
    assert float((m1 - m0).abs().max()) < 2e-6 * float(m0.abs().max())
    av, cv = a.state().double().cpu().numpy(), c.state().double().cpu().numpy()
    assert np.abs(av - cv).max() / np.abs(cv).max() < 2e-5       # rho_v2 is ~0 in this case: normalise globally
