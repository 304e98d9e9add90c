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


def test_fused_kernels_properties_at_c3q_size():
    """The reference's 2D example at its own size (examples/subgrid/main_2d.cu: levels 9-10, 581 632 blocks = 9.3 M subcells,
    fp64): the 2D family kernel + leftover blocks -- conservation, reproducibility, fused == compat."""
    mesh = SynthMesh(2, 9, 10, band=0.1)
    part = mesh.partition(subgrid=True)
    a = SubgridSolver(part, torch.float64, mode="fused")
    b = SubgridSolver(part, torch.float64, mode="fused")
    c = SubgridSolver(part, torch.float64, mode="compat")
    assert a.plan.host.n_families > 0.9 * part.N / 4
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    cellvol = torch.from_numpy(np.repeat(part.volumes / 16, 16)).cuda()
    m0 = (a.state() * cellvol).sum(1)
    for _ in range(2):
        a.iterate(dt)
        b.iterate(dt)
        c.iterate(dt)
    assert torch.equal(a.state(), b.state())
    m1 = (a.state() * cellvol).sum(1)
    assert float((m1 - m0).abs().max()) < 1e-12 * float(m0.abs().max())
    av, cv = a.state().cpu().numpy(), c.state().cpu().numpy()
    assert np.abs(av - cv).max() / np.abs(cv).max() < 1e-11


@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLLC])
def test_long_run_stays_physical_and_conservative(kind):
    """Subgrid<4,4> Kelvin-Helmholtz with hanging block faces to t ~ 1.5 (fp64): positive density and pressure,
    integrals conserved to rounding (up to the RK coefficients' known deficit), total physical entropy never decreasing."""
    mesh = SynthMesh(2, 3, 5, band=0.05)
    part = mesh.partition(subgrid=True)
    s = SubgridSolver(part, torch.float64, flux_kind=kind, mode="fused")
    vol = torch.from_numpy(np.repeat(part.volumes / 16, 16)).cuda()

    def diagnostics():
        u = s.state()
        rho = u[0]
        p = 0.4 * (u[4] - 0.5 * (u[1] ** 2 + u[2] ** 2 + u[3] ** 2) / rho)
        assert bool(torch.isfinite(u).all()) and float(rho.min()) > 0 and float(p.min()) > 0
        return (u * vol).sum(1), float((rho * (torch.log(p) - 1.4 * torch.log(rho)) * vol).sum())

    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    m0, e_prev = diagnostics()
    steps = 0
    for _ in range(8):
        for _ in range(250):
            s.iterate(dt)
        m, e = diagnostics()
        # the reference's truncated third-stage coefficients sum to 1 - 1e-14 (ssp_runge_kutta.inl:12-14):
        # every step scales the integrals by exactly that; what is left after taking it out is rounding (the large
        # uniform regions of the initial state round identically in every element, so allow one ulp per step)
        steps += 250
        assert float((m - m0 * (0.33333333333333 + 0.66666666666666) ** steps).abs().max()) < 2.3e-16 * steps * float(m0.abs().max())
        assert e >= e_prev - 1e-12 * abs(e_prev)
        e_prev = e


_ADDRESSING_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from _gpu import perturbed_state
from t8gpu_amd.solver import SubgridSolver
from t8gpu_amd.synth import SynthMesh
out = []
for dim, args in ((3, dict(base_level=3, max_level=4, band=0.03, periodic=False)), (2, dict(base_level=3, max_level=6, band=0.03))):
    mesh = SynthMesh(dim, **args)
    part = mesh.partition(subgrid=True)
    for dtype in (torch.float32, torch.float64):
        g = SubgridSolver(part, dtype, mode="fused", state=perturbed_state(part, 7))
        for _ in range(3):
            g.iterate(0.1 * 2.0 ** -(mesh.finest_level + 2))
        out.append(g.state().double().cpu().numpy().ravel())
np.save(sys.argv[1], np.concatenate(out))
"""


def test_32_bit_and_64_bit_plane_addressing_agree_bitwise(tmp_path):
    """The block kernel addresses planes shorter than 4 GiB by 32-bit byte offsets; the 64-bit form (T8GPU_SG_WIDE=1,
    what a larger rank gets) must give the same bits. Child processes: the switch is read once per process."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "child.py"
    script.write_text(_ADDRESSING_CHILD.format(root=os.path.dirname(here), tests=here))
    res = []
    for wide in ("0", "1"):
        out = tmp_path / f"state_{wide}.npy"
        subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, T8GPU_SG_WIDE=wide), check=True, timeout=300)
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1])


_FAMILY_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from _gpu import perturbed_state
from t8gpu_amd import hip
from t8gpu_amd.solver import SubgridSolver
from t8gpu_amd.synth import SynthMesh
out = []
for dim, args in ((3, dict(base_level=2, max_level=2)), (3, dict(base_level=3, max_level=4, band=0.03)),
                  (3, dict(base_level=2, max_level=4, band=0.05, periodic=False)), (2, dict(base_level=3, max_level=3)),
                  (2, dict(base_level=3, max_level=6, band=0.03)), (2, dict(base_level=3, max_level=5, band=0.03, periodic=False))):
    mesh = SynthMesh(dim, **args)
    part = mesh.partition(subgrid=True)
    for dtype in (torch.float32, torch.float64):
        for kind in (hip.KEPES, hip.HLL, hip.HLLC):
            g = SubgridSolver(part, dtype, flux_kind=kind, mode="fused", state=perturbed_state(part, 11))
            assert g.plan.host.n_families > 0
            for _ in range(3):
                g.iterate(0.1 * 2.0 ** -(mesh.finest_level + 2))
            out.append(g.state().double().cpu().numpy().ravel())
np.save(sys.argv[1], np.concatenate(out))
"""


def test_family_kernel_and_block_kernel_agree_bitwise(tmp_path):
    """2x2x2 cubes (3D) / 2x2 squares (2D) of same-level blocks run through the family kernels (inner coarse faces
    evaluated once, outward far cells pooled); T8GPU_SG_FAMILY=0 sends every block through the block kernel. Same
    fluxes, same summation order: the states must agree bit for bit (periodic, walled, 2:1 meshes; three fluxes)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "child.py"
    script.write_text(_FAMILY_CHILD.format(root=os.path.dirname(here), tests=here))
    res = []
    # the family kernels; every block through the block kernel
    for tag, env in (("family", {}), ("block", dict(T8GPU_SG_FAMILY="0"))):
        out = tmp_path / f"state_{tag}.npy"
        subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, **env), check=True, timeout=600)
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    for other in res[1:]:
        assert np.array_equal(res[0], other), int((res[0] != other).sum())
