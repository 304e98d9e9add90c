"""-m gpu: the plain-element kernels on the mesh class of the reference's own example (prisms + hexahedra,
curved geometry: 5 / 6 faces per element, a different oblique normal on every face, walls) against the
CPU oracle -- compat tier, the three fused kernel variants, a partitioned run, conservation."""
import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.unstructured import PrismHexMesh, shell_map, wavy_map

pytestmark = pytest.mark.gpu
DTYPES = [torch.float64, torch.float32]
VARIANTS = {"dictionary": {}, "per-face geometry": dict(dictionary=False), "generic": dict(compressed=False),
            "four passes": dict(fcap=1024), "four passes, per-face geometry": dict(fcap=1024, dictionary=False)}


def time_step(part):
    return 0.1 * float(np.cbrt(part.volumes.min()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
def test_compat_tier_on_curved_prisms(dtype, kind):
    part = PrismHexMesh((8, 8, 4), split="checker", mapping=shell_map).partition()
    st = perturbed_state(part, 5)
    g, o = PlainSolver(part, dtype, flux_kind=kind, state=st), O.PlainCase(part, NP[dtype], state=st)
    dt = time_step(part)
    for _ in range(3):
        g.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, : part.N]) < 3 * TOL1[dtype]
    if kind == hip.KEPES:
        assert rel_err(g.speed.cpu().numpy()[None, : part.F + part.B], o.speed[None]) < 10 * TOL1[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("mesh", [dict(n=(8, 8, 8), split=0.5, mapping=shell_map), dict(n=(16, 8, 4), split="all", mapping=wavy_map, periodic=True)])
def test_fused_tier_on_curved_prisms(dtype, kind, variant, mesh):
    part = PrismHexMesh(**mesh).partition()
    st = perturbed_state(part, 6)
    g = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=st, plan_options=VARIANTS[variant])
    o = O.PlainCase(part, NP[dtype], state=st)
    dt = time_step(part)
    g.iterate(dt)
    o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, : part.N]) < TOL1[dtype]
    if kind == hip.KEPES:
        assert rel_err(g.speed.cpu().numpy()[None, : part.F + part.B], o.speed[None]) < 10 * TOL1[dtype]
    for _ in range(9):
        g.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, : part.N]) < TOL10[dtype]


def test_fused_variants_agree_and_conserve():
    mesh = PrismHexMesh((16, 16, 8), split=0.5, mapping=wavy_map, periodic=True)
    part = mesh.partition()
    dt = time_step(part)
    res = []
    for opt in VARIANTS.values():
        g = PlainSolver(part, torch.float64, mode="fused", plan_options=opt)
        before = [g.compute_integral(k) for k in range(5)]
        for _ in range(5):
            g.iterate(dt)
        after = [g.compute_integral(k) for k in range(5)]
        assert max(abs(a - b) for a, b in zip(after, before)) < 1e-12 * max(abs(b) for b in before)
        res.append(g.state().clone())
    # the dictionary holds face frames computed on the host; the other two build them on the device: a few ulp
    for r in res[1:]:
        assert rel_err(r.cpu().numpy(), res[0].cpu().numpy()) < 1e-13
    uni = np.tile(np.array([[1.3], [0.2], [-0.1], [0.05], [3.0]]), (1, part.N))
    g = PlainSolver(part, torch.float64, mode="fused", state=uni)
    g.iterate(dt)
    assert np.abs(g.state().cpu().numpy() - uni).max() < 1e-12


@pytest.mark.parametrize("nranks", [2, 5])
def test_partitioned_run_equals_single_rank_bitwise(nranks):
    """Contiguous ranges of the element numbering, halo through the one-GPU loopback transport."""
    from t8gpu_amd.halo import HaloExchange
    from test_gpu_halo import loopback
    mesh = PrismHexMesh((16, 16, 8), split=0.5, mapping=shell_map)
    whole = mesh.partition()
    dt = time_step(whole)
    single = PlainSolver(whole, torch.float64, mode="fused")
    parts = [mesh.partition(r, nranks) for r in range(nranks)]
    solvers, halos = [], []
    for part in parts:
        local = part.kh_initial_state().copy()
        local[:, part.N:] = np.nan                              # ghost values must arrive through the exchange
        solvers.append(PlainSolver(part, torch.float64, mode="fused", state=local, plan_options=dict(tmax=32, fcap=120)))
        halos.append(HaloExchange(part, torch.float64, dist=None, overlap=False))
    assert all(s.plan.host.n_interior < s.plan.host.ntiles for s in solvers) and any(s.plan.host.n_interior > 0 for s in solvers)
    for _ in range(4):
        single.iterate(dt)
        for s in solvers:
            s.begin_step()
        for k in range(3):
            for s, h in zip(solvers, halos):
                h._pack(s.step_planes(s.stage_steps(k)[0]))
            loopback(halos)
            for s, h in zip(solvers, halos):
                h._unpack(s.step_planes(s.stage_steps(k)[0]))
            for s in solvers:
                s.run_stage(k, dt, split=True)
    torch.cuda.synchronize()
    got = torch.cat([s.state() for s in solvers], dim=1).cpu().numpy()
    assert np.array_equal(got, single.state().cpu().numpy())


# ---- BASELINE config 5's mesh class: mixed tetrahedra / hexahedra --------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL])
@pytest.mark.parametrize("mode,options", [("compat", None), ("fused", {}), ("fused", dict(compressed=False))])
def test_tet_hex_mesh_vs_oracle(dtype, kind, mode, options):
    """Tetrahedra (4 faces) and hexahedra (6-12 faces where they meet tetrahedra) on a curved shell with walls, every
    kernel tier against the oracle on the same arrays."""
    from t8gpu_amd.unstructured import TetHexMesh
    part = TetHexMesh((8, 8, 4), tets="blocks", mapping=shell_map).partition()
    st = perturbed_state(part, 9)
    g = PlainSolver(part, dtype, flux_kind=kind, mode=mode, state=st, plan_options=options)
    o = O.PlainCase(part, NP[dtype], state=st)
    dt = 0.5 * time_step(part)
    g.iterate(dt)
    o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, : part.N]) < 3 * TOL1[dtype]
    for _ in range(9):
        g.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, : part.N]) < TOL10[dtype]


def test_tet_hex_partitioned_run_equals_single_rank_bitwise():
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.unstructured import TetHexMesh
    from test_gpu_halo import loopback
    mesh = TetHexMesh((8, 8, 8), tets="half", mapping=shell_map)
    whole = mesh.partition()
    st = perturbed_state(whole, 3)
    ref = PlainSolver(whole, torch.float64, mode="fused", state=st)
    parts = [mesh.partition(r, 3) for r in range(3)]
    solvers = [PlainSolver(p, torch.float64, mode="fused", state=st[:, np.concatenate([p.first_global + np.arange(p.N), p.ghost_global])])
               for p in parts]
    halos = [HaloExchange(p, torch.float64, dist=None, overlap=False) for p in parts]
    dt = 0.5 * time_step(whole)
    for _ in range(3):
        ref.iterate(dt)
        for s in solvers:
            s.begin_step()
        for k in range(3):
            for s, h in zip(solvers, halos):
                h._pack(s.step_planes(s.stage_steps(k)[0]))
            loopback(halos)
            for s, h in zip(solvers, halos):
                h._unpack(s.step_planes(s.stage_steps(k)[0]))
            for s in solvers:
                s.run_stage(k, dt, split=True)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([s.state() for s in solvers], dim=1), ref.state())


def test_c5t_full_size_properties():
    """BASELINE config 5's mesh class at size: `bench.py --workload c5t` -- 4.13 M tetrahedra + hexahedra on a curved shell
    sector (4 / 6-12 faces per element, every normal oblique, reflective walls all round), fp64. Size-independent
    properties: run-to-run bitwise identical, mass and energy conserved between walls, finite, fused tier = reference data
    flow (compat tier) within the fp64 tolerance. (The real t8code cmesh of examples/compressible_euler/main.cu:23 needs
    t8code; this is its geometry-synthetic stand-in)"""
    from t8gpu_amd.unstructured import TetHexMesh
    part = TetHexMesh((96, 96, 128), tets="blocks").partition()
    assert part.N == 4128768 and part.B > 0
    a = PlainSolver(part, torch.float64, mode="fused")
    b = PlainSolver(part, torch.float64, mode="fused")
    c = PlainSolver(part, torch.float64, mode="compat")
    a.use_native_stepper()
    m0 = [a.compute_integral(k) for k in (0, 4)]
    dt = 0.05 * float(part.volumes.min()) ** (1 / 3)
    a.iterate_steps(3, dt)
    for _ in range(3):
        b.iterate(dt)
        c.iterate(dt)
    torch.cuda.synchronize()
    m1 = [a.compute_integral(k) for k in (0, 4)]
    assert bool(torch.isfinite(a.state()).all())
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
    assert max(abs(x - y) / abs(x) for x, y in zip(m0, m1)) < 1e-12                  # reflective walls: no mass / energy flux
    assert rel_err(a.state().cpu().numpy(), c.state().cpu().numpy()) < 1e-12
