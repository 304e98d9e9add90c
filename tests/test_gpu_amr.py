"""-m gpu: SURVEY 8f-3 -- AMR indicator, data-transfer kernel and the adapt loop against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL10, perturbed_state, rel_err
from t8gpu_amd import amr, hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_indicator_kernels_vs_oracle(dtype):
    mesh = SynthMesh(2, 3, 6, band=0.06)
    part = mesh.partition()
    st = perturbed_state(part, 3)
    g = PlainSolver(part, dtype, state=st)
    crit = amr.refinement_criteria(g).cpu().numpy()
    npdt = NP[dtype]
    rho = st[0].astype(npdt)
    grad = np.zeros(part.N, npdt)
    sf = O.suf(npdt)
    getattr(O.lib(), "oracle_estimate_gradient_" + sf)(part.F, O.p(part.face_neighbors), None, O.p(rho), O.p(grad))
    want = np.zeros(part.N, npdt)
    vol = part.volumes.astype(npdt)
    getattr(O.lib(), "oracle_refinement_criteria_" + sf)(part.N, O.p(grad), O.p(vol), O.p(want))
    assert np.abs(crit - want).max() <= (1e-12 if dtype == torch.float64 else 1e-4) * np.abs(want).max()
    assert (g.planes[20, :part.N] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("dim", [2, 3])
def test_transfer_kernel_vs_oracle_and_conservation(dtype, dim):
    mesh = SynthMesh(dim, 3, 5 if dim == 2 else 4, band=0.06)
    part = mesh.partition()
    rng = np.random.default_rng(5)
    x = part.centres[:part.N, 0]
    marks = np.where(x < 0.5, -1, np.where(x > 0.75, rng.integers(0, 2, part.N), 0)).astype(np.int8)   # coarsen left, refine right
    new_mesh, ad = mesh.adapt(marks)
    assert set(np.unique(np.diff(ad))) >= {0, 1, 2 ** dim}
    new_part = new_mesh.partition()
    npdt = NP[dtype]
    st = perturbed_state(part, 8).astype(npdt)
    vol = part.volumes.astype(npdt)
    old = torch.from_numpy(np.vstack([st, vol[None]])).cuda().contiguous()
    new = torch.zeros((6, new_part.N), dtype=dtype, device="cuda")
    dad = torch.from_numpy(ad).cuda()
    hip.call("t8gpu_hip_adapt_variables_and_volume", dtype, new_part.N, dim, hip.ptr(dad), hip.vars_of(old), hip.vars_of(new),
             hip.ptr(old[5]), hip.ptr(new[5]), hip.stream_ptr())
    torch.cuda.synchronize()
    got = new.cpu().numpy()
    want = np.zeros((5, new_part.N), npdt)
    wvol = np.zeros(new_part.N, npdt)
    getattr(O.lib(), "oracle_adapt_variables_and_volume_" + O.suf(npdt))(new_part.N, dim, O.p(ad), O.p(st), C.c_size_t(part.N),
                                                                           O.p(want), C.c_size_t(new_part.N), O.p(vol), O.p(wvol))
    assert np.array_equal(got[:5], want) and np.array_equal(got[5], wvol)       # same operation order: bit-exact
    assert np.allclose(got[5], new_part.volumes, rtol=1e-6)                      # the transferred volumes ARE the new cells' volumes
    m_old = (st.astype(np.float64) * part.volumes).sum(1)
    m_new = (got[:5].astype(np.float64) * new_part.volumes).sum(1)
    assert np.abs(m_new - m_old).max() < (1e-13 if dtype == torch.float64 else 1e-6) * np.abs(m_old).max()


@pytest.mark.parametrize("dim", [2, 3])
def test_adaptive_run_follows_the_oracle(dim):
    """iterate / adapt / iterate ... on the device vs the same sequence with the oracle's kernels on the host
    (dim = 3: the small version of bench.py's c5a loop -- hexahedral forest, adapt every few steps)."""
    mesh = SynthMesh(2, 4, 6, band=0.03) if dim == 2 else SynthMesh(3, 3, 4, band=0.05)
    part = mesh.partition()
    st = None if dim == 2 else perturbed_state(part, 8)      # (3D KH has rho v2 = 0 exactly: no scale for a relative error)
    g = PlainSolver(part, torch.float64, mode="fused", state=st)
    g.use_native_stepper()
    o = O.PlainCase(part, np.float64, state=st)
    m0 = g.compute_integral(0)
    sizes = [part.N]
    for cycle in range(3):
        dt = 0.1 * 2.0 ** -g.part.mesh.finest_level
        for _ in range(5):
            g.iterate(dt)
            o.iterate(dt)
        # device adapt
        g, marks, ad = amr.adapt(g, threshold=10.0, min_level=3 if dim == 2 else 2, max_level=7 if dim == 2 else 5)
        # host adapt with the oracle's kernels on the oracle's state
        opart = o.part
        rho = o.current()[0, :opart.N].copy()
        grad = np.zeros(opart.N)
        O.lib().oracle_estimate_gradient_f64(opart.F, O.p(opart.face_neighbors), None, O.p(rho), O.p(grad))
        crit = np.zeros(opart.N)
        O.lib().oracle_refinement_criteria_f64(opart.N, O.p(grad), O.p(opart.volumes), O.p(crit))
        omarks = opart.mesh.marks_from_criteria(crit, 10.0, 3 if dim == 2 else 2, 7 if dim == 2 else 5)
        assert np.array_equal(omarks, marks)                                      # same decisions on both sides
        nmesh, oad = opart.mesh.adapt(omarks)
        npart = nmesh.partition()
        cur = np.ascontiguousarray(o.current()[:, :opart.N])
        nst = np.zeros((5, npart.N))
        nvol = np.zeros(npart.N)
        O.lib().oracle_adapt_variables_and_volume_f64(npart.N, dim, O.p(oad), O.p(cur), C.c_size_t(opart.N), O.p(nst), C.c_size_t(npart.N),
                                                      O.p(opart.volumes), O.p(nvol))
        nxt, prv = o.next, o.prev
        o = O.PlainCase(npart, np.float64, state=np.zeros((5, npart.N)))
        o.next, o.prev = nxt, prv
        o.planes[5 * o.next:5 * o.next + 5, :npart.N] = nst
        sizes.append(npart.N)
        assert g.N == npart.N
        assert rel_err(g.state().cpu().numpy(), o.current()[:, :npart.N]) < TOL10[torch.float64]
    assert len(set(sizes)) > 1                                                     # the mesh really changed
    assert abs(g.compute_integral(0) - m0) < 1e-12 * abs(m0)                       # mass conserved through adapt cycles


@pytest.mark.parametrize("world", [2, 5])
def test_partitioned_adapt_and_repartition_equals_single_rank(world):
    """k ranks on one GPU (loopback transport): criteria all-gathered, families cut by a rank boundary kept,
    local transfer, element runs shipped to their owners in the new equal split."""
    mesh = SynthMesh(2, 4, 6, band=0.03)
    whole = mesh.partition()
    st = perturbed_state(whole, 12)
    st[0] += 1.0 * (np.abs(whole.centres[:, 0] - 0.3) < 0.1)          # a density bump: refinement away from the KH layers too
    st[4] += 0.5
    ref = PlainSolver(whole, torch.float64, mode="fused", state=st)
    parts = [mesh.partition(r, world) for r in range(world)]
    solvers = []
    for part in parts:
        gidx = np.concatenate([part.first_global + np.arange(part.N), part.ghost_global])
        solvers.append(PlainSolver(part, torch.float64, mode="fused", state=st[:, gidx]))
    kw = dict(threshold=10.0, min_level=3, max_level=7)
    crits = [amr.refinement_criteria(s).double().cpu().numpy() for s in solvers]
    all_crit = np.concatenate(crits)
    ref_crit = amr.refinement_criteria(ref).double().cpu().numpy()
    assert np.allclose(all_crit, ref_crit, rtol=1e-12, atol=1e-12)         # ghost values are current: same indicator
    pas = [amr.PartitionedAdapt(s, all_crit, **kw) for s in solvers]
    by_rank = {p.rank: p for p in pas}
    for p in pas:                                                         # loopback transport
        for q, _, n in p.sends:
            if q != p.rank:
                by_rank[q].recvbufs[p.rank].copy_(p.sendbufs[q])
    news = [p.finish() for p in pas]
    # reference: single-rank adapt with the same (split-family-aware) marks
    new_mesh, ad = mesh.adapt(pas[0].marks)
    npart = new_mesh.partition()
    want = torch.zeros((6, npart.N), dtype=torch.float64, device="cuda")
    hip.call("t8gpu_hip_adapt_variables_and_volume", torch.float64, npart.N, 2, hip.ptr(torch.from_numpy(ad).cuda()),
             ref.get_own_variables(ref.next), hip.vars_of(want), hip.ptr(ref.planes[25]), hip.ptr(want[5]), hip.stream_ptr())
    torch.cuda.synchronize()
    got = torch.cat([n.state() for n in news], dim=1)
    gvol = torch.cat([n.planes[25, :n.N] for n in news])
    assert sum(n.N for n in news) == npart.N and max(n.N for n in news) - min(n.N for n in news) <= 1   # balanced again
    assert torch.equal(got, want[:5]) and torch.equal(gvol, want[5])
    assert any(len(p.sends) > 1 for p in pas)                              # elements really changed owner


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("dim", [2, 3])
def test_subgrid_indicator_and_transfer_vs_oracle(dtype, dim):
    from t8gpu_amd.solver import SubgridSolver
    mesh = SynthMesh(dim, 3, 5 if dim == 2 else 4, band=0.03)
    part = mesh.partition(subgrid=True)
    S = 4 ** dim
    npdt = NP[dtype]
    st = perturbed_state(part, 15)
    g = SubgridSolver(part, dtype, state=st)
    # indicator
    crit = amr.subgrid_refinement_criteria(g).cpu().numpy()
    want = np.zeros(part.N, npdt)
    sf = O.suf(npdt)
    rho = np.ascontiguousarray(st[0].astype(npdt))
    vol = part.volumes.astype(npdt)
    getattr(O.lib(), "oracle_subgrid_refinement_criteria_" + sf)(dim, part.N, O.p(rho), O.p(vol), O.p(want))
    assert np.abs(crit - want).max() <= (1e-12 if dtype == torch.float64 else 2e-5) * np.abs(want).max()
    # transfer: coarsen the left half, refine part of the right half
    rng = np.random.default_rng(6)
    x = part.centres[:part.N, 0]
    marks = np.where(x < 0.5, -1, np.where(x > 0.75, rng.integers(0, 2, part.N), 0)).astype(np.int8)
    new_mesh, ad = mesh.adapt(marks)
    assert set(np.unique(np.diff(ad))) >= {0, 1, 2 ** dim}
    npart = new_mesh.partition(subgrid=True)
    new = torch.zeros((5, npart.N * S), dtype=dtype, device="cuda")
    nvol = torch.zeros(npart.N, dtype=dtype, device="cuda")
    hip.call("t8gpu_hip_subgrid_adapt_variables_and_volume", dtype, dim, npart.N, hip.ptr(torch.from_numpy(ad).cuda()),
             g.get_own_variables(0), hip.vars_of(new), hip.ptr(g.volumes), hip.ptr(nvol), hip.stream_ptr())
    torch.cuda.synchronize()
    old = np.ascontiguousarray(st.astype(npdt))
    wst = np.zeros((5, npart.N * S), npdt)
    wvol = np.zeros(npart.N, npdt)
    getattr(O.lib(), "oracle_subgrid_adapt_variables_and_volume_" + sf)(dim, npart.N, O.p(ad), O.p(old), C.c_size_t(part.N * S), O.p(wst),
                                                                        C.c_size_t(npart.N * S), O.p(vol), O.p(wvol))
    assert np.array_equal(new.cpu().numpy(), wst) and np.array_equal(nvol.cpu().numpy(), wvol)
    assert np.allclose(wvol, npart.volumes, rtol=1e-6)
    m_old = (old.astype(np.float64) * np.repeat(part.volumes / S, S)).sum(1)
    m_new = (wst.astype(np.float64) * np.repeat(npart.volumes / S, S)).sum(1)
    assert np.abs(m_new - m_old).max() < (1e-13 if dtype == torch.float64 else 1e-6) * np.abs(m_old).max()


def test_subgrid_adaptive_run_stays_conservative():
    from t8gpu_amd.solver import SubgridSolver
    mesh = SynthMesh(2, 3, 3)
    g = SubgridSolver(mesh.partition(subgrid=True), torch.float64, mode="fused")
    S = g.S

    def mass(s):
        return float((s.state()[0] * torch.from_numpy(np.repeat(s.part.volumes[:s.N] / S, S)).cuda()).sum())

    m0 = mass(g)
    sizes = [g.N]
    for _ in range(3):
        g, _, _ = amr.adapt_subgrid(g, threshold=0.02, min_level=3, max_level=5)
        sizes.append(g.N)
        dt = 0.1 * 2.0 ** -(g.part.mesh.finest_level + 2)
        for _ in range(4):
            g.iterate(dt)
    assert len(set(sizes)) > 1 and bool(torch.isfinite(g.state()).all())
    assert abs(mass(g) - m0) < 1e-12 * abs(m0)


@pytest.mark.parametrize("world,dim", [(2, 2), (3, 3)])
def test_partitioned_subgrid_adapt_and_repartition_equals_single_rank(world, dim):
    """Subgrid blocks: k ranks on one GPU (loopback transport) against the single-rank adapt with the same marks;
    the repartitioned run then advances and stays bitwise equal to the single-rank run on the adapted mesh."""
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.solver import SubgridSolver
    from test_gpu_halo import loopback
    mesh = SynthMesh(dim, 2, 3, band=0.1)
    whole = mesh.partition(subgrid=True)
    S = 4 ** dim
    st = whole.kh_initial_state().copy()
    rough = np.arange(whole.N * S) < (whole.N // 5) * S                  # a rough patch at the start of the curve:
    st[0, rough] *= 1 + 0.3 * np.random.default_rng(3).random(int(rough.sum()))   # refinement there only => blocks move
    ref = SubgridSolver(whole, torch.float64, mode="fused", state=st)
    parts = [mesh.partition(r, world, subgrid=True) for r in range(world)]
    solvers = []
    for p in parts:
        gidx = np.concatenate([p.first_global + np.arange(p.N), p.ghost_global])
        cells = (gidx[:, None] * S + np.arange(S)[None, :]).reshape(-1)
        solvers.append(SubgridSolver(p, torch.float64, mode="fused", state=st[:, cells]))
    kw = dict(threshold=0.02, min_level=2, max_level=4)
    all_crit = np.concatenate([amr.subgrid_refinement_criteria(s).double().cpu().numpy() for s in solvers])
    assert np.array_equal(all_crit, amr.subgrid_refinement_criteria(ref).double().cpu().numpy())
    pas = [amr.PartitionedSubgridAdapt(s, all_crit, **kw) for s in solvers]
    by_rank = {p.rank: p for p in pas}
    for p in pas:
        for q, _, n in p.sends:
            if q != p.rank:
                by_rank[q].recvbufs[p.rank].copy_(p.sendbufs[q])
    news = [p.finish() for p in pas]
    new_mesh, ad = mesh.adapt(pas[0].marks)
    npart = new_mesh.partition(subgrid=True)
    assert new_mesh.num_elements != mesh.num_elements
    want = SubgridSolver(npart, torch.float64, mode="fused", state=np.zeros((5, npart.N * S)))
    hip.call("t8gpu_hip_subgrid_adapt_variables_and_volume", torch.float64, dim, npart.N, hip.ptr(torch.from_numpy(ad).cuda()),
             ref.get_own_variables(ref.next), want.get_own_variables(want.next), hip.ptr(ref.volumes), hip.ptr(want.volumes),
             hip.stream_ptr())
    torch.cuda.synchronize()
    assert sum(n.N for n in news) == npart.N and max(n.N for n in news) - min(n.N for n in news) <= 1
    assert torch.equal(torch.cat([n.state() for n in news], dim=1), want.state())
    assert torch.equal(torch.cat([n.volumes[: n.N] for n in news]), want.volumes[: npart.N])
    assert any(len(p.sends) > 1 for p in pas)
    # advance both: ghosts of the new partition arrive through the exchange
    halos = [HaloExchange(n.part, torch.float64, dist=None, overlap=False) for n in news]
    dt = 0.1 * 2.0 ** -(new_mesh.finest_level + 2)
    for _ in range(2):
        want.iterate(dt)
        for s in news:
            s.begin_step()
        for k in range(3):
            for s, h in zip(news, halos):
                h._pack(s.step_planes(s.stage_steps(k)[0]))
            loopback(halos)
            for s, h in zip(news, halos):
                h._unpack(s.step_planes(s.stage_steps(k)[0]))
            for s in news:
                s.run_stage(k, dt, split=True)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([n.state() for n in news], dim=1), want.state())
