"""-m gpu: the reference-dataflow HIP kernels (C-ABI) against the CPU oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import FLUXES, PlainSolver, SubgridSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu
DTYPES = [torch.float64, torch.float32]


def plain_pair(mesh, dtype, seed=1, **kw):
    part = mesh.partition()
    st = perturbed_state(part, seed)
    return PlainSolver(part, dtype, state=st, **kw), O.PlainCase(part, NP[dtype], state=st), part


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
def test_plain_flux_kernels_vs_oracle(dtype, kind):
    mesh = SynthMesh(2, 3, 6, band=0.06, periodic=False)          # AMR + hanging faces + walls
    g, o, part = plain_pair(mesh, dtype, flux_kind=kind)
    assert part.B > 0 and part.F > 0
    s = hip.stream_ptr()
    st, fl = g.get_own_variables(0), g.get_own_variables(FLUXES)
    hip.call("t8gpu_hip_flux_faces", dtype, kind, g.F, g.ndim, hip.ptr(g.fn), None, hip.ptr(g.normals),
             hip.ptr(g.areas), st, fl, hip.ptr(g.speed), s)
    hip.call("t8gpu_hip_flux_boundary", dtype, kind, g.F, g.B, g.ndim, hip.ptr(g.fn), hip.ptr(g.normals),
             hip.ptr(g.areas), st, fl, hip.ptr(g.speed), s)
    torch.cuda.synchronize()
    sf = O.suf(NP[dtype])
    getattr(O.lib(), "oracle_plain_interior_faces_" + sf)(kind, part.F, 3, O.p(o.fn), O.p(part.indices), O.p(o.normals),
                                                          O.p(o.areas), O.p(o.planes[0:5]), O.p(o.planes[20:25]),
                                                          C.c_size_t(o.stride), O.p(o.speed))
    getattr(O.lib(), "oracle_plain_boundary_faces_" + sf)(kind, part.F, part.B, 3, O.p(o.fn), O.p(o.normals), O.p(o.areas),
                                                          O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride),
                                                          O.p(o.speed))
    got = g.planes[20:25, :part.N].cpu().numpy()
    want = o.planes[20:25, :part.N]
    # fluxes are differences of O(area) terms: normalise by the largest single-face contribution
    scale = np.abs(want).max(axis=1, keepdims=True) + part.areas.max()
    assert (np.abs(got - want) / scale).max() < TOL1[dtype]
    # per-face wave-speed estimates for every flux kind (HLL / HLLC: max |S| of the wave-speed bounds)
    assert o.speed.min() > 0 and rel_err(g.speed.cpu().numpy()[None], o.speed[None]) < TOL1[dtype]
    # conservation: interior faces add -F and +F
    if part.B == 0:
        assert np.abs(got.sum(1)).max() < 1e-10


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stage", [1, 2, 3])
def test_plain_rk_stage_vs_oracle(dtype, stage):
    n, cap = 1000, 1111                                           # ragged: N not a multiple of the block, stride > N
    rng = np.random.default_rng(stage)
    host = rng.standard_normal((26, cap)).astype(NP[dtype])
    host[25] = rng.uniform(0.5, 2.0, cap)
    dev = torch.from_numpy(host.copy()).cuda()
    dt = 1e-3
    hip.call("t8gpu_hip_rk3_stage", dtype, stage, n, hip.vars_of(dev, 0), hip.vars_of(dev, 1), hip.vars_of(dev, 2),
             hip.vars_of(dev, 4), hip.ptr(dev[25]), hip.fscalar(dtype, dt), hip.stream_ptr())
    torch.cuda.synchronize()
    ref = host.copy()
    getattr(O.lib(), "oracle_plain_rk_stage_" + O.suf(NP[dtype]))(stage, n, O.p(ref[0:5]), O.p(ref[5:10]), O.p(ref[10:15]),
                                                                  O.p(ref[20:25]), C.c_size_t(cap), O.p(ref[25]),
                                                                  O.fs(NP[dtype], dt))
    got = dev.cpu().numpy()
    assert np.array_equal(got[10:15, :n], ref[10:15, :n])          # same operation order, no contraction: bit-exact
    assert (got[20:25, :n] == 0).all()                             # fluxes zeroed
    assert np.array_equal(got[:, n:], host[:, n:])                 # nothing written past N
    assert np.array_equal(got[0:10], host[0:10])


def test_empty_inputs_are_no_ops():
    dev = torch.zeros((26, 8), dtype=torch.float64, device="cuda")
    v = hip.vars_of(dev, 0)
    hip.call("t8gpu_hip_flux_faces", torch.float64, 0, 0, 3, None, None, None, None, v, v, None, hip.stream_ptr())
    hip.call("t8gpu_hip_rk3_stage", torch.float64, 1, 0, v, v, v, v, None, C.c_double(0.1), hip.stream_ptr())
    hip.call("t8gpu_hip_subgrid_outer", torch.float64, 0, 3, 0, None, None, None, None, None, None, v, v, hip.stream_ptr())
    torch.cuda.synchronize()
    with pytest.raises(hip.T8gpuHipError):
        hip.call("t8gpu_hip_rk3_stage", torch.float64, 7, 4, v, v, v, v, hip.ptr(dev[25]), C.c_double(0.1), hip.stream_ptr())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mesh_args", [dict(base_level=5, max_level=5), dict(base_level=3, max_level=6, band=0.06),
                                       dict(base_level=3, max_level=5, band=0.06, periodic=False)])
def test_plain_iterate_vs_oracle(dtype, mesh_args):
    mesh = SynthMesh(2, **mesh_args)
    g, o, part = plain_pair(mesh, dtype)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    g.iterate(dt)
    o.iterate(dt)
    torch.cuda.synchronize()
    assert (g.next, g.prev) == (o.next, o.prev) == (3, 0)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :part.N]) < TOL1[dtype]
    for _ in range(9):
        g.iterate(dt)
        o.iterate(dt)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :part.N]) < TOL10[dtype]
    if part.B == 0:                                                # conservation of sum(vol * u) on periodic meshes
        u0 = (perturbed_state(part, 1) * part.volumes).sum(1)
        u1 = (g.state().double().cpu().numpy() * part.volumes).sum(1)
        assert np.abs(u1 - u0).max() < (1e-12 if dtype == torch.float64 else 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim,mesh_args", [(3, dict(base_level=3, max_level=4, band=0.03)),
                                            (2, dict(base_level=3, max_level=5, band=0.03)),
                                            (3, dict(base_level=3, max_level=4, band=0.03, periodic=False)),
                                            (2, dict(base_level=3, max_level=5, band=0.03, periodic=False))])
def test_subgrid_iterate_vs_oracle(dtype, dim, mesh_args):
    mesh = SynthMesh(dim, **mesh_args)
    part = mesh.partition(subgrid=True)
    assert (part.level_diff != 0).any()                            # hanging faces present
    st = perturbed_state(part, 5)
    g = SubgridSolver(part, dtype, state=st)
    o = O.SubgridCase(part, NP[dtype], state=st)
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)                     # main_2d.cu:27-30
    g.iterate(dt)
    o.iterate(dt)
    torch.cuda.synchronize()
    n = part.N * g.S
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :n]) < TOL1[dtype]
    for _ in range(4):
        g.iterate(dt)
        o.iterate(dt)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :n]) < TOL10[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_subgrid_kernels_individually(dtype):
    mesh = SynthMesh(3, 3, 4, band=0.03, periodic=False)
    part = mesh.partition(subgrid=True)
    assert (part.level_diff != 0).any() and part.B > 0
    st = perturbed_state(part, 9)
    g = SubgridSolver(part, dtype, state=st)
    o = O.SubgridCase(part, NP[dtype], state=st)
    s = hip.stream_ptr()
    sf = O.suf(NP[dtype])
    stv, fl = g.get_own_variables(0), g.get_own_variables(FLUXES)
    checks = []
    hip.call("t8gpu_hip_subgrid_inner", dtype, 0, 3, g.N, stv, fl, hip.ptr(g.volumes), s)
    getattr(O.lib(), "oracle_subgrid_inner_" + sf)(0, 3, part.N, O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride), O.p(o.volumes))
    checks.append((g.planes[20:25].cpu().numpy(), o.planes[20:25].copy()))
    hip.call("t8gpu_hip_subgrid_boundary", dtype, 0, 3, g.F, g.B, hip.ptr(g.fn), hip.ptr(g.normals), hip.ptr(g.areas), stv, fl, s)
    getattr(O.lib(), "oracle_subgrid_boundary_" + sf)(0, 3, part.F, part.B, O.p(o.fn), O.p(o.normals), O.p(o.areas), O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride))
    checks.append((g.planes[20:25].cpu().numpy(), o.planes[20:25].copy()))
    hip.call("t8gpu_hip_subgrid_outer", dtype, 0, 3, g.F, hip.ptr(g.fn), None, hip.ptr(g.level_diff), hip.ptr(g.nb_offset), hip.ptr(g.normals), hip.ptr(g.areas), stv, fl, s)
    getattr(O.lib(), "oracle_subgrid_outer_" + sf)(0, 3, part.F, O.p(o.fn), O.p(part.indices), O.p(part.level_diff), O.p(part.nb_offset), O.p(o.normals), O.p(o.areas), O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride))
    checks.append((g.planes[20:25].cpu().numpy(), o.planes[20:25].copy()))
    area = part.areas.max() / 16
    for got, want in checks:
        scale = np.abs(want).max(axis=1, keepdims=True) + area
        assert (np.abs(got - want) / scale).max() < TOL1[dtype]


def test_full_size_c1_properties():
    """BASELINE C1 (256^2 quads): size-independent properties at full size -- conservation and free stream."""
    mesh = SynthMesh(2, 8, 8)
    part = mesh.partition()
    g = PlainSolver(part, torch.float64)
    dt = 0.1 * 2.0 ** -8
    m0 = (g.state().cpu().numpy() * part.volumes).sum(1)
    for _ in range(3):
        g.iterate(dt)
    m1 = (g.state().cpu().numpy() * part.volumes).sum(1)
    assert np.abs(m1 - m0).max() < 1e-13 * max(1.0, np.abs(m0).max()) * 10
    uniform = np.tile(np.array([[1.3], [0.2], [-0.4], [0.1], [3.0]]), (1, part.N))
    f = PlainSolver(part, torch.float64, state=uniform)
    f.iterate(dt)
    assert rel_err(f.state().cpu().numpy(), uniform) < 1e-13      # uniform state is a fixed point (atomic order: few ulp)


# ---- the HIP kernels against the committed reference vectors (tests/golden/reference_flux_vectors.npz) -------------
def _golden_pair_mesh(g, tag, wall):
    """One face per vector: interior face i joins elements 2i (left) and 2i+1 (right); a wall face i belongs to element i.
    Unit areas, the vector's normal. In the reference's array formats."""
    nrm, L, R = g[f"xyz_n_{tag}"], g[f"xyz_L_{tag}"], g[f"xyz_R_{tag}"]
    n = nrm.shape[0]
    if wall:
        state = np.ascontiguousarray(L.T)                        # [5][n]
        fn = np.arange(n, dtype=np.int32)                        # boundary part of face_neighbors: [2F + B], F = 0
        return state, fn, 0, n
    state = np.empty((5, 2 * n), L.dtype)
    state[:, 0::2], state[:, 1::2] = L.T, R.T
    return state, np.arange(2 * n, dtype=np.int32), n, 0


@pytest.mark.parametrize("dtype,tag", [(torch.float64, "f64"), (torch.float32, "f32")])
@pytest.mark.parametrize("kind,name", [(hip.KEPES, "kepes"), (hip.HLL, "hll")])
@pytest.mark.parametrize("wall", [False, True])
def test_compat_flux_kernels_on_the_reference_vectors(dtype, tag, kind, name, wall):
    """kepes_compute_fluxes / reflective_boundary_condition of the compat tier fed with the generated reference vectors
    (generic, near-equal, strong-jump and supersonic pairs; axis and oblique normals): what element 2i+1 (interior) or
    element i (wall: minus the flux) receives must be the reference's xyz flux. The compat tier keeps the reference's
    operation order; only the device libm (log, sqrt, division sequences) may differ, hence a few ulp, not zero."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_flux_vectors.npz"))
    state, fn, F, B = _golden_pair_mesh(g, tag, wall)
    n = F + B
    nel = state.shape[1]
    planes = torch.zeros((26, nel), dtype=dtype, device="cuda")
    planes[0:5] = torch.from_numpy(state).cuda()
    d_fn = torch.from_numpy(fn).cuda()
    d_n = torch.from_numpy(np.ascontiguousarray(g[f"xyz_n_{tag}"])).cuda()
    d_a = torch.ones(n, dtype=dtype, device="cuda")
    speed = torch.zeros(n, dtype=dtype, device="cuda")
    st, fl = hip.vars_of(planes, 0), hip.vars_of(planes, FLUXES)
    if wall:
        hip.call("t8gpu_hip_flux_boundary", dtype, kind, 0, B, 3, hip.ptr(d_fn), hip.ptr(d_n), hip.ptr(d_a), st, fl,
                 hip.ptr(speed), hip.stream_ptr())
        got = -planes[20:25].cpu().numpy().T                       # the element loses the outward flux
        want = g[f"xyz_{name}_wall_{tag}"]
    else:
        hip.call("t8gpu_hip_flux_faces", dtype, kind, F, 3, hip.ptr(d_fn), None, hip.ptr(d_n), hip.ptr(d_a), st, fl,
                 hip.ptr(speed), hip.stream_ptr())
        fluxes = planes[20:25].cpu().numpy()
        got = fluxes[:, 1::2].T                                     # +F to the right element
        assert np.array_equal(fluxes[:, 0::2], -fluxes[:, 1::2])    # -F to the left one, same bits
        want = g[f"xyz_{name}_{tag}"]
    torch.cuda.synchronize()
    # per-vector scale: the largest flux component of that vector (strong jumps span 6 decades across the set)
    scale = np.abs(want).max(axis=1, keepdims=True) + np.finfo(want.dtype).tiny
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)) / scale
    # worst vector (near-equal and strong-jump pairs amplify the last ulp of the device's log / division sequences
    # through cancellation, most visibly in fp32) and the bulk of the set
    assert err.max() < (5e-13 if dtype == torch.float64 else 1e-4), err.max()
    assert np.quantile(err.max(axis=1), 0.99) < (2e-14 if dtype == torch.float64 else 5e-6), np.quantile(err.max(axis=1), 0.99)
