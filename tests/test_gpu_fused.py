"""-m gpu: fused tile kernels (flux + RK in one launch) against the CPU oracle and the compat tier."""
import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu
DTYPES = [torch.float64, torch.float32]
MESHES = [dict(dim=2, base_level=5, max_level=5), dict(dim=2, base_level=3, max_level=7, band=0.06),
          dict(dim=2, base_level=3, max_level=5, band=0.06, periodic=False), dict(dim=3, base_level=2, max_level=4, band=0.1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
@pytest.mark.parametrize("mesh_args", MESHES)
def test_fused_iterate_vs_oracle(dtype, kind, mesh_args):
    mesh = SynthMesh(**mesh_args)
    part = mesh.partition()
    st = perturbed_state(part, 21)
    g = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=st)
    o = O.PlainCase(part, NP[dtype], state=st)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    g.iterate(dt)
    o.iterate(dt, kind=kind)
    torch.cuda.synchronize()
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :part.N]) < TOL1[dtype]
    if True:                                                     # speed estimates of the last stage, every face, every flux
        assert o.speed.min() > 0
        assert rel_err(g.speed.cpu().numpy()[None, :part.F + part.B], o.speed[None]) < TOL1[dtype] * 10
    assert (g.planes[20:25] == 0).all()                          # Fluxes planes stay zero, as after the reference's RK
    for _ in range(9):
        g.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(g.state().cpu().numpy(), o.current()[:, :part.N]) < TOL10[dtype]


@pytest.mark.parametrize("tmax,fcap,compressed", [(256, 512, False), (64, 100, True), (17, 40, True), (256, 10 ** 6, True),
                                                  (200, 300, False), (256, 700, True), (256, 1024, True)])
def test_fused_is_independent_of_the_tiling(tmax, fcap, compressed):
    """Pipelined (ELL + geometry dictionary) and generic kernel variants, any tiling: bitwise equal."""
    from t8gpu_amd import fused
    mesh = SynthMesh(2, 3, 6, band=0.06, periodic=False)
    part = mesh.partition()
    st = perturbed_state(part, 4)
    ref = PlainSolver(part, torch.float64, mode="fused", state=st)
    alt = PlainSolver(part, torch.float64, mode="fused", state=st)
    alt.plan = fused.PlainPlan(part, torch.float64, tmax=tmax, fcap=fcap, compressed=compressed)
    dt = 0.1 * 2.0 ** -6
    for _ in range(3):
        ref.iterate(dt)
        alt.iterate(dt)
    # per-element sums run in face order whatever the tiling: bitwise identical
    assert torch.equal(ref.state(), alt.state())


def test_fused_is_bitwise_reproducible_and_conservative_at_c2_size():
    """BASELINE C2 (2D KH AMR, ~1.03 M elements, fp64): size-independent properties at full size."""
    mesh = SynthMesh(2, 6, 11, band=0.0596)
    part = mesh.partition()
    a = PlainSolver(part, torch.float64, mode="fused")
    b = PlainSolver(part, torch.float64, mode="fused")
    c = PlainSolver(part, torch.float64, mode="compat")
    dt = 0.1 * 2.0 ** -11
    vol = torch.from_numpy(part.volumes).cuda()
    m0 = (a.state() * vol).sum(1)
    for _ in range(3):
        a.iterate(dt)
        b.iterate(dt)
        c.iterate(dt)
    assert torch.equal(a.state(), b.state())                                   # no atomics: run-to-run identical
    m1 = (a.state() * vol).sum(1)
    assert float((m1 - m0).abs().max()) < 1e-12 * float(m0.abs().max())         # conservation (periodic mesh)
    assert rel_err(a.state().cpu().numpy(), c.state().cpu().numpy()) < 1e-12    # fused == reference dataflow


def test_uniform_state_is_a_fixed_point_fused():
    mesh = SynthMesh(2, 3, 6, band=0.06)
    part = mesh.partition()
    uniform = np.tile(np.array([[1.3], [0.2], [-0.4], [0.1], [3.0]]), (1, part.N))
    f = PlainSolver(part, torch.float64, mode="fused", state=uniform)
    f.iterate(1e-3)
    assert rel_err(f.state().cpu().numpy(), uniform) < 1e-13


@pytest.mark.parametrize("dtype", DTYPES)
def test_native_stepper_equals_python_driven_iterate(dtype):
    mesh = SynthMesh(2, 3, 7, band=0.06, periodic=False)
    part = mesh.partition()
    st = perturbed_state(part, 8)
    a = PlainSolver(part, dtype, mode="fused", state=st)
    b = PlainSolver(part, dtype, mode="fused", state=st)
    stepper = b.use_native_stepper()
    stepper.timing(True)
    dt = 0.1 * 2.0 ** -7
    for _ in range(4):
        a.iterate(dt)
        b.iterate(dt)
    torch.cuda.synchronize()
    assert (a.next, a.prev) == (b.next, b.prev)
    assert torch.equal(a.planes, b.planes) and torch.equal(a.speed, b.speed)
    ms, n = stepper.elapsed()
    assert n == 12 and ms > 0


@pytest.mark.parametrize("n", [1, 2, 5])
def test_native_stepper_many_steps_in_one_call(n):
    mesh = SynthMesh(2, 3, 6, band=0.06)
    part = mesh.partition()
    st = perturbed_state(part, 9)
    a = PlainSolver(part, torch.float64, mode="fused", state=st)
    b = PlainSolver(part, torch.float64, mode="fused", state=st)
    b.use_native_stepper()
    dt = 0.1 * 2.0 ** -6
    for _ in range(n):
        a.iterate(dt)
    b.iterate_steps(n, dt)
    torch.cuda.synchronize()
    assert (a.next, a.prev) == (b.next, b.prev)
    assert torch.equal(a.state(), b.state())
    b.iterate(dt)                                              # and the roles it leaves behind are the right ones
    a.iterate(dt)
    assert torch.equal(a.state(), b.state())


def test_native_comm_single_rank_and_stream_wait():
    from t8gpu_amd import native
    comm = native.NativeComm(0, 1, lambda b, src: b)           # nranks = 1: bootstrap + init only
    part = SynthMesh(2, 3, 5, band=0.06).partition()
    halo = native.NativeHalo(part, torch.float64, comm)
    planes = torch.zeros(5, part.N, dtype=torch.float64, device="cuda")
    halo.exchange(planes)                                      # no peers: must be a no-op
    assert native.stream_wait(torch.cuda.current_stream(), 5.0) == 0
    comm.destroy()


def test_c4_full_size_properties():
    """BASELINE C4 / north-star mesh (9.93 M elements, fp64): reproducibility, conservation, finiteness at full size."""
    mesh = SynthMesh(2, 7, 12, band=0.1472)
    part = mesh.partition()
    assert part.N == 9929728
    a = PlainSolver(part, torch.float64, mode="fused")
    a.use_native_stepper()
    m0 = [a.compute_integral(k) for k in range(5)]
    dt = 0.1 * 2.0 ** -12
    for _ in range(3):
        a.iterate(dt)
    first = a.state().clone()
    m1 = [a.compute_integral(k) for k in range(5)]
    assert max(abs(x - y) for x, y in zip(m0, m1)) < 1e-12 * max(abs(x) for x in m0)
    assert bool(torch.isfinite(first).all())
    a.planes[5 * a.next:5 * a.next + 5, :part.N] = torch.from_numpy(part.kh_initial_state()[:, :part.N]).cuda()
    a.next, a.prev = 0, 3
    a.planes[0:5, :part.N] = a.planes[5 * 0:5, :part.N]
    b = PlainSolver(part, torch.float64, mode="fused")               # python-driven stages, fresh state
    for _ in range(3):
        b.iterate(dt)
    assert torch.equal(b.state(), first)                             # native driver == python driver, run-to-run identical


def test_c5_full_size_properties():
    """BASELINE config 5's family at size -- 3D hexahedral AMR forest, levels 6-8, 3.93 M elements, 6-24 faces per element,
    fp64 (the mesh `bench.py --workload c5` times): run-to-run bitwise identical, conservative on the periodic domain,
    finite, and the fused tier equals the reference data flow (compat tier: face kernel + atomics + RK kernel,
    examples/compressible_euler/kernels.cu:135-309) within the fp64 parity tolerance."""
    mesh = SynthMesh(3, 6, 8, band=0.05)
    part = mesh.partition()
    assert part.N == 3932160
    a = PlainSolver(part, torch.float64, mode="fused")
    b = PlainSolver(part, torch.float64, mode="fused")
    c = PlainSolver(part, torch.float64, mode="compat")
    a.use_native_stepper()
    m0 = [a.compute_integral(k) for k in range(5)]
    dt = 0.1 * 2.0 ** -8
    a.iterate_steps(3, dt)
    for _ in range(3):
        b.iterate(dt)
        c.iterate(dt)
    torch.cuda.synchronize()
    m1 = [a.compute_integral(k) for k in range(5)]
    assert bool(torch.isfinite(a.state()).all())
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)      # native driver == python driver, bitwise
    assert max(abs(x - y) for x, y in zip(m0, m1)) < 1e-12 * max(abs(x) for x in m0)
    # (normalised by the largest value of the whole state: the z-momentum of this z-independent flow is rounding noise in
    #  both tiers, a per-variable norm would compare noise with noise)
    fa, fc = a.state().cpu().numpy(), c.state().cpu().numpy()
    assert np.abs(fa - fc).max() < 1e-12 * np.abs(fc).max()


def test_lds_scatter_add_variant_matches_the_oracle():
    """T8GPU_LDS_SCATTER=1: the accumulation the project brief sketches (ds_add_f64 into per-element LDS
    accumulators instead of the ELL gather). Not bitwise reproducible by construction, so it is checked against
    the oracle within the parity tolerance, in its own process (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
import _oracle as O
from _gpu import perturbed_state, rel_err
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh
mesh = SynthMesh(2, 3, 6, band=0.06, periodic=False)
part = mesh.partition()
st = perturbed_state(part, 21)
g = PlainSolver(part, torch.float64, mode="fused", state=st)
o = O.PlainCase(part, np.float64, state=st)
dt = 0.1 * 2.0 ** -6
m0 = g.compute_integral(0)
for _ in range(10):
    g.iterate(dt); o.iterate(dt)
err = rel_err(g.state().cpu().numpy(), o.current()[:, :part.N])
print("ERR", err, abs(g.compute_integral(0) - m0) / abs(m0))
assert err < 1e-10
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, T8GPU_LDS_SCATTER="1"), capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and "ERR" in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]


@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
def test_long_run_stays_physical_and_conservative(kind):
    """Kelvin-Helmholtz on an AMR mesh to t ~ 1.5 (4 000 steps through the native driver, fp64): density and pressure
    stay positive, the five integrals are conserved to rounding (up to the RK coefficients' known deficit), and the
    total physical entropy never decreases (KEPES is entropy-stable by construction, kernels.cu:38-133; HLL / HLLC
    are dissipative)."""
    mesh = SynthMesh(2, 5, 8, band=0.06)
    part = mesh.partition()
    s = PlainSolver(part, torch.float64, flux_kind=kind, mode="fused")
    s.use_native_stepper()
    vol = torch.from_numpy(part.volumes).cuda()

    def diagnostics():
        u = s.state()
        rho = u[0]
        p = 0.4 * (u[4] - 0.5 * (u[1] ** 2 + u[2] ** 2 + u[3] ** 2) / rho)
        assert bool(torch.isfinite(u).all()) and float(rho.min()) > 0 and float(p.min()) > 0
        entropy = float((rho * (torch.log(p) - 1.4 * torch.log(rho)) * vol).sum())
        return (u * vol).sum(1), entropy

    dt = 0.1 * 2.0 ** -mesh.finest_level
    m0, e_prev = diagnostics()
    steps = 0
    for _ in range(8):
        s.iterate_steps(500, dt)
        torch.cuda.synchronize()
        m, e = diagnostics()
        # the reference's truncated third-stage coefficients sum to 1 - 1e-14 (ssp_runge_kutta.inl:12-14):
        # every step scales the integrals by exactly that; what is left after taking it out is rounding (the large
        # uniform regions of the initial state round identically in every element, so allow one ulp per step)
        steps += 500
        assert float((m - m0 * (0.33333333333333 + 0.66666666666666) ** steps).abs().max()) < 2.3e-16 * steps * float(m0.abs().max())
        assert e >= e_prev - 1e-12 * abs(e_prev)
        e_prev = e


@pytest.mark.parametrize("kind", [hip.HLL, hip.HLLC])
def test_cfl_timestep_works_for_every_flux_kind(kind):
    """compute_timestep (solver.cu:213-229) needs per-face speed estimates; HLL / HLLC write max(|S_l|, |S_r|)."""
    mesh = SynthMesh(2, 4, 6, band=0.06)
    part = mesh.partition()
    g = PlainSolver(part, torch.float64, flux_kind=kind, mode="fused")
    k = PlainSolver(part, torch.float64, flux_kind=hip.KEPES, mode="fused")
    dt = 0.1 * 2.0 ** -mesh.finest_level
    g.iterate(dt)
    k.iterate(dt)
    torch.cuda.synchronize()
    assert float(g.speed[:part.F].min()) > 0
    step, step_kepes = g.compute_timestep(cfl=0.7), k.compute_timestep(cfl=0.7)
    assert 0 < step < float("inf") and 0.5 < step / step_kepes < 2.0      # same physics, two signal-speed estimates


def test_stage_one_requires_prev_and_mid_to_be_the_same_planes():
    """Stage 1 is u1 = u0 + dt/vol f(u0): the pipelined kernel never reads `prev` there, the generic one does, so a
    call with prev != mid is refused (hipErrorInvalidValue = 1) instead of giving variant-dependent results."""
    import ctypes as C
    mesh = SynthMesh(2, 3, 4, band=0.06)
    part = mesh.partition()
    g = PlainSolver(part, torch.float64, mode="fused")
    lib = hip.lib()
    args = lambda prev, mid: (hip.KEPES, 1, C.byref(g.plan.c), 0, g.plan.host.ntiles, g.get_own_variables(prev), g.get_own_variables(mid),
                              g.get_own_variables(1), hip.ptr(g.planes[25]), C.c_double(1e-3), hip.ptr(g.speed), hip.stream_ptr())
    assert lib.t8gpu_hip_plain_fused_stage_f64(*args(0, 0)) == 0
    assert lib.t8gpu_hip_plain_fused_stage_f64(*args(3, 0)) == 1
    torch.cuda.synchronize()


_PERSISTENT_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from _gpu import perturbed_state
from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh
out = []
for dim, args in ((2, dict(base_level=4, max_level=7, band=0.05)), (2, dict(base_level=4, max_level=6, band=0.05, periodic=False)),
                  (3, dict(base_level=3, max_level=5, band=0.05)), (3, dict(base_level=3, max_level=3)),
                  (3, dict(base_level=2, max_level=4, band=0.08, periodic=False))):
    mesh = SynthMesh(dim, **args)
    part = mesh.partition()
    for dtype in (torch.float32, torch.float64):
        for kind in (hip.KEPES, hip.HLL):
            # (patches=False: every tile is a generic tile, i.e. the persistent tile kernel's -- the 2D meshes would
            #  otherwise go through the patch kernel, tests/test_gpu_patch.py)
            g = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=perturbed_state(part, 5), plan_options=dict(patches=False))
            for _ in range(3):
                g.iterate(0.1 * 2.0 ** -(mesh.finest_level + 2))
            out.append(g.state().double().cpu().numpy().ravel())
            out.append(g.speed.double().cpu().numpy().ravel())
np.save(sys.argv[1], np.concatenate(out))
"""


def test_persistent_and_one_tile_kernels_agree_bitwise(tmp_path):
    """T8GPU_PERSISTENT=2 sends every whole-plan launch through the persistent, software-pipelined kernel (2D meshes:
    8-entry face lists; 3D AMR: 16- / 24-entry lists, second chunk prefetched, third on demand), =0 through the one-tile
    kernels. Same arithmetic, same summation order: states and speed estimates must agree bit for bit."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "child.py"
    script.write_text(_PERSISTENT_CHILD.format(root=os.path.dirname(here), tests=here))
    res = []
    for mode in ("2", "0"):
        out = tmp_path / f"state_{mode}.npy"
        # (T8GPU_PERSISTENT_WGS also switches the launcher's size heuristic off: these meshes are small)
        subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, T8GPU_PERSISTENT=mode, T8GPU_PERSISTENT_WGS="3"),
                       check=True, timeout=600)
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1]), int((res[0] != res[1]).sum())


_PERSISTENT_ORACLE_CHILD = """
import ctypes, sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh
q = hip.lib().t8gpu_hip_last_stage_kernel
q.restype = ctypes.c_char_p
worst = 0.0
for dim, args in ((2, dict(base_level=4, max_level=7, band=0.05)), (2, dict(base_level=4, max_level=6, band=0.05, periodic=False)),
                  (3, dict(base_level=3, max_level=5, band=0.05)), (3, dict(base_level=2, max_level=4, band=0.08, periodic=False))):
    mesh = SynthMesh(dim, **args)
    part = mesh.partition()
    st = perturbed_state(part, 23)
    for dtype in (torch.float64, torch.float32):
        for kind in (hip.KEPES, hip.HLL):
            g = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=st, plan_options=dict(patches=False))
            o = O.PlainCase(part, NP[dtype], state=st)
            dt = 0.1 * 2.0 ** -mesh.finest_level
            g.iterate(dt)
            o.iterate(dt, kind=kind)
            torch.cuda.synchronize()
            assert q().decode().startswith("k_plain_persistent<"), q()          # the headline kernel of round 2, directly
            e1 = rel_err(g.state().cpu().numpy(), o.current()[:, :part.N])
            es = rel_err(g.speed.cpu().numpy()[None, :part.F + part.B], o.speed[None])
            assert e1 < TOL1[dtype] and es < 10 * TOL1[dtype], (dim, args, dtype, kind, e1, es)
            for _ in range(9):
                g.iterate(dt)
                o.iterate(dt, kind=kind)
            e10 = rel_err(g.state().cpu().numpy(), o.current()[:, :part.N])
            assert e10 < TOL10[dtype], (dim, args, dtype, kind, e10)
            worst = max(worst, e1 / TOL1[dtype], e10 / TOL10[dtype])
print("worst error / tolerance:", worst)
"""


def test_persistent_kernel_vs_oracle(tmp_path):
    """The persistent tile kernel (T8GPU_PERSISTENT=2: every whole-plan launch, whatever the mesh size) against the CPU
    oracle DIRECTLY -- 2D and 3D AMR meshes, periodic and walled, KEPES / HLL, both precisions, 1 and 10 steps, speed
    estimates included -- not only through its bitwise agreement with the one-tile kernels.
    Reference: examples/compressible_euler/kernels.cu:135-469, ssp_runge_kutta.inl:30-99."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = tmp_path / "child.py"
    script.write_text(_PERSISTENT_ORACLE_CHILD.format(root=os.path.dirname(here), tests=here))
    res = subprocess.run([sys.executable, str(script)], env=dict(os.environ, T8GPU_PERSISTENT="2", T8GPU_PERSISTENT_WGS="3"),
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "worst error / tolerance:" in res.stdout
