"""-m gpu: the structured-patch kernel (kernels_fused_patch.hip) against the CPU oracle and against the tile kernels.

A plan built with patches sends 16 x 16 same-level blocks through the patch kernel and everything else through the tile
kernels; built without (patches=False / T8GPU_PATCH=0) every element goes through the tile kernels. Both evaluate every
face with the same function on the same operands and add a cell's fluxes in ascending face id, so the two agree BIT FOR
BIT -- states and speed estimates -- and both meet the oracle within the parity tolerance
(examples/compressible_euler/kernels.cu:135-309, ssp_runge_kutta.inl:30-99)."""
import os

import numpy as np
import pytest
import torch

import _oracle as O
from _gpu import NP, TOL1, TOL10, perturbed_state, rel_err
from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DTYPES = [torch.float64, torch.float32]
# uniform (inner patches only: the periodic wrap turns the outermost faces round), AMR bands (patches next to coarser and
# finer tiles), walls all round
MESHES = [dict(dim=2, base_level=7, max_level=7), dict(dim=2, base_level=4, max_level=8, band=0.12),
          dict(dim=2, base_level=5, max_level=7, band=0.2, periodic=False)]


def _pair(part, dtype, kind, st, **plan_options):
    a = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=st, plan_options=dict(patches=True, **plan_options))
    b = PlainSolver(part, dtype, flux_kind=kind, mode="fused", state=st, plan_options=dict(patches=False, **plan_options))
    assert a.plan.host.n_patches > 0 and b.plan.host.n_patches == 0
    return a, b


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
@pytest.mark.parametrize("mesh_args", MESHES)
def test_patch_kernel_vs_oracle_and_bitwise_vs_tile_kernels(dtype, kind, mesh_args):
    mesh = SynthMesh(**mesh_args)
    part = mesh.partition()
    st = perturbed_state(part, 31)
    a, b = _pair(part, dtype, kind, st)
    o = O.PlainCase(part, NP[dtype], state=st)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    a.iterate(dt)
    b.iterate(dt)
    o.iterate(dt, kind=kind)
    torch.cuda.synchronize()
    assert rel_err(a.state().cpu().numpy(), o.current()[:, :part.N]) < TOL1[dtype]
    assert rel_err(a.speed.cpu().numpy()[None, :part.F + part.B], o.speed[None]) < TOL1[dtype] * 10
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
    assert (a.planes[20:25] == 0).all()
    for _ in range(9):
        a.iterate(dt)
        b.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(a.state().cpu().numpy(), o.current()[:, :part.N]) < TOL10[dtype]
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)


MESHES3 = [dict(dim=3, base_level=5, max_level=5), dict(dim=3, base_level=4, max_level=6, band=0.1),
           dict(dim=3, base_level=5, max_level=5, periodic=False)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", [hip.KEPES, hip.HLL, hip.HLLC])
@pytest.mark.parametrize("mesh_args", MESHES3)
def test_patch3_kernel_vs_oracle_and_bitwise_vs_tile_kernels(dtype, kind, mesh_args):
    """The 3D patch kernel (kernels_fused_patch3.hip: 8 x 8 x 4 blocks of same-size hexahedra) on uniform, 2:1-refined and
    walled hexahedral meshes: within the parity tolerance of the oracle, bit for bit the tile kernels (states, speeds)."""
    mesh = SynthMesh(**mesh_args)
    part = mesh.partition()
    st = perturbed_state(part, 41)
    a, b = _pair(part, dtype, kind, st, irregular=True)     # (asked for: left alone, fp32 plans drop the form where it does not pay)
    assert a.plan.host.patch_dim == 3
    # every one of these meshes has blocks next to a periodic wrap, a coarser neighbour or a wall: IRREGULAR patches (the
    # irregular instantiation of k_plain_patch3) beside the regular ones
    assert sum(a.plan.host.n_irregular_class) > 0 and (mesh_args["base_level"] < 5 or sum(a.plan.host.n_irregular_class) < a.plan.host.n_patches)
    o = O.PlainCase(part, NP[dtype], state=st)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    a.iterate(dt)
    b.iterate(dt)
    o.iterate(dt, kind=kind)
    torch.cuda.synchronize()
    assert rel_err(a.state().cpu().numpy(), o.current()[:, :part.N]) < TOL1[dtype]
    assert rel_err(a.speed.cpu().numpy()[None, :part.F + part.B], o.speed[None]) < TOL1[dtype] * 10
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
    for _ in range(9):
        a.iterate(dt)
        b.iterate(dt)
        o.iterate(dt, kind=kind)
    assert rel_err(a.state().cpu().numpy(), o.current()[:, :part.N]) < TOL10[dtype]
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)


@pytest.mark.parametrize("mesh_args", [dict(dim=3, base_level=4, max_level=4), dict(dim=3, base_level=4, max_level=6, band=0.1, periodic=False)])
def test_irregular_patches_alone_and_switched_off(mesh_args):
    """A 16^3 periodic cube is irregular patches only (every block touches a wrap); with irregular=False (T8GPU_PATCH_IRREGULAR=0)
    those blocks stay generic tiles. Same bits either way, and through partial launches that cut the irregular range."""
    mesh = SynthMesh(**mesh_args)
    part = mesh.partition()
    st = perturbed_state(part, 43)
    a = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True, irregular=True))
    b = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True, irregular=False))
    c = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True, irregular=True))
    ha, hb = a.plan.host, b.plan.host
    assert sum(ha.n_irregular_class) > 0 and sum(hb.n_irregular_class) == 0 and ha.n_patches > hb.n_patches
    if mesh_args["max_level"] == 4:
        assert sum(ha.n_irregular_class) == ha.n_patches == ha.ntiles and hb.n_patches == 0
    o = O.PlainCase(part, np.float64, state=st)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    for _ in range(3):
        a.iterate(dt)
        b.iterate(dt)
        o.iterate(dt)
    torch.cuda.synchronize()
    assert rel_err(a.state().cpu().numpy(), o.current()[:, :part.N]) < TOL10[torch.float64]
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
    # one stage in pieces that cut through the regular patches, the irregular patches and the generic tiles
    d = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True, irregular=True))
    nt, npatch, nirr = ha.ntiles, ha.n_patches, sum(ha.n_irregular_class)
    cuts = sorted({0, (npatch - nirr) // 2, npatch - nirr + nirr // 3, npatch - 1, npatch + (nt - npatch) // 2, nt})
    s = hip.stream_ptr()
    c.begin_step()
    d.begin_step()
    c.plan.stage(c, 1, c.prev, 1, dt, s)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d.plan.stage(d, 1, d.prev, 1, dt, s, tile_begin=lo, tile_count=hi - lo)
    torch.cuda.synchronize()
    assert not torch.isnan(c.planes).any() and torch.equal(c.planes, d.planes)


def test_patch_kernel_through_the_native_stepper_and_one_patch_per_workgroup():
    """The C++ step driver (whole-plan launches: persistent grids) and explicit partial ranges (one patch per workgroup,
    what a class-split multi-rank stage launches) give the same bits."""
    mesh = SynthMesh(2, 4, 8, band=0.12)
    part = mesh.partition()
    st = perturbed_state(part, 5)
    a, b = _pair(part, torch.float64, hip.KEPES, st)
    a.use_native_stepper()
    dt = 0.1 * 2.0 ** -8
    a.iterate_steps(3, dt)
    for _ in range(3):
        b.iterate(dt)
    torch.cuda.synchronize()
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
    # the same stage in three pieces that cut through the patch tiles and the generic tiles
    c = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True))
    d = PlainSolver(part, torch.float64, mode="fused", state=st, plan_options=dict(patches=True))
    nt, npatch = c.plan.host.ntiles, c.plan.host.n_patches
    cuts = [0, npatch // 3, npatch + (nt - npatch) // 2, nt]
    s = hip.stream_ptr()
    c.begin_step()                                             # (prev = the planes that hold the state, solver.cu:76)
    d.begin_step()
    c.plan.stage(c, 1, c.prev, 1, dt, s)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        d.plan.stage(d, 1, d.prev, 1, dt, s, tile_begin=lo, tile_count=hi - lo)
    torch.cuda.synchronize()
    assert not torch.isnan(c.planes).any() and torch.equal(c.planes, d.planes)


@pytest.mark.parametrize("mesh_args", [dict(dim=2, base_level=4, max_level=8, band=0.12), dict(dim=3, base_level=4, max_level=6, band=0.1)])
def test_patch_plan_on_a_partition_is_bitwise_the_single_rank_run(mesh_args):
    """A 3-way SFC partition with patches on every rank (ghosts across the + sides of patches, patch tiles in several
    tile classes, interior / ghost-reading ranges launched separately) through the loopback transport of
    tests/test_gpu_halo.py: bitwise the single-rank run without patches. 2D and 3D patches."""
    from test_gpu_halo import loopback
    from t8gpu_amd.halo import HaloExchange
    mesh = SynthMesh(**mesh_args)
    whole = mesh.partition()
    st = perturbed_state(whole, 17)
    ref = PlainSolver(whole, torch.float64, mode="fused", state=st, plan_options=dict(patches=False))
    world = 3
    parts = [mesh.partition(r, world) for r in range(world)]
    solvers, halos = [], []
    for part in parts:
        gidx = np.concatenate([part.first_global + np.arange(part.N), part.ghost_global])
        local = st[:, gidx].copy()
        local[:, part.N:] = np.nan
        solvers.append(PlainSolver(part, torch.float64, mode="fused", state=local, plan_options=dict(patches=True)))
        halos.append(HaloExchange(part, torch.float64, dist=None, overlap=False))
    assert all(s.plan.host.n_patches > 0 for s in solvers)
    assert sum(s.plan.host.n_patch_class[1] + s.plan.host.n_patch_class[2] for s in solvers) > 0   # not only deep patches
    dt = 0.1 * 2.0 ** -mesh.finest_level
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
    full = torch.cat([s.state() for s in solvers], dim=1)
    assert torch.equal(full, ref.state())


_STREAM_CHILD = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from t8gpu_amd.solver import PlainSolver, SubgridSolver
from t8gpu_amd.synth import SynthMesh
out = []
for what, mesh, sub in (("2d", SynthMesh(2, 4, 7, band=0.12), False), ("3d", SynthMesh(3, 3, 5, band=0.1), False), ("sg", SynthMesh(3, 2, 3, band=0.12), True)):
    for dt_ in (torch.float64, torch.float32):
        part = mesh.partition(subgrid=True) if sub else mesh.partition()
        s = (SubgridSolver if sub else PlainSolver)(part, dt_, mode="fused")
        dt = 0.1 * 2.0 ** -(mesh.finest_level + (2 if sub else 0))
        for _ in range(3):
            s.iterate(dt)
        torch.cuda.synchronize()
        out.append(f"{what} {dt_} {hashlib.sha256(s.state().cpu().numpy().tobytes()).hexdigest()}")
print("\n".join(out))
"""


@pytest.mark.gpu
def test_non_temporal_instantiations_give_the_same_bits(tmp_path):
    """The patch kernels (2D, 3D) and the Subgrid family kernels exist twice: with ordinary and with non-temporal stores of the stage
    results / loads of the previous state (flux_math.hpp: stream_store; chosen per launch from the size of the stage's planes,
    T8GPU_STREAM_MB). Same arithmetic, so the same bits: small meshes through both (threshold 1 MB = always, 0 = never), in
    child processes (the threshold is read once)."""
    import subprocess
    import sys
    script = tmp_path / "stream_child.py"
    script.write_text(_STREAM_CHILD)
    outs = {}
    for mb in ("0", "1"):
        env = dict(os.environ, T8GPU_STREAM_MB=mb)
        res = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=280, env=env)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        outs[mb] = [" ".join(ln.split()[:3]) for ln in res.stdout.strip().splitlines()]
    assert len(outs["0"]) == 6 and outs["0"] == outs["1"], (outs["0"], outs["1"])
