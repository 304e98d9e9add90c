"""CPU check of the tiling pre-pass: a numpy interpreter of the plan (gather, one flux per tile face,
CSR sum with signs) must reproduce the oracle's face loop, and the packing invariants must hold."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from _gpu import perturbed_state
from t8gpu_amd.plan import HostPlainPlan
from t8gpu_amd.synth import SynthMesh


def interpret(plan, part, state, speed_out=None):
    """Net flux per owned element exactly as the fused kernel accumulates it (fp64, oracle flux)."""
    N = part.N
    net = np.zeros((5, N))
    for t in range(plan.ntiles):
        e0, e1 = plan.elem_off[t], plan.elem_off[t + 1]
        halo = plan.halo_ids[plan.halo_off[t]:plan.halo_off[t + 1]]
        slots = np.concatenate([np.arange(e0, e1), halo])
        f0, f1 = plan.face_off[t], plan.face_off[t + 1]
        lr = plan.face_lr[f0:f1]
        l, r = (lr & 0xFFFF).astype(np.int64), (lr >> 16).astype(np.int64)
        wall = r == 0xFFFF
        r = np.where(wall, l, r)
        geo = plan.face_geo[f0:f1]
        sL, sR = state[:, slots[l]].T.copy(), state[:, slots[r]].T.copy()
        ff = np.zeros((f1 - f0, 5))
        if (~wall).any():
            ff[~wall] = O.xyz_face_flux(0, geo[~wall, :3], sL[~wall], sR[~wall])
        if wall.any():
            ff[wall] = O.xyz_face_flux(0, geo[wall, :3], sL[wall], sL[wall], mirror=True)
        ff *= geo[:, 3:4]
        for e in range(e0, e1):
            for ent in plan.csr_ent[plan.csr_off[e]:plan.csr_off[e + 1]]:
                f = int(ent) & 0x7FFF
                net[:, e] += ff[f] if (int(ent) & 0x8000) else -ff[f]
    return net


@pytest.mark.parametrize("mesh_args,ranks", [(dict(dim=2, base_level=3, max_level=6, band=0.06), 1),
                                             (dict(dim=2, base_level=3, max_level=5, band=0.06, periodic=False), 1),
                                             (dict(dim=3, base_level=2, max_level=3, band=0.2), 1),
                                             (dict(dim=2, base_level=3, max_level=6, band=0.06), 3)])
def test_plan_reproduces_the_face_loop(mesh_args, ranks):
    mesh = SynthMesh(**mesh_args)
    for rk in range(ranks):
        part = mesh.partition(rk, ranks)
        plan = HostPlainPlan.from_partition(part, tmax=64, fcap=150)
        st = perturbed_state(part, 11 + rk)
        o = O.PlainCase(part, np.float64, state=st)
        getattr(O.lib(), "oracle_plain_interior_faces_f64")(0, part.F, 3, O.p(o.fn), O.p(part.indices), O.p(o.normals), O.p(o.areas),
                                                            O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride), O.p(o.speed))
        if part.B:
            getattr(O.lib(), "oracle_plain_boundary_faces_f64")(0, part.F, part.B, 3, O.p(o.fn), O.p(o.normals), O.p(o.areas),
                                                                O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride), O.p(o.speed))
        want = o.planes[20:25, :part.N]
        got = interpret(plan, part, st)
        assert np.abs(got - want).max() < 1e-13 * max(1.0, np.abs(want).max())
        # invariants of the packed format
        assert plan.elem_off[0] == 0 and plan.elem_off[-1] == part.N and (np.diff(plan.elem_off) > 0).all()
        assert np.diff(plan.elem_off).max() <= 64 and plan.max_faces <= max(150, np.diff(plan.csr_off).max())
        rep = plan.face_orig[plan.face_orig >= 0]
        assert np.array_equal(np.sort(rep), np.arange(part.F + part.B))      # every face has exactly one reporter
        order = plan.tile_order
        assert np.array_equal(np.sort(order), np.arange(plan.ntiles))
        reads_ghost = np.array([(plan.halo_ids[plan.halo_off[t]:plan.halo_off[t + 1]] >= part.N).any() for t in range(plan.ntiles)])
        assert not reads_ghost[order[:plan.n_interior]].any() and reads_ghost[order[plan.n_interior:]].all()
        if ranks == 1:
            assert plan.n_interior == plan.ntiles


def test_plan_rejects_oversized_tiles():
    part = SynthMesh(2, 4, 4).partition()
    with pytest.raises(ValueError):
        HostPlainPlan.from_partition(part, tmax=2000, fcap=10 ** 6)


def test_tile_cap_heuristics_of_the_device_plan():
    """t8gpu_amd/fused.py picks 768-face tiles for fp64 on 3D meshes the persistent kernel cannot take: elements with more
    than 8 faces (16-entry ELL rows) or more distinct {normal, area} rows than its LDS dictionary holds."""
    from t8gpu_amd.fused import PlainPlan
    from t8gpu_amd.synth import SynthMesh
    from t8gpu_amd.unstructured import PrismHexMesh
    amr2 = SynthMesh(2, base_level=4, max_level=7, band=0.05).partition()
    amr3 = SynthMesh(3, base_level=3, max_level=5, band=0.05).partition()
    uni3 = SynthMesh(3, base_level=3, max_level=3).partition()
    curved = PrismHexMesh((8, 8, 10)).partition()
    assert not PlainPlan._wide_rows(amr2) and not PlainPlan._many_geometries(amr2)
    assert PlainPlan._wide_rows(amr3) and not PlainPlan._many_geometries(amr3)
    assert not PlainPlan._wide_rows(uni3) and not PlainPlan._many_geometries(uni3)
    assert not PlainPlan._wide_rows(curved) and PlainPlan._many_geometries(curved)
