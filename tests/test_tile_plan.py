"""CPU check of the tiling pre-pass: a numpy interpreter of the plan (gather, one flux per tile face,
CSR sum with signs) must reproduce the oracle's face loop, and the packing invariants must hold."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from _gpu import perturbed_state
from t8gpu_amd.plan import HostPlainPlan
from t8gpu_amd.synth import SynthMesh


def _morton(i, j):
    t = 0
    for b in range(4):
        t |= ((i >> b) & 1) << (2 * b) | ((j >> b) & 1) << (2 * b + 1)
    return t


def _ctz4(v):
    return 4 if v == 0 else (v & -v).bit_length() - 1


def interpret_patch(plan, t, state, net, reporters):
    """A patch tile as kernels_fused_patch.hip evaluates it: nothing but the descriptor and the 64 elements across the
    sides; neighbours, face ids and the summation order follow from the position (i, j) in the 16 x 16 block."""
    pos = int(np.flatnonzero(plan.tile_order == t)[0])
    d = plan.tile_desc[pos]
    e0, ne, h0, nh, fbase, flags = (int(x) for x in d[:6])
    assert (flags & 0x100) and e0 == plan.elem_off[t] and h0 == plan.halo_off[t]
    if flags & 0x400:      # uniform volume in words 1 and 3 (instead of the implied counts 256 and 64)
        assert float(np.array([d[1], d[3]], np.int32).view(np.float64)[0]) > 0
    else:
        assert ne == 256 and nh == 64
    area = float(d[6:8].copy().view(np.float64)[0])
    halo = plan.halo_ids[h0:h0 + 64]
    ex, ey = np.array([[1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0]])

    def flux(n, l, r):
        return area * O.xyz_face_flux(0, n, state[:, [l]].T.copy(), state[:, [r]].T.copy())[0]

    for i in range(16):
        for j in range(16):
            e = e0 + _morton(i, j)
            px = e0 + _morton(i + 1, j) if i < 15 else halo[16 + j]
            mx = e0 + _morton(i - 1, j) if i > 0 else halo[j]
            py = e0 + _morton(i, j + 1) if j < 15 else halo[48 + i]
            my = e0 + _morton(i, j - 1) if j > 0 else halo[32 + i]
            f_mx, f_my, f_px, f_py = flux(ex, mx, e), flux(ey, my, e), flux(ex, e, px), flux(ey, e, py)
            yfirst = bool(flags & 1) if (i, j) == (0, 0) else _ctz4(j) >= _ctz4(i)
            first, second = (f_my, f_mx) if yfirst else (f_mx, f_my)
            net[:, e] = ((first + second) - f_px) - f_py
            reporters += [fbase + 2 * _morton(i, j), fbase + 2 * _morton(i, j) + 1]


def _morton3(i, j, k):
    t = 0
    for b in range(3):
        t |= ((i >> b) & 1) << (3 * b) | ((j >> b) & 1) << (3 * b + 1)
    for b in range(2):
        t |= ((k >> b) & 1) << (3 * b + 2)
    return t


def interpret_patch3(plan, t, state, net, reporters):
    """An 8 x 8 x 4 patch tile as kernels_fused_patch3.hip evaluates it: the descriptor and the 256 cells across the six
    sides; neighbours, face ids and the order of the three - faces follow from (i, j, k) and three flag bits."""
    pos = int(np.flatnonzero(plan.tile_order == t)[0])
    d = plan.tile_desc[pos]
    e0, ne, h0, nh, fbase, flags = (int(x) for x in d[:6])
    assert (flags & 0x300) == 0x300 and e0 == plan.elem_off[t] and h0 == plan.halo_off[t]
    if flags & 0x400:
        assert float(np.array([d[1], d[3]], np.int32).view(np.float64)[0]) > 0
    else:
        assert ne == 256 and nh == 256
    area = float(d[6:8].copy().view(np.float64)[0])
    halo = plan.halo_ids[h0:h0 + 256]
    ax = [np.array([[1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0]]), np.array([[0.0, 0.0, 1.0]])]
    if flags & 0x800:
        # IRREGULAR patch: word 4 is where its per-cell words start in face_lr / face_orig; every side face in the orientation
        # the words give (the lister is the left operand, the normal points away from it), the six added in the listed order
        w = fbase
        assert plan.face_off[t] == w and plan.face_off[t + 1] - w == 512
        info, first, wfirst = plan.face_lr[w:w + 256], plan.face_orig[w:w + 256], plan.face_orig[w + 256:w + 512]
        for i in range(8):
            for j in range(8):
                for k in range(4):
                    c = _morton3(i, j, k)
                    e = e0 + c
                    nb = [e0 + _morton3(i - 1, j, k) if i > 0 else halo[j + 8 * k], e0 + _morton3(i + 1, j, k) if i < 7 else halo[32 + j + 8 * k],
                          e0 + _morton3(i, j - 1, k) if j > 0 else halo[64 + i + 8 * k], e0 + _morton3(i, j + 1, k) if j < 7 else halo[96 + i + 8 * k],
                          e0 + _morton3(i, j, k - 1) if k > 0 else halo[128 + i + 8 * j], e0 + _morton3(i, j, k + 1) if k < 3 else halo[192 + i + 8 * j]]
                    own, wall, order = int(info[c]) & 63, (int(info[c]) >> 6) & 63, int(info[c]) >> 12
                    assert wall & ~own == 0 and sorted((order >> (3 * q)) & 7 for q in range(6)) == list(range(6))
                    acc = np.zeros(5)
                    for q in range(6):
                        sd = (order >> (3 * q)) & 7
                        mine, is_wall, below = bool((own >> sd) & 1), bool((wall >> sd) & 1), (1 << sd) - 1
                        s = 1.0 if mine == bool(sd & 1) else -1.0
                        L, R = (e, nb[sd]) if mine else (nb[sd], e)
                        if is_wall:
                            f = area * O.xyz_face_flux(0, s * ax[sd // 2], state[:, [L]].T.copy(), state[:, [L]].T.copy(), mirror=True)[0]
                        else:
                            f = area * O.xyz_face_flux(0, s * ax[sd // 2], state[:, [L]].T.copy(), state[:, [R]].T.copy())[0]
                        acc = acc - f if mine else acc + f
                        if mine:
                            reporters.append(int(wfirst[c]) + bin(wall & below).count("1") if is_wall
                                             else int(first[c]) + bin(own & ~wall & below).count("1"))
                    net[:, e] = acc
        return

    def flux(a, l, r):
        return area * O.xyz_face_flux(0, ax[a], state[:, [l]].T.copy(), state[:, [r]].T.copy())[0]

    def ctz(v):
        return 8 if v == 0 else (v & -v).bit_length() - 1

    for i in range(8):
        for j in range(8):
            for k in range(4):
                c = _morton3(i, j, k)
                e = e0 + c
                plus = [e0 + _morton3(i + 1, j, k) if i < 7 else halo[32 + j + 8 * k],
                        e0 + _morton3(i, j + 1, k) if j < 7 else halo[96 + i + 8 * k],
                        e0 + _morton3(i, j, k + 1) if k < 3 else halo[192 + i + 8 * j]]
                minus = [e0 + _morton3(i - 1, j, k) if i > 0 else halo[j + 8 * k],
                         e0 + _morton3(i, j - 1, k) if j > 0 else halo[64 + i + 8 * k],
                         e0 + _morton3(i, j, k - 1) if k > 0 else halo[128 + i + 8 * j]]
                fm = [flux(a, minus[a], e) for a in range(3)]
                fp = [flux(a, e, plus[a]) for a in range(3)]
                yx = bool(flags & 1) if (i == 0 and j == 0) else ctz(j) >= ctz(i)
                zx = bool(flags & 2) if (i == 0 and k == 0) else ctz(k) >= ctz(i)
                zy = bool(flags & 4) if (j == 0 and k == 0) else ctz(k) >= ctz(j)
                px, py = int(yx) + int(zx), int(not yx) + int(zy)
                pz = 3 - px - py
                order = sorted(range(3), key=lambda a: (px, py, pz)[a])
                acc = fm[order[0]] + fm[order[1]]
                acc = acc + fm[order[2]]
                net[:, e] = ((acc - fp[0]) - fp[1]) - fp[2]
                reporters += [fbase + 3 * c, fbase + 3 * c + 1, fbase + 3 * c + 2]


def interpret(plan, part, state, reporters=None):
    """Net flux per owned element exactly as the fused kernels accumulate it (fp64, oracle flux)."""
    N = part.N
    net = np.zeros((5, N))
    reporters = [] if reporters is None else reporters
    for t in range(plan.ntiles):
        if plan.tile_patch[t]:
            (interpret_patch3 if plan.patch_dim == 3 else interpret_patch)(plan, t, state, net, reporters)
            continue
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
        orig = plan.face_orig[f0:f1]
        reporters += [int(x) for x in orig[orig >= 0]]
    return net


def order_of(plan):
    return plan.tile_order


@pytest.mark.parametrize("mesh_args,ranks,patches", [(dict(dim=2, base_level=3, max_level=6, band=0.06), 1, False),
                                                     (dict(dim=2, base_level=3, max_level=5, band=0.06, periodic=False), 1, False),
                                                     (dict(dim=3, base_level=2, max_level=3, band=0.2), 1, False),
                                                     (dict(dim=2, base_level=3, max_level=6, band=0.06), 3, False),
                                                     (dict(dim=2, base_level=4, max_level=7, band=0.12), 1, True),
                                                     (dict(dim=2, base_level=6, max_level=6, periodic=False), 1, True),
                                                     (dict(dim=2, base_level=4, max_level=7, band=0.12), 3, True),
                                                     (dict(dim=3, base_level=3, max_level=5, band=0.12), 1, True),
                                                     (dict(dim=3, base_level=5, max_level=5, periodic=False), 2, True),
                                                     # every block next to a periodic wrap; walls, coarser neighbours and a partition cut
                                                     (dict(dim=3, base_level=4, max_level=4), 1, True),
                                                     (dict(dim=3, base_level=3, max_level=5, band=0.12, periodic=False), 2, True)])
def test_plan_reproduces_the_face_loop(mesh_args, ranks, patches):
    mesh = SynthMesh(**mesh_args)
    irregular_expected = True if (mesh_args["dim"] == 3 and patches) else (False if patches else None)
    n_patches = 0
    for rk in range(ranks):
        part = mesh.partition(rk, ranks)
        plan = HostPlainPlan.from_partition(part, tmax=64, fcap=150, patches=patches)
        n_patches += plan.n_patches
        assert plan.n_patches == int(plan.tile_patch.sum()) == sum(plan.n_patch_class) and (plan.n_patches > 0) == patches
        if patches:          # the synthetic meshes' patches are blocks of equal cells: every patch carries its volume
            assert plan.n_patches_uniform_volume == plan.n_patches
            pd = plan.tile_desc[:plan.ntiles][(plan.tile_desc[:plan.ntiles, 5] & 0x100) != 0]
            vols = np.stack([pd[:, 1], pd[:, 3]], axis=1).astype(np.int32).copy().view(np.float64)[:, 0]
            e0s = pd[:, 0]
            assert np.array_equal(vols, part.volumes[e0s])
        st = perturbed_state(part, 11 + rk)
        o = O.PlainCase(part, np.float64, state=st)
        getattr(O.lib(), "oracle_plain_interior_faces_f64")(0, part.F, 3, O.p(o.fn), O.p(part.indices), O.p(o.normals), O.p(o.areas),
                                                            O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride), O.p(o.speed))
        if part.B:
            getattr(O.lib(), "oracle_plain_boundary_faces_f64")(0, part.F, part.B, 3, O.p(o.fn), O.p(o.normals), O.p(o.areas),
                                                                O.p(o.planes[0:5]), O.p(o.planes[20:25]), C.c_size_t(o.stride), O.p(o.speed))
        want = o.planes[20:25, :part.N]
        rep = []
        got = interpret(plan, part, st, rep)
        assert np.abs(got - want).max() < 1e-13 * max(1.0, np.abs(want).max())
        # invariants of the packed format
        assert plan.elem_off[0] == 0 and plan.elem_off[-1] == part.N and (np.diff(plan.elem_off) > 0).all()
        generic = ~plan.tile_patch
        assert not generic.any() or np.diff(plan.elem_off)[generic].max() <= 64
        assert plan.max_faces <= max(150, np.diff(plan.csr_off).max())
        irregular = np.zeros(plan.ntiles, bool)
        irregular[order_of(plan)] = (plan.tile_desc[:plan.ntiles, 5] & 0x800) != 0
        irregular &= plan.tile_patch             # (word 5 of a generic tile is its face count)
        assert (np.diff(plan.elem_off)[plan.tile_patch] == 256).all()
        assert (np.diff(plan.face_off)[plan.tile_patch & ~irregular] == 0).all() and (np.diff(plan.face_off)[irregular] == 512).all()
        assert irregular.sum() == sum(plan.n_irregular_class) and (irregular.sum() > 0) == (irregular_expected is True or irregular.sum() > 0)
        if irregular_expected is not None:
            assert (irregular.sum() > 0) == irregular_expected
        assert np.array_equal(np.sort(rep), np.arange(part.F + part.B))      # every face has exactly one reporter
        order = plan.tile_order
        assert np.array_equal(np.sort(order), np.arange(plan.ntiles))
        reads_ghost = np.array([(plan.halo_ids[plan.halo_off[t]:plan.halo_off[t + 1]] >= part.N).any() for t in range(plan.ntiles)])
        assert not reads_ghost[order[:plan.n_interior]].any() and reads_ghost[order[plan.n_interior:]].all()
        if ranks == 1:
            assert plan.n_interior == plan.ntiles
        # inside every class of tile_order the patch tiles come first, the irregular ones last among them
        for c, (a, b) in enumerate(((0, plan.n_deep), (plan.n_deep, plan.n_interior), (plan.n_interior, plan.ntiles))):
            flags = plan.tile_patch[order[a:b]]
            assert flags[:plan.n_patch_class[c]].all() and not flags[plan.n_patch_class[c]:].any()
            irr = irregular[order[a:b]][:plan.n_patch_class[c]]
            assert not irr[:plan.n_patch_class[c] - plan.n_irregular_class[c]].any() and irr[plan.n_patch_class[c] - plan.n_irregular_class[c]:].all()
    assert (n_patches > 0) == patches


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


def test_tile_cap_heuristics_ask_the_launchers_own_test():
    """fused.PlainPlan keeps 480-face tiles on a 3D AMR mesh only if the persistent tile kernel will take the plan, and asks
    the launcher itself (t8gpu_hip_plain_persistent_accepts; no GPU needed): small launches (at most one tile per resident
    workgroup) go to it except fp64 KEPES, MID-SIZE launches (fewer than 8 tiles per resident workgroup) never do, large ones
    do (the Python copy of the test missed the tile-count gate and the flux kind)."""
    import torch
    from t8gpu_amd import hip
    from t8gpu_amd.fused import PlainPlan
    small = HostPlainPlan.from_partition(SynthMesh(3, 3, 5, band=0.05).partition(), tmax=256, fcap=480, want_face_geo=False)
    mid = HostPlainPlan.from_partition(SynthMesh(3, 5, 7, band=0.05).partition(), tmax=256, fcap=480, want_face_geo=False)
    large = HostPlainPlan.from_partition(SynthMesh(3, 6, 8, band=0.05).partition(), tmax=256, fcap=480, want_face_geo=False)
    assert small.ntiles < 768 < mid.ntiles < 8 * 768 < large.ntiles
    acc = PlainPlan._persistent_accepts
    assert not acc(small, torch.float64, hip.KEPES) and acc(small, torch.float64, hip.HLL) and acc(small, torch.float32, hip.KEPES)
    assert not acc(mid, torch.float64, hip.KEPES) and not acc(mid, torch.float64, hip.HLL) and not acc(mid, torch.float32, hip.KEPES)
    assert acc(large, torch.float64, hip.KEPES) and acc(large, torch.float64, hip.HLL)
    wide = HostPlainPlan.from_partition(SynthMesh(3, 6, 8, band=0.05).partition(), tmax=256, fcap=512, want_face_geo=False)
    assert not acc(wide, torch.float64, hip.KEPES)          # 512-face 3D tiles: the third workgroup per CU does not fit


def test_patch_form_and_cap_rules_of_the_device_plan():
    """fused.PlainPlan.on_host (no GPU): on the c5 benchmark mesh -- patches carry 83 % of it, the generic tiles are the short
    stretches between them -- both float types settle on 384-face tiles; fp64 keeps the irregular patch form, fp32 drops it
    (the plan without it is large enough for the persistent kernel); a uniform periodic box keeps it in both (the wrap
    layers alone would be a launch too small for that kernel); asked for explicitly it always stays."""
    import torch
    from t8gpu_amd.fused import PlainPlan
    c5 = SynthMesh(3, 6, 8, band=0.05).partition()
    p64 = PlainPlan.on_host(c5, torch.float64)
    assert p64.auto_fcap == 384 and p64.irregular is True and p64.auto_irregular is True
    assert sum(p64.host.n_irregular_class) > 0 and p64.host.n_patches * 256 > 0.8 * c5.N and p64.host.max_faces <= 384
    p32 = PlainPlan.on_host(c5, torch.float32)
    assert p32.auto_fcap == 384 and p32.irregular is False and sum(p32.host.n_irregular_class) == 0
    assert PlainPlan.on_host(c5, torch.float32, irregular=True).host.n_irregular_class != (0, 0, 0)
    box = SynthMesh(3, 7, 7).partition()
    for dt in (torch.float64, torch.float32):
        b = PlainPlan.on_host(box, dt)
        assert b.irregular is True and b.host.n_patches == b.host.ntiles and sum(b.host.n_irregular_class) > 0
    # a plan whose patches are nearly all irregular hands "no irregular form" down to the meshes adapted from it
    sheet = SynthMesh(3, 4, 6, band=0.02).partition()
    s = PlainPlan.on_host(sheet, torch.float64)
    n_irr = sum(s.host.n_irregular_class)
    assert s.auto_irregular == (n_irr <= 4 * (s.host.n_patches - n_irr))
