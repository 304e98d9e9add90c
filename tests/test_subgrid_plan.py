"""Host plan of the fused Subgrid kernels (csrc/host/subgrid_plan.cpp): every coarse face must reach each owned
block it touches exactly once -- in the block record's + slot, its - slot, or the block's generic list (faces towards
finer blocks only) -- on the right side of the block and with the right orientation flag."""
import numpy as np
import pytest

from t8gpu_amd.plan import HostSubgridPlan
from t8gpu_amd.synth import SynthMesh


@pytest.mark.parametrize("dim,args,parts", [(2, dict(base_level=3, max_level=3), 1), (2, dict(base_level=3, max_level=6, band=0.03), 1),
                                            (2, dict(base_level=3, max_level=5, band=0.03, periodic=False), 3),
                                            (3, dict(base_level=2, max_level=2), 1), (3, dict(base_level=3, max_level=4, band=0.03), 2),
                                            (3, dict(base_level=2, max_level=4, band=0.05, periodic=False), 1)])
def test_every_face_reaches_each_of_its_blocks_once(dim, args, parts):
    mesh = SynthMesh(dim, **args)
    for rank in range(parts):
        part = mesh.partition(rank, parts, subgrid=True) if parts > 1 else mesh.partition(subgrid=True)
        plan = HostSubgridPlan(part)
        block_rec, bf_rec = plan.records(part.areas, 8)
        N, F, B = part.N, part.F, part.B
        fn = np.asarray(part.face_neighbors, np.int64).reshape(-1)
        left = np.concatenate([fn[0:2 * F:2], fn[2 * F:2 * F + B]])
        right = np.concatenate([fn[1:2 * F:2], np.full(B, -1)])
        normals = np.asarray(part.normals, np.float64).reshape(F + B, dim)
        axis = np.abs(normals).argmax(1)
        positive = normals[np.arange(F + B), axis] > 0
        hanging = np.concatenate([np.asarray(part.level_diff).reshape(-1)[:F] != 0, np.zeros(B, bool)])
        areas = np.asarray(part.areas, np.float64)
        # expected (block, side axis, side sign, other block, is right side, area) per (face, adjacent owned block)
        want = {}
        for f in range(F + B):
            l, r = int(left[f]), int(right[f])
            if l < N:
                want.setdefault((l, int(axis[f]), bool(positive[f])), []).append((r, False, areas[f], bool(hanging[f])))
            if 0 <= r < N and r != l:
                want.setdefault((r, int(axis[f]), not bool(positive[f])), []).append((l, True, areas[f], bool(hanging[f])))
        S = 4 ** dim
        offs = np.asarray(part.nb_offset, np.int64).reshape(F, dim)

        def check_cells(w, e_is_right, f_axis, f_positive, f_hanging, anchor):
            """The two cell recipes of a row against kernels.inl:710-758 spelled out for every sub-face."""
            code, base = int(w[1]), int(w[0])
            la, lb = ((code >> 21) & 1) * 2, 4 - ((code >> 22) & 1) * 2
            hf, ho, c0 = (code >> 19) & 1, (code >> 20) & 1, (code >> 13) & 63
            ta, tb = (1 if f_axis == 0 else 0), (1 if f_axis == 2 else 2)
            for sj in range(4 if dim == 3 else 1):
                for si in range(4):
                    left = (3 if f_positive else 0) * 4 ** f_axis + si * 4 ** ta + sj * 4 ** tb
                    hs = (lambda x: x // 2) if f_hanging else (lambda x: x)
                    rgt = anchor[f_axis] * 4 ** f_axis + (anchor[ta] + hs(si)) * 4 ** ta + ((anchor[tb] if tb < dim else 0) + hs(sj)) * 4 ** tb
                    own, far = (rgt, left) if e_is_right else (left, rgt)
                    assert c0 + ((si >> ho) << la) + ((sj >> ho) << lb) == own
                    if base >= 0:
                        assert (base % S) + ((si >> hf) << la) + ((sj >> hf) << lb) == far

        # (face, side) -> anchor, to look the offsets up again from what a row says
        anchor_of = {}
        for f in range(F):
            anchor_of[(int(left[f]), int(right[f]), int(axis[f]), bool(positive[f]))] = [int(x) for x in offs[f]] + [0] * (3 - dim)
        got = {}
        assert sorted(block_rec[:N, 0].tolist()) == list(range(N))          # a permutation of the owned blocks
        first = 0
        for pos in range(N):
            rec = block_rec[pos]
            e, nbf = int(rec[0]), int(rec[1])
            assert int(rec[2]) == first
            for side, base in ((True, 4), (False, 16)):
                for d in range(3):
                    w = rec[base + 4 * d:base + 4 * d + 4]
                    if d >= dim or w[0] == -2:
                        assert w[0] == -2
                        continue
                    code = int(w[1])
                    assert (code & 3) == d
                    is_right = bool((code >> 12) & 1)
                    # the side the face lies on, seen from this block: the normal points away from the LEFT block
                    assert (bool((code >> 2) & 1) != is_right) == side
                    area = np.frombuffer(w[2:4].tobytes(), np.float64)[0]
                    other = int(w[0]) // S if w[0] >= 0 else int(w[0])
                    got.setdefault((e, d, side), []).append((other, is_right, area, bool((code >> 3) & 1)))
                    assert not (is_right and (code >> 3) & 1)               # the coarse side of a hanging face is never folded
                    lr = (other, e) if is_right else (e, other)
                    check_cells(w, is_right, d, bool((code >> 2) & 1), bool((code >> 3) & 1),
                                anchor_of.get((lr[0], lr[1], d, bool((code >> 2) & 1)), [0, 0, 0]))
            for j in range(first, first + nbf):
                w = bf_rec[j]
                code = int(w[1])
                is_right = bool((code >> 12) & 1)
                side = bool((code >> 2) & 1) != is_right
                area = np.frombuffer(w[2:4].tobytes(), np.float64)[0]
                other = int(w[0]) // S
                got.setdefault((e, code & 3, side), []).append((other, is_right, area, bool((code >> 3) & 1)))
                assert is_right and (code >> 3) & 1                          # generic rows: towards finer blocks only
                check_cells(w, True, code & 3, bool((code >> 2) & 1), True, anchor_of[(other, e, code & 3, bool((code >> 2) & 1))])
            first += nbf
        assert first == plan.n_entries
        assert set(got) == set(want)
        for key in want:
            assert sorted(got[key]) == sorted(want[key]), key


@pytest.mark.parametrize("args,parts", [(dict(base_level=2, max_level=2), 1), (dict(base_level=3, max_level=4, band=0.03), 1),
                                        (dict(base_level=2, max_level=4, band=0.05, periodic=False), 1),
                                        (dict(base_level=3, max_level=4, band=0.03), 2)])
def test_families_are_cubes_of_same_level_blocks_and_cover_the_plan_with_the_rest(args, parts):
    """2x2x2 families (subgrid_plan.cpp): eight consecutive owned blocks whose mutual faces follow the Morton pattern at
    equal level; every block is in exactly one family or in the rest list; the family record's 36 rows are the faces the
    block records hold for the same (block, side)."""
    mesh = SynthMesh(3, **args)
    for rank in range(parts):
        part = mesh.partition(rank, parts, subgrid=True) if parts > 1 else mesh.partition(subgrid=True)
        plan = HostSubgridPlan(part)
        block_rec, _ = plan.records(part.areas, 8)
        fam_rec, rest_rec = plan.family_records(part.areas, 8)
        N = part.N
        assert 8 * plan.n_families + plan.n_rest == N
        by_block = {int(r[0]): r for r in block_rec[:N]}
        seen = np.zeros(N, int)
        expand = lambda j, d: (j << 1) if d == 0 else ((j & 1) | ((j >> 1) << 2) if d == 1 else j)
        for q in range(plan.n_families):
            rec = fam_rec[q]
            e0 = int(rec[0])
            seen[e0:e0 + 8] += 1
            levels = {int(part.levels[e0 + w]) for w in range(8)}
            assert len(levels) == 1
            for w in range(8):
                assert by_block[e0 + w][1] == 0                      # no generic faces inside a family
            for d in range(3):
                for j in range(4):
                    lo, hi = expand(j, d), expand(j, d) | (1 << d)
                    assert (rec[4 + 4 * (d * 4 + j):][:4] == by_block[e0 + hi][4 + 4 * d:][:4]).all()           # outward +
                    assert (rec[4 + 4 * (12 + d * 4 + j):][:4] == by_block[e0 + lo][16 + 4 * d:][:4]).all()    # outward -
                    inner = rec[4 + 4 * (24 + d * 4 + j):][:4]
                    assert (inner == by_block[e0 + lo][4 + 4 * d:][:4]).all()                                  # inner
                    assert int(inner[0]) // 64 == e0 + hi and not (int(inner[1]) >> 3) & 1                     # the sibling, same level
        for r in range(plan.n_rest):
            e = int(rest_rec[r][0])
            seen[e] += 1
            assert (rest_rec[r] == by_block[e]).all()
        assert (seen == 1).all()
        if parts == 1 and "band" not in args:
            assert plan.n_rest == 0                                   # a uniform mesh is all families
