"""CPU: the prism / hexahedron mesh provider (t8gpu_amd/unstructured.py) -- topology, closed cells, the
reference's listing rule, partitions whose halo lists agree across ranks, and the oracle on that geometry."""
import numpy as np
import pytest

import _oracle as O
from t8gpu_amd.unstructured import PrismHexMesh, shell_map, wavy_map


def outward_sums(part):
    F, B = part.F, part.B
    A = (part.normals.reshape(-1, 3) * part.areas[:, None])
    acc = np.zeros((part.N + part.G, 3))
    fn = part.face_neighbors
    np.add.at(acc, fn[0:2 * F:2], A[:F])
    np.add.at(acc, fn[1:2 * F:2], -A[:F])
    np.add.at(acc, fn[2 * F:], A[F:])
    return acc[: part.N]


@pytest.mark.parametrize("split,periodic,mapping", [("checker", False, shell_map), ("all", False, shell_map),
                                                    (0.4, True, wavy_map), ("none", True, wavy_map)])
def test_cells_are_closed_and_faces_follow_the_listing_rule(split, periodic, mapping):
    m = PrismHexMesh((8, 4, 4), split=split, mapping=mapping, periodic=periodic)
    p = m.partition()
    nsplit = int(m.split.sum()) * 4
    assert p.N == 8 * 4 * 4 + nsplit and p.G == 0
    # every cell closed: the outward area vectors of its faces sum to zero
    assert np.abs(outward_sums(p)).max() < 1e-15
    # 5 faces per prism, 6 per hexahedron
    cnt = np.bincount(p.face_neighbors, minlength=p.N)
    assert sorted(set(cnt.tolist())) == ([5] if split == "all" else [6] if split == "none" else [5, 6])
    assert (cnt == 5).sum() == 2 * nsplit
    # interior faces: listed once, by the lower index; unit normals
    l, r = p.face_neighbors[0:2 * p.F:2], p.face_neighbors[1:2 * p.F:2]
    assert (l < r).all() and len({(a, b) for a, b in zip(l.tolist(), r.tolist())}) == p.F
    assert np.allclose(np.linalg.norm(p.normals.reshape(-1, 3), axis=1), 1, atol=1e-14)
    assert (p.B == 0) == periodic and (p.volumes > 0).all()
    # the geometry is genuinely oblique: most faces have a normal with three non-zero components
    if mapping is shell_map:
        assert (np.abs(p.normals.reshape(-1, 3)) > 1e-3).all(axis=1).mean() > 0.5
        assert np.isclose(p.volumes.sum(), m.volumes.sum()) and len(np.unique(np.round(p.areas, 12))) > 30


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_partitions_cover_the_mesh_and_agree_on_the_halo(nranks):
    m = PrismHexMesh((8, 8, 4), split=0.5, mapping=shell_map)
    parts = [m.partition(r, nranks) for r in range(nranks)]
    assert sum(p.N for p in parts) == m.num_elements
    for p in parts:
        assert np.abs(outward_sums(p)).max() < 1e-15          # every owned cell sees all of its faces
        for k, q in enumerate(p.peers.tolist()):
            other = parts[q]
            kk = other.peers.tolist().index(p.rank)
            sent = other.first_global + other.send_idx[other.send_off[kk]: other.send_off[kk + 1]]
            assert np.array_equal(sent, p.ghost_global[p.recv_off[k]: p.recv_off[k + 1]])
        assert p.recv_off[-1] == p.G
    # the state a rank starts from is the global one
    ic = m.initial_state()
    for p in parts:
        assert np.array_equal(p.kh_initial_state()[:, : p.N], ic[:, p.first_global: p.first_global + p.N])


def test_oracle_on_curved_prisms_keeps_a_uniform_state_and_conserves():
    m = PrismHexMesh((8, 8, 4), split="checker", mapping=wavy_map, periodic=True)
    p = m.partition()
    uni = np.tile(np.array([[1.3], [0.2], [-0.1], [0.05], [3.0]]), (1, p.N))
    o = O.PlainCase(p, np.float64, state=uni)
    o.iterate(1e-3)
    assert np.abs(o.current()[:, : p.N] - uni).max() < 1e-12
    o = O.PlainCase(p, np.float64)
    before = (o.current()[:, : p.N] * p.volumes[: p.N]).sum(axis=1)
    for _ in range(3):
        o.iterate(2e-3)
    after = (o.current()[:, : p.N] * p.volumes[: p.N]).sum(axis=1)
    assert np.abs(after - before).max() < 1e-13 * np.abs(before).max()
    assert np.isfinite(o.current()).all()


# ---- mixed tetrahedron / hexahedron meshes (BASELINE config 5's mesh class) -------------------------------------------
@pytest.mark.parametrize("tets", ["blocks", "half", "all", "none"])
def test_tet_hex_mesh_cells_are_closed_conforming_and_fill_the_domain(tets):
    from t8gpu_amd.unstructured import TetHexMesh
    m = TetHexMesh((4, 6, 4), tets=tets, mapping=shell_map)
    p = m.partition()
    ncell = 4 * 6 * 4
    ntet_cells = {"all": ncell, "none": 0}.get(tets, ncell // 2)
    assert p.N == 6 * ntet_cells + (ncell - ntet_cells) and m.num_tets == 6 * ntet_cells and p.G == 0
    assert np.abs(outward_sums(p)).max() < 1e-15                              # every element closed
    cnt = np.bincount(p.face_neighbors, minlength=p.N)
    tet = np.zeros(p.N, bool)
    first = 0
    # 4 faces per tetrahedron; a hexahedron has 6 sides, each listed as one quadrilateral or as the two triangles of a
    # neighbouring tetrahedron cell: 6 .. 12 faces
    assert set(cnt.tolist()) <= set(range(4, 13)) and (cnt == 4).sum() == m.num_tets
    l, r = p.face_neighbors[0:2 * p.F:2], p.face_neighbors[1:2 * p.F:2]
    assert (l < r).all() and len({(a, b, tuple(np.round(c, 9))) for a, b, c in zip(l.tolist(), r.tolist(), m.face_centroid[: p.F])}) == p.F
    assert np.allclose(np.linalg.norm(p.normals.reshape(-1, 3), axis=1), 1, atol=1e-14) and (p.volumes > 0).all()
    # same domain whatever the mix: total volume and total wall area
    ref = TetHexMesh((4, 6, 4), tets="none", mapping=shell_map).partition()
    assert np.isclose(p.volumes.sum(), ref.volumes.sum(), rtol=2e-3)         # (curved sides: the cut changes the volume slightly)
    flat = TetHexMesh((4, 4, 4), tets=tets, mapping=lambda x, y, z: np.stack([x, y, z], axis=-1)).partition()
    assert abs(flat.volumes.sum() - 1.0) < 1e-14 and abs(flat.areas[flat.F:].sum() - 6.0) < 1e-13


def test_tet_hex_partitions_agree_on_the_halo_and_the_oracle_conserves():
    from t8gpu_amd.unstructured import TetHexMesh
    m = TetHexMesh((8, 4, 4), tets="blocks", mapping=shell_map)
    parts = [m.partition(r, 3) for r in range(3)]
    assert sum(p.N for p in parts) == m.num_elements
    for p in parts:
        assert np.abs(outward_sums(p)).max() < 1e-15
        for k, q in enumerate(p.peers.tolist()):
            other = parts[q]
            kk = other.peers.tolist().index(p.rank)
            sent = other.first_global + other.send_idx[other.send_off[kk]: other.send_off[kk + 1]]
            assert np.array_equal(sent, p.ghost_global[p.recv_off[k]: p.recv_off[k + 1]])
    p = m.partition()
    uni = np.tile(np.array([[1.3], [0.2], [-0.1], [0.05], [3.0]]), (1, p.N))
    o = O.PlainCase(p, np.float64, state=uni)                                  # walls: only the pressure acts on them
    o.iterate(1e-3)
    assert np.isfinite(o.current()).all()
    o = O.PlainCase(p, np.float64)
    before = (o.current()[[0, 4], : p.N] * p.volumes[: p.N]).sum(axis=1)
    for _ in range(3):
        o.iterate(1e-3)
    after = (o.current()[[0, 4], : p.N] * p.volumes[: p.N]).sum(axis=1)
    assert np.abs(after - before).max() < 1e-13 * np.abs(before).max()       # mass and energy: walls let nothing through
