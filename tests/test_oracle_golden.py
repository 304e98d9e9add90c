"""Pins the CPU oracle: the SURVEY 8c known-answer vectors (the only ones that exist for this path)
plus the algebraic invariants the scheme guarantees (last rows)."""
import json
import os

import numpy as np
import pytest

import _oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_8c_kat.json")))


def test_kat_kepes_f64_every_digit():
    F = O.face_frame_flux(0, np.array([GOLD["uL"]], np.float64), np.array([GOLD["uR"]], np.float64))[0]
    # the survey printed 17 significant digits: bit-exact round trip
    assert [float(x) for x in F] == GOLD["kepes_f64"]


def test_kat_kepes_f32_nine_digits():
    F = O.face_frame_flux(0, np.array([GOLD["uL"]], np.float32), np.array([GOLD["uR"]], np.float32))[0]
    assert [float("%.9g" % x) for x in F] == GOLD["kepes_f32"]


def test_kat_hll_f32_nine_digits():
    F = O.face_frame_flux(1, np.array([GOLD["uL"]], np.float32), np.array([GOLD["uR"]], np.float32))[0]
    assert [float("%.9g" % x) for x in F] == GOLD["hll_f32"]


REFVEC = os.path.join(os.path.dirname(__file__), "golden", "reference_flux_vectors.npz")


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_is_bit_exact_on_the_generated_reference_vectors(tag):
    """tests/golden/reference_flux_vectors.npz = outputs of the reference's own kernels.inl:1-332 compiled for the host
    (tests/golden/make_reference_vectors.py, run in the build container; 10 240 vectors per float type: ln_mean on both
    branches, KEPES and the dead-code HLL in the face frame on generic / near-equal / strong-jump / supersonic pairs, and
    the kernels' xyz pipeline basis -> rotate | reflect -> flux -> rotate back on axis-aligned and oblique normals).
    The oracle must reproduce every one of them BIT FOR BIT in both precisions."""
    g = np.load(REFVEC)
    lm = O.ln_mean(g[f"lm_a_{tag}"], g[f"lm_b_{tag}"])
    assert np.array_equal(lm, g[f"lm_out_{tag}"])
    small = np.abs(g[f"lm_b_{tag}"] / g[f"lm_a_{tag}"] - 1) < 0.02      # both branches of kernels.inl:26-32 are in the set
    assert 200 < small.sum() < small.size - 200
    for kind, name in ((0, "kepes"), (1, "hll")):
        F = O.face_frame_flux(kind, g[f"ff_L_{tag}"], g[f"ff_R_{tag}"])
        assert np.array_equal(F, g[f"ff_{name}_{tag}"]), name
        X = O.xyz_face_flux(kind, g[f"xyz_n_{tag}"], g[f"xyz_L_{tag}"], g[f"xyz_R_{tag}"])
        assert np.array_equal(X, g[f"xyz_{name}_{tag}"]), name
        W = O.xyz_face_flux(kind, g[f"xyz_n_{tag}"], g[f"xyz_L_{tag}"], g[f"xyz_R_{tag}"], mirror=True)
        assert np.array_equal(W, g[f"xyz_{name}_wall_{tag}"]), name
    assert g[f"ff_L_{tag}"].shape[0] + g[f"xyz_L_{tag}"].shape[0] >= 3000 and np.isfinite(g[f"ff_kepes_{tag}"]).all()


def test_reference_vector_set_covers_the_hard_cases():
    g = np.load(REFVEC)
    L, R = g["ff_L_f64"], g["ff_R_f64"]
    assert (L == R).all(axis=1).sum() >= 16                               # identical states (consistency, 0/0 guards)
    assert (np.maximum(L[:, 0] / R[:, 0], R[:, 0] / L[:, 0]) > 100).sum() >= 50   # strong density jumps
    n = g["xyz_n_f64"]
    assert (np.abs(n).max(axis=1) == 1.0).sum() >= 300 and (np.abs(n).max(axis=1) < 0.99).sum() >= 300   # axis + oblique
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-15)
    assert "identical to kernels.inl:21-130 modulo whitespace: True" in str(g["meta"][0])


def test_ln_mean_branches():
    # series branch (u < 1e-4, kernels.cu:29-32) and log branch agree where they meet; a == b -> a
    a = np.array([1.0, 1.0, 1.0, 2.0])
    b = np.array([1.0, 1.0 + 1e-9, 1.0201, 5.0])
    m = O.ln_mean(a, b)
    assert m[0] == 1.0
    exact = (b[1:] - a[1:]) / np.log(b[1:] / a[1:])
    assert np.allclose(m[1:], exact, rtol=1e-9)
    # either side of the branch switch (xi ~ 1.0202 gives u ~ 1e-4)
    xi = np.array([1.02019, 1.02021])
    assert abs(np.diff(O.ln_mean(np.ones(2), xi) - (xi - 1) / np.log(xi))[0]) < 1e-10


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_flux_consistency_and_antisymmetry(kind, dtype, tol):
    n = 2000
    sL, sR = O.random_states(n, 2024, dtype), O.random_states(n, 2025, dtype)
    rng = np.random.default_rng(7)
    nrm = rng.normal(size=(n, 3))
    nrm[: n // 4] = np.eye(3)[rng.integers(0, 3, n // 4)] * rng.choice([-1.0, 1.0], (n // 4, 1))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm.astype(dtype)
    # F(uL, uR; n) = -F(uR, uL; -n)
    F = O.xyz_face_flux(kind, nrm, sL, sR)
    G = O.xyz_face_flux(kind, -nrm, sR, sL)
    scale = np.abs(F).max()
    assert np.abs(F + G).max() <= 50 * tol * scale
    # consistency F(u, u; n) = f(u).n
    C = O.xyz_face_flux(kind, nrm, sL, sL).astype(np.float64)
    s = sL.astype(np.float64)
    v = s[:, 1:4] / s[:, :1]
    pr = 0.4 * (s[:, 4] - 0.5 * (s[:, 1:4] * v).sum(1))
    vn = (v * nrm).sum(1)
    exact = np.stack([s[:, 0] * vn] + [s[:, 1 + d] * vn + pr * nrm[:, d] for d in range(3)] + [(s[:, 4] + pr) * vn], 1)
    assert np.abs(C - exact).max() <= 50 * tol * np.abs(exact).max()


def test_wall_flux_has_no_mass_or_energy():
    s = O.random_states(500, 3, np.float64)
    nrm = np.tile(np.array([[0.0, -1.0, 0.0]]), (500, 1))
    W = O.xyz_face_flux(0, nrm, s, s, mirror=True)
    assert np.abs(W[:, 0]).max() < 1e-13 and np.abs(W[:, 4]).max() < 1e-12
    assert np.abs(W[:, 1]).max() < 1e-13 and np.abs(W[:, 3]).max() < 1e-13   # only normal momentum (pressure)
    at_rest = s.copy()
    at_rest[:, 1:4] = 0
    at_rest[:, 4] = 2.5
    Wr = O.xyz_face_flux(0, nrm, at_rest, at_rest, mirror=True)
    assert np.allclose(Wr[:, 2], -1.0)                                      # fluid at rest: p * n_y, p = 0.4 * 2.5


def test_rk3_truncated_coefficients_quirk_q1():
    # ssp_runge_kutta.inl:12-14,23-25: 0.33333333333333 / 0.66666666666666, not 1/3, 2/3
    import ctypes as C
    n = 4
    prev = np.full((5, n), 3.0)
    mid = np.full((5, n), 6.0)
    out = np.zeros((5, n))
    flux = np.full((5, n), 9.0)
    vol = np.full(n, 2.0)
    O.lib().oracle_plain_rk_stage_f64(3, n, O.p(prev), O.p(mid), O.p(out), O.p(flux), C.c_size_t(n), O.p(vol), C.c_double(0.5))
    assert out[0, 0] == 0.33333333333333 * 3.0 + 0.66666666666666 * 6.0 + 0.66666666666666 * 0.5 / 2.0 * 9.0
    assert out[0, 0] != 3.0 / 3 + 6.0 * 2 / 3 + (2.0 / 3) * 0.5 / 2.0 * 9.0
    assert (flux == 0).all()                      # every stage zeroes the flux planes


# ---- HLLC (an addition: the reference has no HLLC) pinned by its defining properties -------------------
HLLC = 2


def _state(rho, v, p):
    v = np.asarray(v, float)
    return np.array([rho, rho * v[0], rho * v[1], rho * v[2], p / 0.4 + 0.5 * rho * (v ** 2).sum()])


def _physical_flux(u):
    rho, v = u[0], u[1:4] / u[0]
    p = 0.4 * (u[4] - 0.5 * rho * (v ** 2).sum())
    return np.array([u[1], u[1] * v[0] + p, u[1] * v[1], u[1] * v[2], v[0] * (u[4] + p)])


def test_hllc_is_consistent_and_upwinds():
    rng = np.random.default_rng(7)
    for _ in range(200):
        u = _state(rng.uniform(0.5, 2), rng.uniform(-1, 1, 3), rng.uniform(0.5, 3))
        F = O.face_frame_flux(HLLC, u[None], u[None])[0]
        assert np.allclose(F, _physical_flux(u), rtol=1e-13, atol=1e-13)
    # supersonic to the right / left: the flux of the upwind state, exactly
    L, R = _state(1.0, [5.0, 0.3, -0.2], 1.0), _state(0.7, [4.5, 0.1, 0.0], 0.8)
    assert np.allclose(O.face_frame_flux(HLLC, L[None], R[None])[0], _physical_flux(L), rtol=1e-14)
    L, R = _state(1.0, [-5.0, 0.3, -0.2], 1.0), _state(0.7, [-4.5, 0.1, 0.0], 0.8)
    assert np.allclose(O.face_frame_flux(HLLC, L[None], R[None])[0], _physical_flux(R), rtol=1e-14)


def test_hllc_keeps_contact_discontinuities_that_hll_smears():
    # stationary contact: density and tangential velocity jump, u = 0, equal pressure -> only the pressure term
    L, R = _state(2.0, [0.0, 0.7, -0.3], 1.5), _state(0.5, [0.0, -0.2, 0.4], 1.5)
    F = O.face_frame_flux(HLLC, L[None], R[None])[0]
    assert np.allclose(F, [0, 1.5, 0, 0, 0], atol=1e-15)
    assert abs(O.face_frame_flux(1, L[None], R[None])[0][0]) > 0.1          # HLL leaks mass through it
    # contact moving to the right with speed 0.3: everything is carried from the left state
    L, R = _state(2.0, [0.3, 0.7, -0.3], 1.5), _state(0.5, [0.3, -0.2, 0.4], 1.5)
    assert np.allclose(O.face_frame_flux(HLLC, L[None], R[None])[0], _physical_flux(L), rtol=1e-14, atol=1e-15)
    L, R = _state(2.0, [-0.3, 0.7, -0.3], 1.5), _state(0.5, [-0.3, -0.2, 0.4], 1.5)
    assert np.allclose(O.face_frame_flux(HLLC, L[None], R[None])[0], _physical_flux(R), rtol=1e-14, atol=1e-15)


def test_hllc_is_symmetric_under_reflection_and_gives_a_pure_pressure_wall_flux():
    rng = np.random.default_rng(8)
    flip = np.array([1, -1, 1, 1, 1.0])
    for _ in range(100):
        L = _state(rng.uniform(0.5, 2), rng.uniform(-1, 1, 3), rng.uniform(0.5, 3))
        R = _state(rng.uniform(0.5, 2), rng.uniform(-1, 1, 3), rng.uniform(0.5, 3))
        F = O.face_frame_flux(HLLC, L[None], R[None])[0]
        G = O.face_frame_flux(HLLC, (R * flip)[None], (L * flip)[None])[0]   # mirror image of the same problem
        assert np.allclose(G, -F * flip, rtol=1e-12, atol=1e-13)
        W = O.face_frame_flux(HLLC, L[None], (L * flip)[None])[0]            # reflective wall
        assert abs(W[0]) < 1e-14 and abs(W[4]) < 1e-13 and abs(W[2]) < 1e-14 and abs(W[3]) < 1e-14
