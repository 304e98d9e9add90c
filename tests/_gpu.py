"""Helpers shared by the -m gpu parity tests: everything on the device goes through the C-ABI."""
import numpy as np
import torch

# Parity tolerances: max error normalised by max|u| per variable.
TOL1 = {torch.float64: 1e-12, torch.float32: 2e-5}     # after 1 step / one kernel
TOL10 = {torch.float64: 1e-10, torch.float32: 2e-4}    # after 10 steps
NP = {torch.float64: np.float64, torch.float32: np.float32}


def rel_err(got, want):
    """max_k max_i |got - want| / max_i |want| over the 5 variables (rows)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300)
    return float((np.abs(got - want) / scale).max())


def perturbed_state(part, seed, cells=1):
    """KH state plus a smooth random perturbation so that no flux component is identically zero."""
    u = part.kh_initial_state().copy()
    rng = np.random.default_rng(seed)
    n = u.shape[1]
    rho = u[0] * (1 + 0.05 * rng.standard_normal(n))
    v = np.stack([u[1] / u[0], u[2] / u[0], u[3] / u[0]]) + 0.2 * rng.standard_normal((3, n))
    p = 1.0 + 0.1 * rng.uniform(-1, 1, n)
    E = p / 0.4 + 0.5 * rho * (v ** 2).sum(0)
    return np.stack([rho, rho * v[0], rho * v[1], rho * v[2], E])
