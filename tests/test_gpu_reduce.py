"""-m gpu: device-side max-speed and conservation-integral reductions against numpy."""
import ctypes as C

import numpy as np
import pytest
import torch

from t8gpu_amd import hip
from t8gpu_amd.solver import PlainSolver, SubgridSolver
from t8gpu_amd.synth import SynthMesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [0, 1, 255, 100003, 1 << 21])
def test_reductions_match_numpy(dtype, n):
    rng = np.random.default_rng(n + 1)
    npdt = np.float64 if dtype == torch.float64 else np.float32
    x = rng.uniform(0, 5, max(n, 1)).astype(npdt)
    v = rng.uniform(0.1, 1, max(n, 1)).astype(npdt)
    dx, dv = torch.from_numpy(x).cuda(), torch.from_numpy(v).cuda()
    f = hip.lib().t8gpu_hip_reduce_workspace_bytes
    f.restype = C.c_size_t
    ws = torch.zeros(f() // 8, dtype=torch.float64, device="cuda")
    res = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
    hip.call("t8gpu_hip_max_speed", dtype, C.c_size_t(n), hip.ptr(dx), hip.ptr(ws), hip.ptr(res), hip.stream_ptr())
    assert float(res.item()) == (float(x[:n].max()) if n else 0.0)
    hip.call("t8gpu_hip_integral", dtype, C.c_size_t(n), 1, hip.ptr(dx), hip.ptr(dv), hip.ptr(ws), hip.ptr(res), hip.stream_ptr())
    want = float((x[:n].astype(np.float64) * v[:n].astype(np.float64)).sum())
    assert abs(float(res.item()) - want) <= 1e-12 * max(1.0, abs(want))
    first = float(res.item())
    hip.call("t8gpu_hip_integral", dtype, C.c_size_t(n), 1, hip.ptr(dx), hip.ptr(dv), hip.ptr(ws), hip.ptr(res), hip.stream_ptr())
    assert float(res.item()) == first                                   # fixed tree: bitwise reproducible


def test_solver_diagnostics_conservation_and_cfl():
    mesh = SynthMesh(2, 4, 7, band=0.06)
    part = mesh.partition()
    g = PlainSolver(part, torch.float64, mode="fused")
    m0 = [g.compute_integral(k) for k in range(5)]
    assert abs(m0[0] - (part.kh_initial_state()[0] * part.volumes).sum()) < 1e-13
    dt = 0.1 * 2.0 ** -7
    for _ in range(5):
        g.iterate(dt)
    m1 = [g.compute_integral(k) for k in range(5)]
    assert max(abs(a - b) for a, b in zip(m0, m1)) < 1e-12 * max(abs(x) for x in m0)   # periodic mesh: conservative to rounding
    smax = g.max_speed()
    assert smax == float(g.speed[:part.F].max().item()) and 1.0 < smax < 3.0
    assert np.isclose(g.compute_timestep(), 0.7 * 0.5 ** 7 / smax)
