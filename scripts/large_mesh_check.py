#!/usr/bin/env python3
"""Scale check beyond the benchmark mesh: the c4 family one level finer everywhere (2D KH AMR, levels 8-13, ~40 M
elements, fp64, ~9 GB of state). Looks for what only shows at size -- 32-bit index arithmetic, plan sizes, LDS windows --
through the size-independent properties: bitwise reproducibility, conservation, finite state; prints the throughput.
usage: large_mesh_check.py [base_level=8] [max_level=13] | large_mesh_check.py subgrid [6] [7]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t8gpu_amd.solver import PlainSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402


def subgrid(base, lmax):
    """the c3 family (3D Subgrid<4,4,4>, fp32) one level finer: ~105 M subcells"""
    import numpy as np
    from t8gpu_amd.solver import SubgridSolver
    t0 = time.time()
    mesh = SynthMesh(3, base, lmax, band=0.17)
    part = mesh.partition(subgrid=True)
    print(f"mesh: {part.N} blocks = {part.N * 64} subcells, {part.F} block faces ({time.time() - t0:.1f} s)", flush=True)
    a = SubgridSolver(part, torch.float32, mode="fused")
    vol = torch.from_numpy(np.repeat(part.volumes / 64, 64)).cuda()
    m0 = (a.state().double() * vol).sum(1)
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    for _ in range(3):
        a.iterate(dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        a.iterate(dt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{part.N * 64 * 10 / el / 1e6:.0f} M subcell-updates/s ({el / 10 * 1e3:.2f} ms/step)", flush=True)
    s1 = a.state().clone()
    assert bool(torch.isfinite(s1).all())
    m1 = (s1.double() * vol).sum(1)
    print(f"integrals: relative drift {float((m1 - m0).abs().max() / m0.abs().max()):.2e} (fp32)")
    del a
    b = SubgridSolver(part, torch.float32, mode="fused")
    for _ in range(13):
        b.iterate(dt)
    torch.cuda.synchronize()
    assert torch.equal(b.state(), s1), "not bitwise reproducible"
    print("bitwise equal to a second run: ok")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "subgrid":
        return subgrid(int(sys.argv[2]) if len(sys.argv) > 2 else 6, int(sys.argv[3]) if len(sys.argv) > 3 else 7)
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    t0 = time.time()
    mesh = SynthMesh(2, base, lmax, band=0.1472)
    part = mesh.partition()
    print(f"mesh: {part.N} elements, {part.F} faces ({time.time() - t0:.1f} s)", flush=True)
    t0 = time.time()
    a = PlainSolver(part, torch.float64, mode="fused")
    a.use_native_stepper()
    print(f"plan: {a.plan.host.ntiles} tiles, max {a.plan.host.max_elems} elements / {a.plan.host.max_faces} faces / "
          f"{a.plan.host.max_slots} slots per tile ({time.time() - t0:.1f} s)", flush=True)
    vol = torch.from_numpy(part.volumes).cuda()
    m0 = (a.state() * vol).sum(1)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    a.iterate_steps(5, dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.iterate_steps(20, dt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{part.N * 20 / el / 1e6:.0f} M cell-updates/s ({el / 20 * 1e3:.2f} ms/step)", flush=True)
    s1 = a.state().clone()
    assert bool(torch.isfinite(s1).all())
    m1 = (s1 * vol).sum(1)
    drift = float((m1 - m0 * (0.33333333333333 + 0.66666666666666) ** 25).abs().max() / m0.abs().max())
    print(f"integrals: relative deviation from the RK coefficients' law {drift:.2e}")
    assert drift < 1e-13
    del a
    b = PlainSolver(part, torch.float64, mode="fused")     # python-driven stages, same plan parameters
    for _ in range(25):
        b.iterate(dt)
    torch.cuda.synchronize()
    assert torch.equal(b.state(), s1), "not bitwise reproducible"
    print("bitwise equal to a second run through the python-driven stages: ok")


if __name__ == "__main__":
    main()
