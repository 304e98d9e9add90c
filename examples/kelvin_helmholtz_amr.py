#!/usr/bin/env python3
"""Adaptive Kelvin-Helmholtz run on one MI355X: the reference's main loop
(examples/compressible_euler/main.cu:30-38 -- adapt every N steps, iterate, periodic output) on the synthetic
2D periodic mesh, with the fused kernels, the native step driver and the device-side adapt path.

    python examples/kelvin_helmholtz_amr.py --steps 400 --adapt-every 50 --min-level 5 --max-level 9
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from t8gpu_amd import amr  # noqa: E402
from t8gpu_amd.solver import PlainSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--adapt-every", type=int, default=50)
    ap.add_argument("--min-level", type=int, default=5)
    ap.add_argument("--max-level", type=int, default=9)
    ap.add_argument("--threshold", type=float, default=10.0)     # mesh_manager.inl:141
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"])
    args = ap.parse_args()
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    mesh = SynthMesh(2, args.min_level, args.min_level)
    solver = PlainSolver(mesh.partition(), dtype, mode="fused")
    solver.use_native_stepper()
    # refine the initial mesh around the shear layers before starting (the reference adapts at step 0)
    for _ in range(args.max_level - args.min_level):
        solver, _, _ = amr.adapt(solver, args.threshold, args.min_level, args.max_level)
        # re-evaluate the initial condition on the refined mesh (sharp layers)
        ic = torch.from_numpy(solver.part.kh_initial_state()).to(dtype).cuda()
        solver.planes[5 * solver.next:5 * solver.next + 5, :solver.N] = ic[:, :solver.N]
    mass0 = [solver.compute_integral(k) for k in range(5)]
    t_iter = t_adapt = 0.0
    cells = 0
    for it in range(args.steps):
        if it % args.adapt_every == 0 and it > 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solver, _, _ = amr.adapt(solver, args.threshold, args.min_level, args.max_level)
            torch.cuda.synchronize()
            t_adapt += time.perf_counter() - t0
        if it == 0 or it % args.adapt_every == 0:
            dt = 0.1 * 2.0 ** -solver.part.mesh.finest_level                 # the reference's fixed step (main_2d.cu:27-30)
        elif it % 10 == 0:
            dt = solver.compute_timestep(cfl=0.35)                            # CFL step from the device-side max speed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.iterate(dt)
        torch.cuda.synchronize()
        t_iter += time.perf_counter() - t0
        cells += solver.N
        if it % 100 == 0:
            drift = max(abs(solver.compute_integral(k) - mass0[k]) for k in range(5))
            print(f"it {it:5d}  elements {solver.N:8d}  finest level {solver.part.mesh.finest_level}  dt {dt:.3e}  "
                  f"conservation drift {drift:.2e}", flush=True)
    assert bool(torch.isfinite(solver.state()).all())
    print(f"iterate: {cells / t_iter / 1e6:.1f} M cell-updates/s (host-synchronised per step), "
          f"adapt: {t_adapt:.2f} s total, iterate: {t_iter:.2f} s total")


if __name__ == "__main__":
    main()
