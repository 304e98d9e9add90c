#!/usr/bin/env python3
"""Adaptive Kelvin-Helmholtz run on one MI355X: the reference's main loop
(examples/compressible_euler/main.cu:30-38 -- adapt every N steps, iterate, periodic output) on the synthetic
2D periodic mesh, with the fused kernels, the native step driver and the device-side adapt path.

    python examples/kelvin_helmholtz_amr.py --steps 400 --adapt-every 50 --min-level 5 --max-level 9 --vtk out/kh

Several GPUs (one process each; adapt + repartition through amr.adapt_partitioned, halo exchange per RK stage):

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 examples/kelvin_helmholtz_amr.py ...

T8GPU_REHEARSAL=1 runs the same on ONE GPU (all ranks share it, gloo + host-staged halos) to try the flow out.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from t8gpu_amd import amr, hostmem, vtk  # noqa: E402
from t8gpu_amd.halo import HaloExchange  # noqa: E402
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
    ap.add_argument("--vtk", default=None, help="prefix of the .vtu / .pvtu files written at the end (density, energy, momentum)")
    args = ap.parse_args()
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    rehearsal = os.environ.get("T8GPU_REHEARSAL", "0") == "1"
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(0 if rehearsal else int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("gloo" if rehearsal else "nccl")
    # (after set_device: pinned memory creates a context on the current device -- every rank's own GPU, not GPU 0)
    hostmem.keep_heap()      # host arrays of an adapt cycle are reused by the next one (t8gpu_amd/hostmem.py)
    if not rehearsal:
        hostmem.use_pinned_uploads()   # ... and uploaded through one pinned staging buffer
    say = print if rank == 0 else (lambda *a, **k: None)

    def adapt(s):
        if world == 1:
            return amr.adapt(s, args.threshold, args.min_level, args.max_level)[0]
        # refreshes the ghost slots of the current state itself (the indicator reads them), through the run's halo object
        return amr.adapt_partitioned(s, dist, host_staged=rehearsal, halo=halo_of.get(id(s)), threshold=args.threshold,
                                     min_level=args.min_level, max_level=args.max_level)

    def total(x, op="sum"):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return float(t.item())

    mesh = SynthMesh(2, args.min_level, args.min_level)
    solver = PlainSolver(mesh.partition(rank, world), dtype, mode="fused")
    # refine the initial mesh around the shear layers before starting (the reference adapts at step 0)
    halo_of = {}           # id(solver) -> its HaloExchange, when one exists already
    for _ in range(args.max_level - args.min_level):
        solver = adapt(solver)
        # re-evaluate the initial condition on the refined mesh (sharp layers)
        ic = torch.from_numpy(solver.part.kh_initial_state()).to(dtype).cuda()
        solver.planes[5 * solver.next:5 * solver.next + 5] = ic
    halo = HaloExchange(solver.part, dtype, dist, stage_through_host=rehearsal) if world > 1 else None
    halo_of = {id(solver): halo}
    if world == 1:
        solver.use_native_stepper()
    mass0 = [total(solver.compute_integral(k)) for k in range(5)]
    t_iter = t_adapt = 0.0
    cells = 0
    for it in range(args.steps):
        if it % args.adapt_every == 0 and it > 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solver = adapt(solver)
            say(f"adapt at it {it}: criteria sum {solver.last_adapt_criteria_sum:.12e}  elements {int(total(solver.N))}", flush=True)
            if world > 1:
                halo = HaloExchange(solver.part, dtype, dist, stage_through_host=rehearsal)
                halo_of = {id(solver): halo}
            else:
                solver.use_native_stepper()
            torch.cuda.synchronize()
            t_adapt += time.perf_counter() - t0
        if it == 0 or it % args.adapt_every == 0:
            dt = 0.1 * 2.0 ** -solver.part.mesh.finest_level                 # the reference's fixed step (main_2d.cu:27-30)
        elif it % 10 == 0:
            dt = solver.compute_timestep(cfl=0.35, dist=dist)                 # CFL step from the device-side max speed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.iterate(dt, halo=halo)
        torch.cuda.synchronize()
        t_iter += time.perf_counter() - t0
        cells += solver.N
        if it % 100 == 0:
            drift = max(abs(total(solver.compute_integral(k)) - mass0[k]) for k in range(5))
            say(f"it {it:5d}  elements {int(total(solver.N)):8d}  finest level {solver.part.mesh.finest_level}  dt {dt:.3e}  "
                f"conservation drift {drift:.2e}", flush=True)
    assert bool(torch.isfinite(solver.state()).all())
    if args.vtk:
        # CompressibleEulerSolver::save_conserved_variables_to_vtk (examples/compressible_euler/solver.cu:177-186)
        os.makedirs(os.path.dirname(os.path.abspath(args.vtk)), exist_ok=True)
        fields = [vtk.get_host_scalar_variable(solver, solver.next, 0, "density"),
                  vtk.get_host_scalar_variable(solver, solver.next, 4, "energy"),
                  vtk.get_host_vector_variable(solver, solver.next, (1, 2, 3), "momentum")]
        say("wrote", vtk.save_variables_to_vtk(solver, fields, args.vtk, dist=dist))
    say(f"iterate: {total(cells) / total(t_iter, 'max') / 1e6:.1f} M cell-updates/s (host-synchronised per step), "
        f"adapt: {t_adapt:.2f} s total, iterate: {t_iter:.2f} s total")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
