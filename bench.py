#!/usr/bin/env python3
"""bench.py -- M cell-updates/s of the flux + SSP-RK3 step on the synthetic Kelvin-Helmholtz AMR mesh.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU. Either launch it under torch.distributed.run yourself (RANK / WORLD_SIZE in the
environment), or just run the line above: bench.py then starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a CHILD process before anything has touched the GPU, forwards its output
and returns its exit code (the reference's analogue is `mpirun -n 8`, README.md:47-58).

One "step" = one full iterate() (3 flux evaluations + 3 RK stages) over the whole mesh; one
cell-update = one element advanced by one step. Inputs are resident in HBM before
the timed region. Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel, HIP-event timed
inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6300.0  # same guide: ~6.3 TB/s achievable (copy); SURVEY 8d asks for both fractions

# Workloads. c4 is the north-star mesh (~10 M elements, strong-scaled over the ranks).
WORKLOADS = {
    "c1": dict(kind="plain", dim=2, base=8, lmax=8, band=0.0, dtype="f64", desc="2D KH uniform 256^2 quads"),
    "c2": dict(kind="plain", dim=2, base=6, lmax=11, band=0.0596, dtype="f64", desc="2D KH AMR levels 6-11 (~1.03 M elements)"),
    "c3": dict(kind="subgrid", dim=3, base=5, lmax=6, band=0.17, dtype="f32", desc="3D Subgrid<4,4,4> AMR levels 5-6"),
    "c3q": dict(kind="subgrid", dim=2, base=9, lmax=10, band=0.1, dtype="f32", desc="2D Subgrid<4,4> AMR levels 9-10 (examples/subgrid/main_2d.cu)"),
    "c4": dict(kind="plain", dim=2, base=7, lmax=12, band=0.1472, dtype="f64", desc="2D KH AMR levels 7-12 (~9.93 M elements)"),
    # BASELINE config 5 needs a real t8code mixed-element cmesh; these are its geometry-synthetic stand-ins (no
    # repartition inside the timed region): c5 = 3D hex AMR (phi = 3, 6-24 faces per element, Cartesian normals),
    # c5p = conforming prisms + hexahedra on a curved shell sector (5 / 6 faces, a different oblique normal on every
    # face: the mesh class of the reference's own example, examples/compressible_euler/main.cu:20-24).
    "c5": dict(kind="plain", dim=3, base=6, lmax=8, band=0.05, dtype="f64", desc="3D hex AMR levels 6-8 (~3.93 M elements), geometry-synthetic"),
    # (diagnostic: the uniform 3D case -- 6 faces per element, one geometry per direction)
    "c5u": dict(kind="plain", dim=3, base=7, lmax=7, band=0.0, dtype="f64", desc="3D hex uniform 128^3 (~2.10 M elements), geometry-synthetic"),
    "c5p": dict(kind="plain", dim=3, prism=(128, 128, 160), dtype="f64",
                desc="3D prisms + hexahedra on a curved shell, 128x128x160 cells half split (~3.93 M elements), geometry-synthetic"),
    # c5t = mixed tetrahedra / hexahedra (4 faces / 6-12 faces where the two kinds meet), curved shell, walls
    "c5t": dict(kind="plain", dim=3, tets=(96, 96, 128), dtype="f64",
                desc="3D tetrahedra + hexahedra on a curved shell, 96x96x128 cells in 2x2x2 blocks of either kind (~4.13 M elements), geometry-synthetic"),
    # c5a = BASELINE config 5's LOOP: adapt (+ repartition over the ranks) every 20 steps INSIDE the measured region,
    # on a 3D hexahedral forest refined by the reference's gradient indicator (handled by bench_adaptive below)
    "c5a": dict(kind="plain", dim=3, adaptive=dict(every=20, min_level=5, max_level=9, threshold=10.0), dtype="f64",
                desc="3D hex AMR levels 5-9 by the reference's indicator, adapt + repartition every 20 steps inside the timed loop"),
}


def algorithmic_bytes(kind, ft, phi, d=3, phi_c=3.0):
    """SURVEY 8d / BASELINE.md: compulsory bytes per cell-update of the reference's dataflow.
    Returns (per cell-update, [per-stage bytes of the flux part, RK part per stage x3])."""
    if kind == "plain":
        flux_stage = (10 * ft + 8) + phi * (8 + (d + 2) * ft)
        rk = [21 * ft, 26 * ft, 26 * ft]
        return 3 * flux_stage + sum(rk), flux_stage, rk
    flux_stage = 10 * ft + (ft + 8 + phi_c * (24 + 4 * ft)) / 64.0
    rk = [(20 + 1 / 64.0) * ft, (25 + 1 / 64.0) * ft, (25 + 1 / 64.0) * ft]
    return 3 * flux_stage + sum(rk), flux_stage, rk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region; the median is reported")
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default=os.environ.get("T8GPU_BENCH_MODE", "auto"), choices=["auto", "compat", "fused"])
    ap.add_argument("--flux", default="kepes", choices=["kepes", "hll", "hllc"])
    ap.add_argument("--dtype", default=None, choices=[None, "f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--prewarm-seconds", type=float, default=1.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # child torchrun; nothing has touched the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU implementation")
    # T8GPU_REHEARSAL=1: all ranks share GPU 0 and talk over gloo (host-staged halos). Lets the N > 1
    # orchestration be rehearsed on a one-GPU box; the numbers it prints are NOT measurements.
    rehearsal = os.environ.get("T8GPU_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from t8gpu_amd import hip
    from t8gpu_amd.solver import PlainSolver, SubgridSolver
    from t8gpu_amd.synth import SynthMesh

    hip.lib()
    dist = None
    # T8GPU_BENCH_FORCE_DIST=1 sends a ONE-rank run through the N > 1 code path (process group on the real RCCL
    # backend, collective decisions, native communicator from a broadcast id, the multi-rank stepper bring-up):
    # the only way to execute that path on a one-GPU box; tests/test_gpu_bench_rehearsal.py uses it.
    distributed = world > 1 or os.environ.get("T8GPU_BENCH_FORCE_DIST", "0") == "1"
    if distributed:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    w = WORKLOADS[args.workload]
    dts = args.dtype or w["dtype"]
    tdtype = torch.float64 if dts == "f64" else torch.float32
    ft = 8 if dts == "f64" else 4
    kindf = {"kepes": hip.KEPES, "hll": hip.HLL, "hllc": hip.HLLC}[args.flux]
    mode = args.mode
    if mode == "auto":
        try:
            from t8gpu_amd import fused  # noqa: F401
            mode = "fused" if w["kind"] == "plain" or hasattr(fused, "SubgridPlan") else "compat"
        except ImportError:
            mode = "compat"

    if "adaptive" in w:
        return bench_adaptive(args, w, dts, tdtype, kindf, mode, rank, world, dist, rehearsal)
    t0 = time.time()
    if "tets" in w:
        from t8gpu_amd.unstructured import TetHexMesh
        mesh = TetHexMesh(w["tets"], tets="blocks")
        part = mesh.partition(rank, world)
    elif "prism" in w:
        from t8gpu_amd.unstructured import PrismHexMesh
        mesh = PrismHexMesh(w["prism"], split=0.5)
        part = mesh.partition(rank, world)
    else:
        mesh = SynthMesh(w["dim"], w["base"], w["lmax"], band=w["band"])
        part = mesh.partition(rank, world, subgrid=(w["kind"] == "subgrid"))
    n_global = mesh.num_elements
    cells = part.cells_per_element
    if w["kind"] == "plain":
        solver = PlainSolver(part, tdtype, flux_kind=kindf, mode=mode)
        delta_t = (0.05 if "tets" in w else 0.1) * float(mesh.volumes.min()) ** (1 / 3) if ("prism" in w or "tets" in w) else 0.1 * 2.0 ** -mesh.finest_level
    else:
        solver = SubgridSolver(part, tdtype, flux_kind=kindf, mode=mode)
        delta_t = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    halo = None
    halo_kind = "none"
    stepper = None
    if distributed:
        from t8gpu_amd import halo as halo_mod
        halo = halo_mod.HaloExchange(part, tdtype, dist, stage_through_host=rehearsal)   # torch.distributed transport
        halo_kind = "torch.distributed (gloo, host-staged REHEARSAL)" if rehearsal else "torch.distributed"
    if mode == "fused" and os.environ.get("T8GPU_STEPPER", "native") == "native":   # plain tiles and Subgrid blocks alike
        native_halo = None
        if distributed and os.environ.get("T8GPU_HALO", "native") == "native":
            native_halo = make_native_halo(part, tdtype, solver, halo, dist, rank, world)
        if not distributed:
            try:
                stepper = solver.use_native_stepper(None)
            except Exception as exc:  # noqa: BLE001  (keep the run alive on the python-driven path)
                print(f"[bench rank {rank}] native stepper unavailable ({exc}); python-driven stages", file=sys.stderr, flush=True)
                solver.stepper, stepper = None, None
        elif native_halo is not None:
            stepper = bring_up_native_stepper(solver, native_halo, delta_t, part, tdtype, dist, rank)
            if stepper is not None:
                halo, halo_kind = None, "native rccl (C++ stepper)"
    setup_s = time.time() - t0

    # HIP-event timing of the dominant kernel (events recorded on the launch stream)
    timers = []
    solver.kernel_timer = None
    py_stride, py_sampled = (8 if world > 1 else 1), [0]   # python-driven stages: events on every 8th step at N > 1

    def run(nsteps, timed):
        if stepper is not None:
            # the native driver takes all steps of a region in ONE call: the exchange stream and the compute stream
            # then meet at the entry and the exit of the region only (csrc/hip/stepper.hip)
            # kernel events on every 4th step (8th at N > 1): an event between two launches keeps them from running
            # back to back, which costs a small mesh up to 2x (c1: 0.050 vs 0.026 ms/step) and a multi-rank run host time
            stepper.timing((4 if world == 1 else 8) if timed else 0)
            solver.iterate_steps(nsteps, delta_t)
            return
        for i in range(nsteps):
            sampled = timed and i % py_stride == 0
            solver.kernel_timer = timers if sampled else None
            py_sampled[0] += 3 if sampled else 0
            solver.iterate(delta_t, halo=halo)

    import contextlib

    @contextlib.contextmanager
    def roctx(name):      # named host range for rocprofv3 --marker-trace (a no-op unless T8GPU_ROCTX=1)
        pushed = hip.lib().t8gpu_hip_range_push(name.encode())
        try:
            yield
        finally:
            if pushed:
                hip.lib().t8gpu_hip_range_pop()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed pre-warm (clocks, caches, lazy RCCL channels) before the W warm-up steps of the contract. Every rank
    # must run the SAME number of steps (each step exchanges halos), so the count comes from an all-reduced timing
    # of a first batch, never from a rank's own clock.
    fence()
    tp = time.perf_counter()
    run(10, False)
    fence()
    t10 = time.perf_counter() - tp
    if dist is not None:
        tt = torch.tensor([t10], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t10 = float(tt.item())
    more = int(min(max(args.prewarm_seconds - t10, 0.0) / max(t10, 1e-6) * 10, 20000))
    with roctx("bench.prewarm"):
        run(more, False)
    prewarm = 10 + more
    with roctx("bench.warmup"):
        run(args.warmup, False)
    # the timed region: EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks -- repeated
    # `--reps` times back to back (median of 5); value / ms_per_step are the MEDIAN repetition's.
    rep_s = []
    for rep in range(max(1, args.reps)):
        fence()
        t1 = time.perf_counter()
        with roctx(f"bench.timed_rep{rep}"):
            run(args.steps, True)
        fence()
        el = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        rep_s.append(el)
    elapsed = float(np.median(rep_s))
    host_enqueue = None
    if stepper is not None and world > 1:
        hms, hsteps = stepper.host_time()
        host_enqueue = round(hms / hsteps, 4) if hsteps else None   # (pre-warm, warm-up and timed steps alike)
    if stepper is not None:
        kernel_ms, kernel_launches = stepper.elapsed()
    else:
        kernel_ms, kernel_launches = sum(a.elapsed_time(b) for a, b in timers), len(timers)

    # sanity: the solution must still be finite (a diverged run is not a measurement)
    finite = bool(torch.isfinite(solver.state()).all().item())

    total_cells = n_global * cells
    value = total_cells * args.steps / elapsed / 1e6
    phi = part.F / max(1, part.N)
    per_update, flux_stage, rk = algorithmic_bytes(w["kind"], ft, phi if w["kind"] == "plain" else 0, 3, phi)
    # dominant kernel: the fused stage kernel (flux + RK of one stage) or the face-flux kernel
    if kernel_launches:
        stride = (8 if world > 1 else 4) if stepper is not None else py_stride    # steps whose stage kernels carry events
        stages_timed = stepper.timed_stages() if stepper is not None else py_sampled[0]
        steps_timed = stages_timed / 3.0
        avg_ms = kernel_ms / max(1, stages_timed)   # one fused stage may be split into several tile ranges
        local_cells = part.N * cells
        if mode == "fused":
            per_launch = local_cells * (flux_stage + sum(rk) / 3.0)
            kname = "fused_stage (flux + RK of one stage)"
        else:
            per_launch = local_cells * flux_stage
            kname = "flux_faces" if w["kind"] == "plain" else "subgrid_inner+outer"
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        # the kernel the last stage call launched for the bulk of its work, as rocprofv3 names it (C-ABI query); the PMC
        # figures of the committed profile are reported only if that profile is of THIS kernel (no stale
        # traffic / VALU / LDS fields)
        launched = None
        if mode == "fused":
            q = hip.lib().t8gpu_hip_last_stage_kernel
            q.restype = ctypes.c_char_p
            launched = (q() or b"").decode() or None
        prof, stale = measured_profile(args.workload, dts, args.flux, mode, world, launched)
        traffic = prof.get("hbm_bytes_per_launch")
        fused_min = fused_min_bytes(solver, w["kind"], ft, part) if mode == "fused" else None
        # `achieved` / `frac`: ALGORITHMIC bytes of the reference's unfused data flow over the measured launch
        # time -- a throughput-equivalent, which a fused kernel can push past 1. The UTILISATION figures are
        # `traffic_GBs` / `frac_traffic` (PMC-measured HBM bytes of the committed profile of this exact workload over
        # this run's launch time) and `valu_busy` / `lds_conflict_frac` from the same profile's SQ pass.
        roof = {"bound": "hbm", "kernel": kname, "kernel_launched": launched, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_copy_rate": round(achieved / HBM_COPY_GBS, 4),
                "achieved_is": "algorithmic bytes of the reference's unfused data flow / launch time (throughput-equivalent)",
                "traffic": traffic, "traffic_source": prof.get("source"),
                "traffic_GBs": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                "frac_traffic": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "fused_min_bytes_per_launch": fused_min,
                "frac_fused_min": round(fused_min / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fused_min else None,
                "valu_busy": prof.get("valu_busy"), "lds_conflict_frac": prof.get("lds_conflict_frac"),
                "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": int(per_launch), "launches_timed": kernel_launches}
        if stale:
            roof["profile_stale"] = stale
        if kernel_launches != 3 * steps_timed:
            roof["note"] = ("stage kernel split into interior / ghost-reading tile ranges; avg_launch_ms is "
                            "their sum per stage")
        if stride > 1:
            roof["note"] = (roof.get("note", "") + f"; kernel events on {stages_timed} of {3 * args.steps} stages of the timed region").lstrip("; ")
    else:
        roof = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(part, w, dts, delta_t, kindf, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "M cell-updates/sec (flux+RK3 step) on Kelvin-Helmholtz AMR",
            "value": round(value, 2), "unit": "M cell-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "repetitions": len(rep_s), "ms_per_step_min": round(min(rep_s) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(rep_s) / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dts, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "elements": int(n_global),
                       "cells": int(total_cells), "faces_per_element": round(phi, 4), "flux": args.flux,
                       "kernels": mode, "driver": "native C++ stepper" if stepper is not None else "python",
                       "halo": halo_kind, "partition": f"sfc-contiguous x{world}", "delta_t": delta_t,
                       "algorithmic_bytes_per_cell_update": round(per_update, 1), "finite": finite,
                       "setup_s": round(setup_s, 1), "prewarm_steps": prewarm},
            "algorithmic_frac_whole_step": round(value * 1e6 * per_update / world / (HBM_PEAK_GBS * 1e9), 4),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if world > 1:
            # what the N > 1 line needs to be read on its own: how the step was driven, what the host paid per
            # step for it, and which RCCL / HIP builds the process bound against which headers the library was compiled with
            from t8gpu_amd import native as _native
            ver = _native.runtime_versions()
            out["multi_gpu"] = {"driver": ("two lanes (interior tiles || RCCL -> ghost-reading tiles), two host threads, ghost window"
                                           if stepper is not None else "python-driven stages"),
                                "host_enqueue_ms_per_step": host_enqueue,
                                "rccl_version": {"compiled": ver["rccl"][0], "runtime": ver["rccl"][1]},
                                "hip_version": {"compiled": ver["hip"][0], "runtime": ver["hip"][1]},
                                "tiles_rank0": {"interior": int(solver.plan.host.n_interior), "ghost_reading": int(solver.plan.host.ntiles - solver.plan.host.n_interior)}
                                if mode == "fused" and w["kind"] == "plain" else None}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torchrun (a child process, not
    an exec: this process has not initialised the GPU and never will), forward its output, return its rc."""
    import socket
    import subprocess
    rehearsal = os.environ.get("T8GPU_REHEARSAL", "0") == "1"
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU on this image
    if not rehearsal and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible (T8GPU_REHEARSAL=1 rehearses the "
              f"N-rank flow on one GPU over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def bench_adaptive(args, w, dts, tdtype, kindf, mode, rank, world, dist, rehearsal):
    """BASELINE config 5's loop: K steps with adapt + repartition every `every` steps INSIDE the timed region
    (MeshManager::adapt + partition + compute_connectivity_information, mesh_manager.inl:196-330,626-723,333-481, and this
    backend's tile plan). One JSON line: value = cell-updates/s of the whole loop; config carries the step and the
    cycle time separately (the device kernels are the small part of a cycle, as in the reference: DESIGN.md section 7)."""
    from t8gpu_amd import amr, hostmem
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.solver import PlainSolver
    from t8gpu_amd.synth import SynthMesh
    if os.environ.get("T8GPU_KEEP_HEAP", "1") != "0":
        hostmem.keep_heap()     # the cycle's host arrays are reused instead of page-faulted in again every adapt
    if os.environ.get("T8GPU_PINNED_UPLOADS", "1") != "0" and not rehearsal:
        hostmem.use_pinned_uploads()   # ~300 MB of plan / connectivity arrays per cycle through one pinned staging buffer
    a = dict(w["adaptive"])
    if os.environ.get("T8GPU_C5A_LEVELS"):      # "min,max": a small version of the same loop (tests)
        a["min_level"], a["max_level"] = (int(x) for x in os.environ["T8GPU_C5A_LEVELS"].split(","))
    t0 = time.time()

    def adapt(s):
        if world == 1:
            new = amr.adapt(s, a["threshold"], a["min_level"], a["max_level"])[0]
        else:
            new = amr.adapt_partitioned(s, dist, host_staged=rehearsal, threshold=a["threshold"], min_level=a["min_level"],
                                        max_level=a["max_level"])
        return new

    def total(x, op="sum"):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return float(t.item())

    mesh = SynthMesh(w["dim"], a["min_level"], a["min_level"])
    solver = PlainSolver(mesh.partition(rank, world), tdtype, flux_kind=kindf, mode=mode)
    for _ in range(a["max_level"] - a["min_level"]):        # the reference adapts before it starts stepping
        solver = adapt(solver)
        ic = torch.from_numpy(solver.part.kh_initial_state()).to(tdtype).cuda()
        solver.planes[5 * solver.next:5 * solver.next + 5] = ic
    setup_s = time.time() - t0

    # Several ranks on the real backend: the C++ two-lane step driver with its native RCCL exchange, as in the fixed-mesh run
    # (main()). ONE communicator for the whole run -- made here, cross-checked once against the torch.distributed exchange and
    # proven by two trial steps, every decision collective (make_native_halo, bring_up_native_stepper) -- and a new halo
    # descriptor + stepper per adapted mesh (attach). Any "no": every rank steps through torch.distributed, stage by stage.
    native_comm, driver = None, "python-driven stages (torch.distributed halo)" if dist is not None else "native C++ stepper"
    if dist is not None and not rehearsal and mode == "fused" and os.environ.get("T8GPU_STEPPER", "native") == "native" \
            and os.environ.get("T8GPU_HALO", "native") == "native":
        first = make_native_halo(solver.part, tdtype, solver, HaloExchange(solver.part, tdtype, dist), dist, rank, world)
        if first is not None:
            dt0 = 0.1 * 2.0 ** -solver.part.mesh.finest_level
            if bring_up_native_stepper(solver, first, dt0, solver.part, tdtype, dist, rank) is not None:
                native_comm, driver = first.comm, "native C++ stepper (native rccl halo)"

    def attach(s):
        if native_comm is not None:
            from t8gpu_amd import native
            if getattr(s, "stepper", None) is None:
                s.use_native_stepper(native.NativeHalo(s.part, tdtype, native_comm))
            return None
        halo = HaloExchange(s.part, tdtype, dist, stage_through_host=rehearsal) if world > 1 else None
        if world == 1 and dist is None and mode == "fused":
            s.use_native_stepper()
        return halo

    halo = attach(solver)
    t_step = t_cycle = 0.0
    cells = cycles = 0
    split = {}

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def chunk(n, timed):
        nonlocal t_step, cells
        dt = 0.1 * 2.0 ** -solver.part.mesh.finest_level
        sync()
        t1 = time.perf_counter()
        if halo is None:
            solver.iterate_steps(n, dt)
        else:
            for _ in range(n):
                solver.iterate(dt, halo=halo)
        sync()
        if timed:
            t_step += time.perf_counter() - t1
            cells += solver.N * n

    chunk(args.warmup, False)
    sync()
    tstart = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(a["every"], args.steps - done)
        chunk(n, True)
        done += n
        if done < args.steps:
            sync()
            t1 = time.perf_counter()
            solver = adapt(solver)
            halo = attach(solver)
            sync()
            t_cycle += time.perf_counter() - t1
            cycles += 1
            for k, v in (getattr(solver, "last_adapt_split", None) or {}).items():
                split[k] = split.get(k, 0.0) + v
    sync()
    elapsed = total(time.perf_counter() - tstart, "max")
    finite = bool(torch.isfinite(solver.state()).all().item())
    # every rank takes part in the reductions; rank 0 prints
    tot_cells, n_end = total(cells), int(total(solver.N))
    step_s, cycle_s = total(t_step, "max"), total(t_cycle, "max")
    if rank == 0:
        print(json.dumps({
            "metric": "M cell-updates/sec (flux+RK3 step) on Kelvin-Helmholtz AMR", "value": round(tot_cells / elapsed / 1e6, 2),
            "unit": "M cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dts, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "elements_at_end": n_end, "flux": args.flux,
                       "kernels": mode, "driver": driver, "adapt_every": a["every"], "adapt_cycles_timed": cycles,
                       "step_ms": round(step_s / max(1, args.steps) * 1e3, 4),
                       "cycle_ms": round(cycle_s / max(1, cycles) * 1e3, 2) if cycles else None,
                       # one rank: where a cycle goes -- indicator kernels + read-back / mesh provider (t8code's share in the
                       # reference) / this backend's tile plan / new planes + uploads / transfer kernel + step driver
                       "cycle_split_ms": {k: round(v / max(1, cycles) * 1e3, 2) for k, v in split.items()} or None,
                       "stepping_only_M_cell_updates_per_s": round(tot_cells / step_s / 1e6, 2),
                       "partition": f"sfc-contiguous x{world}, repartitioned at every adapt", "finite": finite,
                       "setup_s": round(setup_s, 1)},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measured_profile(workload, dts, flux, mode, world, launched=None):
    """The committed rocprofv3 PMC record of exactly this workload / dtype / flux / kernel tier at N = 1
    (profiles/traffic.json, written by scripts/profile_gpu.sh + commit_profile.py): HBM bytes per launch of the
    dominant kernel, VALU-busy and LDS-conflict fractions from the SQ pass. Returns (record, why_not): the record is {}
    -- and the second value says why -- when no profile of this workload exists or when it is a profile of ANOTHER
    kernel than the one this run launched (`launched`: t8gpu_hip_last_stage_kernel), so that no figure of a stale
    profile is ever reported next to this run's timing."""
    if world != 1:
        return {}, None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f).get(f"{workload}|{dts}|{flux}|{mode}") or {}
    except (OSError, ValueError):
        return {}, "profiles/traffic.json missing or unreadable"
    if not rec:
        return {}, "no committed profile of this workload / dtype / flux / tier"
    if launched is not None and launched not in rec.get("kernels", []):
        return {}, f"profile {rec.get('source')} is of {rec.get('kernels')}, this run launched {launched}"
    from t8gpu_amd.build import kernel_source_hash
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return {}, (f"profile {rec.get('source')} was taken with other kernel sources (hash {rec.get('kernel_source_hash')}, now "
                    f"{kernel_source_hash()}): re-profile with scripts/profile_gpu.sh + commit_profile.py")
    return rec, None


def fused_min_bytes(solver, kind, ft, part):
    """Compulsory HBM bytes of ONE fused stage launch, averaged over the three stages: every
    byte the fused data flow must move if each array crossed HBM exactly once -- state in + out, the previous-step
    state of stages 2 and 3, volumes, the per-face speed estimates and the plan arrays the kernel reads. Halo
    re-reads of neighbouring tiles' elements are NOT in it (an ideal cache serves them)."""
    h = solver.plan.host
    if kind == "plain":
        n_tf = int(h.face_lr.size)                       # tile faces (cut faces appear in both tiles)
        geo = 2 * n_tf if h.geo_table.shape[0] else 4 * ft * n_tf
        # speed estimates (+ the original face ids they are scattered by) are written by the third stage only
        # (patch tiles have no face records and read no face-list rows: only the generic tiles' elements count for the ELL rows)
        plan = n_tf * (4 + 4 / 3.0) + geo + h.n_ell_rows * h.ell_width * 2 + int(h.halo_ids.size) * 4 + 32 * h.ntiles
        state = part.N * ft * (5 + 5 + 10.0 / 3.0 + 1)
        return int(state + (part.F + part.B) * ft / 3.0 + plan)
    cells = part.N * part.cells_per_element
    plan = part.N * 64 + h.n_entries * 16
    return int(cells * ft * (5 + 5 + 10.0 / 3.0) + part.N * ft + plan)


def _all_agree(ok, dist):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def make_native_halo(part, tdtype, solver, torch_halo, dist, rank, world):
    """Native RCCL communicator + halo descriptor, cross-checked ONCE against the torch.distributed exchange on
    the initial state. Two collective decisions, so that no rank ever waits for a peer that has given up:
    (1) did every rank get its communicator, (2) did every rank's native exchange complete (60 s stall guard)
    and reproduce the torch.distributed ghosts. Any "no" -> None on all ranks, the run continues on the
    torch.distributed transport."""
    from t8gpu_amd import native

    def complain(exc):
        print(f"[bench rank {rank}] native RCCL halo unavailable ({exc}); using torch.distributed", file=sys.stderr, flush=True)

    comm, nh, ok = None, None, True
    try:
        def bcast(b, src):
            box = [b]
            dist.broadcast_object_list(box, src=src)
            return box[0]
        comm = native.NativeComm(rank, world, bcast)
        nh = native.NativeHalo(part, tdtype, comm)
    except Exception as exc:  # noqa: BLE001
        complain(exc)
        ok = False
    if not _all_agree(ok, dist):
        if comm is not None:
            try:
                comm.abort()
            except Exception:  # noqa: BLE001
                pass
        return None
    src5 = solver.planes[0:5]
    ghosts = slice(part.N, part.N + part.G)
    saved = src5[:, ghosts].clone()
    try:
        torch_halo.exchange(src5)                      # collective: every rank is here (decision 1)
        torch.cuda.synchronize()
        want = src5[:, ghosts].clone()
        src5[:, ghosts] = float("nan")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        nh.exchange(src5, side)
        if native.stream_wait(side, 60.0) != 0:
            raise RuntimeError("native halo exchange did not complete within 60 s")
        torch.cuda.synchronize()
        if part.G and not torch.equal(src5[:, ghosts], want):
            raise RuntimeError("native halo exchange disagrees with the torch.distributed exchange")
    except Exception as exc:  # noqa: BLE001
        complain(exc)
        ok = False
    if ok:
        src5[:, ghosts] = saved
    agreed = _all_agree(ok, dist)
    if not agreed:
        try:
            comm.abort()
        except Exception:  # noqa: BLE001
            pass
        src5[:, ghosts] = saved
        return None
    return nh


def bring_up_native_stepper(solver, native_halo, delta_t, part, tdtype, dist, rank):
    """Every rank must end up on the SAME transport. Bring the C++ driver up, push two trial steps through its
    pipeline under a stall guard, and let an all-reduce decide: any rank that failed or stalled sends everybody
    back to the torch.distributed path (None is returned, the initial state is restored)."""
    from t8gpu_amd import native
    ok, stepper = 1, None
    try:
        stepper = solver.use_native_stepper(native_halo)
        solver.iterate_steps(2, delta_t)
        if native.stream_wait(torch.cuda.current_stream(), 60.0) != 0:
            raise RuntimeError("the native step pipeline did not complete two trial steps within 60 s")
        if not bool(torch.isfinite(solver.state()).all().item()):
            raise RuntimeError("the native step pipeline produced non-finite values")
    except Exception as exc:  # noqa: BLE001
        print(f"[bench rank {rank}] native stepper unavailable ({exc}); python-driven stages", file=sys.stderr, flush=True)
        ok = 0
    if _all_agree(ok == 1, dist):
        return stepper
    try:
        native_halo.comm.abort()
    except Exception:  # noqa: BLE001
        pass
    solver.stepper = None
    ic = torch.from_numpy(part.kh_initial_state()).to(tdtype).cuda()      # [5, (N + G) * cells per element]
    solver.planes[:25].zero_()
    solver.planes[0:5, :ic.shape[1]] = ic
    solver.next, solver.prev = 0, 3   # Step0 / Step3, as after construction (solver.h:100-101)
    return None


def cpu_baseline(part, w, dts, delta_t, kindf, budget_s):
    """The CPU oracle (a port: the reference has no CPU path) timed on this box's host cores on a bounded sample of the same
    mesh: the state planes are first-touched by the OpenMP team (oracle_first_touch), one step each on {16, 32, 64, all}
    threads picks the team size (a rank of a GPU box is entitled to a share of the node's cores, and the node throttles
    well before all of its hardware threads are busy: csrc/host/host_threads.hpp), whole steps with that team until
    ~budget_s of wall time is spent give `value`, then ONE step on one thread (SURVEY 8d asks for both figures).
    The face passes accumulate owner-computes (no atomics; oracle.hpp: OwnerScatter)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    npdt = np.float64 if dts == "f64" else np.float32
    lib = O.lib(omp=True)
    all_threads = lib.oracle_num_threads()
    cells = part.N * part.cells_per_element

    def make():
        return (O.PlainCase(part, npdt, first_touch=lib) if w["kind"] == "plain" else O.SubgridCase(part, npdt, first_touch=lib))

    def timed(case, budget, max_steps):
        t0 = time.perf_counter()
        steps = 0
        while True:
            case.iterate(delta_t, kind=kindf, omp=True)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= max_steps or el / steps * (steps + 1) > 1.5 * budget:
                return steps, el

    sweep = {}
    try:
        for nt in sorted({n for n in (16, 32, 64, all_threads) if 0 < n <= all_threads}):
            lib.oracle_set_num_threads(nt)
            case = make()                                   # (planes first-touched by THIS team)
            case.iterate(delta_t, kind=kindf, omp=True)      # thread start-up, remaining page faults
            st, el = timed(case, 0.0, 1)
            sweep[nt] = cells * st / el / 1e6
            del case
        best = max(sweep, key=sweep.get)
        lib.oracle_set_num_threads(best)
        case = make()
        case.iterate(delta_t, kind=kindf, omp=True)
        steps, el = timed(case, 0.4 * budget_s, 50)
        del case
        lib.oracle_set_num_threads(1)
        case = O.PlainCase(part, npdt) if w["kind"] == "plain" else O.SubgridCase(part, npdt)
        steps1, el1 = timed(case, 0.0, 1)
    finally:
        lib.oracle_set_num_threads(all_threads)
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = "unlimited" if quota[0] == "max" else f"{int(quota[0]) / int(quota[1]):.1f} cpus"
    except (OSError, ValueError, IndexError):
        quota = "unknown"
    return {"value": round(cells * steps / el / 1e6, 3), "unit": "M cell-updates/s", "cores": int(best),
            "kind": "port", "sample": f"{steps} full step(s) of the same mesh ({cells} cells) after one untimed step, OpenMP oracle on {best} threads "
            f"(owner-computes accumulation, no atomics; planes first-touched by the team), {el:.1f} s wall",
            "threads_sweep_M_per_s": {str(k): round(v, 3) for k, v in sweep.items()}, "hardware_threads": int(all_threads),
            "cgroup_cpu_quota": quota,
            "value_1_thread": round(cells * steps1 / el1 / 1e6, 3),
            "sample_1_thread": f"{steps1} full step on 1 thread, {el1:.1f} s wall", "cpu_model": _cpu_model()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
