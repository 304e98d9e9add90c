#!/usr/bin/env python3
"""What one adapt cycle costs next to the steps between two cycles ("honest end-to-end numbers
where adapt runs every 50-100 steps"). Splits CompressibleEulerSolver::adapt into its parts: the device
kernels of this backend (indicator + data transfer), the forest work (here the synthetic provider standing
in for t8code's adapt / balance / ghost / face iteration, mesh_manager.inl:196-481) and the tile planning
pre-pass this backend adds.
usage: amr_cycle_cost.py [workload = c2 | c4]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t8gpu_amd import amr, fused, hip  # noqa: E402
from t8gpu_amd.solver import PlainSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402

MESHES = {"c2": (2, 6, 11, 0.0596), "c4": (2, 7, 12, 0.1472)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    dim, base, lmax, band = MESHES[name]
    mesh = SynthMesh(dim, base, lmax, band=band)
    part = mesh.partition()
    s = PlainSolver(part, torch.float64, mode="fused")
    s.use_native_stepper()
    dt = 0.1 * 2.0 ** -mesh.finest_level
    s.iterate_steps(50, dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.iterate_steps(100, dt)
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0

    def lap(label, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        laps.append((label, time.perf_counter() - t))
        return out

    laps = []
    crit = lap("indicator kernels (estimate_gradient + criteria) + D2H", lambda: amr.refinement_criteria(s).double().cpu().numpy())
    marks = lap("adapt callback over the forest (host)", lambda: mesh.marks_from_criteria(crit, 10.0, base, lmax + 1, 4))
    new_mesh, ad = lap("forest adapt + 2:1 balance + old->new walk (host)", lambda: mesh.adapt(marks))
    new_part = lap("partition + face connectivity (host, stands for t8code)", lambda: new_mesh.partition())
    plan = lap("tile plan (host, this backend's pre-pass) + upload", lambda: fused.PlainPlan(new_part, torch.float64))
    new = lap("new MemoryManager planes", lambda: PlainSolver(new_part, torch.float64, mode="compat", state=np.zeros((5, new_part.N))))
    adt = torch.from_numpy(ad).cuda()
    lap("adapt_variables_and_volume kernel", lambda: hip.call(
        "t8gpu_hip_adapt_variables_and_volume", torch.float64, new_part.N, dim, hip.ptr(adt), s.get_own_variables(s.next),
        new.get_own_variables(new.next), hip.ptr(s.planes[25]), hip.ptr(new.planes[25]), hip.stream_ptr()))
    total = sum(t for _, t in laps)
    print(f"{name}: {part.N} -> {new_part.N} elements; 100 steps take {t_steps * 1e3:.1f} ms ({t_steps * 10:.3f} ms/step)")
    for label, t in laps:
        print(f"  {t * 1e3:9.1f} ms  {label}")
    print(f"  {total * 1e3:9.1f} ms  one adapt cycle = {total / t_steps * 100:.0f} steps' worth of time")


if __name__ == "__main__":
    main()
