"""Rate of the fused plain path (fp64 KEPES, one rank) on a 2D AMR mesh of a chosen size: usage mid_mesh_rate.py base lmax band.\nUsed to place the persistent / one-tile kernel crossover (T8GPU_PERSISTENT=0 / T8GPU_PERSISTENT_WGS=3 force either)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t8gpu_amd.solver import PlainSolver
from t8gpu_amd.synth import SynthMesh
base, lmax, band = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
mesh = SynthMesh(2, base, lmax, band=band)
part = mesh.partition()
s = PlainSolver(part, torch.float64, mode="fused")
s.use_native_stepper()
dt = 0.1 * 2.0 ** -mesh.finest_level
s.iterate_steps(20, dt); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.perf_counter(); s.iterate_steps(100, dt); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
print(f"N={part.N} tiles={s.plan.host.ntiles} {part.N * 100 / best / 1e6:.0f} M/s")
