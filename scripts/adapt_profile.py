#!/usr/bin/env python3
"""cProfile of one steady-state adapt cycle of the c5a workload (one rank): where the PYTHON side of a cycle goes.
usage: adapt_profile.py [min_level=5] [max_level=9]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from t8gpu_amd import amr, hostmem  # noqa: E402
from t8gpu_amd.solver import PlainSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402


def main():
    lo, hi = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 5), (2, 9)))
    hostmem.keep_heap()
    mesh = SynthMesh(3, lo, lo)
    s = PlainSolver(mesh.partition(0, 1), torch.float64, flux_kind=0, mode="fused")
    for _ in range(hi - lo):
        s = amr.adapt(s, 10.0, lo, hi)[0]
        s.planes[5 * s.next:5 * s.next + 5] = torch.from_numpy(s.part.kh_initial_state()).cuda()
    s.use_native_stepper()
    dt = 0.1 * 2.0 ** -s.part.mesh.finest_level
    for cycle in range(3):
        s.iterate_steps(20, dt)
        torch.cuda.synchronize()
        if cycle == 2:
            pr = cProfile.Profile()
            pr.enable()
        s = amr.adapt(s, 10.0, lo, hi)[0]
        torch.cuda.synchronize()
    pr.disable()
    h = s.plan.host
    gen = ~h.tile_patch
    import numpy as np
    print(f"plan: {h.n_patches} patches ({sum(h.n_irregular_class)} irregular) = {h.n_patches * 256 / s.N:.1%} of the elements; "
          f"{int(gen.sum())} generic tiles, {np.diff(h.elem_off)[gen].mean():.0f} elements / {np.diff(h.face_off)[gen].mean():.0f} faces / "
          f"{np.diff(h.halo_off)[gen].mean():.0f} halo entries each; levels {np.bincount(s.part.levels[:s.N])}")
    print("N =", s.N, "split (s):", {k: round(v, 4) for k, v in s.last_adapt_split.items()})
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
