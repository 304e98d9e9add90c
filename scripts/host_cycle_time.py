#!/usr/bin/env python3
"""Host-side cost of one adapt cycle, piece by piece (no GPU needed): the synthetic provider standing where t8code's
adapt / ghost / face iteration run (forest adapt, partition + connectivity arrays) and this backend's tile plan.
usage: host_cycle_time.py [dim=3] [base=6] [lmax=8] [band=0.03]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("T8GPU_PLAN_VERBOSE", "1")
from t8gpu_amd.plan import HostPlainPlan  # noqa: E402
from t8gpu_amd.synth import SynthMesh, _p, lib  # noqa: E402


def lap(what, fn):
    t = time.perf_counter()
    r = fn()
    print(f"{what:42s} {time.perf_counter() - t:7.3f} s", flush=True)
    return r


def main():
    dim, base, lmax = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 3), (2, 6), (3, 8)))
    band = float(sys.argv[4]) if len(sys.argv) > 4 else 0.03
    print(f"cpus: {os.cpu_count()}  OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}")
    mesh = lap("mesh build", lambda: SynthMesh(dim, base, lmax, band=band))
    rng = np.random.default_rng(1)
    marks = np.zeros(mesh.num_elements, np.int8)
    marks[rng.random(mesh.num_elements) < 0.02] = 1
    new, _ = lap("forest adapt + balance + old->new map", lambda: mesh.adapt(marks))
    L = lib()
    h = lap("  part_create (C++: faces, ghosts, peers)", lambda: L.t8gpu_synth_part_create(new._h, 0, 1, 0, 3))
    L.t8gpu_synth_part_destroy(h)
    part = lap("partition() incl. python arrays + IC", lambda: new.partition(0, 1))
    print(f"N = {part.N}  F = {part.F}")
    lap("tile plan (host) total", lambda: HostPlainPlan.from_partition(part, fcap=480 if dim == 3 else 512, want_face_geo=False, patches=True))


if __name__ == "__main__":
    main()
