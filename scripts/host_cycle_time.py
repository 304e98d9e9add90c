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
    if os.environ.get("T8GPU_KEEP_HEAP", "1") != "0":
        from t8gpu_amd import hostmem
        print("keep_heap:", hostmem.keep_heap())
    print(f"cpus: {os.cpu_count()}  OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}")
    mesh = lap("mesh build", lambda: SynthMesh(dim, base, lmax, band=band))
    rng = np.random.default_rng(1)
    L = lib()
    # T8GPU_HCT_REPEAT cycles, each on the mesh the previous one made (the later ones show the steady state of a run: freed
    # arrays of one cycle are the next one's memory when the application keeps its heap, t8gpu_amd/hostmem.py)
    level = np.asarray(mesh.partition(0, 1).levels[:mesh.num_elements])
    for cycle in range(int(os.environ.get("T8GPU_HCT_REPEAT", "1"))):
        print(f"-- cycle {cycle}")
        marks = np.zeros(mesh.num_elements, np.int8)
        marks[(rng.random(mesh.num_elements) < 0.02) & (level < lmax)] = 1
        t0 = time.perf_counter()
        new, _ = lap("forest adapt + balance + old->new map", lambda: mesh.adapt(marks))
        part = lap("partition() incl. python arrays", lambda: new.partition(0, 1))
        print(f"N = {part.N}  F = {part.F}")
        lap("tile plan (host) total", lambda: HostPlainPlan.from_partition(part, fcap=480 if dim == 3 else 512, want_face_geo=False, patches=True))
        print(f"{'cycle (host side)':42s} {time.perf_counter() - t0:7.3f} s")
        mesh, level = new, np.asarray(part.levels[:part.N])
        del part


if __name__ == "__main__":
    main()
