#!/usr/bin/env python3
"""Bitwise comparison of kernel variants on one mesh (debug aid): runs the same 3 steps in child processes with
different T8GPU_* settings / library builds and compares the saved states.
usage: debug_variants.py            (parent)   |   debug_variants.py --child out.npy   (child)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(out):
    import numpy as np
    import torch
    from _gpu import perturbed_state
    from t8gpu_amd.solver import PlainSolver
    from t8gpu_amd.synth import SynthMesh
    mesh = SynthMesh(2, 3, 6, band=0.06, periodic=False)
    part = mesh.partition()
    st = perturbed_state(part, 4)
    g = PlainSolver(part, torch.float64, mode="fused", state=st)
    states = []
    for _ in range(3):
        g.iterate(0.1 * 2.0 ** -6)
        torch.cuda.synchronize()
        states.append(g.state().cpu().numpy().copy())
    np.save(out, np.stack(states))


def main():
    import numpy as np
    runs = {"onetile": dict(T8GPU_PERSISTENT="0"), "persistent": {}, "persistent_wgs1": dict(T8GPU_PERSISTENT_WGS="1"),
            "noaxis": dict(T8GPU_HIP_LIB=os.path.join(ROOT, "t8gpu_amd/lib/variants/libt8gpu_hip_noaxis.so"))}
    res = {}
    for name, env in runs.items():
        out = f"/tmp/dbg_{name}.npy"
        subprocess.run([sys.executable, __file__, "--child", out], env=dict(os.environ, **env), check=True)
        res[name] = np.load(out)
    ref = res["onetile"]
    for name, a in res.items():
        d = np.abs(a - ref)
        bad = np.argwhere(d[0] != 0)
        print(f"{name:>16}: max |diff| per step {[float(x.max()) for x in d]}  differing entries after step 1: {len(bad)}"
              + (f"  first: var {bad[0][0]} elem {bad[0][1]} ref {ref[0][tuple(bad[0])]!r} got {a[0][tuple(bad[0])]!r}" if len(bad) else ""))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
