#!/usr/bin/env python3
"""Randomised differential run: GPU tiers against the CPU oracle over random meshes, states, tilings and
partitions. Prints one line per case and a summary; exits non-zero on the first violation.
usage: fuzz_parity.py [seconds=120] [seed=0] [only_case]   (only_case: replay one case of that seed, verbosely)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from _gpu import NP, TOL1, perturbed_state, rel_err  # noqa: E402
from t8gpu_amd import hip  # noqa: E402
from t8gpu_amd.solver import PlainSolver, SubgridSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402
from t8gpu_amd.unstructured import PrismHexMesh, shell_map, wavy_map  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None
    t0, n, worst = time.time(), 0, {}
    while time.time() - t0 < budget and (only is None or n < only):
        kind = rng.choice(["plain2", "plain3", "prism", "sub2", "sub3"])
        dtype = torch.float64 if rng.random() < 0.5 else torch.float32
        flux = hip.KEPES if rng.random() < 0.7 else (hip.HLL if rng.random() < 0.5 else hip.HLLC)
        periodic = bool(rng.random() < 0.5)
        band = float(rng.choice([0.0, 0.02, 0.05, 0.11, 0.3]))
        seed = int(rng.integers(1 << 30))
        steps = int(rng.integers(1, 4))
        opts = {}
        if kind in ("plain2", "plain3", "sub2", "sub3"):
            dim = 2 if kind.endswith("2") else 3
            # (round 3: large enough for structured patches -- 16 x 16 blocks from level 6 in 2D, 8 x 8 x 4 from level 5 in 3D)
            base = int(rng.integers(1, 7 if dim == 2 else 5))
            lmax = min(base + int(rng.integers(0, 3 if dim == 2 else 2)), 8 if dim == 2 else 5)
            if kind.startswith("sub"):
                base, lmax = min(base, 3), min(lmax, 3 if dim == 3 else 4)
            mesh = SynthMesh(dim, base, lmax, band=band, periodic=periodic)
            part = mesh.partition(subgrid=kind.startswith("sub"))
            dt = 0.1 * 2.0 ** -(mesh.finest_level + (2 if kind.startswith("sub") else 0))
            desc = f"{kind} base {base} max {lmax} band {band} periodic {periodic}"
        else:
            n3 = tuple(int(x) for x in rng.choice([2, 4, 8], 3))
            split = rng.choice(["all", "none", "checker", "0.3"])
            mesh = PrismHexMesh(n3, split=split if split in ("all", "none", "checker") else float(split),
                                mapping=wavy_map if periodic else shell_map, periodic=periodic, seed=seed)
            part = mesh.partition()
            dt = 0.03 * float(np.cbrt(part.volumes.min()))   # coarse curved cells + a random state: stay well inside stability
            desc = f"prism {n3} split {split} periodic {periodic}"
        cells = part.cells_per_element
        mode = "fused" if rng.random() < 0.7 else "compat"
        if not kind.startswith("sub") and mode == "fused" and rng.random() < 0.5:
            opts = dict(tmax=int(rng.choice([16, 50, 256])), fcap=int(rng.choice([40, 130, 512, 1024])),
                        compressed=bool(rng.random() < 0.8), dictionary=bool(rng.random() < 0.7), patches=bool(rng.random() < 0.7))
        if only is not None and n + 1 != only:
            n += 1
            continue
        st = perturbed_state(part, seed, cells)
        if kind.startswith("sub"):
            g = SubgridSolver(part, dtype, flux_kind=flux, mode=mode, state=st)
            o = O.SubgridCase(part, NP[dtype], state=st)
        else:
            g = PlainSolver(part, dtype, flux_kind=flux, mode=mode, state=st, plan_options=opts if mode == "fused" else None)
            o = O.PlainCase(part, NP[dtype], state=st)
        for _ in range(steps):
            g.iterate(dt)
            o.iterate(dt, kind=flux)
        torch.cuda.synchronize()
        ncell = part.N * cells
        if only is not None:
            gs, os_ = g.state().cpu().numpy()[:, :ncell], o.current()[:, :ncell]
            print("seed", seed, "gpu finite", np.isfinite(gs).all(), "oracle finite", np.isfinite(os_).all(), "dt", dt,
                  "min vol", part.volumes.min(), "min rho/p of the state", st[0].min(), flush=True)
            bad = np.argwhere(~np.isfinite(gs))
            print("non-finite GPU entries", bad[:10].tolist(), "oracle there", [os_[tuple(b)] for b in bad[:5]], flush=True)
        if not np.isfinite(o.current()[:, :ncell]).all():
            n += 1          # the random state blew up in the ORACLE too (physics, not parity): not a case
            continue
        err = rel_err(g.state().cpu().numpy()[:, :ncell], o.current()[:, :ncell])
        tol = TOL1[dtype] * (steps + 2) * (3 if mode == "compat" else 1)
        key = (kind, str(dtype).split(".")[-1], mode)
        worst[key] = max(worst.get(key, 0.0), err / tol)
        n += 1
        line = f"[{n:4d}] {desc:52s} N={part.N:7d} {str(dtype)[-7:]} flux {flux} {mode:6s} {opts} steps {steps}: err {err:.2e} (tol {tol:.1e})"
        if not (err < tol) or not np.isfinite(err):
            print("VIOLATION " + line, flush=True)
            sys.exit(1)
        if n % 25 == 0:
            print(line, flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s, no violation; worst err/tol per (mesh, dtype, tier):")
    for k in sorted(worst):
        print(f"  {k}: {worst[k]:.3f}")


if __name__ == "__main__":
    main()
