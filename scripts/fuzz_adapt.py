#!/usr/bin/env python3
"""Randomised check of the adapt + repartition machinery on ONE GPU: random mesh, random refinement criteria
(so: arbitrary refine / coarsen patterns, families cut by rank boundaries, balance cascades), random number of
ranks with a loopback transport. The partitioned adapt must give, bitwise, the state and volumes of the
single-rank adapt with the same marks, the element counts must re-balance, mass must be conserved, and the
adapted mesh must still be 2:1 balanced with a consistent old -> new map.
usage: fuzz_adapt.py [seconds=120] [seed=0]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _gpu import perturbed_state  # noqa: E402
from t8gpu_amd import amr, hip  # noqa: E402
from t8gpu_amd.solver import PlainSolver, SubgridSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        sub = bool(rng.random() < 0.4)
        dim = int(rng.choice([2, 3]))
        base = int(rng.integers(1, 4 if dim == 2 else 3))
        lmax = base + int(rng.integers(0, 3 if dim == 2 else 2))
        if sub:
            base, lmax = min(base, 3 if dim == 2 else 2), min(lmax, 4 if dim == 2 else 3)
        mesh = SynthMesh(dim, base, lmax, band=float(rng.choice([0.0, 0.05, 0.15])), periodic=bool(rng.random() < 0.5))
        world = int(rng.integers(2, 7))
        if mesh.num_elements < world:
            continue
        seed = int(rng.integers(1 << 30))
        whole = mesh.partition(subgrid=sub)
        S = whole.cells_per_element
        st = perturbed_state(whole, seed, S)
        crit = rng.random(mesh.num_elements) ** 3 * 40.0          # mostly small, some far above the threshold
        kw = dict(threshold=10.0, min_level=max(1, base - 1), max_level=lmax + 1, family_members_averaged=int(rng.choice([0, 4])))
        Solver = SubgridSolver if sub else PlainSolver
        Adapt = amr.PartitionedSubgridAdapt if sub else amr.PartitionedAdapt
        ref = Solver(whole, torch.float64, mode="fused", state=st)
        parts = [mesh.partition(r, world, subgrid=sub) for r in range(world)]
        solvers = []
        for p in parts:
            gidx = np.concatenate([p.first_global + np.arange(p.N), p.ghost_global])
            cells = (gidx[:, None] * S + np.arange(S)[None, :]).reshape(-1)
            solvers.append(Solver(p, torch.float64, mode="fused", state=st[:, cells]))
        pas = [Adapt(s, crit, **kw) for s in solvers]
        by_rank = {p.rank: p for p in pas}
        for p in pas:
            for q, _, cnt in p.sends:
                if q != p.rank:
                    by_rank[q].recvbufs[p.rank].copy_(p.sendbufs[q])
        news = [p.finish() for p in pas]
        marks = pas[0].marks
        new_mesh, ad = mesh.adapt(marks)
        npart = new_mesh.partition(subgrid=sub)
        adt = torch.from_numpy(ad).cuda()
        if sub:
            want = SubgridSolver(npart, torch.float64, mode="fused", state=np.zeros((5, npart.N * S)))
            hip.call("t8gpu_hip_subgrid_adapt_variables_and_volume", torch.float64, dim, npart.N, hip.ptr(adt), ref.get_own_variables(ref.next),
                     want.get_own_variables(want.next), hip.ptr(ref.volumes), hip.ptr(want.volumes), hip.stream_ptr())
            wvol, gvol = want.volumes[: npart.N], torch.cat([x.volumes[: x.N] for x in news])
        else:
            want = PlainSolver(npart, torch.float64, mode="fused", state=np.zeros((5, npart.N)))
            hip.call("t8gpu_hip_adapt_variables_and_volume", torch.float64, npart.N, dim, hip.ptr(adt), ref.get_own_variables(ref.next),
                     want.get_own_variables(want.next), hip.ptr(ref.planes[25]), hip.ptr(want.planes[25]), hip.stream_ptr())
            wvol, gvol = want.planes[25, : npart.N], torch.cat([x.planes[25, : x.N] for x in news])
        torch.cuda.synchronize()
        got = torch.cat([x.state() for x in news], dim=1)
        n += 1
        line = (f"[{n:4d}] {'sub' if sub else 'plain'}{dim} base {base} max {lmax} world {world} N {mesh.num_elements} -> {new_mesh.num_elements} "
                f"refine {int((marks > 0).sum())} coarsen {int((marks < 0).sum())}")
        sizes = [x.N for x in news]
        checks = {"state": torch.equal(got, want.state()), "volumes": torch.equal(gvol, wvol),
                  "balance": sum(sizes) == npart.N and max(sizes) - min(sizes) <= 1,
                  # geometry of the new mesh: volumes tile the unit domain, old -> new map covers every old element once
                  "tiling": abs(float(npart.volumes[: npart.N].sum()) - 1.0) < 1e-12,
                  # old -> new map: monotone, steps of 0 (children of a refined element), 1 (kept) or 2^dim (a coarsened family)
                  "map": bool(ad[0] == 0 and ad[-1] == mesh.num_elements and np.isin(np.diff(ad), (0, 1, 2 ** dim)).all())}
        ok = all(checks.values())
        # mass: refinement and (with all family members averaged) coarsening conserve it exactly up to rounding
        if kw["family_members_averaged"] == 0 or not (marks < 0).any():
            m_old = (st.reshape(5, -1, S).mean(axis=2) * whole.volumes[None, : whole.N]).sum(axis=1)
            gs = got.cpu().numpy().reshape(5, -1, S).mean(axis=2)
            m_new = (gs * npart.volumes[None, : npart.N]).sum(axis=1)
            checks["mass"] = bool(np.abs(m_new - m_old).max() < 1e-12 * np.abs(m_old).max())
            ok = ok and checks["mass"]
        if not ok:
            print("VIOLATION " + line, checks, flush=True)
            sys.exit(1)
        if n % 20 == 0:
            print(line, flush=True)
    print(f"{n} adapt + repartition cases in {time.time() - t0:.0f} s: all equal to the single-rank adapt")


if __name__ == "__main__":
    main()
