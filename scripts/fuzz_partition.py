#!/usr/bin/env python3
"""Randomised check of the multi-rank data path on ONE GPU: random mesh, random number of ranks, random tile
caps; all ranks live in this process and exchange through a loopback (what RCCL send/recv does across GPUs).
The gathered result of the partitioned run must be BITWISE the single-rank run (plain elements and Subgrid
blocks, fused tier), and every rank's tile classes must respect their definition.
usage: fuzz_partition.py [seconds=120] [seed=0]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _gpu import perturbed_state  # noqa: E402
from t8gpu_amd import fused  # noqa: E402
from t8gpu_amd.halo import HaloExchange  # noqa: E402
from t8gpu_amd.solver import PlainSolver, SubgridSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402
from t8gpu_amd.unstructured import PrismHexMesh, shell_map, wavy_map  # noqa: E402


def loopback(halos):
    by_rank = {h.rank: h for h in halos}
    for h in halos:
        for j, p in enumerate(h.peers):
            peer = by_rank[p]
            jj = peer.peers.index(h.rank)
            w = 5 * h.cells
            src = h.sendbuf[w * h.send_off[j]:w * h.send_off[j + 1]]
            dst = peer.recvbuf[w * peer.recv_off[jj]:w * peer.recv_off[jj + 1]]
            assert src.numel() == dst.numel() > 0
            dst.copy_(src)


def check_classes(plan, part):
    """A = reads a ghost slot, B = reads an element an A tile owns, C = neither; tile_order = C, B, A."""
    h = plan.host
    owner = np.empty(part.N, np.int64)
    reads_ghost = np.zeros(h.ntiles, bool)
    for t in range(h.ntiles):
        owner[h.elem_off[t]:h.elem_off[t + 1]] = t
        ids = h.halo_ids[h.halo_off[t]:h.halo_off[t + 1]]
        reads_ghost[t] = bool((ids >= part.N).any())
    cls = np.zeros(h.ntiles, int)                       # 0 = C, 1 = B, 2 = A
    cls[reads_ghost] = 2
    for t in np.flatnonzero(~reads_ghost):
        ids = h.halo_ids[h.halo_off[t]:h.halo_off[t + 1]]
        if reads_ghost[owner[ids]].any():
            cls[t] = 1
    order = h.tile_order
    assert sorted(order.tolist()) == list(range(h.ntiles))
    assert (cls[order[: h.n_deep]] == 0).all() and (cls[order[h.n_deep: h.n_interior]] == 1).all() and (cls[order[h.n_interior:]] == 2).all()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        kind = rng.choice(["plain2", "plain3", "prism", "sub2", "sub3"])
        world = int(rng.integers(2, 10))
        steps = int(rng.integers(1, 4))
        seed = int(rng.integers(1 << 30))
        sub = kind.startswith("sub")
        if kind == "prism":
            n3 = tuple(int(x) for x in rng.choice([4, 8, 16], 3))
            periodic = bool(rng.random() < 0.5)
            mesh = PrismHexMesh(n3, split=float(rng.choice([0.0, 0.4, 1.0])), mapping=wavy_map if periodic else shell_map, periodic=periodic, seed=seed)
            whole, parts = mesh.partition(), [mesh.partition(r, world) for r in range(world)]
            dt = 0.03 * float(np.cbrt(whole.volumes.min()))
            desc = f"prism {n3} periodic {periodic}"
        else:
            dim = 2 if kind.endswith("2") else 3
            base = int(rng.integers(2, 7 if dim == 2 else 5))            # (round 3: large enough for structured patches)
            lmax = min(base + int(rng.integers(0, 3 if dim == 2 else 2)), 8 if dim == 2 else 5)
            if sub:
                base, lmax = min(base, 4 if dim == 2 else 3), min(lmax, 5 if dim == 2 else 3)
            mesh = SynthMesh(dim, base, lmax, band=float(rng.choice([0.0, 0.03, 0.08, 0.2])), periodic=bool(rng.random() < 0.5))
            if mesh.num_elements < world:
                continue
            whole, parts = mesh.partition(subgrid=sub), [mesh.partition(r, world, subgrid=sub) for r in range(world)]
            dt = 0.1 * 2.0 ** -(mesh.finest_level + (2 if sub else 0))
            desc = f"{kind} base {base} max {lmax}"
        S = whole.cells_per_element
        st = perturbed_state(whole, seed, S)
        opts = dict(tmax=int(rng.choice([8, 32, 256])), fcap=int(rng.choice([30, 100, 512])), patches=bool(rng.random() < 0.7))
        Solver = SubgridSolver if sub else PlainSolver
        ref = Solver(whole, torch.float64, mode="fused", state=st, **({} if sub else dict(plan_options=dict(patches=False))))
        solvers, halos = [], []
        for p in parts:
            gidx = np.concatenate([p.first_global + np.arange(p.N), p.ghost_global])
            cellsidx = (gidx[:, None] * S + np.arange(S)[None, :]).reshape(-1)
            local = st[:, cellsidx].copy()
            local[:, p.N * S:] = np.nan
            s = Solver(p, torch.float64, mode="fused", state=local)
            if not sub:
                s.plan = fused.PlainPlan(p, torch.float64, **opts)
                check_classes(s.plan, p)
            solvers.append(s)
            halos.append(HaloExchange(p, torch.float64, dist=None, overlap=False))
        for _ in range(steps):
            ref.iterate(dt)
            for s in solvers:
                s.begin_step()
            for k in range(3):
                for s, h in zip(solvers, halos):
                    h._pack(s.step_planes(s.stage_steps(k)[0]))
                loopback(halos)
                for s, h in zip(solvers, halos):
                    h._unpack(s.step_planes(s.stage_steps(k)[0]))
                for s in solvers:
                    s.run_stage(k, dt, split=True)
        torch.cuda.synchronize()
        got = torch.cat([s.state() for s in solvers], dim=1)
        n += 1
        line = f"[{n:4d}] {desc:34s} N={whole.N:7d} world {world} {opts if not sub else ''} steps {steps}"
        if not torch.equal(got, ref.state()):
            print("VIOLATION " + line, "max abs diff", float((got - ref.state()).abs().max()), flush=True)
            sys.exit(1)
        if n % 20 == 0:
            print(line, flush=True)
    print(f"{n} partitioned cases in {time.time() - t0:.0f} s: all bitwise equal to the single-rank run")


if __name__ == "__main__":
    main()
