"""N > 1 path on CPU: world_size 2 and 3 over gloo. Each rank owns an SFC-contiguous share with ghost
mirror slots, refreshes them through t8gpu_amd.halo.HaloExchange (the same class the GPU run uses,
there with RCCL + HIP pack kernels) and advances with the oracle's stage functions; the gathered
result must equal the single-rank oracle run on the whole mesh."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MESH = dict(dim=2, base_level=3, max_level=6, band=0.06)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, out_path):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import _oracle as O
    from _gpu import perturbed_state
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.synth import SynthMesh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mesh = SynthMesh(**MESH)
    whole = mesh.partition()
    glob_state = perturbed_state(whole, 33)
    part = mesh.partition(rank, world)
    gidx = np.concatenate([part.first_global + np.arange(part.N), part.ghost_global])
    st = glob_state[:, gidx].copy()
    st[:, part.N:] = np.nan                                   # ghosts must come from the exchange, not from the IC
    case = O.PlainCase(part, np.float64, state=st)
    planes = torch.from_numpy(case.planes)                    # shares memory with the oracle's arrays
    halo = HaloExchange(part, torch.float64, dist, device="cpu")
    dt = 0.1 * 2.0 ** -mesh.finest_level
    lib = O.lib()
    for _ in range(STEPS):
        case.next, case.prev = case.prev, case.next
        srcs, dsts = (case.prev, 1, 2), (1, 2, case.next)
        for k in range(3):
            src = planes[5 * srcs[k]:5 * srcs[k] + 5]
            halo.exchange(src)
            lib.oracle_plain_interior_faces_f64(0, part.F, 3, O.p(case.fn), O.p(part.indices), O.p(case.normals), O.p(case.areas),
                                                O.p(case.planes[5 * srcs[k]:]), O.p(case.planes[20:25]), C.c_size_t(case.stride), O.p(case.speed))
            lib.oracle_plain_rk_stage_f64(k + 1, part.N, O.p(case.planes[5 * case.prev:]), O.p(case.planes[5 * srcs[k]:]),
                                          O.p(case.planes[5 * dsts[k]:]), O.p(case.planes[20:25]), C.c_size_t(case.stride),
                                          O.p(case.planes[25]), C.c_double(dt))
    mine = case.current()[:, :part.N].copy()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((part.first_global, mine), gathered, dst=0)
    if rank == 0:
        full = np.zeros((5, mesh.num_elements))
        for first, arr in gathered:
            full[:, first:first + arr.shape[1]] = arr
        ref = O.PlainCase(whole, np.float64, state=glob_state)
        for _ in range(STEPS):
            ref.iterate(dt)
        err = np.abs(full - ref.current()).max() / np.abs(ref.current()).max()
        np.save(out_path, np.array([err, float(np.isnan(full).sum())]))
    dist.barrier()
    dist.destroy_process_group()


def _subgrid_rank_main(rank, world, port, out_path):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import _oracle as O
    from _gpu import perturbed_state
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.synth import SynthMesh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mesh = SynthMesh(2, 3, 5, band=0.03)
    S = 16
    whole = mesh.partition(subgrid=True)
    glob_state = perturbed_state(whole, 34)
    part = mesh.partition(rank, world, subgrid=True)
    blocks = np.concatenate([part.first_global + np.arange(part.N), part.ghost_global])
    cells = (blocks[:, None] * S + np.arange(S)[None, :]).reshape(-1)
    st = glob_state[:, cells].copy()
    st[:, part.N * S:] = np.nan
    case = O.SubgridCase(part, np.float64, state=st)
    planes = torch.from_numpy(case.planes)
    halo = HaloExchange(part, torch.float64, dist, device="cpu")
    dt = 0.1 * 2.0 ** -(mesh.finest_level + 2)
    lib = O.lib()
    for _ in range(2):
        case.prev, case.next = case.next, case.prev
        srcs, dsts = (case.prev, 1, 2), (1, 2, case.next)
        for k in range(3):
            halo.exchange(planes[5 * srcs[k]:5 * srcs[k] + 5])
            st5, fl5 = case.planes[5 * srcs[k]:], case.planes[20:25]
            lib.oracle_subgrid_inner_f64(0, 2, part.N, O.p(st5), O.p(fl5), C.c_size_t(case.stride), O.p(case.volumes))
            lib.oracle_subgrid_outer_f64(0, 2, part.F, O.p(case.fn), O.p(part.indices), O.p(part.level_diff), O.p(part.nb_offset),
                                         O.p(case.normals), O.p(case.areas), O.p(st5), O.p(fl5), C.c_size_t(case.stride))
            lib.oracle_subgrid_rk_stage_f64(k + 1, 2, part.N, O.p(case.planes[5 * case.prev:]), O.p(st5), O.p(case.planes[5 * dsts[k]:]),
                                            O.p(fl5), C.c_size_t(case.stride), O.p(case.volumes), C.c_double(dt))
    mine = case.current()[:, :part.N * S].copy()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((part.first_global, mine), gathered, dst=0)
    if rank == 0:
        full = np.zeros((5, mesh.num_elements * S))
        for first, arr in gathered:
            full[:, first * S:first * S + arr.shape[1]] = arr
        ref = O.SubgridCase(whole, np.float64, state=glob_state)
        for _ in range(2):
            ref.iterate(dt)
        err = np.abs(full - ref.current()).max() / np.abs(ref.current()).max()
        np.save(out_path, np.array([err, float(np.isnan(full).sum())]))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_subgrid_run_equals_single_rank(tmp_path):
    """Ghost blocks travel whole (Subgrid::size cells each) through the same HaloExchange."""
    out = str(tmp_path / "err.npy")
    mp.spawn(_subgrid_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    err, nans = np.load(out)
    assert nans == 0 and err < 1e-13


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_run_equals_single_rank(world, tmp_path):
    out = str(tmp_path / "err.npy")
    mp.spawn(_rank_main, args=(world, _free_port(), out), nprocs=world, join=True)
    err, nans = np.load(out)
    assert nans == 0
    assert err < 1e-13        # same fluxes, same orientation; only the summation order differs


def test_halo_exchange_single_rank_is_a_noop():
    sys.path.insert(0, ROOT)
    from t8gpu_amd.halo import HaloExchange
    from t8gpu_amd.synth import SynthMesh
    part = SynthMesh(2, 3, 4, band=0.1).partition()
    h = HaloExchange(part, torch.float64, dist, device="cpu")
    x = torch.zeros(5, part.N)
    h.exchange(x)
    assert h.peers == [] and not x.any()
