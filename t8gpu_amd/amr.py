"""Adaptation of a running plain-element solver: CompressibleEulerSolver::adapt
(examples/compressible_euler/solver.cu:243-277) + MeshManager::adapt (t8gpu/mesh/mesh_manager.inl:196-330).

Indicator and data transfer are HIP kernels behind the C-ABI (estimate_gradient, refinement criteria,
adapt_variables_and_volume); the forest operations (adapt callback, refine / coarsen, 2:1 balance, the
old->new correspondence) run on the host in the mesh provider, where the reference calls t8code.
"""
import ctypes as C

import numpy as np
import torch

from . import hip
from .solver import FLUXES, PlainSolver


def _inherited_plan_options(solver):
    """Tile caps and patch forms of the adapted mesh's plan: those the previous plan settled on (fused.PlainPlan may build a
    plan twice to find out which kernel a mesh class gets; an adaptive run should pay for that once, not at every adapt)."""
    plan = getattr(solver, "plan", None)
    if plan is None:
        return None
    opts = {}
    if getattr(plan, "auto_fcap", None) is not None:
        opts["fcap"] = plan.auto_fcap
        opts["fcap_elements"] = getattr(plan, "auto_fcap_elements", None)     # (the mesh size the cap was chosen for)
    if getattr(plan, "auto_irregular", None) is not None:
        opts["irregular"] = plan.auto_irregular
        # (like the cap, the patch form is decided again once the mesh size has changed by more than a factor two since it was
        #  chosen: fused.PlainPlan drops both under that rule)
        opts["fcap_elements"] = getattr(plan, "auto_fcap_elements", None)
    return opts or None


def refinement_criteria(solver):
    """estimate_gradient + compute_refinement_criteria (solver.cu:245-263); returns a device tensor [N]."""
    s = hip.stream_ptr()
    grad = solver.planes[5 * FLUXES]          # the reference reuses the Fluxes/Rho plane (solver.cu:249)
    rho = solver.planes[5 * solver.next]
    hip.call("t8gpu_hip_estimate_gradient", solver.dtype, solver.F, hip.ptr(solver.fn), None, hip.ptr(rho), hip.ptr(grad), s)
    crit = torch.empty(solver.N, dtype=solver.dtype, device="cuda")
    hip.call("t8gpu_hip_refinement_criteria", solver.dtype, solver.N, hip.ptr(grad), hip.ptr(solver.planes[25]), hip.ptr(crit), s)
    grad[:solver.N + solver.G].zero_()        # solver.cu:269-270
    return crit


def adapt(solver, threshold=10.0, min_level=1, max_level=4, family_members_averaged=4, volume_dim=None):
    """Refine / coarsen the mesh of a single-rank PlainSolver by the reference's criterion and transfer the
    current solution. Returns the new solver (same dtype, flux, kernel tier, step bookkeeping)."""
    import time
    part = solver.part
    assert part.nranks == 1, "multi-rank adaptation goes through amr.adapt_partitioned"
    mesh = part.mesh
    t0 = time.perf_counter()
    crit = refinement_criteria(solver).double().cpu().numpy()
    t1 = time.perf_counter()
    marks = mesh.marks_from_criteria(crit, threshold, min_level, max_level, family_members_averaged)
    new_mesh, adapt_data = mesh.adapt(marks)
    new_part = new_mesh.partition(0, 1, subgrid=False, normal_dim=part.normal_dim)
    t2 = time.perf_counter()
    dim = mesh.dim if volume_dim is None else volume_dim
    new = PlainSolver(new_part, solver.dtype, flux_kind=solver.kind, mode=solver.mode,
                      state="zeros", plan_options=_inherited_plan_options(solver))
    t3 = time.perf_counter()
    new.next, new.prev = solver.next, solver.prev
    ad = torch.from_numpy(adapt_data).cuda()
    hip.call("t8gpu_hip_adapt_variables_and_volume", solver.dtype, new_part.N, dim, hip.ptr(ad),
             solver.get_own_variables(solver.next), new.get_own_variables(new.next), hip.ptr(solver.planes[25]),
             hip.ptr(new.planes[25]), hip.stream_ptr())
    torch.cuda.synchronize()
    if getattr(solver, "stepper", None) is not None:
        new.use_native_stepper()
    t4 = time.perf_counter()
    new.last_adapt_criteria_sum = float(crit.sum())
    # where the cycle went (seconds): the indicator kernels + read-back, the mesh provider (forest adapt + balance,
    # partition + connectivity: t8code's share in the reference), this backend's tile plan + upload + new planes, the
    # transfer kernel + step driver
    plan_s = getattr(new, "plan_build_s", 0.0)
    new.last_adapt_split = {"indicator": t1 - t0, "provider": t2 - t1, "plan": plan_s, "planes_upload": (t3 - t2) - plan_s,
                            "transfer": t4 - t3}
    return new, marks, adapt_data


class PartitionedAdapt:
    """adapt() + partition() for an SFC-partitioned plain-element run (MeshManager::adapt followed by
    MeshManager::partition, t8gpu/mesh/mesh_manager.inl:196-330, 645-723), one rank per GPU.

    Every rank holds the (cheap) forest description, so the forest operations are replicated and only the
    element payloads travel: each rank adapts its own elements on the device, then ships contiguous runs of
    the adapted elements (5 variables + volume, one message per destination) to their owners in the new
    equal split -- the reference instead lets the new owner PULL through CUDA-IPC pointers
    (partition_data<<<>>>, mesh_manager.inl:626-643). Split in prepare / transport / finish so that the
    transport can be RCCL (torch.distributed P2P), gloo (CPU tests) or a loopback (one-GPU tests).
    """

    def __init__(self, solver, all_criteria, threshold=10.0, min_level=1, max_level=4, family_members_averaged=4):
        part = solver.part
        self.solver, self.rank, self.world = solver, part.rank, part.nranks
        mesh = part.mesh
        old_off = mesh.partition_offsets(self.world)
        marks = mesh.marks_from_criteria(all_criteria, threshold, min_level, max_level, family_members_averaged)
        marks = mesh.unmark_split_families(marks, old_off[1:-1])
        self.marks = marks
        self.new_mesh, adapt_data = mesh.adapt(marks)
        n_new = self.new_mesh.num_elements
        # new elements made from rank p's old elements: [have_off[p], have_off[p+1])
        self.have_off = np.searchsorted(adapt_data[:-1], old_off, side="left").astype(np.int64)
        self.have_off[-1] = n_new
        self.new_off = self.new_mesh.partition_offsets(self.world)
        a, b = int(self.have_off[self.rank]), int(self.have_off[self.rank + 1])
        self.n_have = b - a
        dtype, dev = solver.dtype, solver.planes.device
        # 1. local data transfer (adapt_variables_and_volume) into 6 temporary planes
        self.tmp = torch.zeros((6, max(1, self.n_have)), dtype=dtype, device=dev)
        ad_local = torch.from_numpy((adapt_data[a:b + 1] - old_off[self.rank]).astype(np.int32)).to(dev)
        if self.n_have:
            if dev.type == "cuda":
                hip.call("t8gpu_hip_adapt_variables_and_volume", dtype, self.n_have, mesh.dim, hip.ptr(ad_local),
                         solver.get_own_variables(solver.next), hip.vars_of(self.tmp), hip.ptr(solver.planes[25]),
                         hip.ptr(self.tmp[5]), hip.stream_ptr())
            else:
                raise hip.T8gpuHipError("the data transfer kernel needs a GPU")
        # 2. the new partition and an empty solver for it
        self.new_part = self.new_mesh.partition(self.rank, self.world, subgrid=False, normal_dim=part.normal_dim)
        self.new_solver = PlainSolver(self.new_part, dtype, flux_kind=solver.kind, mode=solver.mode,
                                      state="zeros",
                                      plan_options=_inherited_plan_options(solver))
        self.new_solver.next, self.new_solver.prev = solver.next, solver.prev
        # 3. message plan: intersections of what I have with what every rank will own (and vice versa)
        self.sends, self.recvs = [], []
        lo_r, hi_r = int(self.new_off[self.rank]), int(self.new_off[self.rank + 1])
        for q in range(self.world):
            s0, s1 = max(a, int(self.new_off[q])), min(b, int(self.new_off[q + 1]))
            if s1 > s0:
                self.sends.append((q, s0 - a, s1 - s0))                  # (peer, first in tmp, count)
            r0, r1 = max(int(self.have_off[q]), lo_r), min(int(self.have_off[q + 1]), hi_r)
            if r1 > r0:
                self.recvs.append((q, r0 - lo_r, r1 - r0))               # (peer, first in the new planes, count)
        self.sendbufs = {q: torch.empty(6 * n, dtype=dtype, device=dev) for q, _, n in self.sends if q != self.rank}
        self.recvbufs = {q: torch.empty(6 * n, dtype=dtype, device=dev) for q, _, n in self.recvs if q != self.rank}
        s = hip.stream_ptr()
        for q, first, n in self.sends:
            if q != self.rank:
                hip.call("t8gpu_hip_gather_elements", dtype, n, first, hip.vars_of(self.tmp), hip.ptr(self.tmp[5]),
                         hip.ptr(self.sendbufs[q]), s)

    def transport(self, dist, host_staged=False):
        """host_staged: move the messages through host memory (gloo, which cannot send device buffers)."""
        torch.cuda.synchronize()
        rbuf = {q: (b.cpu() if host_staged else b) for q, b in self.recvbufs.items()}
        sbuf = {q: (b.cpu() if host_staged else b) for q, b in self.sendbufs.items()}
        ops = []
        for q, _, n in self.recvs:
            if q != self.rank:
                ops.append(dist.P2POp(dist.irecv, rbuf[q], q))
        for q, _, n in self.sends:
            if q != self.rank:
                ops.append(dist.P2POp(dist.isend, sbuf[q], q))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if host_staged:
            for q, b in self.recvbufs.items():
                b.copy_(rbuf[q])

    def finish(self):
        new, dtype = self.new_solver, self.solver.dtype
        s = hip.stream_ptr()
        nv = new.get_own_variables(new.next)
        for q, first, n in self.recvs:
            if q == self.rank:       # stays here: straight from the temporary planes
                src_first = [f for p, f, m in self.sends if p == self.rank][0]
                new.planes[5 * new.next:5 * new.next + 5, first:first + n] = self.tmp[0:5, src_first:src_first + n]
                new.planes[25, first:first + n] = self.tmp[5, src_first:src_first + n]
            else:
                hip.call("t8gpu_hip_scatter_elements", dtype, n, first, hip.ptr(self.recvbufs[q]), nv, hip.ptr(new.planes[25]), s)
        torch.cuda.synchronize()
        return new


def _gather_criteria(crit, solver, dist, host_staged):
    world = solver.part.nranks
    off = solver.part.mesh.partition_offsets(world)
    crit = crit.cpu() if host_staged else crit
    sizes = [int(off[r + 1] - off[r]) for r in range(world)]
    width = max(sizes)                       # all_gather wants equal contributions: pad to the largest share
    mine = torch.zeros(width, dtype=torch.float64, device=crit.device)
    mine[: crit.numel()] = crit
    chunks = [torch.empty(width, dtype=torch.float64, device=crit.device) for _ in range(world)]
    dist.all_gather(chunks, mine)
    return torch.cat([c[:n] for c, n in zip(chunks, sizes)]).cpu().numpy()


def adapt_partitioned(solver, dist, host_staged=False, halo=None, **kw):
    """Collective: every rank calls it with its own solver; returns the rank's new solver.
    host_staged: collectives and messages through host memory (gloo).

    The indicator (estimate_gradient, solver.cu:245-263) reads rho of GHOST elements of the `next` state, and neither
    step driver refreshes those slots -- a stage only exchanges the ghosts of its SOURCE state, so after a step the
    last stage's output carries ghost values two stages old. The exchange is therefore part of this function (through
    `halo`, the run's HaloExchange of this partition, or a temporary one): refinement marks must not depend on the
    number of ranks."""
    if solver.part.nranks > 1:
        if halo is None:
            from .halo import HaloExchange
            halo = HaloExchange(solver.part, solver.dtype, dist, overlap=False, stage_through_host=host_staged)
        halo.exchange(solver.step_planes(solver.next))
    all_crit = _gather_criteria(refinement_criteria(solver).double(), solver, dist, host_staged)
    pa = PartitionedAdapt(solver, all_crit, **kw)
    pa.transport(dist, host_staged)
    new = pa.finish()
    new.last_adapt_criteria_sum = float(all_crit.sum())      # diagnostic: equal (to rounding) for every rank count
    return new


def subgrid_refinement_criteria(solver):
    """compute_refinement_criteria<Subgrid> (examples/subgrid/solver.inl:329-340); device tensor [N]."""
    crit = torch.empty(solver.N, dtype=solver.dtype, device="cuda")
    hip.call("t8gpu_hip_subgrid_refinement_criteria", solver.dtype, solver.rank, solver.N, hip.ptr(solver.planes[5 * solver.next]),
             hip.ptr(solver.volumes), hip.ptr(crit), hip.stream_ptr())
    return crit


def adapt_subgrid(solver, threshold=0.02, min_level=1, max_level=6, family_members_averaged=4):
    """SubgridCompressibleEulerSolver::adapt (examples/subgrid/solver.inl:327-345) for a single-rank SubgridSolver:
    H1-seminorm indicator, the forest adapt, block-wise data transfer (subgrid_mesh_manager.inl:246-425)."""
    from .solver import SubgridSolver
    part = solver.part
    assert part.nranks == 1
    mesh = part.mesh
    crit = subgrid_refinement_criteria(solver).double().cpu().numpy()
    marks = mesh.marks_from_criteria(crit, threshold, min_level, max_level, family_members_averaged)
    new_mesh, adapt_data = mesh.adapt(marks)
    new_part = new_mesh.partition(0, 1, subgrid=True)
    S = solver.S
    new = SubgridSolver(new_part, solver.dtype, flux_kind=solver.kind, mode=solver.mode, state=np.zeros((5, new_part.N * S)))
    new.next, new.prev = solver.next, solver.prev
    ad = torch.from_numpy(adapt_data).cuda()
    hip.call("t8gpu_hip_subgrid_adapt_variables_and_volume", solver.dtype, solver.rank, new_part.N, hip.ptr(ad),
             solver.get_own_variables(solver.next), new.get_own_variables(new.next), hip.ptr(solver.volumes),
             hip.ptr(new.volumes), hip.stream_ptr())
    torch.cuda.synchronize()
    return new, marks, adapt_data


class PartitionedSubgridAdapt:
    """adapt() + partition() for an SFC-partitioned Subgrid run (SubgridMeshManager::adapt followed by
    SubgridMeshManager::partition, t8gpu/mesh/subgrid_mesh_manager.inl:428-558, 1217-1369), one rank per GPU.

    Same scheme as PartitionedAdapt: the forest operations are replicated, each rank transfers its own blocks
    on the device (block-wise injection / mean, subgrid_mesh_manager.inl:246-425) and ships contiguous runs of
    adapted blocks -- 5 x 4^rank values + one volume per block, one message per destination -- to their
    owners in the new equal split. A run of blocks is contiguous in every variable plane, so a message is
    six slices; no gather kernel is needed."""

    def __init__(self, solver, all_criteria, threshold=0.02, min_level=1, max_level=6, family_members_averaged=4):
        from .solver import SubgridSolver
        part = solver.part
        self.solver, self.rank, self.world = solver, part.rank, part.nranks
        mesh, S = part.mesh, solver.S
        self.S = S
        old_off = mesh.partition_offsets(self.world)
        marks = mesh.marks_from_criteria(all_criteria, threshold, min_level, max_level, family_members_averaged)
        self.marks = mesh.unmark_split_families(marks, old_off[1:-1])
        self.new_mesh, adapt_data = mesh.adapt(self.marks)
        n_new = self.new_mesh.num_elements
        self.have_off = np.searchsorted(adapt_data[:-1], old_off, side="left").astype(np.int64)
        self.have_off[-1] = n_new
        self.new_off = self.new_mesh.partition_offsets(self.world)
        a, b = int(self.have_off[self.rank]), int(self.have_off[self.rank + 1])
        self.n_have = b - a
        dtype, dev = solver.dtype, solver.planes.device
        self.tmp = torch.zeros((5, max(1, self.n_have) * S), dtype=dtype, device=dev)
        self.tmp_vol = torch.zeros(max(1, self.n_have), dtype=dtype, device=dev)
        if self.n_have:
            ad_local = torch.from_numpy((adapt_data[a:b + 1] - old_off[self.rank]).astype(np.int32)).to(dev)
            hip.call("t8gpu_hip_subgrid_adapt_variables_and_volume", dtype, solver.rank, self.n_have, hip.ptr(ad_local),
                     solver.get_own_variables(solver.next), hip.vars_of(self.tmp), hip.ptr(solver.volumes), hip.ptr(self.tmp_vol),
                     hip.stream_ptr())
        self.new_part = self.new_mesh.partition(self.rank, self.world, subgrid=True)
        tot = self.new_part.N + self.new_part.G
        self.new_solver = SubgridSolver(self.new_part, dtype, flux_kind=solver.kind, mode=solver.mode, state=np.zeros((5, tot * S)))
        self.new_solver.next, self.new_solver.prev = solver.next, solver.prev
        self.sends, self.recvs = [], []
        lo_r, hi_r = int(self.new_off[self.rank]), int(self.new_off[self.rank + 1])
        for q in range(self.world):
            s0, s1 = max(a, int(self.new_off[q])), min(b, int(self.new_off[q + 1]))
            if s1 > s0:
                self.sends.append((q, s0 - a, s1 - s0))                  # (peer, first block in tmp, count)
            r0, r1 = max(int(self.have_off[q]), lo_r), min(int(self.have_off[q + 1]), hi_r)
            if r1 > r0:
                self.recvs.append((q, r0 - lo_r, r1 - r0))               # (peer, first block in the new planes, count)
        w = 5 * S + 1
        self.sendbufs, self.recvbufs = {}, {}
        for q, first, n in self.sends:
            if q != self.rank:
                buf = torch.empty(w * n, dtype=dtype, device=dev)
                buf[: 5 * S * n].view(5, n * S).copy_(self.tmp[:, first * S:(first + n) * S])
                buf[5 * S * n:].copy_(self.tmp_vol[first:first + n])
                self.sendbufs[q] = buf
        for q, _, n in self.recvs:
            if q != self.rank:
                self.recvbufs[q] = torch.empty(w * n, dtype=dtype, device=dev)

    transport = PartitionedAdapt.transport

    def finish(self):
        new, S = self.new_solver, self.S
        dst = new.planes[5 * new.next:5 * new.next + 5]
        for q, first, n in self.recvs:
            if q == self.rank:
                src_first = [f for p, f, m in self.sends if p == self.rank][0]
                dst[:, first * S:(first + n) * S] = self.tmp[:, src_first * S:(src_first + n) * S]
                new.volumes[first:first + n] = self.tmp_vol[src_first:src_first + n]
            else:
                buf = self.recvbufs[q]
                dst[:, first * S:(first + n) * S] = buf[: 5 * S * n].view(5, n * S)
                new.volumes[first:first + n] = buf[5 * S * n:]
        torch.cuda.synchronize()
        return new


def adapt_subgrid_partitioned(solver, dist, host_staged=False, **kw):
    """Collective: every rank calls it with its own SubgridSolver; returns the rank's new solver."""
    all_crit = _gather_criteria(subgrid_refinement_criteria(solver).double(), solver, dist, host_staged)
    pa = PartitionedSubgridAdapt(solver, all_crit, **kw)
    pa.transport(dist, host_staged)
    return pa.finish()
