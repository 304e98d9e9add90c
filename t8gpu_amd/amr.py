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
    part = solver.part
    assert part.nranks == 1, "multi-rank adaptation goes through amr.adapt_partitioned"
    mesh = part.mesh
    crit = refinement_criteria(solver).double().cpu().numpy()
    marks = mesh.marks_from_criteria(crit, threshold, min_level, max_level, family_members_averaged)
    new_mesh, adapt_data = mesh.adapt(marks)
    new_part = new_mesh.partition(0, 1, subgrid=False, normal_dim=part.normal_dim)
    dim = mesh.dim if volume_dim is None else volume_dim
    new = PlainSolver(new_part, solver.dtype, flux_kind=solver.kind, mode=solver.mode,
                      state=np.zeros((5, new_part.N + new_part.G)))
    new.next, new.prev = solver.next, solver.prev
    ad = torch.from_numpy(adapt_data).cuda()
    hip.call("t8gpu_hip_adapt_variables_and_volume", solver.dtype, new_part.N, dim, hip.ptr(ad),
             solver.get_own_variables(solver.next), new.get_own_variables(new.next), hip.ptr(solver.planes[25]),
             hip.ptr(new.planes[25]), hip.stream_ptr())
    torch.cuda.synchronize()
    if getattr(solver, "stepper", None) is not None:
        new.use_native_stepper()
    return new, marks, adapt_data
