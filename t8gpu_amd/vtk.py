"""VTK output and host read-back: mirrors of
  MeshManager::get_host_scalar_variable / get_host_vector_variable / save_variables_to_vtk
      t8gpu/mesh/mesh_manager.inl:516-623
  SubgridMeshManager::save_variable_to_vtk / save_mesh_to_vtk / get_host_*_variable
      t8gpu/mesh/subgrid_mesh_manager.inl:1051-1206
The device side (cast to double, xyz interleave, column-major -> z-order) runs in
csrc/hip/kernels_readback.hip; the file is written by csrc/host/vtk_writer.cpp (a .vtu piece per rank and
a .pvtu from rank 0), which stands where the reference calls t8_forest_write_vtk_ext.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import hip, synth

SCALAR, VECTOR = 1, 3   # T8_VTK_SCALAR / T8_VTK_VECTOR: components per cell


@dataclass
class HostVariableInfo:
    """mesh_manager.h: HostVariableInfo -- a named host array of doubles ready for the writer."""
    type: int
    data: np.ndarray
    name: str


def _host_lib():
    lib = synth.lib()
    lib.t8gpu_host_write_vtu.restype = C.c_int
    lib.t8gpu_host_write_vtu.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.t8gpu_host_write_pvtu.restype = C.c_int
    lib.t8gpu_host_write_pvtu.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def _own_cells(solver):
    return solver.N * getattr(solver, "S", 1)


def _variable(solver, step, variable):
    return solver.step_planes(step)[variable][: _own_cells(solver)]


def get_host_scalar_variable(solver, step, variable, name):
    """One variable of one step as doubles on the host (owned cells, storage order)."""
    n = _own_cells(solver)
    v = _variable(solver, step, variable)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    hip.call("t8gpu_hip_host_scalar_variable", v.dtype, C.c_size_t(n), hip.ptr(v), hip.ptr(out), hip.stream_ptr())
    return HostVariableInfo(SCALAR, out.cpu().numpy(), name)


def get_host_vector_variable(solver, step, variables, name):
    """Three variables as interleaved xyz doubles on the host."""
    n = _own_cells(solver)
    v = [_variable(solver, step, k) for k in variables]
    out = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    hip.call("t8gpu_hip_host_vector_variable", v[0].dtype, C.c_size_t(n), hip.ptr(v[0]), hip.ptr(v[1]), hip.ptr(v[2]), hip.ptr(out),
             hip.stream_ptr())
    return HostVariableInfo(VECTOR, out.cpu().numpy().reshape(n, 3), name)


def column_major_to_z_order(solver, values):
    """Per-cell device array of a Subgrid solver, storage order -> z-order of the twice-refined blocks."""
    out = torch.empty_like(values)
    hip.call("t8gpu_hip_column_major_to_z_order", values.dtype, C.c_int(solver.rank), C.c_int(solver.N), hip.ptr(values), hip.ptr(out),
             hip.stream_ptr())
    return out


def _write(solver, fields, prefix, dist, ascii):
    part = solver.part
    rank, nranks = part.rank, part.nranks
    piece = f"{prefix}_{rank:04d}.vtu" if nranks > 1 else f"{prefix}.vtu"
    names = (C.c_char_p * len(fields))(*[f.name.encode() for f in fields])
    comps = np.array([f.type for f in fields], np.int32)
    arrays = [np.ascontiguousarray(f.data, np.float64) for f in fields]
    ptrs = (C.c_void_p * len(fields))(*[a.ctypes.data for a in arrays])
    centres = np.ascontiguousarray(part.centres[: part.N])
    levels = np.ascontiguousarray(part.levels[: part.N])
    rc = _host_lib().t8gpu_host_write_vtu(piece.encode(), part.mesh.dim, part.N, centres.ctypes.data, levels.ctypes.data,
                                          4 if part.subgrid else 1, rank, part.first_global, len(fields), names,
                                          comps.ctypes.data, ptrs, int(ascii))
    if rc != 0:
        raise OSError(f"t8gpu_host_write_vtu({piece}) failed with code {rc}")
    if nranks > 1:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            files = (C.c_char_p * nranks)(*[os.path.basename(f"{prefix}_{r:04d}.vtu").encode() for r in range(nranks)])
            rc = _host_lib().t8gpu_host_write_pvtu(f"{prefix}.pvtu".encode(), nranks, files, len(fields), names, comps.ctypes.data)
            if rc != 0:
                raise OSError(f"t8gpu_host_write_pvtu({prefix}.pvtu) failed with code {rc}")
    return piece


def save_variables_to_vtk(solver, host_variables, prefix, dist=None, ascii=False):
    """MeshManager::save_variables_to_vtk (mesh_manager.inl:588-623). For a Subgrid solver the arrays must
    already be in z-order (see save_variable_to_vtk)."""
    return _write(solver, list(host_variables), prefix, dist, ascii)


def save_variable_to_vtk(solver, step, variable, prefix, dist=None, ascii=False):
    """SubgridMeshManager::save_variable_to_vtk (subgrid_mesh_manager.inl:1051-1138): the variable on the
    twice-refined forest, field name "variables"."""
    z = column_major_to_z_order(solver, _variable(solver, step, variable).contiguous())
    out = torch.empty(z.numel(), dtype=torch.float64, device="cuda")
    hip.call("t8gpu_hip_host_scalar_variable", z.dtype, C.c_size_t(z.numel()), hip.ptr(z), hip.ptr(out), hip.stream_ptr())
    return _write(solver, [HostVariableInfo(SCALAR, out.cpu().numpy(), "variables")], prefix, dist, ascii)


def save_mesh_to_vtk(solver, prefix, dist=None, ascii=False):
    """SubgridMeshManager::save_mesh_to_vtk (subgrid_mesh_manager.inl:1185-1206): the forest, no data."""
    part = solver.part
    centres = np.ascontiguousarray(part.centres[: part.N])
    levels = np.ascontiguousarray(part.levels[: part.N])
    piece = f"{prefix}_{part.rank:04d}.vtu" if part.nranks > 1 else f"{prefix}.vtu"
    rc = _host_lib().t8gpu_host_write_vtu(piece.encode(), part.mesh.dim, part.N, centres.ctypes.data, levels.ctypes.data, 1, part.rank,
                                          part.first_global, 0, None, None, None, int(ascii))
    if rc != 0:
        raise OSError(f"t8gpu_host_write_vtu({piece}) failed with code {rc}")
    return piece
