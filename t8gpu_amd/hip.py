"""ctypes mirror of include/t8gpu_hip.h. The HIP library is mandatory: import fails loudly without it."""
import ctypes as C
import os

import numpy as np

from . import build as _build

ABI_VERSION = 8   # t8gpu_hip_abi_version() of include/t8gpu_hip.h: the layout of the plan structs mirrored in fused.py
KEPES, HLL, HLLC = 0, 1, 2   # HLLC is an addition: the reference has none


class Vars32(C.Structure):
    _fields_ = [("p", C.c_void_p * 5)]


class Vars64(C.Structure):
    _fields_ = [("p", C.c_void_p * 5)]


class T8gpuHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64 / librccl (same SONAMEs as /opt/rocm): load torch FIRST so that
        # this library binds to the runtime that owns torch's streams and allocations.
        import torch  # noqa: F401
        path = os.environ.get("T8GPU_HIP_LIB", _build.HIP_LIB)   # override: A/B runs of two builds
        if not os.path.exists(path):
            raise T8gpuHipError(f"{path} is missing: run `python -m t8gpu_amd.build` (hipcc --offload-arch=gfx950). "
                                "There is no CPU fallback for the hot path.")
        _lib = C.CDLL(path)
        _lib.t8gpu_hip_abi_version.restype = C.c_int
        if _lib.t8gpu_hip_abi_version() != ABI_VERSION:   # a stale build would misread the plan structs
            got, _lib = _lib.t8gpu_hip_abi_version(), None
            raise T8gpuHipError(f"{path} has ABI version {got}, this package needs {ABI_VERSION}: rebuild it "
                                "(`python -m t8gpu_amd.build`)")
        _lib.t8gpu_hip_error_string.restype = C.c_char_p
        _lib.t8gpu_hip_error_string.argtypes = [C.c_int]
    return _lib


def check(code):
    if code != 0:
        raise T8gpuHipError(f"t8gpu_hip call failed: {code} ({lib().t8gpu_hip_error_string(code).decode()})")


def suffix(dtype):
    import torch
    if dtype in (torch.float32, np.float32):
        return "f32"
    if dtype in (torch.float64, np.float64):
        return "f64"
    raise TypeError(f"float_type must be float32 or float64, got {dtype}")


def vars_of(planes, step=0):
    """planes: torch tensor [nplanes, stride]; returns the T8gpuVars of planes[5*step : 5*step+5]."""
    import torch
    cls = Vars32 if planes.dtype == torch.float32 else Vars64
    v = cls()
    for k in range(5):
        v.p[k] = planes[5 * step + k].data_ptr()
    return v


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def fscalar(dtype, x):
    import torch
    return C.c_float(x) if dtype == torch.float32 else C.c_double(x)


def call(name, dtype, *args):
    fn = getattr(lib(), f"{name}_{suffix(dtype)}")
    fn.restype = C.c_int
    check(fn(*args))
