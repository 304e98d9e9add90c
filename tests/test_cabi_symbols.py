"""The C-ABI library must load without a GPU and export every symbol include/t8gpu_hip.h declares."""
import ctypes
import os
import re

from t8gpu_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            names |= set(re.findall(r"\b(t8gpu_hip_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_builds_loads_and_exports_the_header():
    path = build.build_hip()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 50
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    lib.t8gpu_hip_abi_version.restype = ctypes.c_int
    assert lib.t8gpu_hip_abi_version() >= 1


def test_host_library_exports_its_header():
    lib = ctypes.CDLL(build.build_host())
    text = open(os.path.join(ROOT, "include", "t8gpu_host.h")).read()
    names = set(re.findall(r"\b(t8gpu_(?:synth|plan|host)_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_host_library_has_no_hip_dependency():
    import subprocess
    out = subprocess.check_output(["ldd", build.build_host()]).decode()
    assert "amdhip" not in out and "rccl" not in out
