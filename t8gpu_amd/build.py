"""In-tree build of the native libraries (no cmake; g++ / hipcc directly).

  t8gpu_amd/lib/libt8gpu_host.so   host-only C++ (synthetic mesh provider, halo/tile plans)
  t8gpu_amd/lib/libt8gpu_hip.so    HIP kernels for gfx950 + the C-ABI of include/t8gpu_hip.h
  oracle/liboracle*.so             CPU checker (test infrastructure; built here, used only by tests/bench)

hipcc cross-compiles gfx950 without a GPU, so all of this runs in the build container.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "t8gpu_amd")
LIB = os.path.join(PKG, "lib")
CSRC = os.path.join(PKG, "csrc")
HOST_LIB = os.path.join(LIB, "libt8gpu_host.so")
HIP_LIB = os.path.join(LIB, "libt8gpu_hip.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
ORACLE_OMP_LIB = os.path.join(ROOT, "oracle", "liboracle_omp.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _glob(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def _run(cmd):
    print("[build]", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd)


def build_host(force=False):
    srcs = _glob(os.path.join(CSRC, "host"), (".cpp",))
    deps = srcs + _glob(os.path.join(CSRC, "host"), (".h", ".hpp")) + [os.path.join(ROOT, "include", "t8gpu_host.h")]
    if force or _newer(HOST_LIB, deps):
        os.makedirs(LIB, exist_ok=True)
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fopenmp", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", HOST_LIB] + srcs)
    return HOST_LIB


def build_hip(force=False):
    srcs = _glob(os.path.join(CSRC, "hip"), (".hip", ".cpp"))
    deps = srcs + _glob(os.path.join(CSRC, "hip"), (".h", ".hpp")) + _glob(os.path.join(ROOT, "include"), (".h",))
    if force or _newer(HIP_LIB, deps):
        os.makedirs(LIB, exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize",
              "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "hip"),
              "-Wall", "-Wno-unused-function", "-o", HIP_LIB] + srcs + ["-L/opt/rocm/lib", "-lrccl"])
    return HIP_LIB


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    deps = [os.path.join(odir, "oracle.hpp"), os.path.join(odir, "oracle_capi.cpp")]
    if force or _newer(ORACLE_LIB, deps) or _newer(ORACLE_OMP_LIB, deps):
        _run(["make", "-C", odir, "-B" if force else "-s"])
    return ORACLE_LIB


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
