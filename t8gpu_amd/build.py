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
        _run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-fopenmp", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", HOST_LIB] + srcs)
    return HOST_LIB


HIP_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall",
             "-Wno-unused-function"]


def build_hip(force=False, variant=None, defines=(), only=None):
    """One object per .hip file (compiled in parallel, rebuilt only when the file or a header changed), then one link.
    variant / defines: an experiment build with extra -D flags into lib/variants/libt8gpu_hip_<variant>.so
    (load it with T8GPU_HIP_LIB=...; used for A/B measurements and diagnostics, never by the product path).
    only: basenames of the sources the defines matter for -- the other objects are taken from the default build."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = _glob(os.path.join(CSRC, "hip"), (".hip", ".cpp"))
    hdrs = _glob(os.path.join(CSRC, "hip"), (".h", ".hpp")) + _glob(os.path.join(ROOT, "include"), (".h",))
    objdir = os.path.join(LIB, "obj", variant or "default")
    target = HIP_LIB if variant is None else os.path.join(LIB, "variants", f"libt8gpu_hip_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(target), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stamp = os.path.join(objdir, "flags.txt")
    flags = HIP_FLAGS + list(defines)
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs = []
    default_objdir = os.path.join(LIB, "obj", "default")
    if variant is not None and only is not None:
        build_hip()                                            # the shared objects must be current
    def objpath(src):
        shared = variant is not None and only is not None and os.path.basename(src) not in only
        return os.path.join(default_objdir if shared else objdir, os.path.basename(src) + ".o")
    for src in srcs:
        obj = objpath(src)
        if obj.startswith(default_objdir + os.sep) and variant is not None:
            continue
        if force or _newer(obj, [src] + hdrs):
            jobs.append([hipcc] + flags + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "hip"), "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(_run, jobs))
        open(stamp, "w").write(" ".join(flags))
    objs = [objpath(src) for src in srcs]
    if jobs or _newer(target, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-L/opt/rocm/lib", "-lrccl"])
    return target


def kernel_source_hash():
    """sha256[:16] over the CODE of the kernel sources (csrc/hip/kernels_*.hip and every header beside them; the step driver,
    which holds no kernel, is left out; `//` comments and white space do not count): what a committed rocprofv3 profile is tied
    to (profiles/traffic.json, bench.py)."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(CSRC, "hip")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hpp", ".h")) or (f.startswith("kernels_") and f.endswith(".hip")):
            text = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            text = re.sub(r"//[^\n]*", "", text)            # (no kernel source has `//` inside a string literal)
            text = re.sub(r"\s+", " ", text)
            h.update(f.encode())
            h.update(text.encode())
    return h.hexdigest()[:16]


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    deps = [os.path.join(odir, "oracle.hpp"), os.path.join(odir, "oracle_capi.cpp")]
    if force or _newer(ORACLE_LIB, deps) or _newer(ORACLE_OMP_LIB, deps):
        _run(["make", "-C", odir, "-B" if force else "-s"])
    return ORACLE_LIB


NO_RCCL_LIB = os.path.join(LIB, "variants", "libt8gpu_hip_norccl.so")


def build_diagnostic_variants(force=False):
    """Diagnostic builds the GPU test-suite loads in child processes (never the product path): `norccl` = the step driver
    with the RCCL group compiled out (tests/test_gpu_graph.py: the three-stream capture without RCCL)."""
    return build_hip(force, variant="norccl", defines=["-DT8GPU_EXP_NO_RCCL"], only=["stepper.hip"])


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_diagnostic_variants(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
