"""The drop-in boundary, checked against the reference's OWN translation units: the device-code TUs of the two
examples are compiled with hipcc for gfx950 against include/t8gpu, read from /root/reference where they lie
(never copied into this repo; the test is skipped where the reference tree is absent, e.g. on the GPU box).
`<t8.h>` comes from tests/compat/t8_stub (opaque handles, a compile-only TEST stub: nothing is linked or run).

  examples/compressible_euler/kernels.cu   must compile with ZERO errors (t8gpu::MeshManager is a complete type,
                                           every accessor member the kernels use exists with the same signature)
  examples/subgrid/kernels_{2d,3d}.cu      every diagnostic must be one of the two places where the REFERENCE source
                                           omits the `template` keyword after `typename SubgridType::` (accepted by
                                           nvcc, an error for clang / hipcc; INTEGRATION.md section 1 tells a
                                           maintainer to add it) -- nothing may point at this backend's headers
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples")), reason="reference tree not present")


def hipcc(src, example_dir, mode, out):
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-x", "hip", "-ferror-limit=0", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "compat", "t8_stub"), "-I", os.path.join(REF, "examples", example_dir),
           os.path.join(REF, "examples", example_dir, src)] + (["-fsyntax-only"] if mode == "syntax" else ["-c", "-o", out])
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900)


def errors(stderr):
    return [ln for ln in stderr.splitlines() if re.search(r": (fatal )?error:", ln)]


def test_reference_plain_kernels_tu_compiles_unchanged(tmp_path):
    """kepes_compute_fluxes / reflective_boundary_condition / estimate_gradient ... exactly as the reference wrote
    them, down to a gfx950 object file."""
    res = hipcc("kernels.cu", "compressible_euler", "object", str(tmp_path / "kernels.o"))
    assert res.returncode == 0 and not errors(res.stderr), res.stderr[-4000:]
    assert os.path.getsize(tmp_path / "kernels.o") > 10000


@pytest.mark.parametrize("src", ["kernels_3d.cu", "kernels_2d.cu"])
def test_reference_subgrid_kernels_tu_only_trips_over_its_own_missing_template_keywords(src):
    res = hipcc(src, "subgrid", "syntax", None)
    errs = errors(res.stderr)
    ours = [e for e in errs if "/root/repo/" in e or "include/t8gpu" in e]
    assert not ours, "\n".join(ours)
    known = re.compile(r"examples/subgrid/(kernels\.h:60|kernels\.inl:1111):\d+: error: .*template")
    unknown = [e for e in errs if not known.search(e)]
    assert not unknown, "\n".join(unknown[:20])
