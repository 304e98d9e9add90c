"""The oracle's KERNEL-LEVEL functions against executions of the reference's own kernel bodies.

tests/golden/reference_kernel_vectors.npz (tests/golden/make_reference_kernel_vectors.py, build container only) holds inputs and
outputs of the reference's `__global__` kernels run on the host through the reference's own accessor classes on tiny AMR meshes:
kepes_compute_fluxes + reflective_boundary_condition (examples/compressible_euler/kernels.cu:135-469), compute_inner /
boundary / outer_fluxes for Subgrid<4,4,4> and Subgrid<4,4> (examples/subgrid/kernels.inl:335-1107: inner-face pattern, walls,
the 2:1 hanging map) and the six SSP-RK3 stage kernels (t8gpu/timestepping/ssp_runge_kutta.inl:30-221). Here the oracle's
per-kernel entry points (oracle/oracle_capi.cpp) run on the same inputs.

What this closes: until round 4 only the flux FUNCTIONS had ever been executed; the index logic around them -- which element a face
reads and which it adds to, with which sign, the far-cell map of a hanging sub-face, the wall's mirror state -- was
pinned by reading alone, in the oracle and in the kernels alike. Tolerance: NONE -- the reference adds its fluxes in thread order
(the generator runs the threads of a launch one after another), the oracle walks faces and sub-faces in the same order ("one loop
iteration per CUDA thread"), and every array of every case agrees BIT FOR BIT in fp32 and fp64.
The stand-ins the generator needs (CUDA built-ins, the accessor-owning friend classes) keep parity "unpinned" by the project
rules."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "reference_kernel_vectors.npz")
KEPES = 0

_cache = {}


def vectors():
    if "z" not in _cache:
        _cache["z"] = np.load(PATH, allow_pickle=False)
    return _cache["z"]


def case_names():
    return sorted({k.split("|")[0] for k in vectors().files if "|" in k})


def load(case, tag):
    z = vectors()
    ins = {k.split("|")[3]: z[k] for k in z.files if k.startswith(f"{case}|{tag}|in|")}
    outs = {k.split("|")[3]: z[k] for k in z.files if k.startswith(f"{case}|{tag}|out|")}
    return ins, outs


def rel(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / scale)


TOL = {"f64": 2e-14, "f32": 2e-6}      # (only for the derived checks below: conservation, the outer kernel's own contribution)


def same(a, b, what):
    assert np.array_equal(a, b), (what, "max rel. difference", rel(a, b))


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("case", [c for c in case_names() if c.startswith("plain")])
def test_plain_kernels_against_the_executed_reference_kernels(case, tag):
    ins, outs = load(case, tag)
    N, G, F, B, stride = (int(x) for x in ins["sizes"])
    dt_ = ins["state"].dtype
    suf = O.suf(dt_)
    lib = O.lib()
    ndim = int(ins["normal_dim"][0])
    state = np.ascontiguousarray(ins["state"].reshape(5, stride))
    flux = np.zeros((5, stride), dt_)
    speed = np.zeros(F + B, dt_)
    # kepes_compute_fluxes (kernels.cu:135-309)
    getattr(lib, "oracle_plain_interior_faces_" + suf)(KEPES, F, ndim, O.p(ins["fn"]), O.p(ins["idx"]), O.p(ins["normals"]), O.p(ins["areas"]),
                                                      O.p(state), O.p(flux), C.c_size_t(stride), O.p(speed))
    want = outs["flux_interior"].reshape(5, stride)
    same(flux[:, :N], want[:, :N], (case, "interior faces"))
    # reflective_boundary_condition (kernels.cu:311-469)
    if B:
        getattr(lib, "oracle_plain_boundary_faces_" + suf)(KEPES, F, B, ndim, O.p(ins["fn"]), O.p(ins["normals"]), O.p(ins["areas"]), O.p(state),
                                                          O.p(flux), C.c_size_t(stride), O.p(speed))
    want = outs["flux_all"].reshape(5, stride)
    same(flux[:, :N], want[:, :N], (case, "interior + wall faces"))
    same(speed, outs["speed"], (case, "speed estimates"))
    # every element received something, and the reference's flux planes are conservative on a periodic mesh
    assert np.abs(want[:, :N]).max(axis=0).min() > 0
    if B == 0:
        assert np.abs(want[:, :N].astype(np.float64).sum(axis=1)).max() < 1e3 * TOL[tag] * np.abs(want).max() * N
    # SSP_3RK_step1 / 2 / 3 (ssp_runge_kutta.inl:30-99)
    mid = np.ascontiguousarray(ins["rk_mid"].reshape(5, stride))
    vol = np.ascontiguousarray(ins["volume"])
    for stage in (1, 2, 3):
        fl = np.ascontiguousarray(ins["rk_flux"].reshape(5, stride).copy())
        out = np.zeros((5, stride), dt_)
        getattr(lib, "oracle_plain_rk_stage_" + suf)(stage, N, O.p(state), O.p(mid) if stage > 1 else None, O.p(out), O.p(fl), C.c_size_t(stride),
                                                    O.p(vol), O.fs(dt_, float(ins["dt"][0])))
        want = outs[f"rk_out{stage}"].reshape(5, stride)
        assert np.array_equal(out[:, :N], want[:, :N]), (case, f"RK stage {stage}", rel(out[:, :N], want[:, :N]))
        assert np.array_equal(fl[:, :N], outs[f"rk_flux_after{stage}"].reshape(5, stride)[:, :N])      # the flux planes are zeroed


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("case", [c for c in case_names() if c.startswith("sub")])
def test_subgrid_kernels_against_the_executed_reference_kernels(case, tag):
    ins, outs = load(case, tag)
    N, G, F, B, stride = (int(x) for x in ins["sizes"])
    rank = int(ins["rank"][0])
    S = 4 ** rank
    dt_ = ins["state"].dtype
    suf = O.suf(dt_)
    lib = O.lib()
    state = np.ascontiguousarray(ins["state"].reshape(5, stride))
    vol = np.ascontiguousarray(ins["volume"])
    assert int((ins["level_diff"] != 0).sum()) > 0                       # the mesh has hanging faces
    flux = np.zeros((5, stride), dt_)
    # compute_inner_fluxes (kernels.inl:335-662)
    getattr(lib, "oracle_subgrid_inner_" + suf)(KEPES, rank, N, O.p(state), O.p(flux), C.c_size_t(stride), O.p(vol))
    want = outs["flux_inner"].reshape(5, stride)
    same(flux[:, :N * S], want[:, :N * S], (case, "inner faces"))
    # compute_boundary_fluxes (kernels.inl:913-1107)
    if B:
        getattr(lib, "oracle_subgrid_boundary_" + suf)(KEPES, rank, F, B, O.p(ins["fn"]), O.p(ins["normals"]), O.p(ins["areas"]), O.p(state),
                                                      O.p(flux), C.c_size_t(stride))
    want = outs["flux_inner_boundary"].reshape(5, stride)
    same(flux[:, :N * S], want[:, :N * S], (case, "inner + wall faces"))
    # compute_outer_fluxes (kernels.inl:664-911): same-level faces and the 2:1 hanging map
    getattr(lib, "oracle_subgrid_outer_" + suf)(KEPES, rank, F, O.p(ins["fn"]), O.p(ins["idx"]), O.p(ins["level_diff"]), O.p(ins["nb_offset"]),
                                               O.p(ins["normals"]), O.p(ins["areas"]), O.p(state), O.p(flux), C.c_size_t(stride))
    want = outs["flux_all"].reshape(5, stride)
    same(flux[:, :N * S], want[:, :N * S], (case, "all faces"))
    # the outer kernel alone changed the surface cells of every block and nothing else: compare the DIFFERENCE too (a wrong
    # far cell on a hanging face would drown in the inner fluxes' magnitude otherwise)
    d_ref = want.astype(np.float64) - outs["flux_inner_boundary"].reshape(5, stride).astype(np.float64)
    flux2 = np.zeros((5, stride), dt_)
    getattr(lib, "oracle_subgrid_outer_" + suf)(KEPES, rank, F, O.p(ins["fn"]), O.p(ins["idx"]), O.p(ins["level_diff"]), O.p(ins["nb_offset"]),
                                               O.p(ins["normals"]), O.p(ins["areas"]), O.p(state), O.p(flux2), C.c_size_t(stride))
    assert np.abs(flux2[:, :N * S].astype(np.float64) - d_ref[:, :N * S]).max() < 50 * TOL[tag] * np.abs(want).max(), (case, "outer faces alone")
    if B == 0:
        assert np.abs(want[:, :N * S].astype(np.float64).sum(axis=1)).max() < 1e3 * TOL[tag] * np.abs(want).max() * N * S
    # subgrid::SSP_3RK_step1 / 2 / 3 (ssp_runge_kutta.inl:101-221)
    mid = np.ascontiguousarray(ins["rk_mid"].reshape(5, stride))
    for stage in (1, 2, 3):
        fl = np.ascontiguousarray(ins["rk_flux"].reshape(5, stride).copy())
        out = np.zeros((5, stride), dt_)
        getattr(lib, "oracle_subgrid_rk_stage_" + suf)(stage, rank, N, O.p(state), O.p(mid) if stage > 1 else None, O.p(out), O.p(fl),
                                                      C.c_size_t(stride), O.p(vol), O.fs(dt_, float(ins["dt"][0])))
        want = outs[f"rk_out{stage}"].reshape(5, stride)
        assert np.array_equal(out[:, :N * S], want[:, :N * S]), (case, f"RK stage {stage}", rel(out[:, :N * S], want[:, :N * S]))
        assert np.array_equal(fl[:, :N * S], outs[f"rk_flux_after{stage}"].reshape(5, stride)[:, :N * S])


def test_fixture_is_what_it_says():
    z = vectors()
    assert "executed on the host through the reference's own accessor classes" in str(z["meta"][0])
    assert len(case_names()) >= 7
