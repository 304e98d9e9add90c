"""Accuracy of the fast-tier scalar helpers (csrc/hip/flux_math.hpp) against the host libm, through the
diagnostic entry t8gpu_hip_math_probe_*: the fused kernels replace IEEE division / sqrt / log by
reciprocal + Newton sequences and a short log, which is only admissible if they stay within a few ulp on
the operand ranges the solver produces."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RCP, DIV, SQRT, LOG, LN_MEAN, LN_MEAN_REF, LOG_TAB, SQRT_RATIO, DIV_SHARED = range(9)


def probe(op, a, b=None):
    import torch
    from t8gpu_amd import hip
    ta = torch.from_numpy(a).cuda()
    tb = torch.from_numpy(b).cuda() if b is not None else None
    out = torch.empty_like(ta)
    hip.call("t8gpu_hip_math_probe", ta.dtype, C.c_int(op), C.c_int(ta.numel()), hip.ptr(ta), hip.ptr(tb), hip.ptr(out),
             hip.stream_ptr())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def ulps(got, want):
    want = want.astype(got.dtype)
    return np.abs(got.astype(np.float64) - want.astype(np.float64)) / np.spacing(np.abs(want)).astype(np.float64)


@pytest.mark.parametrize("dtype,limit", [(np.float64, 2.0), (np.float32, 2.5)])
def test_division_sqrt_within_ulps(dtype, limit):
    rng = np.random.default_rng(3)
    a = np.exp(rng.uniform(-12, 12, 200000)).astype(dtype) * rng.choice([-1.0, 1.0], 200000).astype(dtype)
    b = np.exp(rng.uniform(-12, 12, 200000)).astype(dtype)
    exact_div = (a.astype(np.longdouble) / b.astype(np.longdouble))
    assert ulps(probe(DIV, a, b), exact_div.astype(dtype)).max() <= limit
    assert ulps(probe(RCP, b), (1 / b.astype(np.longdouble)).astype(dtype)).max() <= limit
    assert ulps(probe(SQRT, b), np.sqrt(b.astype(np.longdouble)).astype(dtype)).max() <= limit
    assert ulps(probe(DIV_SHARED, a, b), exact_div.astype(dtype)).max() <= limit
    c = np.exp(rng.uniform(-12, 12, 200000)).astype(dtype)
    ratio = np.sqrt(c.astype(np.longdouble) / b.astype(np.longdouble))
    assert ulps(probe(SQRT_RATIO, c, b), ratio.astype(dtype)).max() <= limit + 1.0


@pytest.mark.parametrize("dtype,k,op", [(np.float64, 1.5, LOG), (np.float32, 4.0, LOG),   # fp32: hardware log2 x ln 2
                                        (np.float64, 2.0, LOG_TAB)])   # fp64 in the plain tile kernels: 128-entry table
def test_log_matches_libm(dtype, k, op):
    rng = np.random.default_rng(4)
    span = 40 if dtype == np.float64 else 20      # (fp32: the range of densities / pressures / ratios with a wide margin)
    x = np.concatenate([np.exp(rng.uniform(-span, span, 300000)), rng.uniform(0.5, 2.0, 300000),
                        1.0 + rng.uniform(-1e-6, 1e-6, 1000), [1.0, 0.5, 2.0, np.sqrt(0.5), np.sqrt(2.0)]]).astype(dtype)
    want = np.log(x.astype(np.longdouble))
    got = probe(op, x).astype(np.longdouble)
    # relative to max(|log x|, ulp-scale of the argument error): 1 ulp of the result, or 1 ulp of x near x = 1
    tol = k * np.maximum(np.spacing(np.abs(want).astype(dtype)).astype(np.longdouble), np.finfo(dtype).eps / 2)
    assert (np.abs(got - want) <= tol).all(), float((np.abs(got - want) / tol).max())


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 2e-13), (np.float32, 3e-5)])
def test_ln_mean_fast_vs_reference_formula(dtype, rtol):
    """ln_mean of the fast tier against the exact (aR - aL) / log(aR / aL) and against the compat tier's
    restatement of kernels.cu:24-36, across the series / log branch switch (u = 1e-4 <=> ratio ~ 1.02)."""
    rng = np.random.default_rng(5)
    aL = np.exp(rng.uniform(-3, 3, 400000))
    ratio = np.concatenate([np.exp(rng.uniform(-3, 3, 200000)), 1 + rng.uniform(-0.05, 0.05, 199000), np.ones(1000)])
    aR = aL * ratio
    aL, aR = aL.astype(dtype), aR.astype(dtype)
    L, R = aL.astype(np.longdouble), aR.astype(np.longdouble)
    with np.errstate(invalid="ignore", divide="ignore"):
        exact = np.where(L == R, L, (R - L) / np.log(R / L))
    fast, ref = probe(LN_MEAN, aL, aR), probe(LN_MEAN_REF, aL, aR)
    assert (np.abs(fast - exact) <= rtol * np.abs(exact)).all(), float((np.abs(fast - exact) / np.abs(exact)).max())
    assert (np.abs(ref - exact) <= rtol * np.abs(exact)).all(), float((np.abs(ref - exact) / np.abs(exact)).max())
