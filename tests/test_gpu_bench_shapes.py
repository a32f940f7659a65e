"""Fit-level parity at the BENCHMARK's trailing shapes (VERDICT r1 "What's weak" #4, SURVEY 8(d): "parity at
scale is GPU-vs-CPU-restatement on a row-subsampled replica").

The kernels the benchmark shapes select -- KC / FULL score_deflate variants, the 1024-thread deflate_rows path,
the parked-LDS half row at P = 65536, "u = Y q up front" for >= 16 column tiles, xcov / MTTKRP with M = 32 and
two response tiles -- are only reached with J = K = 128 or 256.  Here whole fits (direct AND xcov, coupled,
30 % NaN) at exactly those trailing shapes, with fewer rows so that the float64 oracle finishes in seconds,
are compared with the oracle value by value.

Tolerance (north star: "factors matching the CPU reference to rtol=1e-5"): f32 storage of f32-representable
inputs, every sum in f64; scores / loadings / q within 1e-5 of the oracle's, relative to each component's
own scale (column-wise max), loadings after the per-component paired sign.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def api():
    import cmtf_pls_amd
    return cmtf_pls_amd


def _f32(a):
    return a.astype(np.float32).astype(np.float64)


def col_close(got, want, rtol=RTOL, sign=None):
    """|got - want| <= rtol * (|want| + max|column|), column by column (normwise per component)."""
    if sign is not None:
        got = got * sign
    scale = np.abs(want).max(axis=0, keepdims=True)
    err = np.abs(got - want) / (np.abs(want) + scale)
    assert err.max() <= rtol, f"max column-relative error {err.max():.3e} > {rtol:.0e} (component {int(err.max(axis=0).argmax())})"
    return err.max()


def check_fit(m, fit, block=0, rtol=RTOL):
    Xf = m.X_factors if hasattr(m, "X_factors") else m.Xs_factors[block]
    col_close(Xf[0], fit.T, rtol)
    loads = fit.loadings[block]
    s = np.sign(np.sum(Xf[1] * loads[0], axis=0))
    s[s == 0] = 1
    for mode, L in enumerate(loads):
        col_close(Xf[1 + mode], L, rtol, sign=s if len(loads) == 2 else None)
    col_close(m.Y_factors[1], fit.Q, rtol)
    col_close(m.Y_factors[0], fit.U, rtol)
    assert_allclose(m.R2Y, fit.r2y, rtol=rtol, atol=rtol)
    r2x = m.R2X if hasattr(m, "R2X") else m.R2Xs[block]
    assert_allclose(r2x, fit.r2x[block], rtol=rtol, atol=rtol)
    assert_allclose(m.coef_, fit.coef, rtol=10 * rtol, atol=10 * rtol * np.abs(fit.coef).max())
    assert all(abs(a - b) <= 1 for a, b in zip(m.n_iter_, fit.n_iter))


# The R = 3 replicas of BASELINE configs[1..4] that lived here (round 2) repeated tests/test_gpu_r10_parity.py's inputs with a
# smaller rank and their own CPU oracle fits (~55 s of the suite); their extra assertions -- transform with a planted NaN,
# predict, float32 graph replay -- moved onto the R = 10 fits there (round 4).

# ---- BASELINE configs[4] at FULL size on one GPU through size-independent properties ---------------------
def test_cfg5_full_size_properties(api):
    """X 262144 x 256 x 256 f32 (68.7 GB), Y 262144 x 32: direct == xcov (two different kernel sets: sweeps with
    "u up front" vs. S on the matrix cores with two response tiles + carried S), unit-norm loadings, monotone R2,
    transform(training X) == training scores (one-pass MTTKRP over 68.7 GB), R2X against the literal
    reconstruction formula on a row sample."""
    import torch
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~140 GB of free HBM")
    X, Y = synthetic_shard_device((262144, 256, 256), 32, 10, error=0.1, device="cuda:0")
    m = api.tPLS(2, dtype="float32")
    m.fit(X, Y, max_iter=12)
    xc = api.tPLS(2, dtype="float32", algorithm="xcov")
    xc.fit(X, Y, max_iter=12)
    assert m.n_iter_ == xc.n_iter_
    # round 4: the 256 x 256 row is served by the one-read form (the row split over four workgroups), on the uncentred tensor
    assert xc.fit_report_["one_read"] and xc.fit_report_["raw"] and not xc.fit_report_["x_written"], xc.fit_report_
    col_close(xc.X_factors[0], m.X_factors[0])
    assert_allclose(xc.R2X, m.R2X, rtol=1e-6)
    assert_allclose(xc.R2Y, m.R2Y, rtol=1e-6)
    for f in m.X_factors[1:]:
        assert_allclose(np.linalg.norm(f, axis=0), 1, rtol=1e-12)
    assert np.all(np.diff(m.R2X) > 0) and np.all(np.diff(m.R2Y) > 0)
    s = np.abs(m.X_factors[0]).max()
    assert_allclose(m.transform(X), m.X_factors[0], rtol=1e-4, atol=1e-5 * s)
    rows = torch.arange(0, 262144, 256, device="cuda:0")          # 1024 rows
    Xs = X[rows].double() - torch.from_numpy(m.X_mean).cuda()
    Tsub = torch.from_numpy(m.X_factors[0]).cuda()[rows]
    rec = torch.einsum("ir,jr,kr->ijk", Tsub, torch.from_numpy(m.X_factors[1]).cuda(), torch.from_numpy(m.X_factors[2]).cuda())
    r2_sample = 1 - float(((Xs - rec) ** 2).sum()) / float((Xs ** 2).sum())
    assert abs(r2_sample - m.R2X[-1]) < 1e-2


# ---- trailing modes beyond every BASELINE config: rows of 196 608 - 655 360 elements -----------------
import functools


@functools.lru_cache(maxsize=None)
def _long_rows_oracle(shape, M, dt):
    x, y, _ = O.import_synthetic(shape, M, 3, error=0.1, seed=31)
    if dt == "float32":
        x, y = _f32(x), _f32(y)
    return O.fit_tpls(x, y, 2, max_iter=8)


@pytest.mark.parametrize("shape,M,dt", [((48, 512, 384), 5, "float32"), ((24, 1024, 640), 3, "float64")])
@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_very_long_rows_fit_vs_oracle(api, shape, M, dt, algorithm):
    """Rows of 0.75 - 5 MB (several 1024-thread segments per row in the row-per-workgroup sweeps, the loadings
    read from global memory where they exceed the LDS, rank-1 extraction at n = 384 / 640 up to its 1024 limit).
    transform / predict: at 512 x 384 the one-pass MTTKRP with 112 KB of loadings in LDS (one workgroup per CU); at
    1024 x 640 the loadings (208 KB) exceed the MTTKRP's LDS budget, which the engine must notice by itself and
    take the SEQUENTIAL project-and-deflate path."""
    x, y, _ = O.import_synthetic(shape, M, 3, error=0.1, seed=31)
    if dt == "float32":
        x, y = _f32(x), _f32(y)
    fit = _long_rows_oracle(shape, M, dt)                        # max_iter = 8: parity at a fixed iteration count is as strict
    m = api.tPLS(2, dtype=dt, algorithm=algorithm)
    m.fit(x, y, max_iter=8)
    check_fit(m, fit, rtol=RTOL if dt == "float32" else 1e-7)
    T = m.transform(x[:16])
    col_close(T, O.transform(fit, x[:16]), RTOL if dt == "float32" else 1e-7)
    col_close(m.predict(x[:16]), O.predict(fit, x[:16]), RTOL if dt == "float32" else 1e-7)
    xr = m.X_reconstructed()
    assert xr.shape == shape
    want = O.reconstruct(fit)
    assert_allclose(xr, want, rtol=0, atol=(1e-4 if dt == "float32" else 1e-6) * np.abs(want).max())
