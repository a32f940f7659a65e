"""Opt-in mixed-precision matrix-core kernels (f32 MFMA, csrc/mixed.hip) for f32-stored X.

Stated accuracy: the second operand is rounded once to f32 and products are chained in f32 over at
most 64 rows / 256 columns before entering f64, so S and M agree with the f64 kernels to
rtol 2e-6 relative to sum |terms|; a fit through them agrees with the f64 path to 2e-6 on scores
(the default path stays f64-exact, tests/test_gpu_kernels.py).
"""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from cmtf_pls_amd.backend import HipBackend
    return HipBackend("cuda:0")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).to("cuda:0")


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,M", [((37, 10, 8), 4), ((100, 38, 65), 3), ((300, 16, 16), 17), ((1030, 128, 128), 16),
                                     ((257, 24, 12), 33), ((70, 8, 8), 64), ((64, 1, 20), 16)])
def test_xcov_mixed(be, shape, M, masked):
    rng = np.random.default_rng(61)
    I, A, B = shape
    x = rng.normal(size=(I, A * B)).astype(np.float32).astype(np.float64)
    if masked:
        x[rng.random(x.shape) < 0.3] = np.nan
    y = rng.normal(size=(I, M))
    S = be.xcov(dev(x, torch.float32), dev(y), masked, mixed=True).cpu().numpy()
    x0 = np.nan_to_num(x)
    want = y.T @ x0
    scale = np.abs(y).T @ np.abs(x0)                       # sum |terms| per entry
    assert np.max(np.abs(S - want) / (scale + 1e-30)) < 2e-6
    # same tile -> output mapping as the f64 kernel (asymmetric data would expose a swapped map)
    S64 = be.xcov(dev(x, torch.float32), dev(y), masked).cpu().numpy()
    assert np.max(np.abs(S - S64) / (scale + 1e-30)) < 2e-6


@pytest.mark.parametrize("shape,R", [((37, 10, 8), 3), ((100, 38, 65), 8), ((64, 1, 20), 5), ((50, 128, 128), 10),
                                     ((70, 12, 8), 17), ((300, 16, 16), 32),
                                     # round 3, the k-row form: 128- and 64-column passes, two passes, more than one f32 chain (A > 256)
                                     ((41, 128, 128), 10), ((23, 32, 64), 7), ((19, 16, 256), 16), ((9, 272, 128), 5)])
def test_mttkrp_mixed(be, shape, R):
    rng = np.random.default_rng(62)
    I, A, B = shape
    x = rng.normal(size=(I, A * B)).astype(np.float32).astype(np.float64)
    WA, WB = rng.normal(size=(A, R)), rng.normal(size=(B, R))
    out = be.mttkrp(dev(x, torch.float32), A, B, dev(WA), dev(WB), be.empty(I, R), mixed=True).cpu().numpy()
    W = (WA[:, None, :] * WB[None, :, :]).reshape(A * B, R)
    want = x @ W
    scale = np.abs(x) @ np.abs(W)
    assert np.max(np.abs(out - want) / (scale + 1e-30)) < 2e-6


def test_fit_with_mixed_matrix_precision():
    from cmtf_pls_amd import tPLS
    x, y, _ = O.import_synthetic((2048, 32, 32), 16, 6, error=0.1, seed=33)
    x = x.astype(np.float32).astype(np.float64)
    a = tPLS(4, dtype="float32", algorithm="xcov")
    b = tPLS(4, dtype="float32", algorithm="xcov", matrix_precision="f32")
    a.fit(x, y)
    b.fit(x, y)
    s = np.abs(a.X_factors[0]).max()
    assert all(abs(i - j) <= 1 for i, j in zip(a.n_iter_, b.n_iter_))
    np.testing.assert_allclose(b.X_factors[0], a.X_factors[0], rtol=2e-6, atol=2e-6 * s)
    np.testing.assert_allclose(b.R2X, a.R2X, rtol=1e-6)
    np.testing.assert_allclose(b.R2Y, a.R2Y, rtol=1e-6)
    fit = O.fit_tpls(x, y, 4)
    np.testing.assert_allclose(b.X_factors[0], fit.T, rtol=1e-5, atol=1e-5 * s)      # the north-star tolerance still holds
    xt = np.random.default_rng(5).normal(size=(40, 32, 32)).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(b.transform(xt), a.transform(xt), rtol=2e-6, atol=2e-6 * s)
