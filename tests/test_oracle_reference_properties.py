"""Seeded ports of the reference's own tests, run against the oracle.

The reference's tests are property / known-answer tests on small random inputs (no golden
vectors); several use unseeded RNG.  They are the only value-level statements the reference makes
about its fit loop, so they are what pins the oracle's restatement of tpls.py / cmtf.py.
Each test names the reference test it ports.
"""
import numpy as np
import pytest
from numpy.linalg import norm
from numpy.testing import assert_allclose
from sklearn.decomposition import PCA

import oracle as O

DIMS, N_RESPONSE, N_LATENT = (100, 38, 65), 4, 8


def _congruence_min(A, B):
    """|cos| between best-matched columns (what tensorly's congruence_coefficient reports)."""
    from scipy.optimize import linear_sum_assignment
    A = A / norm(A, axis=0)
    B = B / norm(B, axis=0)
    C = np.abs(A.T @ B)
    r, c = linear_sum_assignment(-C)
    return C[r, c].mean()


@pytest.fixture(scope="module")
def standard():
    x, y, cp = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT)
    return x, y, cp, O.fit_tpls(x, y, N_LATENT)


def test_factor_normality(standard):                       # tests/test_tpls.py:31
    fit = standard[3]
    for L in fit.loadings[0]:
        assert_allclose(norm(L, axis=0), 1)
    assert_allclose(norm(fit.Q, axis=0), 1)


def test_factor_orthogonality(standard):                   # tests/test_tpls.py:41
    fit = standard[3]
    facs = [f / norm(f, axis=0) for f in fit.x_factors(0)]
    R = fit.n_components
    for c1 in range(R):
        for c2 in range(c1 + 1, R):
            prod = 1.0
            for f in facs:
                prod *= f[:, c1] @ f[:, c2]
            assert abs(prod) < 1e-2


def test_consistent_components(standard):                  # tests/test_tpls.py:54
    fit = standard[3]
    assert all(f.shape[1] == N_LATENT for f in fit.x_factors(0) + fit.y_factors)


def test_same_x_y():                                       # tests/test_tpls.py:84
    x, _, _ = O.import_synthetic((100, 100), N_RESPONSE, N_LATENT)
    fit = O.fit_tpls(x, x, N_LATENT)
    pca = PCA(N_LATENT)
    scores = pca.fit_transform(x)
    assert_allclose(fit.T, fit.U, rtol=0, atol=1e-4)
    assert_allclose(fit.loadings[0][0], fit.Q, rtol=0, atol=1e-4)
    assert _congruence_min(fit.T, scores) > 0.95
    assert _congruence_min(fit.loadings[0][0], pca.components_.T) > 0.95


def test_zero_covariance_x():                              # tests/test_tpls.py:98
    x, y, _ = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT)
    x[:, 0, :] = 1
    fit = O.fit_tpls(x, y, N_LATENT)
    assert_allclose(fit.loadings[0][0][0, :], 0)


@pytest.mark.parametrize("n_response", [5, 7, 9])
def test_increasing_r2_random(n_response):                 # tests/test_tpls.py:132
    rng = np.random.default_rng(100 + n_response)
    fit = O.fit_tpls(rng.random((20, 8, 6, 4)), rng.random((20, n_response)), 12)
    assert np.all(np.diff(fit.r2x[0]) >= 0.0)
    assert np.all(np.diff(fit.r2y) >= 0.0)


@pytest.mark.parametrize("n_response", [5, 7, 9])
def test_increasing_r2_synthetic(n_response):              # tests/test_tpls.py:139
    X, Y, _ = O.import_synthetic((20, 8, 6, 4), n_response, 5)
    fit = O.fit_tpls(X, Y, 12)
    assert np.all(np.diff(fit.r2x[0]) >= 0.0)
    assert np.all(np.diff(fit.r2y) >= 0.0)


def test_transform():                                      # tests/test_tpls.py:145
    rng = np.random.default_rng(7)
    X, Y = rng.random((20, 8, 6, 4)), rng.random((20, 5))
    fit = O.fit_tpls(X, Y, 6)
    order = rng.permutation(20)
    xs, ys = O.transform(fit, X[order], Y[order])
    assert np.allclose(xs, fit.T[order])
    assert np.allclose(ys, fit.U[order])


def test_tpls_ctpls_equivalence():                         # tests/test_cmtf.py:8
    rng = np.random.default_rng(8)
    X, Y = rng.random((10, 9, 8, 7)), rng.random((10, 5))
    assert np.allclose(O.fit_tpls(X, Y, 6).r2x[0], O.fit_ctpls([X], Y, 6).r2x[0])


@pytest.mark.parametrize("dims", [[(10, 9, 8, 7), (10, 8, 7), (10, 8)],
                                  [(10, 9, 8, 7, 6), (10, 9, 8, 7), (10, 9, 8)]])
def test_ctpls_dimensions(dims):                           # tests/test_cmtf.py:18-29
    rng = np.random.default_rng(9)
    Xs = [rng.random(d) for d in dims]
    Y = rng.random((10, 5))
    fit = O.fit_ctpls(Xs, Y, 6)
    assert np.allclose(fit.T, O.transform(fit, Xs))
    assert np.all(np.diff(fit.r2y))


def test_ctpls_missingvals():                              # tests/test_cmtf.py:53
    rng = np.random.default_rng(10)
    Xs = [rng.random((10, 9, 8, 7)), rng.random((10, 8, 7))]
    Y = rng.random((10, 5))
    full = O.fit_ctpls(Xs, Y, 3)
    Xs[0][5, 4, 3, 2] = np.nan
    Xs[1][6, 5, 4] = np.nan
    miss = O.fit_ctpls(Xs, Y, 3)
    assert O.calc_r2x(full.T, miss.T) > 0.9


def test_miss_tensordot_equivalence():                     # tests/test_missingvals.py:13
    rng = np.random.default_rng(11)
    X = rng.random((10, 5, 4, 3))
    X[rng.random(X.shape) < 0.1] = np.nan
    u = rng.random(10)
    w = O.masked_mode0_contract(X, u)
    w2 = np.einsum("i...,i...->...", X, u)
    assert np.allclose(w * ~np.isnan(w2), np.nan_to_num(w2))


def test_miss_x_transform():                               # tests/test_missingvals.py:70
    rng = np.random.default_rng(12)
    X, Y = rng.random((10, 7, 6, 5)), rng.random((10, 4))
    X[rng.random(X.shape) < 0.2] = np.nan
    fit = O.fit_tpls(X, Y, 7)
    assert np.all(np.diff(fit.r2x[0]) >= 0.0)
    assert np.all(np.diff(fit.r2y) >= 0.0)
    xs, ys = O.transform(fit, X, Y)
    assert np.allclose(fit.T, xs)
    assert np.allclose(fit.U, ys)


def test_miss_x_imputation():                              # tests/test_missingvals.py:83
    X, Y, _ = O.import_synthetic((10, 9, 8, 7), 4, 3, seed=123)
    Xm = X.copy()
    pos = np.random.default_rng(13).random(X.shape) < 0.25
    Xm[pos] = np.nan
    fit = O.fit_tpls(Xm, Y, 3)
    assert O.calc_r2x(X[pos], O.reconstruct(fit)[pos]) > 0.8


def test_synthetic_contract():                             # tests/test_synthetic.py:9-49
    x, y, cp = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT, error=0)
    assert x.shape == DIMS and y.shape == (DIMS[0], N_RESPONSE)
    assert all(f.shape[1] == N_LATENT for f in cp.factors) and cp.y_factor.shape[1] == N_LATENT
    x1, y1, _ = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT, seed=42)
    x2, y2, _ = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT, seed=42)
    x3, _, _ = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT, seed=43)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2) and not np.array_equal(x1, x3)
    xs, ys, c = O.import_synthetic((10, 10), 10, 10, error=0, seed=42)
    assert np.allclose(xs @ np.linalg.inv(c.factors[1].T), ys @ np.linalg.inv(c.y_factor.T))
    xt, yt, ct = O.make_synthetic_test(c, 10, 0)
    assert ct.factors[0].shape == (10, 10)
