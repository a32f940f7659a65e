"""The reference's own test suite, test for test, on the HIP estimators and the HIP missing-value contractions.

One function per test of meyer-lab/cmtf-pls `tests/` (same name, same shapes, same assertions, file:line cited), with two
differences only: the imports come from `cmtf_pls_amd`, and the unseeded `np.random` draws are seeded generators, so a
failure reproduces.  Nothing here touches `oracle/`: this is the drop-in claim itself -- a user of the reference swaps the
import and the suite passes.  (`_test_dimension_compatibility` and `_test_decomposition_accuracy` are disabled in the
reference by their leading underscore, tests/test_tpls.py:66,106; they are not ported.)
"""
import numpy as np
import pytest
from numpy.linalg import norm
from numpy.testing import assert_allclose

pytestmark = [pytest.mark.gpu, pytest.mark.small_fit]      # the product's default behaviour, small fits included ...

TENSOR_DIMENSIONS = (100, 38, 65)      # tests/test_tpls.py:13-15, tests/test_synthetic.py:4-6
N_RESPONSE = 4
N_LATENT = 8


@pytest.fixture(scope="module", params=["product_default", "regular_engine"], autouse=True)
def engine_mode(request):
    """... and the whole suite a second time on the multi-launch engine (a small float64 fit takes the one-launch kernel
    by default, which most other suites switch off): the drop-in claim must hold on both paths."""
    from cmtf_pls_amd.engine import EngineOptions, set_default_options
    old = set_default_options(EngineOptions(small_fit=(request.param == "product_default")))
    yield request.param
    set_default_options(old)


@pytest.fixture(scope="module")
def pkg():
    import cmtf_pls_amd
    from cmtf_pls_amd import missingvals, synthetic, util
    return dict(tPLS=cmtf_pls_amd.tPLS, ctPLS=cmtf_pls_amd.ctPLS, import_synthetic=synthetic.import_synthetic,
                make_synthetic_test=synthetic.make_synthetic_test, calcR2X=util.calcR2X, factors_to_tensor=util.factors_to_tensor,
                miss_tensordot=missingvals.miss_tensordot, miss_mmodedot=missingvals.miss_mmodedot)


@pytest.fixture(scope="module")
def standard(pkg, engine_mode):                                # tests/test_tpls.py:21-25
    x, y, cp_tensor = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT)
    pls = pkg["tPLS"](N_LATENT)
    pls.fit(x, y)
    return x, y, cp_tensor, pls


def congruence(A, B):
    """Mean |cosine| of the optimally matched columns (what tensorly's congruence_coefficient returns first)."""
    from scipy.optimize import linear_sum_assignment
    C = np.abs((A / norm(A, axis=0)).T @ (B / norm(B, axis=0)))
    r, c = linear_sum_assignment(-C)
    return C[r, c].mean()


# ---- tests/test_tpls.py ---------------------------------------------------------------------------
def test_factor_normality(standard):                           # tests/test_tpls.py:31-36
    pls = standard[3]
    for x_factor in pls.X_factors[1:]:
        assert_allclose(norm(x_factor, axis=0), 1)
    for y_factor in pls.Y_factors[1:]:
        assert_allclose(norm(y_factor, axis=0), 1)


def test_factor_orthogonality(standard):                       # tests/test_tpls.py:41-51
    pls = standard[3]
    x_cp = [f / norm(f, axis=0) for f in pls.X_factors]        # cp_normalize
    for component_1 in range(N_LATENT):
        for component_2 in range(component_1 + 1, N_LATENT):
            factor_product = 1
            for factor in x_cp:
                factor_product *= np.dot(factor[:, component_1], factor[:, component_2])
            assert abs(factor_product) < 1e-2


def test_consistent_components(standard):                      # tests/test_tpls.py:54-61
    pls = standard[3]
    for x_factor in pls.X_factors:
        assert x_factor.shape[1] == N_LATENT
    for y_factor in pls.Y_factors:
        assert y_factor.shape[1] == N_LATENT


def test_same_x_y(pkg):                                        # tests/test_tpls.py:84-95
    from sklearn.decomposition import PCA
    x, _, _ = pkg["import_synthetic"]((100, 100), N_RESPONSE, N_LATENT)
    pls = pkg["tPLS"](N_LATENT)
    pca = PCA(N_LATENT)
    pls.fit(x, x)
    scores = pca.fit_transform(x)
    assert_allclose(pls.X_factors[0], pls.Y_factors[0], rtol=0, atol=1e-4)
    assert_allclose(pls.X_factors[1], pls.Y_factors[1], rtol=0, atol=1e-4)
    assert congruence(pls.X_factors[0], scores) > 0.95
    assert congruence(pls.X_factors[1], pca.components_.T) > 0.95


def test_zero_covariance_x(pkg):                               # tests/test_tpls.py:98-104
    x, y, _ = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT)
    x[:, 0, :] = 1
    pls = pkg["tPLS"](N_LATENT)
    pls.fit(x, y)
    assert_allclose(pls.X_factors[1][0, :], 0)                 # rtol 1e-7, atol 0: exactly zero


def _test_increasing_R2X(pkg, X, Y):                           # tests/test_tpls.py:119-129
    tpls = pkg["tPLS"](12)
    tpls.fit(X, Y)
    assert np.all(np.diff(tpls.R2X) >= 0.0), "R2X is not monotonically increasing"
    assert np.all(np.diff(tpls.R2Y) >= 0.0), "R2Y is not monotonically increasing"


@pytest.mark.parametrize("n_response", [5, 7, 9])
def test_increasing_R2X_random(pkg, n_response):               # tests/test_tpls.py:132-136
    rng = np.random.default_rng(n_response)
    _test_increasing_R2X(pkg, rng.random((20, 8, 6, 4)), rng.random((20, n_response)))


@pytest.mark.parametrize("n_response", [5, 7, 9])
def test_increasing_R2X(pkg, n_response, n_latent=5):          # tests/test_tpls.py:139-142
    X, Y, _ = pkg["import_synthetic"]((20, 8, 6, 4), n_response, n_latent)
    _test_increasing_R2X(pkg, X, Y)


def test_transform(pkg):                                       # tests/test_tpls.py:145-155
    rng = np.random.default_rng(145)
    X, Y = rng.random((20, 8, 6, 4)), rng.random((20, 5))
    tpls = pkg["tPLS"](6)
    tpls.fit(X, Y)
    rord = np.arange(20)
    rng.shuffle(rord)
    X_scores, Y_scores = tpls.transform(X[rord, :], Y[rord, :])
    assert np.allclose(X_scores, tpls.X_factors[0][rord, :])
    assert np.allclose(Y_scores, tpls.Y_factors[0][rord, :])


# ---- tests/test_cmtf.py ---------------------------------------------------------------------------
def test_tPLS_equivalence(pkg):                                # tests/test_cmtf.py:8-15
    rng = np.random.default_rng(8)
    X, Y = rng.random((10, 9, 8, 7)), rng.random((10, 5))
    pls0 = pkg["tPLS"](6)
    pls0.fit(X, Y)
    pls1 = pkg["ctPLS"](6)
    pls1.fit([X], Y)
    assert np.allclose(pls0.R2X, pls1.R2Xs[0])


@pytest.mark.parametrize("X0dim", [(10, 9, 8, 7), (10, 9, 8, 7, 6)])
@pytest.mark.parametrize("X1dim", [(10, 8, 7), (10, 9, 8, 7)])
@pytest.mark.parametrize("X2dim", [(10, 8), (10, 9, 8)])
def test_ctPLS_dimensions(pkg, X0dim, X1dim, X2dim):           # tests/test_cmtf.py:18-29
    rng = np.random.default_rng(len(X0dim) * 100 + len(X1dim) * 10 + len(X2dim))
    Xs = [rng.random(d) for d in (X0dim, X1dim, X2dim)]
    Y = rng.random((10, 5))
    pls = pkg["ctPLS"](6)
    pls.fit(Xs, Y)
    assert np.allclose(pls.factor_T, pls.transform(Xs))
    assert np.all(np.diff(pls.R2Y))


def test_ctPLS_increasing_R2Y_synthetic(pkg):                  # tests/test_cmtf.py:32-41
    rng = np.random.default_rng(32)
    dims = [(10, 9, 8, 7), (10, 8, 7)]
    n_latent = 4
    Xs = [pkg["factors_to_tensor"]([rng.random((d, n_latent)) for d in ds]) for ds in dims]
    Y = rng.random((10, 4)) @ rng.random((5, 4)).T
    pls = pkg["ctPLS"](6)
    pls.fit(Xs, Y)
    assert np.all(np.diff(pls.R2Y))


def test_ctPLS_transform(pkg):                                 # tests/test_cmtf.py:44-50
    rng = np.random.default_rng(44)
    Xs = [rng.random(d) for d in [(10, 9, 8, 7), (10, 8, 7)]]
    Y = rng.random((10, 5))
    pls = pkg["ctPLS"](3)
    pls.fit(Xs, Y)
    assert np.allclose(pls.factor_T, pls.transform(Xs))


def test_ctPLS_missingvals(pkg):                               # tests/test_cmtf.py:53-66
    rng = np.random.default_rng(53)
    Xs = [rng.random(d) for d in [(10, 9, 8, 7), (10, 8, 7)]]
    Y = rng.random((10, 5))
    pls = pkg["ctPLS"](3)
    pls.fit(Xs, Y)
    Xs[0][5, 4, 3, 2] = np.nan
    Xs[1][6, 5, 4] = np.nan
    pls_m = pkg["ctPLS"](3)
    pls_m.fit(Xs, Y)
    assert pkg["calcR2X"](pls.factor_T, pls_m.factor_T) > 0.9


# ---- tests/test_missingvals.py --------------------------------------------------------------------
def test_miss_tensordot(pkg):                                  # tests/test_missingvals.py:13-33
    miss_tensordot = pkg["miss_tensordot"]
    rng = np.random.default_rng(13)
    X = rng.random((10, 5, 4, 3))
    X[rng.random(X.shape) < 0.1] = np.nan
    missX = np.isnan(X)
    u = rng.random(10)
    w = miss_tensordot(X, u, missX.reshape(X.shape[0], -1))
    w2 = np.einsum("i...,i...->...", X, u)
    assert w.shape == X.shape[1:]
    assert np.allclose(w * ~np.isnan(w2), np.nan_to_num(w2))
    assert np.array_equal(w, miss_tensordot(X, u))             # the mask defaults to isnan(X) (missingvals.py:11-12)

    total_error = 0
    for _ in range(10):
        X = rng.random((20, 1)) @ rng.random((8, 1)).T
        u = rng.random(20)
        w = X.T @ u
        X[rng.random(X.shape) < 0.2] = np.nan
        w1 = miss_tensordot(X, u)
        w2 = np.nan_to_num(X.T) @ u
        assert norm(w - w1) / norm(w) < norm(w - w2) / norm(w) + 0.01
        total_error += norm(w - w1) / norm(w)
    assert total_error < 1.2


def test_miss_mmodedot(pkg):                                   # tests/test_missingvals.py:36-49
    miss_mmodedot = pkg["miss_mmodedot"]
    rng = np.random.default_rng(36)
    total_error = 0
    for _ in range(10):
        X = rng.random((10, 9, 8, 7))
        facs = [rng.random(lf) for lf in X.shape[1:]]
        t = np.einsum("ijkl,j,k,l->i", X, *facs)               # multi_mode_dot(X, facs, range(1, X.ndim))
        X[rng.random(X.shape) < 0.1] = np.nan
        missX = np.isnan(X)
        t1 = miss_mmodedot(X, facs, missX)
        t2 = np.einsum("ijkl,j,k,l->i", np.nan_to_num(X), *facs)
        assert norm(t - t1) / norm(t) < norm(t - t2) / norm(t) + 0.01
        total_error += norm(t - t1) / norm(t)
    assert total_error < 1.2


@pytest.mark.parametrize("Xshape", [(10, 9, 8), (10, 9, 8, 7), (10, 9, 8, 7, 6)])
def test_miss_X_synthetic(pkg, Xshape):                        # tests/test_missingvals.py:52-67
    X, Y, _ = pkg["import_synthetic"](Xshape, 4, 1, seed=52 + len(Xshape))
    tpls = pkg["tPLS"](1)
    tpls.fit(X, Y)
    X[np.random.default_rng(len(Xshape)).random(X.shape) < 0.1] = np.nan
    tpls1 = pkg["tPLS"](1)
    tpls1.fit(X, Y)
    # The reference compares the loadings sign-sensitively.  The score and Y sides are sign-invariant; the trailing
    # loadings carry parafac's PAIRED sign, which tensorly does not pin ("parity unpinned"), so an
    # even number of flips between the two fits is allowed for here -- and checked to be even.
    flips = 0
    for i in range(X.ndim):
        fac, fac1 = tpls.X_factors[i], tpls1.X_factors[i]
        s = 1.0 if i == 0 else float(np.sign(np.sum(fac * fac1)))
        flips += s < 0
        assert (norm(fac - s * fac1) / norm(fac)) < 0.2
    assert flips % 2 == 0
    for i in range(Y.ndim):
        fac, fac1 = tpls.Y_factors[i], tpls1.Y_factors[i]
        assert (norm(fac - fac1) / norm(fac)) < 0.01


def test_miss_X_transform(pkg):                                # tests/test_missingvals.py:70-80
    rng = np.random.default_rng(70)
    X, Y = rng.random((10, 7, 6, 5)), rng.random((10, 4))
    X[rng.random(X.shape) < 0.2] = np.nan
    tpls = pkg["tPLS"](7)
    tpls.fit(X, Y)
    assert np.all(np.diff(tpls.R2X) >= 0.0)
    assert np.all(np.diff(tpls.R2Y) >= 0.0)
    Xsc, Ysc = tpls.transform(X, Y)
    assert np.allclose(tpls.X_factors[0], Xsc)
    assert np.allclose(tpls.Y_factors[0], Ysc)


def test_miss_X_imputation(pkg):                               # tests/test_missingvals.py:83-91
    X, Y, _ = pkg["import_synthetic"]((10, 9, 8, 7), 4, 3, seed=83)
    Xmiss = X.copy()
    missPos = np.random.default_rng(84).random(X.shape) < 0.25
    Xmiss[missPos] = np.nan
    tpls = pkg["tPLS"](3)
    tpls.fit(Xmiss, Y)
    assert pkg["calcR2X"](X[missPos], tpls.X_reconstructed()[missPos]) > 0.8


# ---- tests/test_synthetic.py (host recipe; the device generator has its own file, test_gpu_synthetic.py) ------
def test_synthetic_dimensions(pkg):                            # tests/test_synthetic.py:9-15
    x, y, cp_tensor = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT, error=0)
    assert all([factor.shape[1] == N_LATENT for factor in cp_tensor.factors])
    assert cp_tensor.y_factor.shape[1] == N_LATENT
    assert x.shape == TENSOR_DIMENSIONS
    assert y.shape == (TENSOR_DIMENSIONS[0], N_RESPONSE)


def test_synthetic_test_dimensions(pkg):                       # tests/test_synthetic.py:18-24
    n_test = 10
    x, y, cp_tensor = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT, error=0)
    x_test, y_test, test_tensor = pkg["make_synthetic_test"](cp_tensor, n_test, 0)
    assert cp_tensor.factors[0].shape[1] == test_tensor.factors[0].shape[1]
    assert test_tensor.factors[0].shape[0] == n_test


def test_reproducibility(pkg):                                 # tests/test_synthetic.py:27-41
    x1, y1, _ = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT, error=0, seed=42)
    x2, y2, _ = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT, error=0, seed=42)
    x3, y3, _ = pkg["import_synthetic"](TENSOR_DIMENSIONS, N_RESPONSE, N_LATENT, error=0, seed=43)
    assert np.array_equal(x1, x2)
    assert np.array_equal(y1, y2)
    assert not np.array_equal(x1, x3)
    assert not np.array_equal(y1, y3)


def test_shared_factor(pkg):                                   # tests/test_synthetic.py:44-49
    x, y, cp_tensor = pkg["import_synthetic"]((10, 10), 10, 10, error=0, seed=42)
    inv_x_factor = np.linalg.inv(cp_tensor.factors[1].T)
    inv_y_factor = np.linalg.inv(cp_tensor.y_factor.T)
    assert np.allclose(np.matmul(x, inv_x_factor), np.matmul(y, inv_y_factor))


# ---- the missing-value API against the REFERENCE-GENERATED vectors (tests/golden/ref_missingvals.npz) -----------
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_missingvals_api_matches_reference_vectors(pkg, golden_dir, tag):
    import os
    g = np.load(os.path.join(golden_dir, "ref_missingvals.npz"))
    X = g[f"{tag}_X"]
    facs = [g[f"{tag}_w{m}"] for m in range(X.ndim - 1)]
    w = pkg["miss_tensordot"](X, g[f"{tag}_u"])
    want = g[f"{tag}_tensordot"]
    assert w.shape == want.shape
    assert_allclose(w, want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
    t = pkg["miss_mmodedot"](X, facs)
    want = g[f"{tag}_mmodedot"]
    assert np.array_equal(np.isnan(t), np.isnan(want))
    ok = ~np.isnan(want)
    assert_allclose(t[ok], want[ok], rtol=1e-11, atol=1e-11 * np.abs(want[ok]).max())
    # an explicit mask: flag extra positions as missing, leave the stored values in place (missingvals.py:11-14)
    rng = np.random.default_rng(5)
    extra = np.isnan(X) | (rng.random(X.shape) < 0.1)
    Xh = np.where(extra, np.nan, X)
    assert_allclose(pkg["miss_tensordot"](X, g[f"{tag}_u"], extra), pkg["miss_tensordot"](Xh, g[f"{tag}_u"]), rtol=0, atol=0)
    Xfilled = np.nan_to_num(X)                                  # no NaN in band at all: the mask alone decides
    assert_allclose(pkg["miss_tensordot"](Xfilled, g[f"{tag}_u"], np.isnan(X)), w, rtol=0, atol=0)
    t_mask = pkg["miss_mmodedot"](Xfilled, facs, np.isnan(X))
    assert np.array_equal(np.isnan(t_mask), np.isnan(t)) and np.array_equal(t_mask[ok], t[ok])
