"""End-to-end parity of the HIP tPLS / ctPLS against the oracle and the committed golden vectors,
plus the reference's own property tests (seeded, order-3 shapes) run on the HIP classes.

Tolerances: float64 storage -> the GPU path is the same arithmetic as the oracle up to summation
order, so factors agree to rtol 1e-7 (component-wise sign canonicalised) and the
iteration counts are equal; float32 storage of fp32-representable inputs -> rtol 1e-5 (the
north-star tolerance), iteration counts within +-1 after the first in-place f32 deflation.
"""
import os

import numpy as np
import pytest
from numpy.linalg import norm
from numpy.testing import assert_allclose

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import cmtf_pls_amd
    return cmtf_pls_amd


def signs(W, Wref):
    s = np.sign(np.sum(W * Wref, axis=0))
    s[s == 0] = 1
    return s


def check_against_oracle(m, fit, rtol, block=0, exact_iters=True):
    Xf = m.X_factors if hasattr(m, "X_factors") else m.Xs_factors[block]
    scale = np.abs(fit.T).max()
    assert_allclose(Xf[0], fit.T, rtol=rtol, atol=rtol * scale)
    loads = fit.loadings[block]
    if len(loads) == 2:
        s = signs(Xf[1], loads[0])
        assert_allclose(Xf[1] * s, loads[0], rtol=rtol, atol=rtol)
        assert_allclose(Xf[2] * s, loads[1], rtol=rtol, atol=rtol)   # paired sign: same s for both modes
    else:
        assert_allclose(Xf[1], loads[0], rtol=rtol, atol=rtol)
    assert_allclose(m.Y_factors[0], fit.U, rtol=rtol, atol=rtol * np.abs(fit.U).max())
    assert_allclose(m.Y_factors[1], fit.Q, rtol=rtol, atol=rtol)
    assert_allclose(m.coef_, fit.coef, rtol=10 * rtol, atol=10 * rtol * np.abs(fit.coef).max())
    assert_allclose(m.R2Y, fit.r2y, rtol=rtol, atol=rtol)
    if exact_iters:
        assert list(m.n_iter_) == list(fit.n_iter)
    else:
        assert all(abs(a - b) <= 1 for a, b in zip(m.n_iter_, fit.n_iter))


@pytest.mark.small_fit
@pytest.mark.parametrize("err", [0.0, 0.1])
def test_tpls_cfg1_f64(api, golden_dir, err, small_fit_mode):
    """BASELINE.json configs[0]: synthetic (200,10,8), M=4, R=3 -- on BOTH paths such a fit can take: the product default (the
    whole fit in one launch, cmtfpls_fit_small_f64) and the regular multi-launch engine."""
    x, y, _ = O.import_synthetic((200, 10, 8), 4, 3, error=err)
    m = api.tPLS(3, options=small_fit_mode)
    m.fit(x, y)
    assert m.fit_report_["form"] == ("small_fit" if small_fit_mode.small_fit else "regular")
    fit = O.fit_tpls(x, y, 3)
    check_against_oracle(m, fit, 1e-7)
    assert_allclose(m.R2X, fit.r2x[0], rtol=1e-8, atol=1e-9)
    g = np.load(os.path.join(golden_dir, "oracle_tpls_cfg1.npz" if err == 0 else "oracle_tpls_cfg1_noise.npz"))
    assert_allclose(m.X_factors[0], g["T"], rtol=1e-7, atol=1e-7)
    assert_allclose(m.predict(g["x_test"]), g["pred_test"], rtol=1e-7, atol=1e-7)
    assert_allclose(m.transform(g["x_test"]), g["scores_test"], rtol=1e-7, atol=1e-7)
    if err == 0:
        assert [n - 1 for n in m.n_iter_] == [6, 35, 1]               # BASELINE.md section 2 (reference run)
        assert_allclose(m.R2X, [0.715, 0.853, 0.940], atol=6e-4)
        assert_allclose(m.R2Y, [0.475, 0.800, 1.000], atol=6e-4)


def test_tpls_f32_storage_golden(api, golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_tpls_f32in.npz"))
    x, y = g["x"], g["y"]
    m = api.tPLS(4, dtype="float32")
    m.fit(x, y)
    fit = O.fit_tpls(x, y, 4)
    check_against_oracle(m, fit, 1e-5, exact_iters=False)
    assert_allclose(m.X_factors[0], g["T"], rtol=1e-5, atol=1e-5 * np.abs(g["T"]).max())
    assert_allclose(m.R2X, g["r2x0"], rtol=1e-5, atol=1e-6)
    # float32 inputs select float32 storage on their own
    m2 = api.tPLS(4)
    m2.fit(x.astype(np.float32), y.astype(np.float32))
    assert_allclose(m2.X_factors[0], m.X_factors[0], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-7), ("float32", 1e-5)])
def test_tpls_nan30_golden(api, golden_dir, dtype, rtol):
    """missingvals path (BASELINE configs[3] in small): 30 % NaN."""
    g = np.load(os.path.join(golden_dir, "oracle_tpls_f32in_nan30.npz"))
    x, y = g["x"], g["y"]
    m = api.tPLS(4, dtype=dtype)
    m.fit(x, y)
    assert m.X_hasMiss
    fit = O.fit_tpls(x, y, 4)
    check_against_oracle(m, fit, rtol, exact_iters=(dtype == "float64"))
    assert_allclose(m.R2X, fit.r2x[0], rtol=rtol, atol=rtol)
    assert_allclose(m.X_factors[0], g["T"], rtol=rtol, atol=rtol * np.abs(g["T"]).max())
    xs, ys = m.transform(x, y)
    assert np.allclose(xs, m.X_factors[0], rtol=10 * rtol, atol=10 * rtol * np.abs(xs).max())      # tests/test_missingvals.py:70-80
    assert np.allclose(ys, m.Y_factors[0], rtol=10 * rtol, atol=10 * rtol * np.abs(ys).max())
    miss = np.isnan(x)
    assert_allclose(m.X_reconstructed()[miss], g["recon"][miss], rtol=100 * rtol, atol=100 * rtol)


@pytest.mark.parametrize("dtype,rtol", [("float64", 1e-7), ("float32", 1e-5)])
def test_ctpls_golden(api, golden_dir, dtype, rtol):
    """Coupled tensor + matrix block (BASELINE configs[2] in small)."""
    g = np.load(os.path.join(golden_dir, "oracle_ctpls_small.npz"))
    x0, x1, y = g["x0"], g["x1"], g["y"]
    if dtype == "float32":
        x0, x1, y = [a.astype(np.float32).astype(np.float64) for a in (x0, x1, y)]
    m = api.ctPLS(4, dtype=dtype)
    m.fit([x0, x1], y)
    fit = O.fit_ctpls([x0, x1], y, 4)
    check_against_oracle(m, fit, rtol, block=0, exact_iters=(dtype == "float64"))
    check_against_oracle(m, fit, rtol, block=1, exact_iters=(dtype == "float64"))
    for b in range(2):
        assert_allclose(m.R2Xs[b], fit.r2x[b], rtol=rtol, atol=rtol)
    if dtype == "float64":
        assert_allclose(m.factor_T, g["T"], rtol=1e-7, atol=1e-7)
    assert np.allclose(m.factor_T, m.transform([x0, x1]), rtol=10 * rtol, atol=10 * rtol * np.abs(m.factor_T).max())
    assert_allclose(m.predict([x0, x1]), O.predict(fit, [x0, x1]), rtol=10 * rtol, atol=10 * rtol * np.abs(y).max())


def test_tpls_ctpls_equivalence(api):                        # tests/test_cmtf.py:8 (order-3 shape)
    rng = np.random.default_rng(8)
    X, Y = rng.random((10, 9, 8)), rng.random((10, 5))
    a, b = api.tPLS(6), api.ctPLS(6)
    a.fit(X, Y)
    b.fit([X], Y)
    assert np.allclose(a.R2X, b.R2Xs[0])


# ---- the reference's property tests on the HIP estimator (tests/test_tpls.py) ----------------
DIMS, N_RESPONSE, N_LATENT = (100, 38, 65), 4, 8


@pytest.fixture(scope="module")
def standard(api):
    x, y, cp = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT)
    m = api.tPLS(N_LATENT)
    m.fit(x, y)
    return x, y, cp, m


def test_factor_normality(standard):                          # tests/test_tpls.py:31
    m = standard[3]
    for f in m.X_factors[1:]:
        assert_allclose(norm(f, axis=0), 1)
    assert_allclose(norm(m.Y_factors[1], axis=0), 1)


def test_factor_orthogonality(standard):                      # tests/test_tpls.py:41
    m = standard[3]
    facs = [f / norm(f, axis=0) for f in m.X_factors]
    for c1 in range(N_LATENT):
        for c2 in range(c1 + 1, N_LATENT):
            prod = 1.0
            for f in facs:
                prod *= f[:, c1] @ f[:, c2]
            assert abs(prod) < 1e-2


def test_standard_matches_oracle(standard):
    x, y, _, m = standard
    fit = O.fit_tpls(x, y, N_LATENT)
    check_against_oracle(m, fit, 1e-6)


def test_same_x_y(api):                                       # tests/test_tpls.py:84
    from sklearn.decomposition import PCA
    x, _, _ = O.import_synthetic((100, 100), N_RESPONSE, N_LATENT)
    m = api.tPLS(N_LATENT)
    m.fit(x, x)
    assert_allclose(m.X_factors[0], m.Y_factors[0], rtol=0, atol=1e-4)
    assert_allclose(m.X_factors[1], m.Y_factors[1], rtol=0, atol=1e-4)
    pca = PCA(N_LATENT)
    scores = pca.fit_transform(x)

    def congruence(A, B):
        from scipy.optimize import linear_sum_assignment
        C = np.abs((A / norm(A, axis=0)).T @ (B / norm(B, axis=0)))
        r, c = linear_sum_assignment(-C)
        return C[r, c].mean()

    assert congruence(m.X_factors[0], scores) > 0.95
    assert congruence(m.X_factors[1], pca.components_.T) > 0.95


def test_zero_covariance_x(api):                              # tests/test_tpls.py:98
    x, y, _ = O.import_synthetic(DIMS, N_RESPONSE, N_LATENT)
    x[:, 0, :] = 1
    m = api.tPLS(N_LATENT)
    m.fit(x, y)
    assert_allclose(m.X_factors[1][0, :], 0)                  # rtol 1e-7, atol 0: exactly zero


@pytest.mark.parametrize("n_response", [5, 7])
def test_increasing_r2(api, n_response):                      # tests/test_tpls.py:132-142 (order-3)
    X, Y, _ = O.import_synthetic((20, 8, 6), n_response, 5)
    m = api.tPLS(5)
    m.fit(X, Y)
    assert np.all(np.diff(m.R2X) >= -1e-12)
    assert np.all(np.diff(m.R2Y) >= -1e-12)


def test_transform_permuted(api):                             # tests/test_tpls.py:145
    rng = np.random.default_rng(7)
    X, Y = rng.random((20, 8, 6)), rng.random((20, 5))
    m = api.tPLS(4)
    m.fit(X, Y)
    order = rng.permutation(20)
    xs, ys = m.transform(X[order], Y[order])
    assert np.allclose(xs, m.X_factors[0][order])
    assert np.allclose(ys, m.Y_factors[0][order])


def test_medium_f32_against_oracle(api):
    """A size where every vector kernel variant and multi-row-block reductions are exercised."""
    x, y, _ = O.import_synthetic((4096, 64, 48), 16, 6, error=0.1, seed=9)
    x = x.astype(np.float32).astype(np.float64)
    y = y.astype(np.float32).astype(np.float64)
    m = api.tPLS(3, dtype="float32")
    m.fit(x, y, max_iter=30)
    fit = O.fit_tpls(x, y, 3, max_iter=30)
    check_against_oracle(m, fit, 1e-5, exact_iters=False)
    assert_allclose(m.R2X, fit.r2x[0], rtol=1e-5, atol=1e-6)


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("shape,dtype,rtol", [((20, 8, 6, 4), "float64", 1e-6), ((12, 5, 4, 3, 2), "float64", 1e-6),
                                              ((20, 8, 6, 4), "float32", 1e-5)])
def test_tpls_higher_order(api, shape, dtype, rtol):
    """X of order 4 and 5 (the shapes of tests/test_tpls.py:132-155, tests/test_missingvals.py:52)."""
    rng = np.random.default_rng(7)
    X, Y = rng.random(shape), rng.random((shape[0], 5))
    if dtype == "float32":
        X, Y = X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64)
    m = api.tPLS(4, dtype=dtype)
    m.fit(X, Y)
    fit = O.fit_tpls(X, Y, 4)
    assert_allclose(m.X_factors[0], fit.T, rtol=rtol, atol=rtol * np.abs(fit.T).max())
    for got, want in zip(m.X_factors[1:], fit.loadings[0]):
        assert_allclose(got, want, rtol=rtol, atol=rtol)
    assert_allclose(m.R2X, fit.r2x[0], rtol=rtol, atol=rtol)
    assert_allclose(m.R2Y, fit.r2y, rtol=rtol, atol=rtol)
    order = rng.permutation(shape[0])
    xs, ys = m.transform(X[order], Y[order])                  # tests/test_tpls.py:145-155
    assert np.allclose(xs, m.X_factors[0][order], rtol=10 * rtol, atol=10 * rtol)
    assert np.allclose(ys, m.Y_factors[0][order], rtol=10 * rtol, atol=10 * rtol)


def test_ctpls_mixed_orders(api):                             # tests/test_cmtf.py:18-29
    rng = np.random.default_rng(9)
    Xs = [rng.random(d) for d in [(10, 9, 8, 7, 6), (10, 9, 8, 7), (10, 9, 8)]]
    Y = rng.random((10, 5))
    m = api.ctPLS(6)
    m.fit(Xs, Y)
    fit = O.fit_ctpls(Xs, Y, 6)
    assert_allclose(m.factor_T, fit.T, rtol=1e-5, atol=1e-6)
    assert np.allclose(m.factor_T, m.transform(Xs))
    assert np.all(np.diff(m.R2Y))
    assert_allclose(m.R2Y, fit.r2y, rtol=1e-5, atol=1e-7)


def test_miss_x_higher_order(api):                            # tests/test_missingvals.py:70-80
    rng = np.random.default_rng(12)
    X, Y = rng.random((10, 7, 6, 5)), rng.random((10, 4))
    X[rng.random(X.shape) < 0.2] = np.nan
    m = api.tPLS(7)
    m.fit(X, Y)
    assert np.all(np.diff(m.R2X) >= -1e-12) and np.all(np.diff(m.R2Y) >= -1e-12)
    xs, ys = m.transform(X, Y)
    assert np.allclose(m.X_factors[0], xs) and np.allclose(m.Y_factors[0], ys)


@pytest.mark.parametrize("case", ["plain", "nan", "matrix", "coupled", "f32"])
def test_xcov_algorithm_on_gpu(api, case):
    """algorithm="xcov": inner loop on S = X^T Y (f64 MFMA).  Same iteration counts and factors as the
    direct loop and as the oracle."""
    rng = np.random.default_rng(31)
    x, y, cp = O.import_synthetic((300, 12, 8), 5, 3, error=0.2, seed=8)
    dtype, rtol = ("float32", 1e-5) if case == "f32" else ("float64", 1e-7)
    if case == "f32":
        x, y = x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    if case == "nan":
        x[rng.random(x.shape) < 0.25] = np.nan
    if case == "matrix":
        x = x.reshape(300, 96)
    if case == "coupled":
        xm = cp.factors[0] @ rng.normal(size=(11, 3)).T + 0.1 * rng.normal(size=(300, 11))
        m, d = api.ctPLS(3, algorithm="xcov"), api.ctPLS(3)
        m.fit([x, xm], y)
        d.fit([x, xm], y)
        fit = O.fit_ctpls([x, xm], y, 3)
        check_against_oracle(m, fit, rtol, block=0)
        check_against_oracle(m, fit, rtol, block=1)
        assert_allclose(m.R2Xs[1], fit.r2x[1], rtol=rtol, atol=rtol)
        assert_allclose(m.factor_T, d.factor_T, rtol=1e-9, atol=1e-9)
    else:
        m, d = api.tPLS(3, dtype=dtype, algorithm="xcov"), api.tPLS(3, dtype=dtype)
        m.fit(x, y)
        d.fit(x, y)
        fit = O.fit_tpls(x, y, 3)
        if case == "matrix":
            assert_allclose(m.X_factors[0], fit.T, rtol=rtol, atol=rtol * np.abs(fit.T).max())
            assert_allclose(m.X_factors[1], fit.loadings[0][0], rtol=rtol, atol=rtol)
            assert m.n_iter_ == list(fit.n_iter)
        else:
            check_against_oracle(m, fit, rtol, exact_iters=(case != "f32"))
        assert_allclose(m.R2X, fit.r2x[0], rtol=rtol, atol=rtol)
        assert_allclose(m.X_factors[0], d.X_factors[0], rtol=rtol, atol=rtol * np.abs(fit.T).max())
        assert_allclose(m.transform(x), m.X_factors[0], rtol=10 * rtol, atol=10 * rtol * np.abs(fit.T).max())


def test_graph_replay_equals_eager(api):
    """FitRun.use_graphs: the iteration's launch sequence captured into a HIP graph and replayed must
    give bit-identical iterates to eager launches."""
    import torch
    from cmtf_pls_amd.backend import HipBackend
    from cmtf_pls_amd.engine import NipalsEngine
    x, y, _ = O.import_synthetic((512, 16, 16), 4, 3, error=0.2, seed=5)
    outs = []
    for graphs in (False, True):
        eng = NipalsEngine(HipBackend("cuda:0"))
        X = torch.from_numpy(x).to("cuda:0", torch.float32)
        Y = torch.from_numpy(y).to("cuda:0")
        run = eng.begin([X], Y, 3, coupled=False)
        run.use_graphs = graphs
        run.start_component(0)
        dus = [run.iterate(it) for it in range(12)]
        assert run._graph_error is None
        outs.append((dus, run.t.clone(), run.q.clone(), run.wA[0].clone()))
    assert outs[0][0][1:] == outs[1][0][1:]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["tpls3", "tpls3_f32", "tpls4", "matrix", "coupled"])
def test_one_pass_projection_on_gpu(api, case):
    """transform/predict through one MFMA MTTKRP pass + the R x R triangular recurrence must equal
    the R sequential project-and-deflate passes (tpls.py:133-142) and the oracle."""
    import torch
    from cmtf_pls_amd.tpls import to_device_copy
    rng = np.random.default_rng(41)
    dtype, rtol = (torch.float32, 1e-5) if case.endswith("f32") else (torch.float64, 1e-8)
    if case == "coupled":
        Xs = [rng.random((40, 6, 5, 4)), rng.random((40, 7, 8)), rng.random((40, 9))]
        Y = rng.random((40, 4))
        m = api.ctPLS(5)
        m.fit(Xs, Y)
        new = [rng.random((24,) + x.shape[1:]) for x in Xs]
        want = O.transform(O.fit_ctpls(Xs, Y, 5), new)
        got_api = m.transform(new)
    else:
        shape = {"tpls3": (64, 16, 12), "tpls3_f32": (64, 16, 12), "tpls4": (30, 6, 5, 4), "matrix": (30, 40)}[case]
        X, Y = rng.random(shape), rng.random((shape[0], 4))
        if case.endswith("f32"):
            X = X.astype(np.float32).astype(np.float64)
        m = api.tPLS(5, dtype="float32" if case.endswith("f32") else "float64")
        m.fit(X, Y)
        new = [rng.random((24,) + shape[1:])]
        if case.endswith("f32"):
            new = [new[0].astype(np.float32).astype(np.float64)]
        want = O.transform(O.fit_tpls(X, Y, 5), new[0])
        got_api = m.transform(new[0])
    eng = m._get_engine()
    dev = lambda xs: [to_device_copy(x, dtype, "cuda:0") for x in xs]
    one = eng.project(m._state, dev(new), one_pass=True).cpu().numpy()
    seq = eng.project(m._state, dev(new), one_pass=False).cpu().numpy()
    scale = np.abs(want).max()
    assert_allclose(one, seq, rtol=rtol, atol=rtol * scale)
    assert_allclose(one, want, rtol=max(rtol, 1e-6), atol=max(rtol, 1e-6) * scale)
    # the API takes the READ-ONLY form (MTTKRP of the uncentred rows, centring applied to its output): identical algebra;
    # with f32 storage it skips the rounding of the centred copy to f32, so it differs from `one` by that rounding
    ro = eng.project_readonly(m._state, dev(new))
    assert ro is not None
    assert_allclose(got_api, ro.cpu().numpy(), rtol=0, atol=0)
    tight = 1e-11 if dtype == torch.float64 else 2e-7
    assert_allclose(got_api, one, rtol=0, atol=tight * scale)
    assert_allclose(got_api, want, rtol=max(rtol, 1e-6), atol=max(rtol, 1e-6) * scale)


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
@pytest.mark.parametrize("kind", ["tpls", "ctpls_mixed"])
def test_fit_with_graphs_is_bit_identical(api, algorithm, kind):
    """graphs=True (every iteration's launch sequence replayed as a HIP graph) changes nothing."""
    rng = np.random.default_rng(51)
    if kind == "tpls":
        x, y, _ = O.import_synthetic((400, 16, 12), 4, 3, error=0.2, seed=9)
        a, b = api.tPLS(3, dtype="float32", algorithm=algorithm), api.tPLS(3, dtype="float32", algorithm=algorithm, graphs=True)
        a.fit(x, y)
        b.fit(x, y)
        Ta, Tb = a.X_factors[0], b.X_factors[0]
    else:
        Xs = [rng.random((30, 6, 5, 4)), rng.random((30, 8, 4)), rng.random((30, 9))]
        Y = rng.random((30, 3))
        a, b = api.ctPLS(3, algorithm=algorithm), api.ctPLS(3, algorithm=algorithm, graphs=True)
        a.fit(Xs, Y)
        b.fit(Xs, Y)
        Ta, Tb = a.factor_T, b.factor_T
    assert a.n_iter_ == b.n_iter_
    assert np.array_equal(Ta, Tb)
    assert np.array_equal(a.coef_, b.coef_) and np.array_equal(a.R2Y, b.R2Y)


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_edge_shapes(api, algorithm):
    """1-D y (M = 1), a single new sample, more responses than the xcov kernel takes (falls back to
    the direct loop), more components than one MTTKRP call takes (falls back to sequential)."""
    x, y, _ = O.import_synthetic((40, 6, 5), 1, 2, error=0.1, seed=3)
    assert y.ndim == 1
    m = api.tPLS(2, algorithm=algorithm)
    m.fit(x, y)
    fit = O.fit_tpls(x, y, 2)
    assert_allclose(m.X_factors[0], fit.T, rtol=1e-7, atol=1e-8)
    assert_allclose(m.R2Y, fit.r2y, rtol=1e-7, atol=1e-9)
    assert_allclose(m.predict(x[:1]), O.predict(fit, x[:1]), rtol=1e-7, atol=1e-8)       # one sample
    # 70 responses: xcov is built for M <= 64 and must fall back, results unchanged
    rng = np.random.default_rng(4)
    X, Y = rng.random((30, 5, 4)), rng.random((30, 70))
    m2 = api.tPLS(3, algorithm=algorithm)
    m2.fit(X, Y)
    f2 = O.fit_tpls(X, Y, 3)
    assert_allclose(m2.X_factors[0], f2.T, rtol=1e-6, atol=1e-8)
    # 35 components (> 32 per MTTKRP call): transform falls back to the sequential path
    Xw, Yw = rng.random((60, 8, 6)), rng.random((60, 40))
    m3 = api.tPLS(35, algorithm=algorithm)
    m3.fit(Xw, Yw, max_iter=5)
    assert np.allclose(m3.transform(Xw), m3.X_factors[0], rtol=1e-6, atol=1e-8)


def test_torch_device_inputs_and_q2y(api):
    """Device tensors are accepted as inputs (cloned, never modified); validate.get_q2y runs its
    leave-one-out refits on the GPU and matches a literal LOO over the oracle."""
    import torch
    from cmtf_pls_amd.validate import get_q2y
    x, y, _ = O.import_synthetic((14, 5, 4), 2, 2, error=0.3, seed=6)
    Xd, Yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m = api.tPLS(2)
    m.fit(Xd, Yd)
    assert torch.equal(Xd.cpu(), torch.from_numpy(x))
    fit = O.fit_tpls(x, y, 2)
    assert_allclose(m.X_factors[0], fit.T, rtol=1e-7, atol=1e-8)
    mh = api.tPLS(2)
    mh.fit(x, y)
    pred = np.zeros_like(y)
    for i in range(14):
        keep = np.arange(14) != i
        pred[i] = O.predict(O.fit_tpls(x[keep], y[keep], 2), x[i:i + 1])[0]
    want = 1 - ((pred - y) ** 2).sum() / (y ** 2).sum()
    assert_allclose(get_q2y(mh), want, rtol=1e-6)


def test_copy_x_false_fits_in_place(api):
    import torch
    x, y, _ = O.import_synthetic((128, 8, 8), 3, 2, error=0.1, seed=4)
    Xd = torch.from_numpy(x).cuda()
    ref = api.tPLS(2)
    ref.fit(x, y)
    m = api.tPLS(2, copy_X=False)
    m.fit(Xd, y)
    assert_allclose(m.X_factors[0], ref.X_factors[0], rtol=1e-12, atol=1e-12)
    assert not torch.equal(Xd.cpu(), torch.from_numpy(x))        # the caller's tensor was centred and deflated
