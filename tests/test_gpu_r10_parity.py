"""R = 10 fits (the benchmark's rank, DEFAULT tol / max_iter) at the benchmark's trailing shapes against the oracle
(VERDICT r2 "Next" #1, north star: "factors matching the CPU reference to rtol=1e-5").

The replicas of tests/test_gpu_bench_shapes.py with every BASELINE config's own rank: f32 storage of f32-representable
inputs, direct AND xcov, at
  (4096, 128, 128) M = 16          BASELINE configs[1] with 1/16 of the rows
  (1024, 256, 256) M = 32          BASELINE configs[4] with 1/256 of the rows
  (1024, 128, 128) + (1024, 512)   BASELINE configs[2], coupled
  (1024, 128, 128), 30 % NaN       BASELINE configs[3]
Asserted metric: NORMWISE 1e-5 per factor column (tests/parity_metrics.py); the plain element-wise figure is computed
and written next to it (table: gpurun_out/r10_parity_table.txt -> profiles/), together with the iteration counts.

Oracle-independent evidence (the oracle's parafac restatement cannot be pinned to tensorly here):
  * order-2 X: the first component against scikit-learn's PLSSVD (w_1 = leading left singular vector of X^T Y);
  * every component of the (4096, 128, 128) R = 10 fit is a FIXED POINT of the reference's loop evaluated with nothing
    but NumPy's LAPACK SVD on the product's own outputs (tpls.py:80-102 line by line);
  * the reference's disabled `_test_decomposition_accuracy` (tests/test_tpls.py:107-117) on the product.
"""
import os

import numpy as np
import pytest
from numpy.linalg import norm

import oracle as O
from golden import make_r10_golden as G10
from parity_metrics import column_errors, fit_error_table, format_table, worst

pytestmark = pytest.mark.gpu

RTOL = 1e-5
R = 10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "gpurun_out", "r10_parity_table.txt")


@pytest.fixture(scope="module")
def api():
    import cmtf_pls_amd
    return cmtf_pls_amd


def _f32(a):
    return a.astype(np.float32).astype(np.float64)


def replica(name):
    """(blocks, y, oracle fit, oracle transform of the first 256 rows): inputs regenerated from their seeds, the oracle's
    outputs from tests/golden/oracle_r10_<name>.npz (the float64 oracle needs minutes per configuration; the fixture was
    written by tests/golden/make_r10_golden.py, which runs exactly `O.fit_tpls / O.fit_ctpls(inputs, 10)` with default tol
    and max_iter) after its input checksums are verified; without a fixture the oracle runs here when R10_RUN_ORACLE=1."""
    blocks, y, coupled = G10.inputs(name)
    got = G10.load(name)
    if got is None:
        if os.environ.get("R10_RUN_ORACLE") != "1":
            pytest.skip(f"tests/golden/oracle_r10_{name}.npz missing (python tests/golden/make_r10_golden.py {name}; or R10_RUN_ORACLE=1)")
        fit = O.fit_ctpls(blocks, y, R) if coupled else O.fit_tpls(blocks[0], y, R)
        return blocks, y, fit, O.transform(fit, [b[:256] for b in blocks] if coupled else blocks[0][:256])
    fit, sums, head = got
    np.testing.assert_allclose(G10.checksums(blocks, y), sums, rtol=1e-12, err_msg="regenerated inputs differ from the fixture's")
    return blocks, y, fit, head


def _record(title, rows, extra):
    text = format_table(title, rows, extra)
    print("\n" + text)
    try:
        os.makedirs(os.path.dirname(TABLE), exist_ok=True)
        with open(TABLE, "a") as f:
            f.write(text + "\n\n")
    except OSError:
        pass


def check(m, fit, title, blocks=(0,)):
    for b in blocks:
        rows, extra = fit_error_table(m, fit, b)
        _record(f"{title} block {b}", rows, extra)
        err, fac, comp = worst(rows)
        # the contract is 1e-5 (RTOL); measured on MI355X (profiles/r03s_r10_parity_table.txt): <= 1.6e-8 on every factor
        # column of every configuration, so the guard sits at 1e-6 -- a regression of two orders fails here long before
        # the contract is at risk
        assert err <= RTOL / 10, f"{title}: normwise error {err:.3e} > {RTOL / 10:.0e} in {fac}[:, {comp}]"
        assert max(extra["R2X_abs"]) <= RTOL / 10 and max(extra["R2Y_abs"]) <= RTOL / 10
        assert extra["coef_normwise"] <= RTOL
        # iteration counts are part of parity: EQUAL for every component of every configuration (f32
        # storage, f64 accumulation: the convergence norm crosses 1e-8 at the same iteration as in the f64 oracle)
        assert extra["n_iter"] == extra["n_iter_oracle"], (extra["n_iter"], extra["n_iter_oracle"])


# ---- BASELINE configs[1] replica ------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg2():
    blocks, y, fit, head = replica("cfg2")
    return blocks[0], y, fit, head


@pytest.mark.parametrize("algorithm,graphs", [("direct", False), ("xcov", False), ("direct", True)])
def test_cfg2_replica_r10(api, cfg2, algorithm, graphs):
    x, y, fit, head = cfg2
    m = api.tPLS(R, dtype="float32", algorithm=algorithm, graphs=graphs)
    m.fit(x, y)
    assert m.fit_report_["graphs"] is graphs
    check(m, fit, f"(4096,128,128) M=16 R=10 f32 {algorithm}{' graphs' if graphs else ''}")
    assert column_errors(m.transform(x[:256]), head)["normwise"].max() <= RTOL
    # transform / predict at this shape: one-pass MTTKRP and, with a NaN planted, the masked sequence for that sample only
    xt = x[:256].copy()
    xt[3, 5, 7] = np.nan
    assert column_errors(m.transform(xt), O.transform(fit, xt))["normwise"].max() <= RTOL
    assert m.projection_report_["incomplete_rows"] == 1
    np.testing.assert_allclose(m.predict(x[:512]), O.predict(fit, x[:512]), rtol=1e-5, atol=1e-5 * np.abs(y).max())


@pytest.mark.parametrize("algorithm,graphs", [("direct", False), ("xcov", False), ("direct", True)])
def test_cfg2_replica_r10_float64_storage(api, cfg2, algorithm, graphs):
    """The same R = 10 fit with float64 storage (what a float64 NumPy input selects by default): no storage rounding at
    all, so the product differs from the oracle only by the order of its float64 sums; also under HIP-graph replay."""
    x, y, fit, head = cfg2
    m = api.tPLS(R, dtype="float64", algorithm=algorithm, graphs=graphs)
    m.fit(x, y)
    rows, extra = fit_error_table(m, fit, 0)
    _record(f"(4096,128,128) M=16 R=10 f64 storage {algorithm}{' graphs' if graphs else ''} block 0", rows, extra)
    err, fac, comp = worst(rows)
    assert err <= 1e-9, f"normwise error {err:.3e} in {fac}[:, {comp}]"
    assert max(extra["R2X_abs"]) <= 1e-11 and max(extra["R2Y_abs"]) <= 1e-11 and extra["coef_normwise"] <= 1e-9
    assert extra["n_iter"] == extra["n_iter_oracle"]
    assert column_errors(m.transform(x[:256]), head)["normwise"].max() <= 1e-9


def test_cfg2_replica_r10_every_component_is_a_fixed_point_of_the_reference_loop(api, cfg2):
    """Oracle-independent: with the product's own (T, W_J, W_K, Q) deflate X and Y in float64 NumPy exactly as
    tpls.py:109-113 does and verify for every component that one more pass of tpls.py:80-102 -- written here with
    np.einsum and np.linalg.svd only -- returns the same loadings, score and q.  Components whose loop stopped at max_iter
    are not fixed points and are skipped."""
    x, y, _, _ = cfg2
    m = api.tPLS(R, dtype="float32")
    m.fit(x, y)
    X = x - x.mean(axis=0)
    Y = y - y.mean(axis=0)
    T, WJ, WK = m.X_factors
    U, Q = m.Y_factors
    checked = 0
    for a in range(R):
        if m.n_iter_[a] < 100:
            u = Y @ Q[:, a]                                                  # tpls.py:102
            Z = np.einsum("i...,i...->...", X, u)                            # tpls.py:83
            Uz, s, Vt = np.linalg.svd(Z)                                     # rank-1 CP of a matrix = leading singular pair
            sgn = np.sign(Uz[:, 0] @ WJ[:, a])
            assert column_errors(WJ[:, a], sgn * Uz[:, 0])["normwise"].max() <= RTOL * s[0] / (s[0] - s[1]), a
            assert column_errors(WK[:, a], sgn * Vt[0])["normwise"].max() <= RTOL * s[0] / (s[0] - s[1]), a
            t = np.einsum("ijk,j,k->i", X, WJ[:, a], WK[:, a])               # tpls.py:97-99
            assert column_errors(T[:, a], t)["normwise"].max() <= RTOL, a
            q = Y.T @ t
            assert column_errors(Q[:, a], q / norm(q))["normwise"].max() <= RTOL, a     # tpls.py:100-101
            assert column_errors(U[:, a], u)["normwise"].max() <= RTOL, a
            checked += 1
        X = X - np.einsum("i,j,k->ijk", T[:, a], WJ[:, a], WK[:, a])         # tpls.py:109
        Y = Y - np.outer(T @ m.coef_[:, a], Q[:, a])                         # tpls.py:113
    assert checked >= 5


# ---- BASELINE configs[4] replica ------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg5():
    blocks, y, fit, head = replica("cfg5")
    return blocks[0], y, fit, head


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_cfg5_replica_r10(api, cfg5, algorithm):
    x, y, fit, head = cfg5
    m = api.tPLS(R, dtype="float32", algorithm=algorithm)
    m.fit(x, y)
    check(m, fit, f"(1024,256,256) M=32 R=10 f32 {algorithm}")
    assert column_errors(m.transform(x[:256]), head)["normwise"].max() <= RTOL
    xt = x[:64].copy()
    xt[1, 2, 3] = np.nan                                         # rows of 256 x 256: the 1024-thread rows-in-registers form
    assert column_errors(m.transform(xt), O.transform(fit, xt))["normwise"].max() <= RTOL


# ---- BASELINE configs[2] replica: coupled ---------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg3():
    blocks, y, fit, head = replica("cfg3")
    return blocks[0], blocks[1], y, fit, head


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_cfg3_replica_coupled_r10(api, cfg3, algorithm):
    x, xm, y, fit, head = cfg3
    m = api.ctPLS(R, dtype="float32", algorithm=algorithm)
    m.fit([x, xm], y)
    check(m, fit, f"(1024,128,128)+(1024,512) coupled M=16 R=10 f32 {algorithm}", blocks=(0, 1))
    assert column_errors(m.transform([x[:256], xm[:256]]), head)["normwise"].max() <= RTOL


# ---- BASELINE configs[3] replica: 30 % NaN ----------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg4():
    blocks, y, fit, head = replica("cfg4")
    return blocks[0], y, fit, head


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_cfg4_replica_nan30_r10(api, cfg4, algorithm):
    x, y, fit, head = cfg4
    m = api.tPLS(R, dtype="float32", algorithm=algorithm)
    m.fit(x, y)
    assert m.X_hasMiss
    check(m, fit, f"(1024,128,128) 30% NaN M=16 R=10 f32 {algorithm}")
    assert column_errors(m.transform(x[:256]), head)["normwise"].max() <= RTOL


# ---- oracle-independent: scikit-learn ---------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_order2_first_component_equals_sklearn_plssvd(api, dtype):
    """For a matrix X the converged first NIPALS component is the leading singular triplet of X_c^T Y_c:
    w_1 = PLSSVD.x_weights_, q_1 = PLSSVD.y_weights_, t_1 = X_c w_1 = PLSSVD.transform(X) (scale=False)."""
    from sklearn.cross_decomposition import PLSSVD
    rng = np.random.default_rng(7)
    lat = rng.normal(size=(1500, 4))
    x = _f32(lat @ rng.normal(size=(4, 96)) + 0.1 * rng.normal(size=(1500, 96)))
    y = _f32(lat @ rng.normal(size=(4, 6)) + 0.1 * rng.normal(size=(1500, 6)))
    ref = PLSSVD(n_components=1, scale=False).fit(x, y)
    m = api.tPLS(1, dtype=dtype)
    m.fit(x, y)
    assert m.n_iter_[0] < 100
    w, q, t = ref.x_weights_[:, 0], ref.y_weights_[:, 0], ref.transform(x)[:, 0] if ref.transform(x).ndim == 2 else ref.transform(x)
    s = np.sign(w @ m.X_factors[1][:, 0])
    tol = RTOL if dtype == "float32" else 1e-7
    assert column_errors(m.X_factors[1][:, 0], s * w)["normwise"].max() <= tol
    assert column_errors(m.Y_factors[1][:, 0], s * q)["normwise"].max() <= tol
    assert column_errors(m.X_factors[0][:, 0], s * np.asarray(t).reshape(-1))["normwise"].max() <= tol


def congruence(A, B):
    """Mean |cosine| of the optimally matched columns (tensorly's congruence_coefficient, first return value)."""
    from scipy.optimize import linear_sum_assignment
    C = np.abs((A / norm(A, axis=0)).T @ (B / norm(B, axis=0)))
    r, c = linear_sum_assignment(-C)
    return C[r, c].mean()


@pytest.mark.parametrize("idims", [(3, 1), (4, 1), (3, 4), (4, 2)])
def test_decomposition_accuracy_port_of_the_references_disabled_test(api, idims):
    """tests/test_tpls.py:107-117 (`_test_decomposition_accuracy`, disabled upstream by its leading underscore), on the
    product: N-way PLS components of a noise-free CP tensor against the generating CP factors.  PLS components are not
    CP components (they maximise covariance with Y and are extracted one at a time), so the upstream threshold 0.95 is
    asserted where PLS and CP coincide -- one latent factor -- and with several the test prints the congruences and
    asserts what does hold for any number of factors: every loading lies in the span of the generating factors of its
    mode (the tensor has no other directions), and R2X / R2Y do not decrease."""
    from cmtf_pls_amd.synthetic import import_synthetic
    x_rank, n_response = idims
    dims = tuple([40] * x_rank)
    # one latent factor: PLS == CP up to scale (the score is the CENTRED sample factor: upstream's 0.95 there)
    x, y, cp = import_synthetic(dims, n_response, 1)
    pls = api.tPLS(1)
    pls.fit(x, y)
    assert congruence(pls.X_factors[0], cp.factors[0]) > 0.95
    for pls_factor, true_factor in zip(pls.X_factors[1:], cp.factors[1:]):
        assert congruence(pls_factor, true_factor) > 1 - 1e-9
    assert congruence(pls.Y_factors[1], cp.y_factor) > 1 - 1e-9
    # several latent factors
    L = 3
    x, y, cp = import_synthetic(dims, n_response, L)
    pls = api.tPLS(L)
    pls.fit(x, y)
    cong = [congruence(f, g) for f, g in zip(pls.X_factors, cp.factors)]
    print(f"idims {idims}: congruence with the generating CP factors per mode {np.round(cong, 4)}, R2X {pls.R2X[-1]:.6f}, R2Y {pls.R2Y[-1]:.6f}")
    for f, g in zip(pls.X_factors[1:], cp.factors[1:]):
        proj = g @ np.linalg.lstsq(g, f, rcond=None)[0]
        assert norm(proj - f) <= 1e-6 * norm(f)
    assert np.all(np.diff(pls.R2X) >= -1e-12) and np.all(np.diff(pls.R2Y) >= -1e-12)
