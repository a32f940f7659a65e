"""Evidence for the fit that passes through NEITHER the oracle's `parafac` restatement NOR any kernel of
`libcmtfpls.so` as the checker (VERDICT r3 "Next" #1): every converged component of a product fit is re-derived from the
data by one pass of the reference's loop written with plain float64 torch operations (tests/reference_loop_torch.py:
rocBLAS matmul, LAPACK SVD, `torch.linalg.lstsq`), the masked contractions with the closed forms of `missingvals.py`
(pinned to the real reference by tests/golden/ref_missingvals*.npz).

(a) The R = 10 replicas of BASELINE configs[1..4] (tests/golden/make_r10_golden.py `inputs`), DEFAULT tol / max_iter,
    direct AND xcov: plain, 30 % NaN (tpls.py:80-82,92-95), coupled tensor + matrix (cmtf.py:92-123 incl. the block
    average), the 256 x 256 trailing shape; an order-4 X (rank-1 CP of an order-3 Z: stationarity conditions).
(b) The same check at the FULL size of BASELINE configs[1], [2], [3] (65536 x 128 x 128 f32, + 65536 x 512, 30 % NaN),
    R = 10, default tol / max_iter, on the device -- together with
(c) what bench.py only printed: full-size R = 10 `direct == xcov == direct under graph replay` (T normwise <= 1e-6, equal
    iteration counts, R2 to 1e-9), and the size-independent properties the reference's tests hold (transform of the
    training data reproduces the scores, tests/test_tpls.py:145-155; unit-norm loadings, :31-36; R2 non-decreasing).

Tolerance: 1e-5 normwise per factor column (north star), loadings x sigma_1 / (sigma_1 - sigma_2) of Z (a converged loop
is a fixed point only to its own tol / (1 - rate)).  Tables: gpurun_out/independent_table.txt -> profiles/.
"""
import os

import numpy as np
import pytest
import torch
from numpy.linalg import norm

from golden import make_r10_golden as G10
from parity_metrics import column_errors
from reference_loop_torch import check_estimator, format_records

pytestmark = pytest.mark.gpu

RTOL = 1e-5
R = 10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "gpurun_out", "independent_table.txt")


@pytest.fixture(scope="module")
def api():
    import cmtf_pls_amd
    return cmtf_pls_amd


def _record(title, records):
    text = format_records(title, records)
    print("\n" + text)
    try:
        os.makedirs(os.path.dirname(TABLE), exist_ok=True)
        with open(TABLE, "a") as f:
            f.write(text + "\n\n")
    except OSError:
        pass


# ---- (a) replicas --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def replicas():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = G10.inputs(name)
        return cache[name]
    return get


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg3", "cfg5"])
def test_replica_r10_components_are_fixed_points_of_the_reference_loop(api, replicas, name, algorithm):
    blocks, y, coupled = replicas(name)
    if coupled:
        m = api.ctPLS(R, dtype="float32", algorithm=algorithm)
        m.fit(blocks, y)
        data = blocks
    else:
        m = api.tPLS(R, dtype="float32", algorithm=algorithm)
        m.fit(blocks[0], y)
        data = blocks[0]
    recs = check_estimator(m, data, y, device="cuda", rtol=RTOL, min_checked=5, label=f"{name} {algorithm}")
    _record(f"replica {name} R=10 f32 {algorithm}: product factors vs one pass of the reference loop (torch f64)", recs)


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_order4_components_satisfy_the_rank1_stationarity_conditions(api, algorithm):
    """X of order 4: Z is an order-3 tensor and tpls.py:84-90 takes its rank-1 CP (tensorly's ALS, whose stopping step
    is unpinned).  Whatever the stopping step, a converged rank-1 CP satisfies: every loading is the normalised
    contraction of Z with the other two -- checked here on the product's loadings, then score, q, u, coef, R2 as usual."""
    from cmtf_pls_amd.synthetic import import_synthetic
    x, y, _ = import_synthetic((512, 24, 20, 16), 6, 4, error=0.1, seed=41)
    m = api.tPLS(4, dtype="float64", algorithm=algorithm)
    m.fit(x, y)
    recs = check_estimator(m, x, y, device="cuda", rtol=1e-6, min_checked=3, stationarity_rtol=1e-4, label=f"order4 {algorithm}")
    _record(f"order-4 X (512,24,20,16) R=4 f64 {algorithm}: stationarity of the rank-1 CP + one pass of the loop", recs)


@pytest.mark.parametrize("algorithm", ["direct", "xcov"])
def test_coupled_blocks_of_mixed_orders_with_missing_values(api, algorithm):
    """ctPLS on an order-4 block, an order-3 block with 20 % NaN and a matrix (cmtf.py:92-123: per-block contraction -- masked
    where the block has missing values -- rank-1 CP or Z / |Z|, per-block scores, np.average): every converged component against
    one pass of that loop."""
    rng = np.random.default_rng(51)
    lat = rng.normal(size=(400, 4))
    x4 = np.einsum("ir,jr,kr,lr->ijkl", lat, rng.normal(size=(12, 4)), rng.normal(size=(10, 4)), rng.normal(size=(8, 4))) + 0.1 * rng.normal(size=(400, 12, 10, 8))
    x3 = np.einsum("ir,jr,kr->ijk", lat, rng.normal(size=(16, 4)), rng.normal(size=(32, 4))) + 0.1 * rng.normal(size=(400, 16, 32))
    x3[rng.random(x3.shape) < 0.2] = np.nan
    xm = lat @ rng.normal(size=(4, 40)) + 0.1 * rng.normal(size=(400, 40))
    y = lat @ rng.normal(size=(4, 5)) + 0.1 * rng.normal(size=(400, 5))
    m = api.ctPLS(4, dtype="float64", algorithm=algorithm)
    m.fit([x4, x3, xm], y)
    recs = check_estimator(m, [x4, x3, xm], y, device="cuda", rtol=1e-6, min_checked=3, stationarity_rtol=1e-4, label=f"mixed orders {algorithm}")
    _record(f"coupled (400,12,10,8) + (400,16,32) 20% NaN + (400,40) R=4 f64 {algorithm}: one pass of cmtf.py:92-123 (torch f64)", recs)


# ---- (b) + (c) full size -------------------------------------------------------------------------------------------
def _full(matrix_block=0, nan_fraction=0.0):
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    return synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0",
                                  matrix_block=matrix_block, nan_fraction=nan_fraction)


def _fit(api, coupled, data, Y, **kw):
    m = (api.ctPLS if coupled else api.tPLS)(R, dtype="float32", **kw)
    m.fit(data, Y)
    return m


def _scores(m):
    return m.factor_T if hasattr(m, "factor_T") else m.X_factors[0]


def _r2x(m):
    return np.concatenate([np.asarray(r) for r in m.R2Xs]) if hasattr(m, "R2Xs") else np.asarray(m.R2X)


def _full_size_case(api, title, coupled, data, Y):
    direct = _fit(api, coupled, data, Y)
    xcov = _fit(api, coupled, data, Y, algorithm="xcov")
    graphs = _fit(api, coupled, data, Y, graphs=True)
    # (c) the three forms of the product agree
    T = _scores(direct)
    for name, other in (("xcov", xcov), ("graphs", graphs)):
        assert list(other.n_iter_) == list(direct.n_iter_), (title, name, other.n_iter_, direct.n_iter_)
        err = column_errors(_scores(other), T)["normwise"].max()
        assert err <= 1e-6, (title, name, err)
        assert np.abs(_r2x(other) - _r2x(direct)).max() <= 1e-9 and np.abs(other.R2Y - direct.R2Y).max() <= 1e-9, (title, name)
    # (b) every converged component against the torch-only pass of the reference's loop, on the device
    for name, m in (("direct", direct), ("xcov", xcov)):
        recs = check_estimator(m, data, Y, device="cuda", rtol=RTOL, min_checked=5, label=f"{title} {name}")
        _record(f"FULL SIZE {title} R=10 f32 {name}: product factors vs one pass of the reference loop (torch f64, on the device)", recs)
        torch.cuda.empty_cache()
    # size-independent properties (tests/test_tpls.py:31-36,145-155; tests/test_cmtf.py:44-50)
    loads = [f for fs in direct.Xs_factors for f in fs[1:]] if coupled else direct.X_factors[1:]
    for f in loads:
        np.testing.assert_allclose(norm(f, axis=0), 1, rtol=1e-12)
    assert np.all(np.diff(direct.R2Y) >= -1e-12)
    if not coupled:                                    # (a coupled fit's R2Xs need not increase -- tests/test_cmtf.py:27,39 -- nor be positive)
        assert np.all(np.diff(direct.R2X) >= -1e-12)
    s = np.abs(T).max()
    np.testing.assert_allclose(direct.transform(data), T, rtol=1e-4, atol=1e-5 * s)
    return direct, xcov


def test_cfg2_full_size_r10(api):
    X, Y = _full()
    direct, _ = _full_size_case(api, "configs[1] 65536x128x128 M=16", False, X, Y)
    # R2X bookkeeping against the literal reconstruction formula in one read of X (util.py:7-15 as tpls.py:115-117 calls it)
    assert abs(direct.R2X_literal(X) - direct.R2X[-1]) <= 1e-7


def test_cfg3_full_size_coupled_r10(api):
    X, Y, Xm = _full(matrix_block=512)
    direct, _ = _full_size_case(api, "configs[2] 65536x128x128 + 65536x512 coupled", True, [X, Xm], Y)
    assert direct.Xs_factors[0][0] is direct.Xs_factors[1][0]


def test_cfg4_full_size_nan30_r10(api):
    X, Y = _full(nan_fraction=0.3)
    direct, _ = _full_size_case(api, "configs[3] 65536x128x128 30% NaN", False, X, Y)
    assert direct.X_hasMiss
