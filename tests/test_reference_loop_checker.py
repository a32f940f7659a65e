"""The torch-only reference-loop checker (tests/reference_loop_torch.py) against the oracle's fits, on the CPU.

Two independent restatements of the reference's component loop meet here: the NumPy oracle (its `parafac` restated from
tensorly's definitions) and the checker (LAPACK SVD / stationarity conditions through torch).  Every converged component
of an oracle fit must be a fixed point of the checker's pass -- plain, 30 % NaN, coupled tensor + matrix, order-4 X --
and a wrong factor must be rejected (so that the GPU tests built on the checker can fail)."""
import numpy as np
import pytest
import torch

import oracle as O
from reference_loop_torch import check_fit, format_records


def _run(fit, blocks, y, **kw):
    loadings = [list(L) for L in fit.loadings]
    return check_fit([torch.from_numpy(b) for b in blocks], torch.from_numpy(y.reshape(len(y), -1)), fit.T, loadings, fit.U, fit.Q,
                     fit.coef, fit.n_iter, fit.r2x, fit.r2y, **kw)


def test_checker_accepts_the_oracles_tpls_fit():
    x, y, _ = O.import_synthetic((300, 12, 10), 5, 4, error=0.1, seed=3)
    fit = O.fit_tpls(x, y, 4)
    recs = _run(fit, [x], y, rtol=1e-6, min_checked=3)
    print(format_records("oracle tPLS (300,12,10)", recs))


def test_checker_accepts_the_oracles_masked_fit():
    x, y, _ = O.import_synthetic((300, 12, 10), 5, 4, error=0.1, seed=4)
    x[np.random.default_rng(5).random(x.shape) < 0.3] = np.nan
    fit = O.fit_tpls(x, y, 4)
    _run(fit, [x], y, rtol=1e-6, min_checked=3)


def test_checker_accepts_the_oracles_coupled_fit():
    x, y, cp = O.import_synthetic((300, 12, 10), 5, 4, error=0.1, seed=6)
    xm = cp.factors[0] @ np.random.default_rng(7).normal(size=(20, 4)).T + 0.1 * np.random.default_rng(8).normal(size=(300, 20))
    xm[np.random.default_rng(9).random(xm.shape) < 0.1] = np.nan
    fit = O.fit_ctpls([x, xm], y, 4)
    _run(fit, [x, xm], y, rtol=1e-6, min_checked=3)


def test_checker_accepts_the_oracles_order4_fit():
    x, y, _ = O.import_synthetic((120, 8, 7, 6), 4, 3, error=0.1, seed=10)
    fit = O.fit_tpls(x, y, 3)
    recs = _run(fit, [x], y, rtol=1e-6, min_checked=2, stationarity_rtol=1e-4)
    print(format_records("oracle tPLS order 4", recs))


@pytest.mark.parametrize("what", ["T", "W", "Q", "coef", "R2X"])
def test_checker_rejects_a_wrong_factor(what):
    x, y, _ = O.import_synthetic((200, 10, 8), 4, 3, error=0.1, seed=11)
    fit = O.fit_tpls(x, y, 3)
    if what == "T":
        fit.T[:, 1] *= 1 + 1e-4
    elif what == "W":
        w = fit.loadings[0][1]
        w[:, 0] = np.roll(w[:, 0], 1)
    elif what == "Q":
        fit.Q[0, 2] += 1e-3
    elif what == "coef":
        fit.coef[0, 1] *= 1.01
    else:
        fit.r2x[0][1] += 1e-3
    with pytest.raises(AssertionError):
        _run(fit, [x], y, rtol=1e-6)
