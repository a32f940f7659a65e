#!/usr/bin/env python3
"""cmtf_pls_amd in five minutes (needs one MI355X and the built library: python -c "import __graft_entry__ as g; g.build()").

The estimators keep the reference's API (cmtf_pls.tpls.tPLS / cmtf_pls.cmtf.ctPLS): NumPy in, NumPy out."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import ctPLS, tPLS                                   # noqa: E402
from cmtf_pls_amd.synthetic import import_synthetic, make_synthetic_test   # noqa: E402
from cmtf_pls_amd.validate import get_q2y                               # noqa: E402


def main():
    # 1. the reference's own synthetic recipe (synthetic.py:37-79): X (200, 10, 8), Y (200, 4), 3 latent factors
    X, Y, cp = import_synthetic((200, 10, 8), 4, 3, error=0.1)
    pls = tPLS(3)                                                       # float64 in -> float64 storage, the reference's numerics
    pls.fit(X, Y)
    print("R2X", np.round(pls.R2X, 4), "R2Y", np.round(pls.R2Y, 4), "inner iterations", pls.n_iter_)

    # 2. new samples: transform (scores) and predict (responses), one pass over X_new
    X_new, Y_new, _ = make_synthetic_test(cp, 50, error=0.1)
    scores = pls.transform(X_new)
    Y_hat = pls.predict(X_new)
    print("held-out R2Y", round(1 - ((Y_hat - Y_new) ** 2).sum() / ((Y_new - Y_new.mean(0)) ** 2).sum(), 4), "scores", scores.shape)

    # 3. leave-one-out Q2Y (validate.py): all 200 refits in one launch, one workgroup per fold
    print("Q2Y (leave-one-out)", round(get_q2y(pls), 6))

    # 4. missing values are NaNs in X; the masked contractions of missingvals.py run on the device
    Xm = X.copy()
    Xm[np.random.default_rng(0).random(X.shape) < 0.2] = np.nan
    plsm = tPLS(3)
    plsm.fit(Xm, Y)
    miss = np.isnan(Xm)
    rec = plsm.X_reconstructed()
    print("imputation R2 at the missing entries", round(1 - ((rec[miss] - X[miss]) ** 2).sum() / (X[miss] ** 2).sum(), 4))

    # 5. coupled blocks sharing the sample mode: a tensor and a matrix
    X2, Y2, cp2 = import_synthetic((200, 10, 8), 4, 3, error=0.1, seed=1)
    Xmat = cp2.factors[0] @ np.random.default_rng(1).normal(size=(30, 3)).T
    c = ctPLS(3)
    c.fit([X2, Xmat], Y2)
    print("ctPLS R2Y", np.round(c.R2Y, 4), "shared scores", c.factor_T.shape)

    # 6. benchmark-style use: float32 storage, the cross-covariance form (same iterates, one X read per component)
    fast = tPLS(3, dtype="float32", algorithm="xcov")
    fast.fit(X, Y)
    print("f32 / xcov: max |T - T_f64| relative", float(np.abs(fast.X_factors[0] - pls.X_factors[0]).max() / np.abs(pls.X_factors[0]).max()))


if __name__ == "__main__":
    main()
