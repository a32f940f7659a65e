"""Leave-one-out Q2Y with the reference's definition (cmtf_pls/validate.py:7-37).

The reference reads ``pls_tensor.original_X / original_Y``, which its own ``tPLS.fit`` never stores
(they are locals at tpls.py:74), so ``get_q2y`` fails there on any fitted model; the estimator here
keeps them.  One refit per held-out sample (validate.py:27-33); every refit runs on the GPU engine
with the fitted model's storage type, algorithm and backend.
"""
import numpy as np

from .tpls import tPLS


def get_q2y(pls_tensor):
    assert getattr(pls_tensor, "original_X", None) is not None, "PLS Tensor must be fit prior to calculating Q2Y"
    X = np.asarray(pls_tensor.original_X)
    Y = np.asarray(pls_tensor.original_Y)
    n = X.shape[0]
    refit = tPLS(pls_tensor.n_components, dtype=pls_tensor._dtype, device=pls_tensor._device,
                 backend=pls_tensor._backend, algorithm=pls_tensor._algorithm)
    Y_pred = np.zeros(Y.shape)
    Y_actual = np.zeros(Y.shape)
    keep = np.ones(n, dtype=bool)
    for i in range(n):                                   # LeaveOneOut().split(X, Y)     validate.py:24,27
        keep[i] = False
        refit.fit(X[keep], Y[keep])                      # validate.py:30
        Y_pred[i] = refit.predict(X[i:i + 1]).reshape(Y_pred[i].shape)   # validate.py:32
        Y_actual[i] = Y[i]
        keep[i] = True
    numerator = (Y_pred - Y_actual) ** 2                 # validate.py:35-37
    denominator = Y_actual ** 2
    return 1 - numerator.sum() / denominator.sum()
