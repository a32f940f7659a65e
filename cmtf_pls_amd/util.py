"""Host-side helpers with the reference's names (cmtf_pls/util.py:7-20).  Small-data NumPy code:
the fit path never calls them (R2X / R2Y come out of the deflation sweeps, see engine.py)."""
import numpy as np


def calcR2X(X, Xhat):
    """1 - |Xhat*mask - X|^2 / |X|^2 over finite entries of X (util.py:7-15)."""
    if (Xhat.ndim == 2) and (X.ndim == 1):
        X = X.reshape(-1, 1)
    assert X.shape == Xhat.shape
    finite = np.isfinite(X)
    x0 = np.where(finite, X, 0.0)
    resid = np.where(finite, Xhat, 0.0) - x0
    return 1 - float(np.sum(resid * resid)) / float(np.sum(x0 * x0))


def factors_to_tensor(factors):
    """Dense tensor of CP factors [(d0,R), (d1,R), ...] (util.py:18-20)."""
    letters = "abcdefghijklmnopq"[: len(factors)]
    spec = ",".join(f"{c}z" for c in letters) + "->" + letters
    return np.einsum(spec, *factors)
