"""Leave-one-out Q2Y with the reference's definition (cmtf_pls/validate.py:7-37).

The reference reads ``pls_tensor.original_X / original_Y``, which its own ``tPLS.fit`` never stores
(they are locals at tpls.py:74), so ``get_q2y`` fails there on any fitted model; the estimator here
keeps them.  One refit per held-out sample (validate.py:27-33).  On the GPU the folds run side by side, a workgroup per
fold doing the whole fit for it with the fold's means down-dated from shared column sums: ``cmtfpls_loo_tpls_f64`` when the
fold's vectors fit the LDS (min(J, K) <= 64), ``cmtfpls_loo_xcov_f64`` beyond (min(J, K) <= 256: the fold's NIPALS loop on its
cross-covariance, Gram squarings on the matrix cores) -- X of order 2 or 3 without missing values, M <= 64 / R <= 16 (LDS form), M <= 128 / R <= 64 (xcov form).  Anything else
refits once per fold on the regular engine with the fitted model's storage type, algorithm and backend.  Which form ran is
recorded on the model (``q2y_report_``).
"""
import numpy as np

from .tpls import tPLS


def loo_predictions(pls_tensor, tol: float = 1e-8, max_iter: int = 100):
    """Y_pred[i] = prediction for sample i by the model refitted without it (validate.py:24-33), all folds in one
    launch; None when the device form does not apply (see get_q2y)."""
    import torch

    X = pls_tensor.original_X
    Y = pls_tensor.original_Y
    be = pls_tensor._get_engine().be
    if not hasattr(be, "loo_tpls") or X.ndim not in (2, 3):
        return None
    Xh = X.detach().cpu().numpy() if isinstance(X, torch.Tensor) else np.asarray(X)
    Yh = Y.detach().cpu().numpy() if isinstance(Y, torch.Tensor) else np.asarray(Y)
    if np.isnan(Xh).any() or np.isnan(Yh).any():
        return None
    I = Xh.shape[0]
    A, B = (1, Xh.shape[1]) if Xh.ndim == 2 else (Xh.shape[1], Xh.shape[2])
    with torch.cuda.device(be.device):
        Xd = torch.from_numpy(np.ascontiguousarray(Xh.reshape(I, -1), dtype=np.float64)).to(be.device)
        Yd = torch.from_numpy(np.ascontiguousarray(Yh.reshape(I, -1), dtype=np.float64)).to(be.device)
        out = be.loo_tpls(Xd, Yd, A, B, pls_tensor.n_components, tol, max_iter)
        if out is None:
            return None
        pls_tensor.q2y_report_ = {"form": {"lds": "all folds in one launch, a workgroup per fold, vectors in LDS (cmtfpls_loo_tpls_f64)",
                                           "xcov": "a workgroup per fold on the fold's cross-covariance (cmtfpls_loo_xcov_f64)"}[out[2]],
                                  "folds": int(I), "n_iter_total": int(out[1].sum().item())}
        return out[0].cpu().numpy().reshape(Yh.shape)


def get_q2y(pls_tensor, device_folds: bool = True):
    assert getattr(pls_tensor, "original_X", None) is not None, "PLS Tensor must be fit prior to calculating Q2Y"
    X = np.asarray(pls_tensor.original_X) if not hasattr(pls_tensor.original_X, "cpu") else pls_tensor.original_X.cpu().numpy()
    Y = np.asarray(pls_tensor.original_Y) if not hasattr(pls_tensor.original_Y, "cpu") else pls_tensor.original_Y.cpu().numpy()
    n = X.shape[0]
    Y_pred = loo_predictions(pls_tensor) if device_folds else None
    Y_actual = Y.astype(float)
    if Y_pred is None:
        why = ("device folds switched off" if not device_folds else
               "order > 3, missing values, min(J, K) > 256, M > 128 (or an M x M Gram beyond the LDS) or R > 64: outside both workgroup-per-fold kernels")
        pls_tensor.q2y_report_ = {"form": "one refit per fold on the regular engine", "folds": int(n), "why": why}
        refit = tPLS(pls_tensor.n_components, dtype=pls_tensor._dtype, device=pls_tensor._device,
                     backend=pls_tensor._backend, algorithm=pls_tensor._algorithm)
        Y_pred = np.zeros(Y.shape)
        keep = np.ones(n, dtype=bool)
        for i in range(n):                                   # LeaveOneOut().split(X, Y)     validate.py:24,27
            keep[i] = False
            refit.fit(X[keep], Y[keep])                      # validate.py:30
            Y_pred[i] = refit.predict(X[i:i + 1]).reshape(Y_pred[i].shape)   # validate.py:32
            keep[i] = True
    numerator = (Y_pred - Y_actual) ** 2                     # validate.py:35-37
    denominator = Y_actual ** 2
    return 1 - numerator.sum() / denominator.sum()
