"""The two missing-value contractions of the NIPALS loop under the reference's names and argument meaning
(cmtf_pls/missingvals.py:7-38), computed by the HIP kernels the fit itself uses:

* ``miss_tensordot(X, u, missX=None)``  (missingvals.py:7-20)  -- ``np.einsum("i...,i...->...", X, u)`` over the observed
  samples of every column, rescaled by I / n_observed; a column without observations gives 0.
  = ``cmtfpls_mode0_contract_*`` (masked) + ``cmtfpls_colscale_f64``.
* ``miss_mmodedot(X, facs, missX=None)``  (missingvals.py:23-38) -- ``multi_mode_dot(X, facs, range(1, X.ndim))`` over the
  observed entries of every sample, rescaled by prod(dims[1:]) / n_observed; a sample without observations gives NaN
  (0 / 0 in the reference).  = ``cmtfpls_score_*`` with the per-row counts.

NumPy in -> NumPy out; torch tensors in -> a device tensor out.  There is no CPU path: without a ROCm GPU (or without
libcmtfpls.so) the backend constructor raises.  Arithmetic: X in its own storage type (float32 stays float32, anything
else is float64), every product and sum in float64.
"""
from __future__ import annotations

from functools import reduce
from typing import Optional

import numpy as np
import torch

from .tpls import to_device_copy

_BACKENDS = {}


def _backend(device):
    from .backend import HipBackend
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    key = str(dev)
    if key not in _BACKENDS:
        _BACKENDS[key] = HipBackend(dev)          # raises without a GPU / without the library: no CPU fallback
    return _BACKENDS[key]


def _device_of(*arrays, device=None):
    if device is not None:
        return device
    for a in arrays:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    return None


def _staged(X, missX, be):
    """(X2, stray): X as a fresh (I, P) device tensor with NaN exactly at the missing positions, and -- only when an
    explicit mask leaves NaNs of X unflagged -- the (I, P) bool tensor of those strays (the reference multiplies them
    in, so whatever they touch comes out NaN)."""
    I = X.shape[0]
    dtype = torch.float32 if X.dtype in (np.float32, torch.float32) else torch.float64
    X2 = to_device_copy(X, dtype, be.device).reshape(I, -1)
    if missX is None:
        return X2, None
    m = torch.as_tensor(np.ascontiguousarray(missX) if not isinstance(missX, torch.Tensor) else missX).to(be.device).reshape(I, -1).bool()
    assert m.shape == X2.shape
    stray = torch.isnan(X2) & ~m
    X2.masked_fill_(m, float("nan"))
    return X2, (stray if bool(stray.any().item()) else None)


def _vec(v, be) -> torch.Tensor:
    return to_device_copy(v, torch.float64, be.device).reshape(-1)


def _out(t: torch.Tensor, like):
    return t if isinstance(like, torch.Tensor) else t.cpu().numpy()


def miss_tensordot(X, u, missX=None, device=None):
    """missingvals.py:7-20.  X (I, d1, ...), u (I,), missX: bool array of X's shape or (I, prod(d)) -> array (d1, ...)."""
    Xdim = tuple(X.shape)
    assert Xdim[0] == u.shape[0]
    be = _backend(_device_of(X, u, device=device))
    X2, stray = _staged(X, missX, be)
    _, colcnt = be.colstats(X2)
    Z = be.mode0_contract(X2, _vec(u, be), True)
    be.colscale(Z, colcnt, float(Xdim[0]))                         # x I / n_observed; 0 where nothing was observed
    if stray is not None:
        Z[stray.any(0)] = float("nan")
    return _out(Z.reshape(Xdim[1:]), X)


def miss_mmodedot(X, facs, missX=None, device=None):
    """missingvals.py:23-38.  facs: one vector per trailing mode of X -> array (I,)."""
    Xdim = tuple(X.shape)
    assert len(facs) == len(Xdim) - 1 and all(Xdim[i + 1] == f.shape[0] for i, f in enumerate(facs))
    be = _backend(_device_of(X, *facs, device=device))
    X2, stray = _staged(X, missX, be)
    I, P = X2.shape
    # the kernels take the loading factored as wA (first trailing mode) (x) wB (Kronecker product of the others, C order)
    vs = [_vec(f, be) for f in facs]
    if len(vs) == 1:
        A, wA, wB = 1, torch.ones(1, dtype=torch.float64, device=be.device), vs[0]
    else:
        A, wA = Xdim[1], vs[0]
        wB = reduce(lambda a, b: be.kron(a, b, be.empty(a.numel() * b.numel())), vs[1:])
    rowcnt, _ = be.center(X2, be.zeros(P), True)                   # observed entries per sample (x - 0 leaves X as it is)
    t = be.score(X2, A, P // A, wA, wB, rowcnt, be.empty(I))
    if stray is not None:
        t[stray.any(1)] = float("nan")
    return _out(t, X)
