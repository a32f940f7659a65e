"""NumPy float64 restatement of the reference's NIPALS hot path (TEST INFRASTRUCTURE).

Each function cites the reference lines (under /root/reference/) it follows.
The reference delegates five operations to tensorly 0.9.0 (requirements.lock:34), which is
not available offline; those are restated from tensorly's published definitions:

  unfold(t, m)          = reshape(moveaxis(t, m, 0), (t.shape[m], -1))            (C order)
  multi_mode_dot(X, vs) = contract every trailing mode of X with a vector
  outer([a, b, c])      = a[:,None,None] * b[None,:,None] * c[None,None,:]
  khatri_rao / fold     = column-wise Kronecker (first matrix slowest) / inverse of unfold
  parafac(Z, 1, init="svd", normalize_factors=True, tol)
                        = leading-left-singular-vector init of every unfolding (largest-|.| entry
                          made positive), ALS sweeps, stop when |d rec_error| < tol from the 2nd
                          sweep on, at most 100 sweeps.  For a matrix Z this is exactly the leading
                          singular pair (sigma > 0), the last mode's vector carrying the sign rule.

Nothing here is used by the product path (cmtf_pls_amd/); see oracle/__init__.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import reduce
from typing import List, Optional, Sequence

import numpy as np

__all__ = [
    "OracleFit", "calc_r2x", "cp_factors_to_tensor", "fit_tpls", "fit_ctpls", "mode0_contract",
    "masked_mode0_contract", "masked_score", "score_contract", "rank1_factors", "predict",
    "transform", "reconstruct", "nipals_inner_loop",
]


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def mode0_contract(X: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Z = X x_0 u.  Reference: ``np.einsum("i...,i...->...", X, u)`` tpls.py:83, cmtf.py:94."""
    return np.tensordot(u, X, axes=(0, 0))


def masked_mode0_contract(X: np.ndarray, u: np.ndarray, miss: Optional[np.ndarray] = None) -> np.ndarray:
    """NaN-aware mode-0 contraction.  Reference: ``miss_tensordot`` missingvals.py:7-20.

    Per trailing position c: (sum over observed i of X[i,c] u[i]) / n_obs(c) * I, and 0 when the
    column has no observation (missingvals.py:17-19).  Vectorised; same scaling order as the
    reference (divide by the count, then multiply by I)."""
    shape = X.shape
    if miss is None:
        miss = np.isnan(X)
    X2 = X.reshape(shape[0], -1)
    obs = ~miss.reshape(shape[0], -1)
    n_obs = obs.sum(axis=0)
    dots = np.where(obs, X2, 0.0).T @ u
    out = np.zeros(X2.shape[1])
    nz = n_obs > 0
    out[nz] = dots[nz] / n_obs[nz] * shape[0]
    return out.reshape(shape[1:])


def _kron_all(vecs: Sequence[np.ndarray]) -> np.ndarray:
    return reduce(np.kron, [np.asarray(v).ravel() for v in vecs])


def score_contract(X: np.ndarray, vecs: Sequence[np.ndarray]) -> np.ndarray:
    """t = X x_1 w_1 x_2 w_2 ...  Reference: ``multi_mode_dot(X, vecs, range(1, X.ndim))``
    tpls.py:97-99,139-141,162-164; cmtf.py:107-111.  tensorly contracts mode 1 first, then the
    next (now shifted) mode, and so on; the same order is kept here."""
    out = X
    for v in vecs:
        out = np.tensordot(out, np.asarray(v).ravel(), axes=(1, 0))
    return out


def masked_score(X: np.ndarray, vecs: Sequence[np.ndarray], miss: Optional[np.ndarray] = None) -> np.ndarray:
    """NaN-aware score contraction.  Reference: ``miss_mmodedot`` missingvals.py:23-38.

    Per row i: (sum over observed c of X[i,c] wkron[c]) / n_obs(i) * numel(wkron); a row with no
    observation yields 0/0 = NaN exactly like the reference (missingvals.py:37)."""
    shape = X.shape
    if miss is None:
        miss = np.isnan(X)
    X2 = X.reshape(shape[0], -1)
    obs = ~miss.reshape(shape[0], -1)
    wkron = _kron_all(vecs)
    n_obs = obs.sum(axis=1)
    dots = np.where(obs, X2, 0.0) @ wkron
    with np.errstate(divide="ignore", invalid="ignore"):
        return dots / n_obs * wkron.shape[0]


def _unfold(Z: np.ndarray, mode: int) -> np.ndarray:
    return np.reshape(np.moveaxis(Z, mode, 0), (Z.shape[mode], -1))


def _leading_left_singular(M: np.ndarray) -> np.ndarray:
    U, _, _ = np.linalg.svd(M, full_matrices=False)
    u = U[:, 0].copy()
    if u[np.argmax(np.abs(u))] < 0:
        u = -u
    return u


def rank1_factors(Z: np.ndarray, tol: float = 1e-8, n_iter_max: int = 100) -> List[np.ndarray]:
    """Unit-norm rank-1 factors of the cross-covariance tensor Z.

    Reference call sites: ``Z / norm(Z)`` for a vector (tpls.py:84, cmtf.py:98) and
    ``parafac(Z, 1, tol=tol, init="svd", normalize_factors=True)[1]`` otherwise (tpls.py:86-88,
    cmtf.py:100-102).  The parafac restatement is from tensorly 0.9.0's published algorithm
    (PARITY UNPINNED at value level; for a matrix Z the result is the leading singular pair up to
    the paired sign, which is mathematically pinned)."""
    Z = np.asarray(Z, dtype=float)
    if Z.ndim == 1:
        return [Z / np.linalg.norm(Z)]
    N = Z.ndim
    fac = [_leading_left_singular(_unfold(Z, m)) for m in range(N)]
    weight = 1.0
    norm_Z = np.linalg.norm(Z)
    errs: List[float] = []
    for sweep in range(n_iter_max):
        mttkrp = None
        for m in range(N):
            others = [fac[i] for i in range(N) if i != m]
            gram = weight * weight * float(np.prod([o @ o for o in others]))
            mttkrp = _unfold(Z, m) @ (_kron_all(others) * weight)
            fac[m] = mttkrp / gram
        fnorm2 = weight * weight * float(np.prod([f @ f for f in fac]))
        iprod = float(mttkrp @ fac[-1]) * weight
        errs.append(np.sqrt(abs(norm_Z**2 + fnorm2 - 2.0 * iprod)) / norm_Z)
        if sweep >= 1 and abs(errs[-2] - errs[-1]) < tol:
            break
        norms = [np.linalg.norm(f) for f in fac]
        weight *= float(np.prod(norms))
        fac = [f / n for f, n in zip(fac, norms)]
    return fac


def _outer_all(vecs: Sequence[np.ndarray]) -> np.ndarray:
    """tensorly ``outer``: rank-1 tensor of the given vectors (tpls.py:109,142,165)."""
    out = np.asarray(vecs[0]).ravel()
    for v in vecs[1:]:
        out = np.multiply.outer(out, np.asarray(v).ravel())
    return out


def cp_factors_to_tensor(factors: Sequence[np.ndarray]) -> np.ndarray:
    """Dense tensor of a CP factor list.  Reference: ``factors_to_tensor`` util.py:18-20
    (``factors[0] @ khatri_rao(rest).T`` then ``fold`` along mode 0)."""
    rank = factors[0].shape[1]
    kr = np.ones((1, rank))
    for f in factors[1:]:
        kr = (kr[:, None, :] * f[None, :, :]).reshape(-1, rank)
    return (factors[0] @ kr.T).reshape([f.shape[0] for f in factors])


def calc_r2x(X: np.ndarray, Xhat: np.ndarray) -> float:
    """Reference: ``calcR2X`` util.py:7-15."""
    if Xhat.ndim == 2 and X.ndim == 1:
        X = X.reshape(-1, 1)
    assert X.shape == Xhat.shape
    mask = np.isfinite(X)
    x_in = np.nan_to_num(X)
    top = np.linalg.norm(Xhat * mask - x_in) ** 2.0
    bottom = np.linalg.norm(x_in) ** 2.0
    return 1 - top / bottom


# --------------------------------------------------------------------------------------------
# fitted state
# --------------------------------------------------------------------------------------------
@dataclass
class OracleFit:
    """Everything the reference estimators expose after ``fit`` (tpls.py:44-71, cmtf.py:44-83)."""

    coupled: bool
    n_components: int
    block_shapes: List[tuple]
    y_shape: tuple
    T: np.ndarray                       # (I, R) X scores; shared by all blocks when coupled
    loadings: List[List[np.ndarray]]    # per block, per trailing mode: (dim, R)
    U: np.ndarray                       # (I, R) Y scores   = Y_factors[0]
    Q: np.ndarray                       # (M, R) Y loadings = Y_factors[1]
    coef: np.ndarray                    # (R, R) upper triangular
    r2x: List[np.ndarray]
    r2y: np.ndarray
    x_means: List[np.ndarray]
    y_mean: np.ndarray
    has_miss: List[bool]
    n_iter: List[int] = field(default_factory=list)   # inner iterations executed per component

    # reference-style views ------------------------------------------------------------------
    def x_factors(self, block: int = 0) -> List[np.ndarray]:
        return [self.T] + self.loadings[block]

    @property
    def y_factors(self) -> List[np.ndarray]:
        return [self.U, self.Q]


def _center(blocks, Y):
    x_means = [np.nanmean(X, axis=0) for X in blocks]          # tpls.py:66, cmtf.py:74
    y_mean = np.nanmean(Y, axis=0)                              # tpls.py:67, cmtf.py:75
    return [X - m for X, m in zip(blocks, x_means)], Y - y_mean, x_means, y_mean


def _project(fit: OracleFit, blocks: Sequence[np.ndarray]) -> np.ndarray:
    """Sequential project-and-deflate of new samples (tpls.py:128-142,151-165; cmtf.py:143-177,
    180-210).  Returns the (I', R) X scores; inputs are not modified."""
    work = [np.array(X, dtype=float, copy=True) for X in blocks]
    miss = [np.isnan(X) for X in work]
    any_miss = [bool(m.any()) for m in miss]
    for b, X in enumerate(work):
        if tuple(fit.block_shapes[b][1:]) != tuple(X.shape[1:]):
            raise ValueError(f"block {b}: trained on {fit.block_shapes[b]}, got {X.shape}")
        work[b] = X - fit.x_means[b]
    scores = np.zeros((work[0].shape[0], fit.n_components))
    for a in range(fit.n_components):
        per_block = []
        for b, X in enumerate(work):
            vecs = [L[:, a] for L in fit.loadings[b]]
            per_block.append(masked_score(X, vecs, miss[b]) if any_miss[b] else score_contract(X, vecs))
        scores[:, a] = np.average(per_block, axis=0) if fit.coupled else per_block[0]
        for b in range(len(work)):
            work[b] = work[b] - _outer_all([scores[:, a]] + [L[:, a] for L in fit.loadings[b]])
    return scores


def predict(fit: OracleFit, blocks) -> np.ndarray:
    """Reference: ``tPLS.predict`` tpls.py:122-143 / ``ctPLS.predict`` cmtf.py:142-177."""
    blocks = blocks if isinstance(blocks, (list, tuple)) else [blocks]
    return _project(fit, blocks) @ fit.coef @ fit.Q.T + fit.y_mean


def transform(fit: OracleFit, blocks, Y: Optional[np.ndarray] = None):
    """Reference: ``tPLS.transform`` tpls.py:145-186 / ``ctPLS.transform`` cmtf.py:179-231."""
    blocks = blocks if isinstance(blocks, (list, tuple)) else [blocks]
    x_scores = _project(fit, blocks)
    if Y is None:
        return x_scores
    Y = np.array(Y, dtype=float, copy=True)
    if Y.ndim not in (1, 2):
        raise ValueError("Only a matrix (2-mode tensor) Y is allowed.")
    if Y.ndim == 1:
        Y = Y.reshape(-1, 1)
    if tuple(fit.y_shape[1:]) != tuple(Y.shape[1:]):
        raise ValueError(f"Training Y has shape {fit.y_shape}, while the new Y has shape {Y.shape}")
    Y = Y - fit.y_mean
    y_scores = np.zeros((Y.shape[0], fit.n_components))
    for a in range(fit.n_components):
        y_scores[:, a] = Y @ fit.Q[:, a]
        Y = Y - x_scores @ fit.coef[:, [a]] @ fit.Q[:, [a]].T
    return x_scores, y_scores


def reconstruct(fit: OracleFit, block: int = 0) -> np.ndarray:
    """Reference: ``X_reconstructed`` tpls.py:188-189 / ``Xs_reconstructed`` cmtf.py:233-237."""
    return cp_factors_to_tensor(fit.x_factors(block)) + fit.x_means[block]


# --------------------------------------------------------------------------------------------
# the NIPALS fit (shared by tPLS and ctPLS; a tPLS fit is the one-block, non-averaged case)
# --------------------------------------------------------------------------------------------
def _nipals(blocks, Y, n_components, tol, max_iter, coupled) -> OracleFit:
    originals = [np.array(X, dtype=float, copy=True) for X in blocks]
    Y_in = np.array(Y, dtype=float, copy=True)
    for X in originals:
        assert X.shape[0] == Y_in.shape[0]                       # tpls.py:46, cmtf.py:49
    assert Y_in.ndim <= 2, "Only a matrix (2-mode tensor) Y is acceptable."
    Y2 = Y_in.reshape(-1, 1) if Y_in.ndim == 1 else Y_in          # tpls.py:48-49

    R = n_components
    n_samples, n_resp = Y2.shape
    has_miss = [bool(np.isnan(X).any()) for X in originals]      # tpls.py:61, cmtf.py:77
    miss = [np.isnan(X) for X in originals]                      # tpls.py:64, cmtf.py:80-82
    work, Yc, x_means, y_mean = _center(originals, Y2)
    Yc = Yc.copy()

    fit = OracleFit(
        coupled=coupled, n_components=R, block_shapes=[X.shape for X in originals], y_shape=Y2.shape,
        T=np.zeros((n_samples, R)),
        loadings=[[np.zeros((d, R)) for d in X.shape[1:]] for X in originals],
        U=np.zeros((n_samples, R)), Q=np.zeros((n_resp, R)), coef=np.zeros((R, R)),
        r2x=[np.zeros(R) for _ in originals], r2y=np.zeros(R),
        x_means=x_means, y_mean=y_mean, has_miss=has_miss,
    )

    for a in range(R):
        old_u = np.full(n_samples, np.inf)                       # tpls.py:77
        fit.U[:, a] = Yc[:, 0]                                   # tpls.py:78
        executed = 0
        for _ in range(max_iter):                                # tpls.py:79 / cmtf.py:91
            executed += 1
            per_block = []
            for b, X in enumerate(work):
                Z = (masked_mode0_contract(X, fit.U[:, a], miss[b]) if has_miss[b]
                     else mode0_contract(X, fit.U[:, a]))        # tpls.py:80-83
                for m, f in enumerate(rank1_factors(Z, tol)):    # tpls.py:84-90
                    fit.loadings[b][m][:, a] = np.asarray(f).ravel()
                vecs = [L[:, a] for L in fit.loadings[b]]
                per_block.append(masked_score(X, vecs, miss[b]) if has_miss[b]
                                 else score_contract(X, vecs))   # tpls.py:92-99
            fit.T[:, a] = np.average(per_block, axis=0) if coupled else per_block[0]   # cmtf.py:120
            q = Yc.T @ fit.T[:, a]                               # tpls.py:100
            q = q / np.linalg.norm(q)                            # tpls.py:101
            fit.Q[:, a] = q
            fit.U[:, a] = Yc @ q                                 # tpls.py:102
            if np.linalg.norm(old_u - fit.U[:, a]) < tol:        # tpls.py:103
                break
            old_u = fit.U[:, a].copy()                           # tpls.py:107
        fit.n_iter.append(executed)

        for b in range(len(work)):                               # tpls.py:109 / cmtf.py:130-131
            work[b] = work[b] - _outer_all([fit.T[:, a]] + [L[:, a] for L in fit.loadings[b]])
            fit.r2x[b][a] = calc_r2x(originals[b] - x_means[b],
                                     cp_factors_to_tensor(fit.x_factors(b)))   # tpls.py:115-117
        fit.coef[:, a] = np.linalg.lstsq(fit.T, fit.U[:, a], rcond=-1)[0]     # tpls.py:110-112
        Yc = Yc - fit.T @ fit.coef[:, [a]] @ fit.Q[:, [a]].T                   # tpls.py:113
        y_hat = predict(fit, originals if coupled else originals[0])
        fit.r2y[a] = calc_r2x(Y_in - y_mean, y_hat - y_mean)                   # tpls.py:118-120
    return fit


def nipals_inner_loop(Xc: np.ndarray, Yc: np.ndarray, n_iter: int, tol: float = 1e-8):
    """The bare inner loop of one component on already-centred, NaN-free data (tpls.py:77-107),
    run for exactly ``n_iter`` iterations: contraction, rank-1 extraction, score, Y update,
    convergence norm.  This is the unit bench.py's ``cpu_baseline`` times.  Returns (t, w, q, u, du)."""
    u = Yc[:, 0].copy()
    old_u = np.full(Yc.shape[0], np.inf)
    t = w = q = None
    du = np.inf
    for _ in range(n_iter):
        Z = mode0_contract(Xc, u)                                 # tpls.py:83
        w = rank1_factors(Z, tol)                                 # tpls.py:84-90
        t = score_contract(Xc, w)                                 # tpls.py:97-99
        q = Yc.T @ t                                              # tpls.py:100
        q = q / np.linalg.norm(q)                                 # tpls.py:101
        u = Yc @ q                                                # tpls.py:102
        du = np.linalg.norm(old_u - u)                            # tpls.py:103
        old_u = u.copy()                                          # tpls.py:107
    return t, w, q, u, du


def fit_tpls(X, Y, n_components, tol=1e-8, max_iter=100) -> OracleFit:
    """Reference: ``tPLS.fit`` tpls.py:73-120 (with ``preprocess`` tpls.py:44-71)."""
    return _nipals([X], Y, n_components, tol, max_iter, coupled=False)


def fit_ctpls(Xs, Y, n_components, tol=1e-8, max_iter=100) -> OracleFit:
    """Reference: ``ctPLS.fit`` cmtf.py:85-140 (with ``preprocess`` cmtf.py:44-83)."""
    assert isinstance(Xs, list)                                   # cmtf.py:46
    for X in Xs:
        assert X.ndim >= 2                                        # cmtf.py:50
    return _nipals(Xs, Y, n_components, tol, max_iter, coupled=True)
