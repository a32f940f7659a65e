"""NIPALS engine: the component -> iteration -> block loop of tPLS.fit / ctPLS.fit
(reference cmtf_pls/tpls.py:73-120, cmtf_pls/cmtf.py:85-140) driven over device-resident data.

Layout: each X block lives on the GPU as a C-order (I_local, P) matrix (its mode-0 unfolding, a
free view), f32 or f64; Y, the scores T/U and every reduced quantity are f64.  With several
processes (one per GPU) the SAMPLE mode is sharded: every rank holds I_local rows of every block
and of Y, the loadings are replicated, and the only communication is an all-reduce(sum) of
 - Z (P doubles) and Y^T t (M doubles) and |du|^2 (1 double) per NIPALS iteration,
 - T^T[T|u] and two squared norms per component, column sums/counts once per fit.
All ranks run the identical rank-1 extraction on the identical all-reduced Z, so the loadings stay
bit-identical without being communicated.

Exact identities used instead of extra X passes (all checked against the oracle in tests/):
 - R2X[a] = 1 - |X_{a+1}|^2 / |X_c|^2 over observed entries, because X_c - factors_to_tensor(...)
   IS the deflated tensor (util.py:7-20 with tpls.py:109,115-117): by-product of the deflation sweep;
 - predict(original_X) - Y_mean = T coef Q^T and Y_c - T coef Q^T = Y_{a+1} (tpls.py:113,118-120,
   133-143), so R2Y[a] = 1 - |Y_{a+1}|^2 / |Y_c|^2: by-product of the Y deflation.

The engine only talks to a *backend* object (cmtf_pls_amd.backend.HipBackend in the product); the
tests inject a NumPy backend to exercise the sharded control flow under gloo on CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch


class Comm:
    """Sample-mode data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm)."""

    def __init__(self, group=None, enabled: Optional[bool] = None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        on = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t


class _NoComm:
    world, rank = 1, 0

    def allreduce(self, t):
        return t


def split_trailing(shape: Sequence[int]):
    """(A, B) with A*B = prod(shape[1:]): wA spans the first trailing mode, wB the rest."""
    trailing = list(shape[1:])
    if len(trailing) == 0:
        raise ValueError("X needs at least one trailing mode")
    if len(trailing) == 1:
        return 1, int(trailing[0])
    return int(trailing[0]), int(np.prod(trailing[1:]))


@dataclass
class BlockState:
    shape: tuple                     # local shape (I_local, d1, d2, ...)
    A: int
    B: int
    mean: torch.Tensor               # (P,) f64
    has_miss: bool
    colcnt: Optional[torch.Tensor]   # (P,) global observation counts (masked blocks)
    rowcnt: Optional[torch.Tensor]   # (I_local,)
    ssq0: float
    loadings: List[torch.Tensor] = field(default_factory=list)   # per trailing mode: (dim, R) f64
    r2x: Optional[np.ndarray] = None


@dataclass
class FitState:
    coupled: bool
    n_components: int
    blocks: List[BlockState]
    T: torch.Tensor                  # (I_local, R)
    U: torch.Tensor                  # (I_local, R)
    Q: torch.Tensor                  # (M, R)
    coef: np.ndarray                 # (R, R) host
    r2y: np.ndarray
    y_mean: torch.Tensor
    n_iter: List[int]
    n_samples_total: int


class NipalsEngine:
    def __init__(self, backend, comm=None):
        self.be = backend
        self.comm = comm if comm is not None else _NoComm()

    # ------------------------------------------------------------------------------------
    def _prepare_block(self, X: torch.Tensor, n_total: int) -> BlockState:
        """tpls.py:61-71: NaN statistics, nanmean over samples, centring (in place on the copy)."""
        be, comm = self.be, self.comm
        I = X.shape[0]
        X2 = X.view(I, -1)
        P = X2.shape[1]
        A, B = split_trailing(X.shape)
        colsum, colcnt = be.colstats(X2)
        comm.allreduce(colsum)
        comm.allreduce(colcnt)
        mean = colsum / colcnt                                  # nanmean; 0/0 -> NaN like numpy
        has_miss = bool((colcnt.sum() < float(n_total) * P - 0.5).item())
        rowcnt, ssq0 = be.center(X2, mean, has_miss)
        comm.allreduce(ssq0)
        return BlockState(shape=tuple(X.shape), A=A, B=B, mean=mean, has_miss=has_miss,
                          colcnt=colcnt if has_miss else None, rowcnt=rowcnt, ssq0=float(ssq0.item()))

    def _rank1(self, blk: BlockState, Z: torch.Tensor, wA: torch.Tensor, wB: torch.Tensor) -> None:
        """tpls.py:84-90: Z / norm(Z) for a vector, leading singular pair for a matrix."""
        if len(blk.shape) > 3:
            raise NotImplementedError("X blocks of order >= 4 (cross-covariance tensor of order >= 3) are not built yet")
        if len(blk.shape) == 2:
            wB.copy_(Z)
            self.be.normalize(wB)
            wA.fill_(1.0)
        else:
            self.be.rank1(Z, blk.A, blk.B, wA, wB)

    # ------------------------------------------------------------------------------------
    def fit(self, Xs: List[torch.Tensor], Y: torch.Tensor, n_components: int, tol: float, max_iter: int,
            coupled: bool, verbose: int = 0) -> FitState:
        """Xs: device copies (will be centred and deflated in place); Y: (I_local, M) f64 copy."""
        be, comm = self.be, self.comm
        R = n_components
        I, M = Y.shape
        n_tot = torch.tensor([float(I)], dtype=torch.float64, device=Y.device)
        comm.allreduce(n_tot)
        n_total = int(round(float(n_tot.item())))

        blocks = [self._prepare_block(X, n_total) for X in Xs]
        nb = len(blocks)
        ysum, ycnt = be.colstats(Y)
        comm.allreduce(ysum)
        comm.allreduce(ycnt)
        y_mean = ysum / ycnt                                     # tpls.py:67
        _, ssqy0 = be.center(Y, y_mean, False)
        comm.allreduce(ssqy0)
        ssqy0 = float(ssqy0.item())

        T = be.zeros(I, R)
        U = be.zeros(I, R)
        Q = be.zeros(M, R)
        coef = np.zeros((R, R))
        r2y = np.zeros(R)
        for blk in blocks:
            dims = blk.shape[1:]
            blk.loadings = [be.zeros(d, R) for d in dims]
            blk.r2x = np.zeros(R)
        wA = [be.empty(blk.A) for blk in blocks]
        wB = [be.empty(blk.B) for blk in blocks]
        Zs = [be.empty(blk.A * blk.B) for blk in blocks]
        Ts = be.empty(nb, I)
        t = be.empty(I)
        u = be.empty(I)
        u_new = be.empty(I)
        q = be.empty(M)
        n_iter: List[int] = []

        for a in range(R):
            u.copy_(Y[:, 0])                                     # tpls.py:78
            executed = 0
            for it in range(max_iter):                           # tpls.py:79
                executed += 1
                for b, (blk, X) in enumerate(zip(blocks, Xs)):
                    X2 = X.view(I, -1)
                    be.mode0_contract(X2, u, blk.has_miss, out=Zs[b])           # tpls.py:80-83
                    comm.allreduce(Zs[b])
                    if blk.has_miss:
                        be.colscale(Zs[b], blk.colcnt, n_total)                 # missingvals.py:17-19
                    self._rank1(blk, Zs[b], wA[b], wB[b])                       # tpls.py:84-90
                    be.score(X2, blk.A, blk.B, wA[b], wB[b], blk.rowcnt if blk.has_miss else None, Ts[b])   # tpls.py:92-99
                if coupled:
                    be.scores_mean(Ts, t)                                       # cmtf.py:120
                else:
                    t.copy_(Ts[0])
                qraw = be.gram_tn(Y, t).view(-1)                                # tpls.py:100
                comm.allreduce(qraw)
                q.copy_(qraw)
                be.normalize(q)                                                 # tpls.py:101
                du2 = be.rowdot(Y, q, u_new, u if it > 0 else None)             # tpls.py:102
                u, u_new = u_new, u
                if it > 0:
                    comm.allreduce(du2)
                    if math.sqrt(float(du2.item())) < tol:                      # tpls.py:103
                        if verbose:
                            print("Comp {}: converged after {} iterations".format(a, it))
                        break
            n_iter.append(executed)

            T[:, a].copy_(t)
            U[:, a].copy_(u)
            Q[:, a].copy_(q)
            ssqs = []
            for b, (blk, X) in enumerate(zip(blocks, Xs)):
                if len(blk.shape) == 2:
                    blk.loadings[0][:, a].copy_(wB[b])
                else:
                    blk.loadings[0][:, a].copy_(wA[b])
                    blk.loadings[1][:, a].copy_(wB[b])
                ssqs.append(be.deflate(X.view(I, -1), blk.A, blk.B, t, wA[b], wB[b]))   # tpls.py:109
            # inner regression: coef_[:, a] = lstsq(T, u) with columns > a still zero (tpls.py:110-112)
            Ta = T[:, : a + 1]
            G = be.gram_tn(Ta, Ta)
            g = be.gram_tn(Ta, u)
            packed = torch.cat([G.reshape(-1), g.reshape(-1)] + [s.reshape(-1) for s in ssqs])
            comm.allreduce(packed)
            host = packed.cpu().numpy()
            k = a + 1
            Gh, gh = host[: k * k].reshape(k, k), host[k * k: k * k + k]
            bh = np.linalg.lstsq(Gh, gh, rcond=None)[0]
            coef[:k, a] = bh
            for b, blk in enumerate(blocks):
                blk.r2x[a] = 1.0 - host[k * k + k + b] / blk.ssq0                # tpls.py:115-117
            b_dev = torch.from_numpy(np.ascontiguousarray(bh)).to(Y.device)
            ssqy = be.y_deflate(Y, T, k, b_dev, q)                               # tpls.py:113
            comm.allreduce(ssqy)
            r2y[a] = 1.0 - float(ssqy.item()) / ssqy0                            # tpls.py:118-120

        return FitState(coupled=coupled, n_components=R, blocks=blocks, T=T, U=U, Q=Q, coef=coef, r2y=r2y,
                        y_mean=y_mean, n_iter=n_iter, n_samples_total=n_total)

    # ------------------------------------------------------------------------------------
    def project(self, state: FitState, Xs: List[torch.Tensor]) -> torch.Tensor:
        """Sequential project-and-deflate of new samples (tpls.py:128-142; cmtf.py:143-177).
        Xs are device copies and are consumed.  Rows are independent: no communication."""
        be = self.be
        R = state.n_components
        I = Xs[0].shape[0]
        rowcnts = []
        for blk, X in zip(state.blocks, Xs):
            X2 = X.view(I, -1)
            rowcnt, _ = be.center(X2, blk.mean, True)
            miss = bool((rowcnt.min() < X2.shape[1] - 0.5).item()) or bool(torch.isnan(blk.mean).any().item())
            rowcnts.append(rowcnt if miss else None)
        scores = be.zeros(I, R)
        nb = len(Xs)
        Ts = be.empty(nb, I)
        t = be.empty(I)
        for a in range(R):
            was, wbs = [], []
            for blk in state.blocks:
                if len(blk.shape) == 2:
                    was.append(torch.ones(1, dtype=torch.float64, device=t.device))
                    wbs.append(blk.loadings[0][:, a].contiguous())
                else:
                    was.append(blk.loadings[0][:, a].contiguous())
                    wbs.append(blk.loadings[1][:, a].contiguous())
            if nb == 1:
                blk, X2 = state.blocks[0], Xs[0].view(I, -1)
                if be.score_deflate(X2, blk.A, blk.B, was[0], wbs[0], rowcnts[0], t) is None:
                    be.score(X2, blk.A, blk.B, was[0], wbs[0], rowcnts[0], t)
                    be.deflate(X2, blk.A, blk.B, t, was[0], wbs[0])
            else:
                for b, (blk, X) in enumerate(zip(state.blocks, Xs)):
                    be.score(X.view(I, -1), blk.A, blk.B, was[b], wbs[b], rowcnts[b], Ts[b])
                be.scores_mean(Ts, t)
                for b, (blk, X) in enumerate(zip(state.blocks, Xs)):
                    be.deflate(X.view(I, -1), blk.A, blk.B, t, was[b], wbs[b])
            scores[:, a].copy_(t)
        return scores
