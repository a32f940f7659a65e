"""NIPALS engine: the component -> iteration -> block loop of tPLS.fit / ctPLS.fit
(reference cmtf_pls/tpls.py:73-120, cmtf_pls/cmtf.py:85-140) driven over device-resident data.

Layout: each X block lives on the GPU as a C-order (I_local, P) matrix (its mode-0 unfolding, a
free view), f32 or f64; Y, the scores T/U and every reduced quantity are f64.  With several
processes (one per GPU) the SAMPLE mode is sharded: every rank holds I_local rows of every block
and of Y, the loadings are replicated, and the only communication is an all-reduce(sum) of
 - Z (P doubles) and Y^T t (M doubles) per direct NIPALS iteration (|du|^2 comes from the quadratic
   form dq^T (Y^T Y) dq with the once-per-component all-reduced Gram matrix: no third collective),
 - T^T[T|u] and two squared norms per component, column sums/counts once per fit,
 - with algorithm="xcov": S = X_(0)^T Y (M x P doubles) once per component and NOTHING per iteration.
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

import contextlib
import math
from typing import Dict, List, Optional

import torch

from .fitrun import FitRun
from .options import EngineOptions, default_options, set_default_options  # noqa: F401  (re-exported)
from .projection import ProjectionMixin
from .state import (MAX_COMPONENTS, MAX_ORDER, MAX_RANK1_SIDE, MAX_TENSOR_MODE, BlockState, FitState, split_trailing,  # noqa: F401
                    validate_limits)


class Comm:
    """Sample-mode data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm)."""

    def __init__(self, group=None, enabled: Optional[bool] = None, force: bool = False):
        """force: take the sharded code path (collectives issued, segment-wise graph capture) even when the
        group has a single rank -- how the one-GPU box exercises the RCCL calls of the engine."""
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        on = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.sharded = self.world > 1 or (bool(force) and on)
        self.n_collectives = 0
        # only RCCL enqueues its collectives as kernels on the caller's stream, i.e. can be captured into a HIP graph; gloo stages
        # through the host with stream synchronisations, which a capture forbids (and which leave the runtime in an error state)
        self.backend = str(dist.get_backend(group)) if on else None
        self.capturable = self.backend == "nccl"

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.sharded:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
            self.n_collectives += 1
        return t


class _NoComm:
    world, rank, sharded = 1, 0, False

    def allreduce(self, t):
        return t


class NipalsEngine(ProjectionMixin):
    def __init__(self, backend, comm=None, options: Optional[EngineOptions] = None):
        self.be = backend
        self.comm = comm if comm is not None else _NoComm()
        self.opt = options if options is not None else default_options()
        self.last_projection: Dict[str, object] = {}     # which form the last transform / predict took (estimators: projection_report_)

    def device_ctx(self):
        """Make the backend's GPU the current HIP device for the duration of a call: the kernels are launched
        through ctypes on that device's stream, so fitting on cuda:1 from a process whose current device is
        cuda:0 must not depend on the caller having switched devices."""
        dev = getattr(self.be, "device", None)
        if isinstance(dev, torch.device) and dev.type == "cuda":
            return torch.cuda.device(dev)
        return contextlib.nullcontext()

    # ------------------------------------------------------------------------------------
    def _prepare_block(self, X: torch.Tensor, n_total: int, defer_centring: bool = False) -> BlockState:
        """tpls.py:61-71: NaN statistics, nanmean over samples, centring (in place on the copy).
        defer_centring: only the statistics (one read of X, nothing written); `_centre_block` completes the block later --
        in place, or never when the fit can run on the uncentred tensor (FitRun, algorithm="xcov")."""
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
        blk = BlockState(shape=tuple(X.shape), A=A, B=B, mean=mean, has_miss=has_miss,
                         colcnt=colcnt if has_miss else None, rowcnt=None, ssq0=float("nan"), dtype=X.dtype)
        if not defer_centring:
            self._centre_block(blk, X)
        return blk

    def _centre_block(self, blk: BlockState, X: torch.Tensor) -> None:
        """X -= X_mean in place (tpls.py:71), the per-row observation counts and |X_c|^2 (the R2X denominator)."""
        rowcnt, ssq0 = self.be.center(X.view(X.shape[0], -1), blk.mean, blk.has_miss)
        self.comm.allreduce(ssq0)
        blk.rowcnt, blk.ssq0 = rowcnt, float(ssq0.item())

    def _ssq_uncentred(self, blk: BlockState, X: torch.Tensor) -> bool:
        """|X - X_mean|^2 from ONE READ of the uncentred block, nothing written (cmtfpls_recon_r2_* against an all-zero
        reconstruction); False when the backend / shape has no such form."""
        be = self.be
        I = X.shape[0]
        z = lambda n: be.zeros(n, 1)
        out = be.recon_r2(X.view(I, -1), z(I), z(blk.A), z(blk.B), blk.mean) if hasattr(be, "recon_r2") else None
        if out is None:
            return False
        blk.ssq0 = float(self.comm.allreduce(out)[1].item())
        return True

    def _offset_ratio(self, blocks: List[BlockState], Xs: List[torch.Tensor]) -> float:
        """max over blocks of max|column mean| / rms spread of the centred data, the spread estimated from <= 256 rows strided
        over this rank's shard (an order-of-magnitude guard: a few small elementwise operations, nothing X-sized).  Sharded:
        every rank must take the same decision, so the sample statistics are all-reduced."""
        worst = 0.0
        for blk, X in zip(blocks, Xs):
            I = X.shape[0]
            sample = X.view(I, -1)[:: max(1, I // 256)][:256].to(torch.float64) - blk.mean
            stat = torch.stack([(sample * sample).sum(), torch.tensor(float(sample.numel()), dtype=torch.float64, device=sample.device)])
            self.comm.allreduce(stat)
            spread = math.sqrt(float(stat[0].item()) / max(float(stat[1].item()), 1.0))
            top = float(blk.mean.abs().max().item())
            if top == 0.0:
                continue
            ratio = top / spread if spread > 0.0 else float("inf")
            worst = ratio if not ratio <= worst else worst          # (a NaN ratio wins: the caller then declines)
        return worst

    def _probe_plain(self, Xs: List[torch.Tensor], limit: float = 100.0) -> bool:
        """<= 256 rows strided over this rank's shard of every block: no non-finite value among them and max|sample mean| / rms
        spread <= limit.  What FitRun asks before it takes the column statistics out of the read that builds S (the sums of
        squares lose ~ratio^2 * 1e-16 there; missing values need the masked statistics anyway).  All-reduced: one decision."""
        flag = torch.zeros(2, dtype=torch.float64, device=Xs[0].device)
        for X in Xs:
            I = X.shape[0]
            sample = X.view(I, -1)[:: max(1, I // 256)][:256].to(torch.float64)
            m = sample.mean(dim=0)
            d = sample - m
            spread = torch.sqrt((d * d).mean())
            bad = (~torch.isfinite(sample)).any().to(torch.float64)
            ratio = torch.where(spread > 0, m.abs().max() / spread, torch.zeros_like(spread))
            flag += torch.stack([bad, (torch.nan_to_num(ratio, nan=float("inf"), posinf=float("inf")) > limit).to(torch.float64)])
        self.comm.allreduce(flag)
        return not bool((flag > 0).any().item())

    def _rank1(self, blk: BlockState, Z: torch.Tensor, wA: torch.Tensor, wB: torch.Tensor,
               info: Optional[torch.Tensor] = None, n_squarings: Optional[int] = None,
               fac: Optional[torch.Tensor] = None, tol: float = 1e-8) -> None:
        """tpls.py:84-90: Z / norm(Z) for a vector, leading singular pair for a matrix, rank-1 CP
        (tensorly parafac restated) for a tensor; fills the factored loading (wA, wB)."""
        order = len(blk.shape)
        if order == 2:
            wB.copy_(Z)                                   # wA of a matrix block is the constant [1]: set once in FitRun
            self.be.normalize(wB)
        elif order == 3:
            self.be.rank1(Z, blk.A, blk.B, wA, wB, info=info, n_squarings=n_squarings)
        else:
            if order > MAX_ORDER:
                raise NotImplementedError(f"X blocks of order > {MAX_ORDER} are not supported")
            dims = blk.shape[1:]
            self.be.rank1_tensor(Z, dims, tol, fac, info=info, n_squarings=None)
            wA.copy_(fac[0, : dims[0]])
            self.kron_trailing([fac[m, : dims[m]] for m in range(1, len(dims))], wB)

    def kron_trailing(self, vecs: List[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
        """wB = kron(v_1, v_2, ...) of the trailing-mode loadings except the first (C order)."""
        if len(vecs) == 1:
            out.copy_(vecs[0])
            return out
        acc = vecs[0].contiguous()
        for v in vecs[1:-1]:
            acc = self.be.kron(acc, v.contiguous(), self.be.empty(acc.numel() * v.numel()))
        return self.be.kron(acc, vecs[-1].contiguous(), out)

    # ------------------------------------------------------------------------------------
    def begin(self, Xs: List[torch.Tensor], Y: torch.Tensor, n_components: int, coupled: bool,
              algorithm: str = "direct", owned: Optional[List[bool]] = None, allow_raw: bool = True) -> "FitRun":
        """Preprocess (centre in place) and allocate the per-fit buffers; see FitRun.  owned[b] = False: block b is the
        CALLER's tensor -- it is cloned before anything writes it, and not at all when the fit only reads it."""
        with self.device_ctx():
            return FitRun(self, Xs, Y, n_components, coupled, algorithm, owned, allow_raw)

    def fit(self, Xs: List[torch.Tensor], Y: torch.Tensor, n_components: int, tol: float, max_iter: int,
            coupled: bool, verbose: int = 0, algorithm: str = "direct", use_graphs: bool = False,
            mixed: bool = False, on_preprocessed=None, owned: Optional[List[bool]] = None) -> FitState:
        """Xs: device copies (will be centred and deflated in place); Y: (I_local, M) f64 copy.
        algorithm: "direct" = the reference's loop (two X reads per iteration); "xcov" = the same
        iteration re-associated through S = X_(0)^T Y (one X read + one read/write per component)."""
        with self.device_ctx():
            small = self._fit_small(Xs, Y, n_components, tol, max_iter, coupled, verbose, on_preprocessed)
            if small is not None:
                return small
            # (the f32-MFMA S build accumulates in f32 chains: on an uncentred X its error would scale with the means)
            run = self.begin(Xs, Y, n_components, coupled, algorithm, owned, allow_raw=not mixed)
            if on_preprocessed is not None:                          # the estimators print their missing-value notice
                on_preprocessed(run.blocks)                          # here, where the reference does (tpls.py:62-63)
            run.tol = tol                                            # also handed to parafac (tpls.py:86)
            run.use_graphs = bool(use_graphs) and getattr(self.be, "name", "") == "hip"
            run.mixed = bool(mixed)
            for a in range(n_components):
                run.start_component(a)
                run.inner_loop(a, max_iter, tol, verbose)            # tpls.py:79-107
                run.finish_component(a)
            return run.result()

    def _fit_small(self, Xs, Y, n_components, tol, max_iter, coupled, verbose, on_preprocessed) -> Optional[FitState]:
        """The whole fit in ONE launch of one workgroup (cmtfpls_fit_small_f64) for a single small float64 block of order
        2 or 3 without missing values, unsharded (BASELINE configs[0]); None when it does not apply -- the caller then
        runs the regular loop.  Same operations in the reference's order (tpls.py:73-120); sums are formed in a different
        order than the multi-launch kernels form them, i.e. results agree to rounding."""
        be = self.be
        if not (self.opt.small_fit and hasattr(be, "fit_small")) or len(Xs) != 1 or self.comm.sharded or verbose:
            return None
        X = Xs[0]
        if X.dtype != torch.float64 or X.dim() not in (2, 3) or X.numel() > self.opt.small_fit_elements or X.shape[0] < 2:
            return None
        validate_limits([tuple(X.shape)], n_components)
        I = X.shape[0]
        A, B = split_trailing(X.shape)
        out = be.fit_small(X.view(I, -1), Y, A, B, n_components, tol, max_iter)
        if out is None:
            return None
        R = n_components
        ssq = out["ssq"]
        loadings = [out["WB"]] if X.dim() == 2 else [out["WA"], out["WB"]]
        blk = BlockState(shape=tuple(X.shape), A=A, B=B, mean=out["x_mean"], has_miss=False, colcnt=None, rowcnt=None,
                         ssq0=float(ssq[0, 0]), dtype=X.dtype, loadings=loadings, r2x=1.0 - ssq[1:, 0] / ssq[0, 0])
        if on_preprocessed is not None:
            on_preprocessed([blk])
        report = {"form": "small_fit", "algorithm": "direct", "launches": 1, "storage": ["float64"], "shapes": [tuple(X.shape)],
                  "missing": [False], "sharded": False, "graphs": False,
                  "note": "the whole fit in one launch of one workgroup (cmtfpls_fit_small_f64); `algorithm` and `graphs` do not apply"}
        return FitState(coupled=coupled, n_components=R, blocks=[blk], T=out["T"], U=out["U"], Q=out["Q"], coef=out["coef"],
                        r2y=1.0 - ssq[1:, 1] / ssq[0, 1], y_mean=out["y_mean"], n_iter=out["n_iter"], n_samples_total=I, report=report)
