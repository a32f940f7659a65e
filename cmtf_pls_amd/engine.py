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
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


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


def split_trailing(shape: Sequence[int]):
    """(A, B) with A*B = prod(shape[1:]): wA spans the first trailing mode, wB the rest."""
    trailing = list(shape[1:])
    if len(trailing) == 0:
        raise ValueError("X needs at least one trailing mode")
    if len(trailing) == 1:
        return 1, int(trailing[0])
    return int(trailing[0]), int(np.prod(trailing[1:]))


# Hard limits of the kernel set.  The reference has none (tpls.py:84-90,110-112): they are checked BEFORE
# the first sweep over X, not discovered after a centring pass or 64 components of work.
MAX_COMPONENTS = 1024          # cmtfpls_normal_solve_ws_f64: the (a+1) x (a+1) normal equations in one workgroup
MAX_RANK1_SIDE = 4096          # cmtfpls_rank1_f64: min(J, K) of an order-3 block (Gram squaring of the smaller side)
MAX_TENSOR_MODE = 1024         # cmtfpls_rank1_tensor_f64: every trailing mode of a block of order >= 4
MAX_ORDER = 8                  # cmtfpls_rank1_tensor_f64 takes cross-covariance tensors of order <= 7


def validate_limits(shapes, n_components: int) -> None:
    """Raise ValueError / NotImplementedError for a fit the kernels cannot finish, before any work is done."""
    if n_components < 1:
        raise ValueError("n_components must be >= 1")
    if n_components > MAX_COMPONENTS:
        raise ValueError(f"n_components = {n_components} exceeds this engine's limit of {MAX_COMPONENTS} "
                         "(the inner regression solves the (a+1) x (a+1) normal equations in one workgroup)")
    for shape in shapes:
        order = len(shape)
        if order > MAX_ORDER:
            raise NotImplementedError(f"X blocks of order > {MAX_ORDER} are not supported")
        if order == 3 and min(shape[1:]) > MAX_RANK1_SIDE:
            raise ValueError(f"X block {tuple(shape)}: min(J, K) = {min(shape[1:])} exceeds the rank-1 kernel's limit of {MAX_RANK1_SIDE}")
        if order >= 4 and max(shape[1:]) > MAX_TENSOR_MODE:
            raise ValueError(f"X block {tuple(shape)}: a trailing mode exceeds the order-{order} rank-1 kernel's limit of {MAX_TENSOR_MODE}")


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
    dtype: Optional[torch.dtype] = None                                           # storage type of the block on the GPU
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
    # which form of every step actually ran (algorithm after fallbacks, reads of X per component, centred or raw, pipelined,
    # graph replay, ...): `tPLS.fit_report_`
    report: Dict[str, object] = field(default_factory=dict)


@dataclass(frozen=True)
class EngineOptions:
    """Which exact form of each step the engine takes WHERE THE SHAPE ALLOWS IT.  Every default is the fastest form; each
    switch selects the slower equivalent form the tests compare it with.  One object per engine (`NipalsEngine(backend,
    comm, options)`, `tPLS(..., options=EngineOptions(...))`); what actually ran is written to `FitState.report`
    (`tPLS.fit_report_`), so a path the shape declined is visible instead of silent."""
    # a single small float64 block without missing values: the whole fit in ONE launch (cmtfpls_fit_small_f64); a regular
    # iteration is ~20 launches of pure latency whatever the size, a one-workgroup iteration costs time in proportion to I * P
    small_fit: bool = True
    small_fit_elements: int = 1 << 15    # measured (profiles/r03q_small_fit.txt): 2.2x faster at 16000 elements, slower from 65536 on
    # algorithm="xcov" on blocks without missing values: never deflate X in place (two reads per component instead of a
    # read and a read + write, FitRun._finish_xcov_nowrite); False keeps the deflating form
    xcov_nowrite: bool = True
    # ... and, when X is never written anyway, do not centre it either: the fit runs on the caller's UNCENTRED tensor -- no
    # centring pass, no private copy -- with two rank-one corrections; False keeps the centred copy
    xcov_raw: bool = True
    # the uncentred form works by cancellation: its error grows with max|column mean| / rms spread of the centred data.
    # Beyond this ratio the fit falls back to the centred private copy (report: raw = False, raw_declined = ratio)
    xcov_raw_max_offset: float = 1e4
    # the largest block: score and the contraction with the (block-averaged) score from ONE read of it, the second read per
    # component replaced by a P x a matrix-vector product; False keeps the two reads
    xcov_one_read: bool = True
    # blocks WITH missing values, 2 M <= 64: S = X0^T Y and S2 = X0^T (Y * rowscale) from one matrix-core pass with the
    # I x 2M right-hand side [Y, Y * rowscale]; False builds them one after the other
    xcov_pair_build: bool = True
    # a fit on the uncentred tensor: |X - X_mean|^2 from the read that builds S for the first component instead of a read of
    # its own (backend.xcov_ssq); False keeps the separate pass
    xcov_ssq_with_s: bool = True
    # one block WITH missing values: the deflation happens inside the rebuild of S for the next component (one read + write
    # of X instead of a read + write and a read, FitRun._finish_xcov_masked_fused); False keeps the two passes
    xcov_deflate_build: bool = True
    # the inner loop on S: iteration it + 1 is ENQUEUED before the host has seen iteration it's convergence norm, into a
    # second set of buffers (FitRun._inner_loop_xcov_pipelined); False waits after every iteration
    xcov_pipeline: bool = True
    # sharded direct loop under graph replay: capture the two per-iteration all-reduces INSIDE the iteration's HIP graph (one
    # replay per iteration instead of three segments and two eager collectives); falls back to the segment-wise form when
    # the capture fails (report: collectives_in_graph)
    capture_collectives: bool = False
    # transform / predict of samples with missing values: rows WITHOUT a missing value keep the one-pass MTTKRP result and
    # only the affected rows take the masked sequential form; False runs the sequential form on every row of such a batch
    project_split_rows: bool = True

    def but(self, **changes) -> "EngineOptions":
        return replace(self, **changes)


_DEFAULT_OPTIONS = EngineOptions()


def default_options() -> EngineOptions:
    """The options of an engine constructed without any (the product default: `EngineOptions()`)."""
    return _DEFAULT_OPTIONS


def set_default_options(options: Optional[EngineOptions]) -> EngineOptions:
    """Replace the process-wide default (None restores `EngineOptions()`); returns the previous one.  The test harness uses it
    to keep the small float64 fits of the kernel suites on the multi-launch engine (tests/conftest.py)."""
    global _DEFAULT_OPTIONS
    old, _DEFAULT_OPTIONS = _DEFAULT_OPTIONS, (options if options is not None else EngineOptions())
    return old


class NipalsEngine:
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

    # ------------------------------------------------------------------------------------
    def project(self, state: FitState, Xs: List[torch.Tensor], one_pass: bool = True, mixed: bool = False) -> torch.Tensor:
        """Sequential project-and-deflate of new samples (tpls.py:128-142; cmtf.py:143-177).
        Xs are device copies and are consumed.  Rows are independent: no communication."""
        with self.device_ctx():
            return self._project(state, Xs, one_pass, mixed)

    def project_readonly(self, state: FitState, Xs: List[torch.Tensor]) -> Optional[torch.Tensor]:
        """Scores of new samples from ONE read of every block, the blocks neither copied nor written: the MTTKRP runs on
        the UNCENTRED rows and the centring `X - X_mean` (tpls.py:130,153; cmtf.py:150,187) is applied to its I x R output,
        (X - 1 mean^T) W = X W - 1 (mean^T W)^T.

        Samples are independent (tpls.py:128-142 works row by row).  A missing value in a sample shows as a NaN in its row of
        the MTTKRP output; such samples take the reference's masked sequence -- centre, then R times score with the per-row
        rescale, average the coupled blocks' scores and deflate (missingvals.py:23-38, cmtf.py:143-177) -- while the complete
        samples of the same batch KEEP their one-pass scores (`EngineOptions.project_split_rows`): in registers from one
        more read of just those rows (one block, or two coupled blocks in one workgroup), else on compact private copies of
        those rows through the sequential passes (any number of blocks, any storage types).  A strided sample of the batch
        is probed first: when most samples are incomplete the MTTKRP attempt would be a wasted read and every row goes
        through the masked sequence directly.

        None when no read-only form applies (a training column without observations, a shape neither the MTTKRP nor the
        rows-in-registers kernel takes): the caller then runs `project` on private copies.  `last_projection` records the
        form taken."""
        with self.device_ctx():
            be = self.be
            nb, I, R = len(state.blocks), Xs[0].shape[0], state.n_components
            rep = self.last_projection = {"rows": int(I), "blocks": nb, "form": "sequential passes on private copies", "why": None}
            if any(bool(torch.isnan(blk.mean).any().item()) for blk in state.blocks):
                rep["why"] = "a training column without observations (NaN mean)"
                return None
            can_rows = (nb <= 2 and hasattr(be, "project_rows") and all(X.is_contiguous() for X in Xs)
                        and (nb == 1 or hasattr(be, "project_rows2")))
            ops = None

            def in_registers(out, rows):
                nonlocal ops
                ops = ops or [self._kr_operands(blk, R) for blk in state.blocks]
                if nb == 1:
                    blk = state.blocks[0]
                    return be.project_rows(Xs[0].view(I, -1), blk.A, blk.B, ops[0][0], ops[0][1], blk.mean, out, rows=rows)
                return be.project_rows2([X.view(I, -1) for X in Xs], [b.A for b in state.blocks], [b.B for b in state.blocks],
                                        [o[0].contiguous() for o in ops], [o[1].contiguous() for o in ops],
                                        [b.mean for b in state.blocks], out, rows=rows)

            # probe <= 256 samples strided over the batch: mostly incomplete -> skip the MTTKRP attempt (it would be one wasted read)
            if can_rows and I > 0:
                step = max(1, I // 256)
                bad = None
                for X in Xs:
                    r = torch.isnan(X.view(I, -1)[::step][:256]).any(dim=1)
                    bad = r if bad is None else (bad | r)
                frac = float(bad.double().mean().item())
                rep["probe_incomplete_fraction"] = frac
                if frac > 0.5:
                    out = be.empty(I, R)
                    if in_registers(out, None) is not None:
                        rep.update(form="masked sequence, every row in registers (one read)", why="most samples have a missing value")
                        return out
            flag = torch.zeros(1, dtype=torch.int32, device=be.device)
            scores = self._project_one_pass(state, Xs, False, centred=False, nan_flag=flag)
            if scores is not None and int(flag.item()) == 0:
                rep.update(form="one-pass MTTKRP (one read, nothing written)")
                return scores
            rows = None
            if scores is not None and self.opt.project_split_rows:
                rows = torch.nonzero(torch.isnan(scores).any(dim=1)).view(-1).contiguous()    # samples with a missing value somewhere
                rep["incomplete_rows"] = int(rows.numel())
                if rows.numel() == I:
                    rows = None
            if can_rows:
                out = scores if rows is not None else be.empty(I, R)
                if in_registers(out, rows) is not None:
                    rep.update(form=("one-pass MTTKRP for the complete samples + masked sequence in registers for the incomplete ones"
                                     if rows is not None else "masked sequence, every row in registers (one read)"),
                               why="missing values in the batch")
                    return out
            if rows is not None:
                # any number of blocks / storage types / trailing extents: compact private copies of the incomplete samples only
                sub = [X.index_select(0, rows) for X in Xs]
                scores.index_copy_(0, rows, self._project(state, sub, one_pass=False, mixed=False))
                rep.update(form="one-pass MTTKRP for the complete samples + sequential passes on copies of the incomplete ones",
                           why="missing values in the batch; shape outside the rows-in-registers kernel")
                return scores
            rep["why"] = ("shape outside the MTTKRP and the rows-in-registers kernel" if scores is None
                          else "every sample has a missing value; shape outside the rows-in-registers kernel")
            return None

    def _project(self, state: FitState, Xs: List[torch.Tensor], one_pass: bool, mixed: bool) -> torch.Tensor:
        be = self.be
        R = state.n_components
        I = Xs[0].shape[0]
        rowcnts = []
        for blk, X in zip(state.blocks, Xs):
            X2 = X.view(I, -1)
            rowcnt, _ = be.center(X2, blk.mean, True)
            miss = bool((rowcnt.min() < X2.shape[1] - 0.5).item()) or bool(torch.isnan(blk.mean).any().item())
            rowcnts.append(rowcnt if miss else None)
        if one_pass and all(rc is None for rc in rowcnts):
            scores = self._project_one_pass(state, Xs, mixed)
            if scores is not None:
                return scores
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
                    wbs.append(self.kron_trailing([L[:, a] for L in blk.loadings[1:]], be.empty(blk.B)))
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
        if nb > 1 and any(rc is not None for rc in rowcnts):
            # coupled blocks: a sample whose row is empty in ONE block gets a NaN average (cmtf.py:155,206); the reference's
            # mask comes from the input, so the NaN-deflated rows of its other blocks give NaN scores from then on, while the
            # masked score kernels read those entries as missing: restore the reference's outcome on the I x R result
            scores.masked_fill_(torch.isnan(scores).cumsum(dim=1) > 0, float("nan"))
        return scores

    def _kr_operands(self, blk: BlockState, R: int):
        """(WA, WB): the block's loading matrices as the factored Khatri-Rao operand the matrix kernels take,
        W[c, r] = WA[c / B, r] * WB[c % B, r] (a matrix block: WA = ones; order >= 4: WB = column-wise Kronecker
        product of the trailing modes' loadings, formed on the device)."""
        be = self.be
        loads = blk.loadings
        if len(blk.shape) == 2:
            WA = be.empty(1, R)
            WA.fill_(1.0)
            return WA, loads[0]
        WB = loads[1]
        for L in loads[2:]:
            WB = be.khatri_rao(WB, L)
        return loads[0], WB

    def reconstruct(self, state: FitState, block: int = 0, rows: Optional[slice] = None,
                    dtype: Optional[torch.dtype] = None) -> Optional[torch.Tensor]:
        """Rows of factors_to_tensor(X_factors) + X_mean (util.py:18-20 with tpls.py:188-189 / cmtf.py:233-237) for
        one block, formed on the GPU in `dtype` (default: the block's storage type; the estimators ask for float64 when
        they return a host array, as the reference does): Xhat = T (W_1 (.) W_2 (.) ...)^T + mean with the Khatri-Rao
        operand never materialised (cmtfpls_recon_*).  None when the backend / shape has no device form (the caller
        falls back to the host einsum)."""
        be = self.be
        if not hasattr(be, "recon"):
            return None
        blk = state.blocks[block]
        with self.device_ctx():
            T = state.T if rows is None else state.T[rows]
            WA, WB = self._kr_operands(blk, state.n_components)
            out = be.empty(T.shape[0], blk.A * blk.B, dtype=dtype or blk.dtype or torch.float64)
            if T.shape[0] == 0 or be.recon(T, WA, WB, blk.mean, out) is None:
                return None
            return out.view((T.shape[0],) + tuple(blk.shape[1:]))

    def r2x_literal(self, state: FitState, X: torch.Tensor, block: int = 0) -> Optional[float]:
        """calcR2X(X - X_mean, factors_to_tensor(X_factors)) (util.py:7-15 as called at tpls.py:115-117) for the rows
        X (device, storage type, UNCENTRED, same rows as state.T) in one read of X, the reconstruction never
        materialised (cmtfpls_recon_r2_*).  None when the backend / shape has no device form."""
        be = self.be
        if not hasattr(be, "recon_r2"):
            return None
        blk = state.blocks[block]
        with self.device_ctx():
            WA, WB = self._kr_operands(blk, state.n_components)
            out = be.recon_r2(X.view(X.shape[0], -1), state.T, WA, WB, blk.mean)
            if out is None:
                return None
            res, ssq = self.comm.allreduce(out).cpu().tolist()
            return 1.0 - res / ssq

    def _project_one_pass(self, state: FitState, Xs: List[torch.Tensor], mixed: bool = False, centred: bool = True,
                          nan_flag: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """All R scores from ONE read of every NaN-free block (centred: already centred in place; otherwise the centring
        is applied to the MTTKRP output as the shift mean^T W, itself an MTTKRP of the one-row "tensor" mean).

        The deflations are linear without missing values: X_{b,a+1} = X_{b,a} - t_a w_{b,a}^T with the
        (block-averaged) score t_a, hence X_{b,a} w_{b,a} = M_b[:, a] - sum_{j<a} t_j G_b[j, a] where
        M_b = X_{b,0} (W_A (.) W_B) is one MTTKRP and G_b = W_b^T W_b.  Averaging over blocks
        (cmtf.py:155,206) gives T (I + triu(mean G, 1)) = mean M: an R x R triangular solve.
        Returns None when the MTTKRP kernel does not take the shape (caller falls back)."""
        be = self.be
        R = state.n_components
        I = Xs[0].shape[0]
        nb = len(Xs)
        if R > 64:
            return None
        Ms = be.empty(nb, I * R)
        Gs = be.empty(nb, R * R)
        shifts = None if centred else be.empty(nb, R)
        for b, (blk, X) in enumerate(zip(state.blocks, Xs)):
            loads = blk.loadings
            WA, WB = self._kr_operands(blk, R)
            if be.mttkrp(X.view(I, -1), blk.A, blk.B, WA, WB, Ms[b].view(I, R), mixed=mixed) is None:
                return None
            if not centred and be.mttkrp(blk.mean.view(1, -1), blk.A, blk.B, WA, WB, shifts[b].view(1, R)) is None:
                return None
            for m, L in enumerate(loads):                 # Gram of a Khatri-Rao product = Hadamard product of the mode Grams
                be.kr_gram(L, Gs[b], first=(m == 0))
        Mbar = be.scores_mean(Ms, be.empty(I * R)).view(I, R) if nb > 1 else Ms[0].view(I, R)
        Gbar = be.scores_mean(Gs, be.empty(R * R)).view(R, R) if nb > 1 else Gs[0].view(R, R)
        if centred:
            return be.unit_upper_solve_rows(Mbar, Gbar, None, nan_flag)      # T (I + triu(Gbar, 1)) = Mbar, on the device
        shift = be.scores_mean(shifts, be.empty(R)) if nb > 1 else shifts[0]
        return be.unit_upper_solve_rows(Mbar, Gbar, shift, nan_flag)


class FitRun:
    """One fit in flight: the state between NIPALS iterations.  ``fit`` drives it; bench.py drives
    ``iterate`` directly so that the timed step IS the product's iteration."""

    def __init__(self, eng: NipalsEngine, Xs: List[torch.Tensor], Y: torch.Tensor, n_components: int, coupled: bool,
                 algorithm: str = "direct", owned: Optional[List[bool]] = None, allow_raw: bool = True):
        be, comm = eng.be, eng.comm
        Xs = list(Xs)
        owned = [True] * len(Xs) if owned is None else list(owned)
        if algorithm not in ("direct", "xcov"):
            raise ValueError("algorithm must be 'direct' or 'xcov'")
        self.algorithm_requested = self.algorithm = algorithm
        self.notes: List[str] = []                                # every declined fast form, in words (report["declined"])
        validate_limits([tuple(X.shape) for X in Xs], n_components)   # before the first sweep touches X
        self.eng, self.Xs, self.Y, self.R, self.coupled = eng, Xs, Y, n_components, coupled
        R = n_components
        I, M = Y.shape
        self.I, self.M = I, M
        n_tot = torch.tensor([float(I)], dtype=torch.float64, device=Y.device)
        comm.allreduce(n_tot)
        self.n_total = int(round(float(n_tot.item())))
        # algorithm="xcov" on blocks without missing values reads X and never writes it (_finish_xcov_nowrite): then it need not be
        # centred either.  The statistics pass decides: raw = every block NaN-free and every kernel of that path present.
        want_raw = (allow_raw and algorithm == "xcov" and eng.opt.xcov_raw and eng.opt.xcov_nowrite
                    and n_components <= 64
                    and all(hasattr(be, f) for f in ("axpy_scalar", "total", "recon_r2", "s_downdate", "deflate_contract_yq", "kr_axpy")))
        if not want_raw:
            for b in range(len(Xs)):
                if not owned[b]:
                    Xs[b] = Xs[b].clone()                        # the fit centres and deflates in place: never the caller's tensor
        self.blocks = [eng._prepare_block(X, self.n_total, defer_centring=want_raw) for X in Xs]
        # |X - X_mean|^2 of an uncentred block: from the read that builds S for the first component (backend.xcov_ssq),
        # else from a read of its own (_ssq_uncentred)
        self._ssq_with_s = (want_raw and hasattr(be, "xcov_ssq") and eng.opt.xcov_ssq_with_s)
        self._ssq0_dev = {}
        self.raw = want_raw and not any(blk.has_miss for blk in self.blocks)
        if self.raw:
            # the uncentred form subtracts mean-sized terms from data-sized results: beyond ~1e4 x the spread it loses digits
            # the centred copy keeps (error ~ 1e-16 * ratio), so such data is centred after all
            ratio = eng._offset_ratio(self.blocks, Xs)
            if not ratio <= eng.opt.xcov_raw_max_offset:
                self.raw = False
                self.notes.append(f"uncentred xcov form declined: max|column mean| / spread = {ratio:.3g} > {eng.opt.xcov_raw_max_offset:g}")
        if self.raw and not self._ssq_with_s:
            self.raw = all(eng._ssq_uncentred(blk, X) for blk, X in zip(self.blocks, Xs))
        if want_raw and not self.raw:                            # missing values (or no read-only norm): the deflating form after all
            for b, blk in enumerate(self.blocks):
                if not owned[b]:
                    Xs[b] = Xs[b].clone()
                eng._centre_block(blk, Xs[b])
        self.X2 = [X.view(I, -1) for X in Xs]
        ysum, ycnt = be.colstats(Y)
        comm.allreduce(ysum)
        comm.allreduce(ycnt)
        self.y_mean = ysum / ycnt                                 # tpls.py:67
        _, ssqy0 = be.center(Y, self.y_mean, False)
        comm.allreduce(ssqy0)
        self.ssqy0 = float(ssqy0.item())
        self.T = be.zeros(I, R)
        self.U = be.zeros(I, R)
        self.Q = be.zeros(M, R)
        self.coef = np.zeros((R, R))                              # filled from coef_dev by result()
        self.coef_dev = be.zeros(R, R)
        self.b_dev = be.empty(R)
        # per component: local sums of squares of every deflated block and of the deflated Y (the R2X / R2Y
        # numerators, tpls.py:115-120); all-reduced and read back ONCE, in result()
        self.ssq_log = be.zeros(R, len(self.blocks) + 1)
        self.r2y = np.zeros(R)
        for blk in self.blocks:
            blk.loadings = [be.zeros(d, R) for d in blk.shape[1:]]
            blk.r2x = np.zeros(R)
        self.wA = [be.empty(blk.A) for blk in self.blocks]
        for b, blk in enumerate(self.blocks):
            if len(blk.shape) == 2:
                self.wA[b].fill_(1.0)                     # a matrix block has A = 1 and w = wB: never written again
        self.wB = [be.empty(blk.B) for blk in self.blocks]
        self.Zs = [be.empty(blk.A * blk.B) for blk in self.blocks]
        self.fac = [be.zeros(len(blk.shape) - 1, max(blk.shape[1:])) if len(blk.shape) > 3 else None for blk in self.blocks]
        self.tol = 1e-8
        self.Ts = be.empty(len(self.blocks), I)
        # one block, no averaging: the score kernel writes t directly (no copy)
        self.t = self.Ts[0] if (len(self.blocks) == 1 and not coupled) else be.empty(I)
        # per-iteration status read back in ONE device->host copy: [|du|^2, (converged, squarings) per block]
        self.status = be.zeros(1 + 2 * len(self.blocks))
        self.status[1::2] = 1.0
        # pinned mirror: the read-back is an async copy on the launch stream (a memcpy node when the
        # iteration is replayed as a graph) followed by one stream synchronisation
        self.status_host = None
        if self.status.is_cuda:
            self.status_host = torch.empty(self.status.shape, dtype=torch.float64, pin_memory=True)
        self.sq_max = int(getattr(be, "rank1_squarings", 30))
        self.sq_budget = [self.sq_max] * len(self.blocks)
        self.u = be.empty(I)
        self.u_new = be.empty(I)
        self.q = be.empty(M)
        self.n_iter: List[int] = []
        # the pipelined inner loop on S (report["pipeline"]): iterations accepted, enqueued ahead of the host, enqueued for nothing
        # (the loop had converged), host round trips the GPU idled through (no speculation), tails redone with the full budget
        self.pipeline_stats = {"iterations": 0, "ahead": 0, "unused": 0, "waited": 0, "redone": 0}
        self._executed = 0
        self._parity = 0
        self.mixed = False                        # opt-in f32-MFMA form of the S build (f32 storage only)
        self.use_graphs = False
        self._graphs = {}
        self._graph_error = None
        self._collectives_captured = None         # None: not tried; True / False: the outcome of the first capture
        # Fused Y side (M <= 64): u = Y q is formed inside the contraction and Y^T t inside the score
        # kernel, so an iteration has no launch of its own for either; q lives in two buffers that
        # alternate by parity (a captured graph holds their addresses) and |du|^2 is the quadratic form
        # dq^T (Y^T Y) dq.  More responses keep the separate gram_tn / normalize / rowdot launches.
        # Coupled blocks: normalize(Y^T mean_b t_b) = normalize(sum_b Y^T t_b), so every block's score kernel
        # adds its partial rows and the averaged score itself is only formed once per component.
        self._z_ready = False                     # Zs already hold X x_0 u_0 of the component about to start
        self._fused = (algorithm == "direct" and M <= 64
                       and all(hasattr(be, f) for f in ("mode0_contract_yq", "score_gram", "q_update")))
        if self._fused:
            self.Gy = be.empty(M, M)
            self.qbuf = [be.zeros(M), be.zeros(M)]
            self.qpart = be.empty(len(self.blocks), int(be.n_partials) * M)
        elif algorithm == "direct" and comm.sharded:
            self.Gy = be.empty(M, M)
            self.q_prev = be.zeros(M)
        if algorithm == "xcov":
            nb = len(self.blocks)
            # masked blocks: Y^T t needs the per-row rescale P / n_obs(i) of miss_mmodedot folded into Y -- a second S, built from
            # Y * rowscale.  2 M <= 64 responses: both come from ONE matrix-core pass over X with [Y, Y * rowscale] as its I x 2M
            # right-hand side (S and S2 are the two halves of one 2M x P result)
            self._s_pair = 2 * M <= 64 and eng.opt.xcov_pair_build
            self.S, self.S2, self.S12 = [], [], []
            for blk in self.blocks:
                if blk.has_miss and self._s_pair:
                    both = be.empty(2 * M, blk.A * blk.B)
                    self.S12.append(both)
                    self.S.append(both[:M])
                    self.S2.append(both[M:])
                else:
                    self.S12.append(None)
                    self.S.append(be.empty(M, blk.A * blk.B))
                    self.S2.append(be.empty(M, blk.A * blk.B) if blk.has_miss else None)
            self.rowscale = [(float(blk.A * blk.B) / blk.rowcnt) if blk.has_miss else None for blk in self.blocks]
            any_miss = any(blk.has_miss for blk in self.blocks)
            self.Yw = be.empty(I, 2 * M if self._s_pair else M) if any_miss else None
            self.Gy = be.empty(M, M)
            self.Tq = be.empty(nb, M)
            self.qx = [be.zeros(M), be.zeros(M)]      # q of the current / next iteration, alternating by parity
            self.qc = self.qx[0]
            self.qn = self.Tq[0] if (nb == 1 and not coupled) else be.empty(M)
            # S is carried across a deflation instead of rebuilt when no block has missing values:
            # S+ = S - (Y^T t) w^T - q (X+^T yhat)^T, with X+^T yhat formed inside the deflation sweep
            self._s_carry = (not any(blk.has_miss for blk in self.blocks)
                             and all(hasattr(be, f) for f in ("s_downdate", "deflate_contract_yq")))
            self._s_ready = False
            self._nowrite = False
            if self.raw:
                self.zM = be.zeros(M)
                self.zA = [be.zeros(blk.A) for blk in self.blocks]
                self.zB = [be.zeros(blk.B) for blk in self.blocks]
            if self._s_carry:
                self.yhat = be.empty(I, 1)
                self.one = be.empty(1)
                self.one.fill_(1.0)
                self.vs = [be.empty(blk.A * blk.B) for blk in self.blocks]
                self._nowrite = eng.opt.xcov_nowrite and hasattr(be, "kr_axpy") and R <= 64
                if self._nowrite:
                    # per component [t^T t, t^T t_b per block] (this rank's rows): |X_{b,a+1}|^2 = |X_{b,a}|^2 - 2 t^T t_b + t^T t
                    # is evaluated on the host in result(), from the all-reduced dot products, instead of measured
                    self.dot_log = be.zeros(R, 1 + len(self.blocks))
                    self.Gw = be.empty(R * R)
                    self._G_last = None
                    # the final score and r_a = X_0^T t_a of the LARGEST block come from ONE read of it (_finish_xcov_nowrite),
                    # so its second read per component is a P x a matrix-vector product instead
                    self._one_read = R > 1 and eng.opt.xcov_one_read and hasattr(be, "score_contract")
                    if self._one_read:
                        self._fused_b = max(range(len(self.blocks)), key=lambda b: self.blocks[b].A * self.blocks[b].B)
                        P0 = self.blocks[self._fused_b].A * self.blocks[self._fused_b].B
                        self.ps = be.empty(P0)
                        self.Rm = be.zeros(P0, R)                                # column j: X_0^T t_j
                        self.corr = be.empty(I)
                        self.csum = be.empty(1)
            assert self._nowrite or not self.raw, "an uncentred X needs the form of the loop that never writes it"

    def start_component(self, a: int) -> None:
        self._executed = 0
        be, comm = self.eng.be, self.eng.comm
        if self.algorithm == "direct":
            self.u.copy_(self.Y[:, 0])                            # tpls.py:78
            self._parity = 0
            if self._fused:
                self.qbuf[0].zero_()
                self.qbuf[0][0] = 1.0                             # u_0 = Y[:, 0] = Y e_0 exactly
            if self._fused or comm.sharded:
                be.gram_tn(self.Y, self.Y, out=self.Gy)
                comm.allreduce(self.Gy)
            return
        for b, blk in enumerate(self.blocks):
            if self._s_ready:
                break                                             # S was down-dated by the previous finish_component
            if self.S12[b] is not None:                           # masked block: S and S2 from one pass
                self.Yw[:, :self.M].copy_(self.Y)
                torch.mul(self.Y, self.rowscale[b][:, None], out=self.Yw[:, self.M:])
                be.xcov(self.X2[b], self.Yw, True, out=self.S12[b], mixed=self.mixed)
                comm.allreduce(self.S12[b])
                continue
            if self.raw and self._ssq_with_s and b not in self._ssq0_dev:
                _, ssq = be.xcov_ssq(self.X2[b], self.Y, blk.mean, out=self.S[b])   # S and |X - X_mean|^2 from one read
                self._ssq0_dev[b] = comm.allreduce(ssq)                       # (read back in result(), with everything else)
            else:
                be.xcov(self.X2[b], self.Y, blk.has_miss, out=self.S[b], mixed=self.mixed)
            comm.allreduce(self.S[b])
            if self.raw:
                # X is uncentred: X_c^T Y = X^T Y - mean (1^T Y)^T; the centred Y sums to ~1e-13 per column, not to exactly 0
                ysum, _ = be.colstats(self.Y)
                comm.allreduce(ysum)
                be.s_downdate(self.S[b], blk.A, blk.B, self.zM, self.zA[b], self.zB[b], ysum, blk.mean)
            if blk.has_miss:
                torch.mul(self.Y, self.rowscale[b][:, None], out=self.Yw)
                be.xcov(self.X2[b], self.Yw, True, out=self.S2[b], mixed=self.mixed)
                comm.allreduce(self.S2[b])
        self._s_ready = False
        be.gram_tn(self.Y, self.Y, out=self.Gy)
        comm.allreduce(self.Gy)
        self._parity = 0
        self.qc = self.qx[0]
        self.qc.zero_()
        self.qc[0] = 1.0                                          # u_0 = Y[:, 0] = Y e_0   (tpls.py:78)

    def _iterate_xcov(self, it: int) -> Optional[float]:
        """The same iteration with X x_0 u = sum_m q_m S_m and Y^T t = S_(0) kron(wA, wB): only S is
        touched (no X read, no communication: S is already global).  |u_old - u|^2 = dq^T (Y^T Y) dq."""
        be = self.eng.be
        self._executed += 1

        blk0 = self.blocks[0]
        composite = (len(self.blocks) == 1 and len(blk0.shape) == 3 and not blk0.has_miss and self.M <= 64
                     and self.qn.data_ptr() == self.Tq.data_ptr() and hasattr(be, "xcov_iterate"))

        par = self._parity
        q_cur, q_new = self.qx[par], self.qx[par ^ 1]

        def seg(first: bool):
            if composite:
                # the whole iteration (its kernels are tiny) is issued by one host call; q alternates between
                # two buffers (no copy, and a captured graph keeps their addresses)
                be.xcov_iterate(self.S[0], blk0.A, blk0.B, q_cur, self.Zs[0], self.wA[0], self.wB[0], self.status[1:3],
                                self.sq_budget[0], q_new, self.Gy, self.status[0:1], first)
                return
            if first:
                for b, blk in enumerate(self.blocks):
                    be.mode0_contract(self.S[b], self.qc, False, out=self.Zs[b])     # tpls.py:80-83
                    if blk.has_miss:
                        be.colscale(self.Zs[b], blk.colcnt, self.n_total)            # missingvals.py:17-19
            for b, blk in enumerate(self.blocks):
                self.eng._rank1(blk, self.Zs[b], self.wA[b], self.wB[b], info=self.status[1 + 2 * b: 3 + 2 * b],
                                n_squarings=self.sq_budget[b], fac=self.fac[b], tol=self.tol)   # tpls.py:84-90
                be.score_s(self.S2[b] if blk.has_miss else self.S[b], blk.A, blk.B, self.wA[b], self.wB[b], self.Tq[b])
            if self.qn.data_ptr() != self.Tq.data_ptr():
                be.scores_mean(self.Tq, self.qn)                                 # cmtf.py:120 (linear in t)
            be.normalize(self.qn)                                                # tpls.py:100-101
            if it > 0:
                be.quadform(self.Gy, self.qn, self.qc, self.status[0:1])         # tpls.py:102-103

        first = True
        while True:
            self._run(("xcov", it > 0, tuple(self.sq_budget), first, par if composite else -1), lambda: seg(first))
            host = self._read_status()
            if not self._update_budgets(host):
                break
            first = False
        if composite:
            self._parity ^= 1
            self.qc = q_new
        else:
            self.qc.copy_(self.qn)               # fixed buffers (a captured graph holds their addresses)
        return None if it == 0 else math.sqrt(max(float(host[0]), 0.0))

    def inner_loop(self, a: int, max_iter: int, tol: float, verbose: int = 0) -> None:
        """The NIPALS iterations of component a (tpls.py:79-107): iterate until |u_old - u| < tol or max_iter."""
        if max_iter > 0 and self._pipeline_ok():
            self._pipelined = True
            self._inner_loop_xcov_pipelined(a, max_iter, tol, verbose)
            return
        for it in range(max_iter):                                   # tpls.py:79
            du = self.iterate(it)
            if du is not None and du < tol:                          # tpls.py:103 (first pass: oldU = inf)
                if verbose:
                    print("Comp {}: converged after {} iterations".format(a, it))
                break

    def _single_composite(self) -> bool:
        blk0 = self.blocks[0]
        return (len(self.blocks) == 1 and len(blk0.shape) == 3 and not blk0.has_miss and self.M <= 64
                and self.qn.data_ptr() == self.Tq.data_ptr() and hasattr(self.eng.be, "xcov_iterate"))

    def _pipeline_ok(self) -> bool:
        be = self.eng.be
        if self.algorithm != "xcov" or self.use_graphs or not self.eng.opt.xcov_pipeline:
            return False
        if self.M > 64 or not all(hasattr(be, f) for f in ("status_snapshot", "status_wait")):
            return False
        return self._single_composite() or (hasattr(be, "xcov_blocks_plan") and all(len(blk.shape) in (2, 3) for blk in self.blocks))

    def _inner_loop_xcov_pipelined(self, a: int, max_iter: int, tol: float, verbose: int) -> None:
        """The inner loop on S with iteration it + 1 in flight while the host looks at iteration it.

        An iteration on S is ~15 dependent launches of a few microseconds each (per order-3 block); waiting for its
        convergence norm (device -> host copy, wake-up, the next launches) left the GPU idle for a quarter of it.  Iteration
        it + 1 only needs q of iteration it, which is on the device: it is enqueued right behind iteration it, writing a
        SECOND set of buffers (Z, wA, wB per block, status: sets alternate with it; q rotates through three buffers so that a
        tail that has to be redone still finds its q_cur).  If iteration it turns out to have converged, set it & 1 holds the
        result and the speculative iteration ran for nothing -- so none is enqueued when the last two norms predict
        convergence.  One host call per iteration with its arguments marshalled once (backend.xcov_iterate_plan for one
        NaN-free order-3 block, backend.xcov_blocks_plan for coupled blocks / blocks with missing values of order 2 or 3).
        Same kernels on the same data in the same order as the waiting loop: identical iteration counts, and for the
        one-block form identical bits."""
        be = self.eng.be
        nb = len(self.blocks)
        single = self._single_composite()
        pp = getattr(self, "_pipe", None)
        if pp is None:
            M = self.M
            second = {"Z": [], "wA": [], "wB": []}
            for b, blk in enumerate(self.blocks):
                second["Z"].append(be.empty(blk.A * blk.B))
                second["wA"].append(self.wA[b].clone())              # (the constant [1] of a matrix block comes along)
                second["wB"].append(be.empty(blk.B))
            pp = self._pipe = {
                "q": [self.qx[0], self.qx[1], be.zeros(M)],
                "Z": [list(self.Zs), second["Z"]], "wA": [list(self.wA), second["wA"]], "wB": [list(self.wB), second["wB"]],
                "status": [be.zeros(1 + 2 * nb), be.zeros(1 + 2 * nb)],
                "plans": {}, "slots": {},                            # (this fit's own pinned status mirrors)
            }
            for st in pp["status"]:
                st[1::2] = 1.0                                       # (blocks without a rank-1 chain never write their flag)
        plans = pp["plans"]

        def make_plan(it: int):
            s = it & 1
            st, q_cur, q_new = pp["status"][s], pp["q"][it % 3], pp["q"][(it + 1) % 3]
            if single:
                blk0 = self.blocks[0]
                if hasattr(be, "xcov_iterate_plan"):                 # arguments marshalled once per (set, q rotation)
                    one = be.xcov_iterate_plan(self.S[0], blk0.A, blk0.B, q_cur, pp["Z"][s][0], pp["wA"][s][0], pp["wB"][s][0], st,
                                               q_new, self.Gy)
                    return lambda nsq, first: one(nsq[0], first)
                return lambda nsq, first: be.xcov_iterate(self.S[0], blk0.A, blk0.B, q_cur, pp["Z"][s][0], pp["wA"][s][0],
                                                          pp["wB"][s][0], st[1:3], nsq[0], q_new, self.Gy, st[0:1], first)
            descr = [dict(S=self.S[b], S2=self.S2[b] if blk.has_miss else None, colcnt=blk.colcnt if blk.has_miss else None,
                          n_samples=self.n_total, order=len(blk.shape), A=blk.A, B=blk.B,
                          Z=pp["Z"][s][b], wA=pp["wA"][s][b], wB=pp["wB"][s][b]) for b, blk in enumerate(self.blocks)]
            return be.xcov_blocks_plan(descr, self.M, q_cur, self.Tq, q_new, self.Gy, st)

        def enqueue(it: int, first: bool = True):
            plan = plans.get(it % 6)
            if plan is None:
                plan = plans[it % 6] = make_plan(it)
            # the first iteration of a component starts from u = Y[:, 0] (tpls.py:78): its Z has another spectrum than the last
            # iterations of the previous component, whose need the budget remembers -- 4 spare launches (~4 us each when unused)
            # instead of a tail redone in every other component
            plan([n if it > 0 else min(self.sq_max, n + 4) for n in self.sq_budget], first)
            return be.status_snapshot(pp["status"][it & 1], it & 1, slots=pp["slots"])

        stats = self.pipeline_stats
        it, tok = 0, enqueue(0)
        du_prev = du = None
        while True:
            ahead = None
            if it + 1 < max_iter:
                # |du| shrinks geometrically: no speculation when the next norm is predicted below tol (the wait costs less
                # than an iteration run for nothing)
                predicted = None if (du is None or du_prev is None or du_prev <= 0.0) else du * (du / du_prev)
                if it == 0 or predicted is None or predicted >= tol:
                    ahead = enqueue(it + 1)
                    stats["ahead"] += 1
            host = be.status_wait(tok)
            short = [b for b in range(nb) if not host[1 + 2 * b] > 0.5 and self.sq_budget[b] < self.sq_max]
            if short:
                # a rank-1 extraction ran out of squarings: redo the tail of iteration it with the full budget (Z of set it & 1
                # is intact; whatever was enqueued ahead was built on the unfinished loadings and is overwritten later)
                for b in short:
                    self.sq_budget[b] = self.sq_max
                tok = enqueue(it, first=False)
                stats["redone"] += 1
                continue
            for b, blk in enumerate(self.blocks):
                if len(blk.shape) == 3 and host[1 + 2 * b] > 0.5:
                    self.sq_budget[b] = min(self.sq_max, int(host[2 + 2 * b]) + 1)
            self._executed += 1
            stats["iterations"] += 1
            du_prev, du = du, (None if it == 0 else math.sqrt(max(float(host[0]), 0.0)))
            if (du is not None and du < tol) or it + 1 >= max_iter:  # tpls.py:103 (first pass: oldU = inf)
                if verbose and du is not None and du < tol:
                    print("Comp {}: converged after {} iterations".format(a, it))
                stats["unused"] += ahead is not None                  # an iteration that ran for nothing
                break
            it += 1
            stats["waited"] += ahead is None                         # the GPU idled through one host round trip
            tok = ahead if ahead is not None else enqueue(it)
        if it & 1:                                                   # the engine's own buffers are set 0
            for b in range(nb):
                self.wA[b].copy_(pp["wA"][1][b])
                self.wB[b].copy_(pp["wB"][1][b])
        self.qc = pp["q"][(it + 1) % 3]
        self._parity = 0

    def _update_budgets(self, host) -> bool:
        """Adapt the squaring budget of every order-3 block; True if the iteration tail must be redone."""
        retry = False
        for b in range(len(self.blocks)):
            conv, used = host[1 + 2 * b] > 0.5, int(host[2 + 2 * b])
            if not conv and self.sq_budget[b] < self.sq_max:
                self.sq_budget[b] = self.sq_max
                retry = True
            elif conv and len(self.blocks[b].shape) == 3:
                # the last computing launch (`used`) declares its own output final, or launch used + 1 sees it; keep
                # one spare.  Under graph replay the launch sequence is part of the captured graph: hysteresis
                # (re-plan only outside [used+1, used+3]) keeps it stable; eager launches follow the need exactly
                # (every spare launch is ~4 us of an idle GPU)
                if not self.use_graphs or used + 1 > self.sq_budget[b] or used + 3 < self.sq_budget[b]:
                    self.sq_budget[b] = min(self.sq_max, used + 1)
        return retry

    def _read_status(self) -> np.ndarray:
        if self.status_host is None:
            return self.status.cpu().numpy()
        self.status_host.copy_(self.status, non_blocking=True)
        torch.cuda.current_stream(self.status.device).synchronize()
        return self.status_host.numpy()

    def _run(self, key, fn) -> None:
        """Run one launch sequence; with use_graphs it is captured once per key into a HIP graph
        (torch.cuda.CUDAGraph on the launch stream) and replayed afterwards: one host call instead of
        ~20 kernel launches, which is what bounds a strongly-scaled iteration."""
        if not self.use_graphs:
            fn()
            return
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
            return
        fn()                                   # eager: does this call's work and sizes every workspace
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: an RCCL watchdog thread polling events must not invalidate the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self._graphs[key] = g
        except Exception as e:                 # capture unsupported in this context: stay eager
            self.use_graphs = False
            self._graph_error = repr(e)
            self._after_failed_capture()

    def _run_with_collectives(self, key, fn) -> bool:
        """`EngineOptions.capture_collectives`: one sharded iteration INCLUDING its all-reduces as ONE HIP graph (RCCL enqueues
        its kernels on the capturing stream), instead of three captured segments with two eager collectives between them.
        True when `fn`'s work was done (eagerly the first time, by replay afterwards); False when this form is not available
        -- not asked for, no graph replay, or a capture that failed once (the communicator's backend cannot be captured, e.g.
        gloo): the caller then runs the segment-wise form, which is what every earlier round ran."""
        if not (self.use_graphs and self.eng.opt.capture_collectives and self._collectives_captured is not False):
            return False
        if not getattr(self.eng.comm, "capturable", False):
            self._collectives_captured = False
            self.notes.append(f"all-reduces not captured into the iteration's graph: the {getattr(self.eng.comm, 'backend', None)} "
                              "backend stages through the host (only RCCL collectives are stream-ordered kernels)")
            return False
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
            return True
        fn()                                   # eager: does this call's work (collectives included) and sizes every workspace
        try:
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self._graphs[key] = g
            self._collectives_captured = True
        except Exception as e:                 # the segment-wise form from the next iteration on
            self._collectives_captured = False
            self.notes.append("all-reduces not captured into the iteration's graph: " + repr(e)[:200])
            self._after_failed_capture()
        return True

    def _after_failed_capture(self) -> None:
        """A capture that failed leaves the stream's work undone (nothing of the captured pass ran: the eager pass before it did
        the iteration's work) and the runtime's last error set: drain the device and clear the error before the next launch."""
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        if hasattr(self.eng.be, "clear_error"):
            self.eng.be.clear_error()

    def iterate(self, it: int) -> Optional[float]:
        """One NIPALS inner iteration (tpls.py:80-107).  Returns |u_old - u|_2 (None on the first
        pass of a component, where the reference compares against +inf).

        The rank-1 extraction is launched with a squaring budget learnt from the previous iteration
        (+3); its convergence flag comes back with the convergence norm in the single device->host
        copy the iteration needs anyway, and in the rare case the budget was too small the tail of the
        iteration is redone with the full budget (identically on every rank: the flag is a
        deterministic function of the all-reduced Z)."""
        if self.algorithm == "xcov":
            return self._iterate_xcov(it)
        be, comm = self.eng.be, self.eng.comm
        if self._fused:
            return self._iterate_fused(it)
        self._executed += 1
        sharded = comm.sharded
        par = self._parity                       # which of the two u buffers holds the current u
        u, u_new = (self.u, self.u_new) if par == 0 else (self.u_new, self.u)

        def seg_contract():
            for b, blk in enumerate(self.blocks):
                be.mode0_contract(self.X2[b], u, blk.has_miss, out=self.Zs[b])   # tpls.py:80-83

        def seg_colscale():
            for b, blk in enumerate(self.blocks):
                if blk.has_miss:
                    be.colscale(self.Zs[b], blk.colcnt, self.n_total)            # missingvals.py:17-19

        def seg_loadings_scores():
            for b, blk in enumerate(self.blocks):
                self.eng._rank1(blk, self.Zs[b], self.wA[b], self.wB[b], info=self.status[1 + 2 * b: 3 + 2 * b],
                                n_squarings=self.sq_budget[b], fac=self.fac[b], tol=self.tol)   # tpls.py:84-90
                be.score(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b],
                         blk.rowcnt if blk.has_miss else None, self.Ts[b])        # tpls.py:92-99
            if self.t.data_ptr() != self.Ts.data_ptr():
                be.scores_mean(self.Ts, self.t)                                  # cmtf.py:120
            be.gram_tn(self.Y, self.t, out=self.q)                               # tpls.py:100

        def seg_y_update():
            be.normalize(self.q)                                                 # tpls.py:101
            if sharded:
                # |u_old - u|^2 = dq^T (Y^T Y) dq with the all-reduced Gram: no third collective
                be.rowdot(self.Y, self.q, u_new, None)                           # tpls.py:102
                if it > 0:
                    be.quadform(self.Gy, self.q, self.q_prev, self.status[0:1])  # tpls.py:103
            else:
                be.rowdot(self.Y, self.q, u_new, u if it > 0 else None, du2=self.status[0:1])   # tpls.py:102-103

        first = True
        while True:
            budgets = tuple(self.sq_budget)
            if not sharded:
                def whole():
                    if first:
                        seg_contract()
                        seg_colscale()
                    seg_loadings_scores()
                    seg_y_update()
                self._run(("iter", it > 0, par, budgets, first), whole)
            else:
                if first:
                    self._run(("contract", par), seg_contract)
                    for b in range(len(self.blocks)):
                        comm.allreduce(self.Zs[b])
                    seg_colscale()
                self._run(("loadings", budgets), seg_loadings_scores)
                comm.allreduce(self.q)
                self._run(("yupdate", it > 0, par), seg_y_update)
            host = self._read_status()
            if not self._update_budgets(host):
                break
            first = False
        if sharded:
            self.q_prev.copy_(self.q)            # only after the accepted attempt (a retry must compare against
        self._parity ^= 1                        # the previous ITERATION's q, not the rejected attempt's)
        return None if it == 0 else math.sqrt(max(float(host[0]), 0.0))          # tpls.py:103

    def _iterate_fused(self, it: int) -> Optional[float]:
        """The direct iteration for one X block with the Y side fused into the two sweeps:
        contraction with u = Y q formed in the kernel (tpls.py:80-83 + 102), rank-1 (84-90), score with the
        partial sums of Y^T t (92-100), and ONE small launch for q = sum / norm and |du|^2 (100-103)."""
        be, comm = self.eng.be, self.eng.comm
        self._executed += 1
        sharded = comm.sharded
        par = self._parity
        q_cur, q_new = self.qbuf[par], self.qbuf[par ^ 1]
        nparts = len(self.blocks) * int(be.n_partials)

        def seg_contract():
            for b, blk in enumerate(self.blocks):
                if be.mode0_contract_yq(self.X2[b], self.Y, q_cur, blk.has_miss, out=self.Zs[b]) is None:
                    be.rowdot(self.Y, q_cur, self.u, None)                       # shape outside the fused form
                    be.mode0_contract(self.X2[b], self.u, blk.has_miss, out=self.Zs[b])

        def seg_colscale():
            for b, blk in enumerate(self.blocks):
                if blk.has_miss:
                    be.colscale(self.Zs[b], blk.colcnt, self.n_total)            # missingvals.py:17-19

        def seg_loadings_scores():
            for b, blk in enumerate(self.blocks):
                self.eng._rank1(blk, self.Zs[b], self.wA[b], self.wB[b], info=self.status[1 + 2 * b: 3 + 2 * b],
                                n_squarings=self.sq_budget[b], fac=self.fac[b], tol=self.tol)
                if be.score_gram(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b], blk.rowcnt if blk.has_miss else None,
                                 self.Ts[b], self.Y, self.qpart[b]) is None:
                    # _fused is only chosen for M <= 64, the one shape limit of score_gram: anything else is a bug,
                    # and q_update must not sum partial rows nobody wrote
                    raise RuntimeError("score_gram refused a shape the fused iteration was planned for")
            if sharded:
                be.q_update(q_new, self.qpart, normalize=False, nparts=nparts)   # local sum_b Y^T t_b; all-reduced next

        def seg_y_update():
            if sharded:
                be.q_update(q_new, None, normalize=True, G=self.Gy, q_prev=q_cur, du2=self.status[0:1])
            else:
                be.q_update(q_new, self.qpart, normalize=True, G=self.Gy, q_prev=q_cur, du2=self.status[0:1], nparts=nparts)

        # the previous component's deflation already produced this contraction (see _finish_fused)
        have_z = self._z_ready and it == 0
        self._z_ready = False
        first = True
        while True:
            budgets = tuple(self.sq_budget)
            if not sharded:
                def whole():
                    if first:
                        if not have_z:
                            seg_contract()
                        seg_colscale()
                    seg_loadings_scores()
                    seg_y_update()
                self._run(("fiter", par, budgets, first, have_z), whole)
            else:
                def whole_sharded():
                    if first:
                        if not have_z:
                            seg_contract()
                        for b in range(len(self.blocks)):
                            comm.allreduce(self.Zs[b])
                        seg_colscale()
                    seg_loadings_scores()
                    comm.allreduce(q_new)
                    seg_y_update()
                if not self._run_with_collectives(("fiter+ar", par, budgets, first, have_z), whole_sharded):
                    if first:
                        if not have_z:
                            self._run(("fcontract", par), seg_contract)
                        for b in range(len(self.blocks)):
                            comm.allreduce(self.Zs[b])
                        seg_colscale()
                    self._run(("floadings", par, budgets), seg_loadings_scores)
                    comm.allreduce(q_new)
                    self._run(("fyupdate", par), seg_y_update)
            host = self._read_status()
            if not self._update_budgets(host):
                break
            first = False
        self._parity ^= 1
        self.q = q_new
        return None if it == 0 else math.sqrt(max(float(host[0]), 0.0))          # tpls.py:103

    def finish_component(self, a: int) -> None:
        be, comm = self.eng.be, self.eng.comm
        self.n_iter.append(self._executed)
        ssqs = []
        if self.algorithm == "xcov" and self._s_carry:
            if self._nowrite:
                self._finish_xcov_nowrite(a)
            else:
                self._finish_xcov_carry(a)
            return
        if (self.algorithm == "xcov" and len(self.blocks) == 1 and self.blocks[0].has_miss and self.S12[0] is not None
                and a + 1 < self.R and hasattr(be, "xcov_deflate") and self.eng.opt.xcov_deflate_build
                and getattr(self, "_deflate_build_ok", True)):
            if self._finish_xcov_masked_fused(a):
                return
        if self.algorithm == "xcov":
            # the final score (tpls.py:92-99 with the converged loadings) and the deflation (tpls.py:109)
            # are the only other passes over X: fused into one read + one write when there is one block
            self.q = self.qc
            if len(self.blocks) == 1:
                blk = self.blocks[0]
                rc = blk.rowcnt if blk.has_miss else None
                s0 = be.score_deflate(self.X2[0], blk.A, blk.B, self.wA[0], self.wB[0], rc, self.t)
                if s0 is None:
                    be.score(self.X2[0], blk.A, blk.B, self.wA[0], self.wB[0], rc, self.t)
                    s0 = be.deflate(self.X2[0], blk.A, blk.B, self.t, self.wA[0], self.wB[0])
                ssqs.append(s0)
            else:
                for b, blk in enumerate(self.blocks):
                    be.score(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b], blk.rowcnt if blk.has_miss else None, self.Ts[b])
                be.scores_mean(self.Ts, self.t)
                for b, blk in enumerate(self.blocks):
                    ssqs.append(be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b]))
            be.rowdot(self.Y, self.q, self.u, None)                              # u = Y q (tpls.py:102)
        if self.algorithm == "direct" and self._fused:
            be.rowdot(self.Y, self.q, self.u, None)      # u = Y q of the last iteration (tpls.py:102), once
            if self.t.data_ptr() != self.Ts.data_ptr():
                be.scores_mean(self.Ts, self.t)          # cmtf.py:120, once per component (the loop needs only Y^T t)
        elif self.algorithm == "direct" and self._parity == 1:
            self.u.copy_(self.u_new)                     # make self.u the current u again; the two buffers keep
            self._parity = 0                             # their identity (captured graphs hold their addresses)
        self.T[:, a].copy_(self.t)
        self.U[:, a].copy_(self.u)
        self.Q[:, a].copy_(self.q)
        for b, blk in enumerate(self.blocks):
            if len(blk.shape) == 2:
                blk.loadings[0][:, a].copy_(self.wB[b])
            elif len(blk.shape) == 3:
                blk.loadings[0][:, a].copy_(self.wA[b])
                blk.loadings[1][:, a].copy_(self.wB[b])
            else:
                for m, d in enumerate(blk.shape[1:]):
                    blk.loadings[m][:, a].copy_(self.fac[b][m, :d])
            if self.algorithm == "direct" and not self._fused:
                ssqs.append(be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b]))   # tpls.py:109
        if self.algorithm == "direct" and self._fused:
            self._finish_fused(a)
            return
        # inner regression: coef_[:, a] = lstsq(T, u) with columns > a still zero (tpls.py:110-112)
        b_dev, _ = self._inner_regression(a)
        for b in range(len(self.blocks)):
            self.ssq_log[a, b].copy_(ssqs[b].reshape(()))                        # tpls.py:115-117 (booked in result())
        ssqy = be.y_deflate(self.Y, self.T, a + 1, b_dev, self.q)                # tpls.py:113
        self.ssq_log[a, len(self.blocks)].copy_(ssqy.reshape(()))                # tpls.py:118-120

    def _finish_xcov_masked_fused(self, a: int) -> bool:
        """finish_component of the xcov algorithm for ONE block WITH missing values, component a < R - 1.  The masked deflation
        is not a rank-one update of S, so S (and S2) are rebuilt for every component; the rebuild reads exactly what the
        deflation has just written.  Here the deflation happens INSIDE the rebuild: the final score (one read), then the Y
        side (inner regression, Y deflation: they need only T and u, tpls.py:110-113), then one read + write of X that
        deflates it (tpls.py:109) and accumulates [S; S2] = [Y, Y * rowscale]^T X0 of the deflated block and its norm on the
        matrix cores (backend.xcov_deflate).  Three passes' worth of traffic per component instead of four.  Returns False
        (nothing done) when the kernel does not take the shape."""
        be, comm = self.eng.be, self.eng.comm
        blk, M = self.blocks[0], self.M
        if blk.A * blk.B % 4 != 0:
            self._deflate_build_ok = False
            return False
        self.q = self.qc
        be.score(self.X2[0], blk.A, blk.B, self.wA[0], self.wB[0], blk.rowcnt, self.t)      # tpls.py:92-99, masked (missingvals.py:23-38)
        be.rowdot(self.Y, self.q, self.u, None)                                  # u = Y q (tpls.py:102)
        self._store_component(a)
        b_dev, _ = self._inner_regression(a)                                     # tpls.py:110-112
        ssqy = be.y_deflate(self.Y, self.T, a + 1, b_dev, self.q)                # tpls.py:113
        self.Yw[:, :M].copy_(self.Y)
        torch.mul(self.Y, self.rowscale[0][:, None], out=self.Yw[:, M:])
        ssq = be.xcov_deflate(self.X2[0], blk.A, blk.B, self.Yw, self.t, self.wA[0], self.wB[0], out=self.S12[0])
        if ssq is None:                                                          # (nothing written) deflate now, rebuild S at start_component
            self._deflate_build_ok = False
            ssq = be.deflate(self.X2[0], blk.A, blk.B, self.t, self.wA[0], self.wB[0])
        else:
            comm.allreduce(self.S12[0])
            self._s_ready = True
        self._log_ssq(a, [ssq], ssqy)
        return True

    def _inner_regression(self, a: int, extra: Optional[torch.Tensor] = None):
        """b = lstsq(T[:, :a+1], u) (tpls.py:110-112) from the normal equations, entirely on the device: Gram
        and right-hand side (all-reduced when sharded, together with `extra`), equilibrated Cholesky in one
        workgroup; the coefficients go into column a of the device coef matrix.  Returns (b, reduced extra)."""
        be, comm = self.eng.be, self.eng.comm
        k = a + 1
        Ta = self.T[:, :k]
        G = be.gram_tn(Ta, Ta)
        g = be.gram_tn(Ta, self.u)
        if comm.sharded:
            packed = torch.cat([G.reshape(-1), g.reshape(-1)] + ([extra.reshape(-1)] if extra is not None else []))
            comm.allreduce(packed)
            G, g = packed[: k * k].view(k, k), packed[k * k: k * k + k]
            extra = packed[k * k + k:] if extra is not None else None
        b_dev = be.normal_solve(G, g.reshape(-1), out=self.b_dev[:k])
        self.coef_dev[:k, a].copy_(b_dev)
        self._G_last = G                                     # T^T T (global): t_j^T (T b) = (G b)_j without another reduction
        return b_dev, extra

    def _finish_fused(self, a: int) -> None:
        """Tail of finish_component on the fused direct path.  The inner regression and the Y deflation
        (tpls.py:110-113) depend only on T and u, so they run BEFORE the X deflation (tpls.py:109); the X
        deflation can then be fused with the first contraction of component a+1 (u_0 = Y_new[:, 0] is known):
        one X read less per component.  The deflated norms behind R2X / R2Y (tpls.py:115-120) stay on the
        device (ssq_log) and are read back once, in result(): a component's epilogue has no host round trip."""
        be, comm = self.eng.be, self.eng.comm
        k = a + 1
        b_dev, _ = self._inner_regression(a)                                     # tpls.py:110-112
        ssqy = be.y_deflate(self.Y, self.T, k, b_dev, self.q)                    # tpls.py:113
        ssqs = []
        self._z_ready = False
        if k < self.R:
            # u_0 of the next component is the first column of the deflated Y = Y e_0 (tpls.py:78)
            q0 = self.qbuf[0]
            q0.zero_()
            q0[0] = 1.0
            ready = True
            for b, blk in enumerate(self.blocks):
                s_b = be.deflate_contract_yq(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b], self.Y, q0,
                                             blk.has_miss, out=self.Zs[b])
                if s_b is None:                                                  # shape outside the fused form
                    s_b = be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b])
                    ready = False
                ssqs.append(s_b)
            self._z_ready = ready
        else:
            for b, blk in enumerate(self.blocks):
                ssqs.append(be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b]))   # tpls.py:109
        self._log_ssq(a, ssqs, ssqy)

    def _log_ssq(self, a: int, ssqs, ssqy) -> None:
        for b in range(len(self.blocks)):
            self.ssq_log[a, b].copy_(ssqs[b].reshape(()))                        # tpls.py:115-117 (booked in result())
        self.ssq_log[a, len(self.blocks)].copy_(ssqy.reshape(()))                # tpls.py:118-120

    def _store_component(self, a: int) -> None:
        self.T[:, a].copy_(self.t)
        self.U[:, a].copy_(self.u)
        self.Q[:, a].copy_(self.q)
        self._store_loadings(a)

    def _store_loadings(self, a: int) -> None:
        for b, blk in enumerate(self.blocks):
            if len(blk.shape) == 2:
                blk.loadings[0][:, a].copy_(self.wB[b])
            elif len(blk.shape) == 3:
                blk.loadings[0][:, a].copy_(self.wA[b])
                blk.loadings[1][:, a].copy_(self.wB[b])
            else:
                for m, d in enumerate(blk.shape[1:]):
                    blk.loadings[m][:, a].copy_(self.fac[b][m, :d])

    def _finish_xcov_carry(self, a: int) -> None:
        """finish_component of the xcov algorithm when S is carried across the deflation.  Passes over X:
        the final score (read; tpls.py:92-99 with the converged loadings) and the deflation (read + write;
        tpls.py:109), which also forms v = X+^T yhat for the down-date of S -- no S build on the matrix
        cores for the next component.  R2 bookkeeping is deferred to result() as in _finish_fused."""
        be, comm = self.eng.be, self.eng.comm
        self.q = self.qc
        for b, blk in enumerate(self.blocks):
            be.score(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b], None, self.Ts[b])
        if self.t.data_ptr() != self.Ts.data_ptr():
            be.scores_mean(self.Ts, self.t)                                      # cmtf.py:120
        be.rowdot(self.Y, self.q, self.u, None)                                  # u = Y q (tpls.py:102)
        self._store_component(a)
        k = a + 1
        Ta = self.T[:, :k]
        ya = be.gram_tn(self.Y, self.t).reshape(-1)                              # Y^T t with the not yet deflated Y
        b_dev, ya_g = self._inner_regression(a, extra=ya)                        # tpls.py:110-112; ya_g: all-reduced Y^T t
        ssqs = []
        if k < self.R:
            be.rowdot(Ta, b_dev, self.yhat.view(-1), None)                       # yhat = T b (what Y is deflated by)
        ssqy = be.y_deflate(self.Y, self.T, k, b_dev, self.q)                    # tpls.py:113
        if k < self.R:
            carried = True
            for b, blk in enumerate(self.blocks):
                s_b = be.deflate_contract_yq(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b], self.yhat, self.one,
                                             False, out=self.vs[b])
                if s_b is None:                                                  # shape outside the fused form
                    s_b = be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b])
                    carried = False
                ssqs.append(s_b)
            if carried:
                for b, blk in enumerate(self.blocks):
                    comm.allreduce(self.vs[b])
                    be.s_downdate(self.S[b], blk.A, blk.B, ya_g, self.wA[b], self.wB[b], self.q, self.vs[b])
            self._s_ready = carried
        else:
            for b, blk in enumerate(self.blocks):
                ssqs.append(be.deflate(self.X2[b], blk.A, blk.B, self.t, self.wA[b], self.wB[b]))   # tpls.py:109
        self._log_ssq(a, ssqs, ssqy)

    def _finish_xcov_nowrite(self, a: int) -> None:
        """finish_component of the xcov algorithm WITHOUT writing X (blocks without missing values).

        The deflation X_{a+1} = X_a - t_a w_a^T (tpls.py:109) is linear, so X_a = X_0 - sum_{j<a} t_j w_j^T and nothing the
        loop needs from X_a requires X_a itself:
          score      X_a w_a = X_0 w_a - sum_{j<a} t_j (w_j^T w_a),   w_j^T w_a = prod_modes (W_m^T W_m)[j, a]            [one read]
          down-date  X_{a+1}^T yhat = X_0^T yhat - sum_{j<=a} w_j (t_j^T yhat),   t_j^T yhat = (T^T T b)_j               [one read]
          R2X        |X_{b,a+1}|^2 = |X_{b,a}|^2 - 2 t^T t_b + t^T t   (t_b: the block's own score; |w_b| = 1)   [result()]
        Two reads of X per component instead of a read and a read + write; the last component needs no second pass
        at all.  X stays as centred.  Same S, same iterations, same scores up to f64 rounding (tests compare this form with
        the deflating one, `NipalsEngine.xcov_nowrite = False`).

        `xcov_one_read`: yhat = T b is a combination of the scores, so X_0^T yhat = sum_j b_j r_j with r_j = X_0^T t_j.  The
        score pass over a block can form r_a itself, in the same read, once everything else t_a is made of is known:
        t_a = mean_b t_b (cmtf.py:120), t_b = X_{b,0} w_{b,a} - T[:, :a] g_b -- the correction T g_b is known before the pass
        and so are the other blocks' scores if this block is read LAST.  backend.score_contract forms t_b and X_0^T t_a per
        row (dot product, then the row times the averaged score); the r_j are kept (P x R) and the block's second read per
        component becomes a P x (a+1) matrix-vector product.  Applied to the largest block (the only one of a tPLS fit: ONE
        read of X per component); the other blocks of a coupled fit keep their two reads."""
        be, comm = self.eng.be, self.eng.comm
        self.q = self.qc
        I, R, k = self.I, self.R, a + 1
        nb = len(self.blocks)
        self._store_loadings(a)
        fused_b = self._fused_b if (getattr(self, "_one_read", False) and k < R) else -1
        one_read = False
        for b in [x for x in range(nb) if x != fused_b] + ([fused_b] if fused_b >= 0 else []):      # the fused block is read last
            blk = self.blocks[b]
            mw = None
            if self.raw:                                                         # uncentred X: X_c w = X w - (mean^T w) 1
                mw = be.score_s(blk.mean.view(1, -1), blk.A, blk.B, self.wA[b], self.wB[b], be.empty(1))
            g = None
            if a > 0 and hasattr(be, "kr_gram_row"):
                for m, L in enumerate(blk.loadings):                             # Gram of a Khatri-Rao product = Hadamard product
                    be.kr_gram_row(L, a, self.Gw, first=(m == 0))                # of the mode Grams; only its row a is needed:
                g = self.Gw[:a]                                                  # w_j^T w_a, j < a
            elif a > 0:
                for m, L in enumerate(blk.loadings):
                    be.kr_gram(L, self.Gw, first=(m == 0))
                g = self.Gw.view(R, R)[a, :a]                                    # (row a of the symmetric Gram)
            if b == fused_b:
                # t_b = X_0 w_a - T[:, :a] g and r_a = X_0^T t_a (t_a: the average over the blocks, cmtf.py:120) from the same read
                # of X (this rank's rows; r_a summed over ranks below)
                corr = others = None
                if a > 0:
                    be.rowdot(self.T[:, :a], g, self.corr, None)                 # T[:, :a] g
                    corr = self.corr
                if nb == 2:
                    others = self.Ts[1 - b]                                      # (the other blocks' scores are final by now)
                elif nb > 2:
                    others = self.Ts[[x for x in range(nb) if x != b]].sum(dim=0)
                one_read = be.score_contract(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b], mw, self.Ts[b], self.ps,
                                             sub_own=corr, add_other=others, alpha=1.0 / nb,
                                             csum=self.csum if self.raw else None) is not None
                if one_read:
                    continue
                self._one_read = False                                           # shape outside that kernel: two passes from here on
            be.score(self.X2[b], blk.A, blk.B, self.wA[b], self.wB[b], None, self.Ts[b])       # X_0 w_a
            if mw is not None:
                be.axpy_scalar(self.Ts[b], mw)
            if a > 0:
                be.y_deflate(self.Ts[b].view(I, 1), self.T, a, g, self.one)      # t_b -= T[:, :a] g
        single = self.t.data_ptr() == self.Ts.data_ptr()
        if not single:
            be.scores_mean(self.Ts, self.t)                                      # cmtf.py:120
        if one_read:
            comm.allreduce(self.ps)
            if self.raw:                                                         # X_c^T t = X^T t - (1^T t) mean; 1^T t from the same pass
                be.axpy_scalar(self.ps, comm.allreduce(self.csum), self.blocks[fused_b].mean)
            self.Rm[:, a].copy_(self.ps)
        be.rowdot(self.Y, self.q, self.u, None)                                  # u = Y q (tpls.py:102)
        be.gram_tn(self.t, self.t, out=self.dot_log[a, 0:1])
        for b in range(nb):
            if single:
                self.dot_log[a, 1 + b].copy_(self.dot_log[a, 0])
            else:
                be.gram_tn(self.Ts[b], self.t, out=self.dot_log[a, 1 + b: 2 + b])
        self.T[:, a].copy_(self.t)
        self.U[:, a].copy_(self.u)
        self.Q[:, a].copy_(self.q)
        Ta = self.T[:, :k]
        ya = be.gram_tn(self.Y, self.t).reshape(-1)                              # Y^T t with the not yet deflated Y
        b_dev, ya_g = self._inner_regression(a, extra=ya)                        # tpls.py:110-112; ya_g: all-reduced Y^T t
        if k < R and not (one_read and nb == 1):                                 # (the fused block needs no yhat: X_0^T yhat = Rm b)
            be.rowdot(Ta, b_dev, self.yhat.view(-1), None)                       # yhat = T b (what Y is deflated by)
        ssqy = be.y_deflate(self.Y, self.T, k, b_dev, self.q)                    # tpls.py:113
        if k < R:
            c = be.gram_tn(self._G_last, b_dev).reshape(-1)                      # t_j^T yhat = (T^T T b)_j, j <= a (global)
            for b, blk in enumerate(self.blocks):
                WA, WB = self.eng._kr_operands(blk, R)                           # columns <= a: the components so far
                if one_read and b == fused_b:                                    # X_0^T yhat = sum_j b_j (X_0^T t_j): no read of X
                    be.rowdot(self.Rm[:, :k], b_dev, self.vs[b], None)
                else:
                    be.mode0_contract(self.X2[b], self.yhat.view(-1), False, out=self.vs[b])   # X_0^T yhat
                    comm.allreduce(self.vs[b])
                    if self.raw:                                                 # uncentred X: X_c^T yhat = X^T yhat - (1^T yhat) mean
                        be.axpy_scalar(self.vs[b], comm.allreduce(be.total(self.yhat.view(-1))), blk.mean)
                be.kr_axpy(self.vs[b], blk.A, blk.B, WA, WB, k, c)
                be.s_downdate(self.S[b], blk.A, blk.B, ya_g, self.wA[b], self.wB[b], self.q, self.vs[b])
            self._s_ready = True
        self.ssq_log[a, nb].copy_(ssqy.reshape(()))                              # tpls.py:118-120

    def result(self) -> FitState:
        """The only device -> host traffic of the component epilogues: the R x R coefficients and the
        R x (blocks + 1) deflated norms, all-reduced once, in one copy."""
        if getattr(self, "_state", None) is not None:             # the norms are all-reduced exactly once
            return self._state
        nb = len(self.blocks)
        nowrite = self.algorithm == "xcov" and getattr(self, "_nowrite", False)
        for b, ssq in getattr(self, "_ssq0_dev", {}).items():
            self.blocks[b].ssq0 = float(ssq.item())
        self.eng.comm.allreduce(self.ssq_log)
        parts = [self.coef_dev.reshape(-1), self.ssq_log.reshape(-1)]
        if nowrite:
            self.eng.comm.allreduce(self.dot_log)
            parts.append(self.dot_log.reshape(-1))
        host = torch.cat(parts).cpu().numpy()
        R = self.R
        self.coef[...] = host[: R * R].reshape(R, R)
        ssq = host[R * R: R * R + R * (nb + 1)].reshape(R, nb + 1).copy()
        if nowrite:
            # X was never deflated: |X_{b,a+1}|^2 = |X_{b,a}|^2 - 2 t_a^T t_{b,a} + t_a^T t_a from the logged dot products
            dots = host[R * R + R * (nb + 1):].reshape(R, 1 + nb)
            for b, blk in enumerate(self.blocks):
                run = blk.ssq0
                for a in range(len(self.n_iter)):
                    run = run - 2.0 * dots[a, 1 + b] + dots[a, 0]
                    ssq[a, b] = run
        for a in range(len(self.n_iter)):
            for b, blk in enumerate(self.blocks):
                blk.r2x[a] = 1.0 - ssq[a, b] / blk.ssq0                          # tpls.py:115-117
            self.r2y[a] = 1.0 - ssq[a, nb] / self.ssqy0                          # tpls.py:118-120
        self._state = FitState(coupled=self.coupled, n_components=self.R, blocks=self.blocks, T=self.T, U=self.U, Q=self.Q,
                               coef=self.coef, r2y=self.r2y, y_mean=self.y_mean, n_iter=self.n_iter,
                               n_samples_total=self.n_total, report=self.build_report())
        return self._state

    def build_report(self) -> Dict[str, object]:
        """What actually ran (FitState.report, `tPLS.fit_report_`, bench.py `fit.path`): the algorithm, and for every fast form
        whether it was taken or which condition declined it.  Reads of X are counted per component of the steady state."""
        eng, comm, nb = self.eng, self.eng.comm, len(self.blocks)
        rep: Dict[str, object] = {
            "form": "regular", "algorithm_requested": self.algorithm_requested, "algorithm": self.algorithm,
            "shapes": [tuple(b.shape) for b in self.blocks], "storage": [str(b.dtype).replace("torch.", "") for b in self.blocks],
            "missing": [bool(b.has_miss) for b in self.blocks], "responses": self.M,
            "sharded": bool(comm.sharded), "world": int(comm.world),
            "graphs": bool(self.use_graphs and self._graphs), "graph_error": self._graph_error,
            "collectives_in_graph": self._collectives_captured is True,
            "backend": getattr(eng.be, "name", type(eng.be).__name__),
        }
        if self.algorithm == "direct":
            rep["y_side"] = "fused into the sweeps" if self._fused else "separate launches"
            if not self._fused and self.M > 64:
                self.notes.append("Y side not fused into the sweeps: more than 64 responses")
            rep["x_passes_per_iteration"] = "2 reads"
            rep["x_passes_per_component"] = ("1 read + write (deflation fused with the next contraction)" if self._fused
                                             else "1 read + write (deflation)")
            rep["x_copy"] = "centred private copy"
        else:
            nowrite = bool(getattr(self, "_nowrite", False))
            one_read = nowrite and bool(getattr(self, "_one_read", False))
            any_miss = any(b.has_miss for b in self.blocks)
            rep["x_copy"] = "none: the caller's uncentred tensor is read in place" if self.raw else "centred private copy"
            rep["raw"] = bool(self.raw)
            rep["x_written"] = not nowrite
            rep["s_carried"] = bool(self._s_carry)
            rep["s_build"] = ("per component (missing values)" if any_miss else "first component only") + \
                             (f", {(self.M + 63) // 64} response tiles of <= 64" if self.M > 64 else "") + \
                             (", [Y, Y * rowscale] in one pass" if any(x is not None for x in self.S12) else "")
            rep["one_read"] = one_read
            if nowrite:
                rep["x_passes_per_component"] = ("1 read (largest block); 2 reads (other blocks)" if one_read and nb > 1 else
                                                 "1 read" if one_read else "2 reads")
            elif any_miss and nb == 1 and getattr(self, "_deflate_build_ok", True) and self.S12[0] is not None and eng.opt.xcov_deflate_build:
                rep["x_passes_per_component"] = "1 read + 1 read + write (deflation inside the rebuild of S)"
            elif any_miss:
                rep["x_passes_per_component"] = "1 read + write (score + deflation) + S rebuild reads"
            else:
                rep["x_passes_per_component"] = "1 read + 1 read + write"
            rep["pipelined"] = bool(getattr(self, "_pipelined", False))
            if rep["pipelined"]:
                rep["pipeline"] = dict(self.pipeline_stats)
            if not rep["pipelined"] and eng.opt.xcov_pipeline:
                why = ("graph replay requested" if self.use_graphs else "more than 64 responses" if self.M > 64 else
                       "a block of order > 3 or a backend without the single-call iteration")
                self.notes.append("inner loop on S not pipelined: " + why)
            if self.algorithm_requested == "xcov" and not self.raw and eng.opt.xcov_raw and not any_miss and not any("uncentred" in n for n in self.notes):
                self.notes.append("uncentred xcov form declined: " + ("more than 64 components" if self.R > 64 else
                                                                      "f32 matrix precision or a backend without its kernels"))
            if nowrite and eng.opt.xcov_one_read and self.R > 1 and not one_read:
                self.notes.append("one read per component declined: the row does not fit the registers of one workgroup "
                                  "(rows of 2048..16384 f32 / 1024..8192 f64 elements)")
        rep["declined"] = list(self.notes)
        return rep
