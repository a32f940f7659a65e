"""`FitRun`: one fit in flight -- the state between NIPALS iterations and the component loop of tPLS.fit / ctPLS.fit
(reference cmtf_pls/tpls.py:73-120, cmtf_pls/cmtf.py:85-140).  `NipalsEngine.fit` drives it; bench.py drives `iterate` directly so
that the timed step IS the product's iteration.

Map of this file (the forms are exact re-associations of one loop; DESIGN.md section 5, each parity-tested against the direct form):
  __init__                 preprocess, buffers, which forms apply (recorded for `build_report`)
  start_component          u_0 = Y[:, 0]; algorithm="xcov": S = X_(0)^T Y (built, or carried from the previous component)
  iterate / _iterate_fused the DIRECT iteration (two reads of X), unfused / Y side fused into the sweeps; sharded: two all-reduces
  fitrun_xcov.XcovMixin    _iterate_xcov, _inner_loop_xcov_pipelined: the same iteration on S (no read of X, no communication)
  finish_component         score, deflation, inner regression, Y deflation, in one of five forms:
      (direct, unfused)  in finish_component itself        _finish_fused            direct with the deflation fused into the next contraction
      (fitrun_xcov.py)   _finish_xcov_carry: S down-dated, X deflated; _finish_xcov_nowrite: S down-dated, X never written (raw / one
                         read); _finish_xcov_masked_fused: blocks with missing values, the deflation inside the rebuild of S
  result / build_report    the only device -> host traffic of the epilogues; what actually ran
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .fitrun_xcov import XcovMixin
from .state import BlockState, FitState, split_trailing, validate_limits


class FitRun(XcovMixin):
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
        # the responses first (tpls.py:67-68: nothing of it depends on X): the cross-covariance of a candidate for the uncentred
        # form is taken in the same read as its column statistics, and that needs the centred Y
        ysum, ycnt = be.colstats(Y)
        comm.allreduce(ysum)
        comm.allreduce(ycnt)
        self.y_mean = ysum / ycnt                                 # tpls.py:67
        _, ssqy0 = be.center(Y, self.y_mean, False)
        comm.allreduce(ssqy0)
        self.ssqy0 = float(ssqy0.item())
        # |X - X_mean|^2 of an uncentred block: from the read that builds S for the first component (backend.xcov_ssq),
        # else from a read of its own (_ssq_uncentred)
        self._ssq_with_s = (want_raw and hasattr(be, "xcov_ssq") and eng.opt.xcov_ssq_with_s)
        self._ssq0_dev = {}
        self._s_prebuilt = {}                                     # block -> this rank's S = X^T Y_c from the statistics read
        self.blocks = None
        self._stats_with_s = False
        if (want_raw and self._ssq_with_s and eng.opt.xcov_stats_with_s and hasattr(be, "xcov_stats") and M <= 64
                and eng._probe_plain(Xs)):
            self.blocks = self._blocks_from_one_read(Xs)
            self._stats_with_s = self.blocks is not None
            if self.blocks is None:                               # a missing / non-finite value after all: the statistics pass proper
                self._s_prebuilt, self._ssq0_dev = {}, {}
                self.notes.append("column statistics from the read that builds S declined: a column sum came out non-finite")
        if self.blocks is None:
            self.blocks = [eng._prepare_block(X, self.n_total, defer_centring=want_raw) for X in Xs]
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
        if want_raw and not self.raw:                            # (an S taken from the uncentred tensor serves the uncentred form only)
            self._s_prebuilt, self._ssq0_dev, self._stats_with_s = {}, {}, False
        self.X2 = [X.view(I, -1) for X in Xs]
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
        # spare squarings kept above the last extraction's need.  Where the backend runs the squarings as ONE launch (min(J, K) <=
        # rank1_chain_side: an unused step costs nothing, a missing one a redone iteration) keep three; a launch per squaring: one
        side = int(getattr(be, "rank1_chain_side", 0))
        self.sq_spare = [3 if (len(blk.shape) == 3 and min(blk.A, blk.B) <= side) else 1 for blk in self.blocks]
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
                    pre = self._s_prebuilt.get(len(self.S))
                    self.S.append(pre if pre is not None else be.empty(M, blk.A * blk.B))
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

    def _blocks_from_one_read(self, Xs: List[torch.Tensor]) -> Optional[List[BlockState]]:
        """tpls.py:61-71 for candidates of the uncentred cross-covariance form: ONE read of every block gives its S = X^T Y_c
        (kept for the first component), its column sums (the means) and its column sums of squares (|X - X_mean|^2 = sum_c
        sumsq_c - sum_c^2 / n) -- backend.xcov_stats.  None when a sum is non-finite (missing or infinite values: the regular
        statistics pass decides what to do with them) or the backend declines the shape."""
        be, comm = self.eng.be, self.eng.comm
        blocks, n = [], float(self.n_total)
        for b, X in enumerate(Xs):
            I = X.shape[0]
            X2 = X.view(I, -1)
            P = X2.shape[1]
            out = be.xcov_stats(X2, self.Y, out=be.empty(self.M, P))
            if out is None:
                return None
            S, stats = out
            comm.allreduce(stats)
            colsum, colsq = stats[:P], stats[P:]
            ssq = (colsq - colsum * colsum / n).sum().reshape(1)
            if not bool(torch.isfinite(ssq).all().item()):
                return None
            A, B = split_trailing(X.shape)
            blocks.append(BlockState(shape=tuple(X.shape), A=A, B=B, mean=colsum / n, has_miss=False, colcnt=None, rowcnt=None,
                                     ssq0=float("nan"), dtype=X.dtype))
            self._s_prebuilt[b] = S
            self._ssq0_dev[b] = ssq                               # (summed over the ranks already: the statistics were)
        return blocks

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
            if b in self._s_prebuilt:                             # S and the statistics came from ONE read, in __init__
                del self._s_prebuilt[b]
            elif self.raw and self._ssq_with_s and b not in self._ssq0_dev:
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

    def _update_budgets(self, host) -> bool:
        """Adapt the squaring budget of every order-3 block; True if the iteration tail must be redone."""
        retry = False
        for b in range(len(self.blocks)):
            conv, used = host[1 + 2 * b] > 0.5, int(host[2 + 2 * b])
            if used < 0:                                             # the one-launch chain gave up (a shared GPU): launches from now on
                self._chain_gave_up()
                retry = True
                continue
            if not conv and self.sq_budget[b] < self.sq_max:
                self.sq_budget[b] = self.sq_max
                retry = True
            elif conv and len(self.blocks[b].shape) == 3:
                # the last computing launch (`used`) declares its own output final, or launch used + 1 sees it; keep
                # one spare.  Under graph replay the launch sequence is part of the captured graph: hysteresis
                # (re-plan only outside [used+1, used+3]) keeps it stable; eager launches follow the need exactly
                # (every spare launch is ~4 us of an idle GPU)
                if not self.use_graphs or used + 1 > self.sq_budget[b] or used + 3 < self.sq_budget[b]:
                    self.sq_budget[b] = min(self.sq_max, used + (1 if self.use_graphs else self.sq_spare[b]))
        return retry

    def _peer_failed(self, host, first_attempt: bool) -> bool:
        """Sharded fits only.  Whether an iteration's tail is redone must be the same decision on every rank (the tail holds a
        collective).  Everything a rank decides from is a function of all-reduced data -- except the give-up of the one-launch
        rank-1 chain, which is local.  A rank that gave up leaves NaN loadings, hence a NaN share of Y^T t, hence a NaN q and a NaN
        convergence norm on EVERY rank after the all-reduce: a rank that sees a non-finite norm on its first attempt therefore
        redoes the tail too.  Every rank can give up once (it switches its chain off), possibly during somebody else's repeat: a
        non-finite norm is therefore answered up to world + 1 times per iteration; data that really is non-finite shows again
        after that and is reported by the fit as before."""
        if not self.eng.comm.sharded:
            return False
        if first_attempt:
            self._nan_tails = 0
        if np.isfinite(float(host[0])) or self._nan_tails > int(getattr(self.eng.comm, "world", 1)):
            return False
        self._nan_tails += 1
        return True

    def _chain_gave_up(self) -> None:
        be = self.eng.be
        if hasattr(be, "rank1_chain_gave_up"):
            be.rank1_chain_gave_up()
        self._graphs.clear()                                         # captured sequences hold the chain kernel
        self.sq_spare = [1] * len(self.blocks)
        note = "rank-1 chain of squarings in one launch switched off: a workgroup never became resident (GPU shared with another process)"
        if note not in self.notes:
            self.notes.append(note)

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
                # tpls.py:103; on the first pass (oldU = inf there) the form with q itself: 0, or NaN when q is -- see _peer_failed
                be.quadform(self.Gy, self.q, self.q_prev if it > 0 else self.q, self.status[0:1])
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
            local, peer = self._update_budgets(host), self._peer_failed(host, first)     # (both always evaluated)
            if not (local or peer):
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
            local, peer = self._update_budgets(host), self._peer_failed(host, first)     # (both always evaluated)
            if not (local or peer):
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
            # the column statistics (tpls.py:61-71) out of the read that builds S for the first component: X is read once, not twice,
            # before the first inner loop
            rep["stats_with_s"] = bool(getattr(self, "_stats_with_s", False))
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
                self.notes.append("one read per component declined: the row does not fit the registers of 1..16 workgroups "
                                  "(rows of 2048..131072 f32 -- 262144 when the last mode divides 4096 -- / 1024..131072 f64 elements, "
                                  "last mode a multiple of 4 / 2)")
        # the rank-1 extraction of order-3 blocks with min(J, K) <= 256: every Gram squaring in one launch, unless the process had to
        # switch that off (a GPU shared with another process)
        rep["rank1_one_launch_chain"] = bool(getattr(eng.be, "rank1_chain_side", 0)) and any(
            len(blk.shape) == 3 and min(blk.A, blk.B) <= int(getattr(eng.be, "rank1_chain_side", 0)) for blk in self.blocks)
        rep["declined"] = list(self.notes)
        return rep
