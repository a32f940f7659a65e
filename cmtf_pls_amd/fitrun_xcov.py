"""The cross-covariance form of the NIPALS loop (algorithm="xcov"), as a mixin of `fitrun.FitRun`.

Inside one component X and Y are fixed and u = Y q, so np.einsum(X, u) = sum_m q_m S_m (tpls.py:83), Y.T @ t = S_(0) kron(wA, wB)
(tpls.py:100) and |u_old - u|^2 = dq^T (Y^T Y) dq (tpls.py:103) with S = X_(0)^T Y: the inner loop runs on S alone.
  _iterate_xcov                one iteration on S, waiting for its convergence norm
  _inner_loop_xcov_pipelined   the same with iteration it + 1 enqueued before the host has seen iteration it's norm
  _finish_xcov_carry           component epilogue: S carried across the deflation by a rank-two down-date, X deflated in place
  _finish_xcov_nowrite         ... X never written (nor centred: `raw`), the largest block read ONCE per component
  _finish_xcov_masked_fused    one block WITH missing values: the deflation inside the rebuild of S
"""
from __future__ import annotations

import math
from typing import Optional

import torch


class XcovMixin:
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
            if any(host[2 + 2 * b] < 0 for b in range(nb)):         # the one-launch chain gave up (a shared GPU): launches from now on
                self._chain_gave_up()
                pp["plans"].clear()
                tok = enqueue(it, first=False)
                stats["redone"] += 1
                continue
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
                    self.sq_budget[b] = min(self.sq_max, int(host[2 + 2 * b]) + self.sq_spare[b])
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
