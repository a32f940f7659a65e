"""Projection of new samples, reconstruction and the literal R2X: the engine's `transform` / `predict` side
(reference cmtf_pls/tpls.py:122-189, cmtf.py:142-237), as a mixin of `NipalsEngine`.

Forms, in the order they are tried (`NipalsEngine.last_projection` records which one ran):
  one-pass MTTKRP on the caller's uncentred rows (one read, nothing written)            _project_one_pass / project_readonly
  + the masked sequence for ONLY the incomplete samples, in registers or on compact copies   project_readonly
  the sequential project-and-deflate passes on private copies (the reference's loop)     _project
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .state import BlockState, FitState


class ProjectionMixin:
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
