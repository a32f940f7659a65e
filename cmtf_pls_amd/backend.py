"""HipBackend: the per-GPU kernel set of the NIPALS engine, one method per reference call site.

Every method launches hand-written gfx950 kernels from libcmtfpls.so (include/cmtfpls.h) on the
current torch stream; torch is used only to own device memory and the stream.  There is no CPU
path: constructing the backend without a GPU or without the library raises.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class HipBackend:
    name = "hip"

    def __init__(self, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CmtfplsError("cmtf_pls_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.n_partials = int(self.lib.cmtfpls_sweep_partials())
        self._ws = {}
        self._status_slots = {}        # (slot, words) -> pinned host buffer + event of status_snapshot
        self.rank1_squarings = 30      # budget ceiling: resolves sigma_2/sigma_1 up to 1 - 1e-8
        # min(A, B) up to which cmtfpls_rank1_* runs all squarings in one launch (csrc/rank1.hip); 0 once the process switched it off
        self.rank1_chain_side = 256 if self.lib.cmtfpls_rank1_chain_enabled() else 0

    # -- helpers ---------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = buf
        return buf

    def clear_error(self) -> bool:
        """After a failed HIP-graph capture: reset the runtime's pending error so that the next launch is not blamed for it."""
        return bool(self.lib.cmtfpls_clear_error())

    def empty(self, *shape, dtype=torch.float64) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float64) -> torch.Tensor:
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def _fn(self, base: str, X: torch.Tensor):
        if X.dtype not in _SUFFIX:
            raise TypeError(f"X must be float32 or float64 on device, got {X.dtype}")
        assert X.is_contiguous() and X.device == self.device
        return getattr(self.lib, f"cmtfpls_{base}_{_SUFFIX[X.dtype]}")

    def _close_partials(self, part: torch.Tensor) -> torch.Tensor:
        out = self.empty(1)
        _lib.check(self.lib.cmtfpls_sum_f64(_ptr(part), part.numel(), _ptr(out), self._stream()), "sum")
        return out

    # -- measured HBM ceilings (bench.py: roofline denominators measured in the same run) ---------
    def ceiling(self, op: str, buf: torch.Tensor, row_bytes: int = 0, blocks: int = 2048,
                dst: Optional[torch.Tensor] = None, map: int = 0) -> bool:
        """op: "read" | "rmw" (negates in place; call an even number of times) | "copy" (needs dst); map: 0 flat,
        1 chunked, 2 row per 1024-thread workgroup with a barrier, 3 column owner.  False when the map does not take the shape."""
        nbytes = buf.numel() * buf.element_size()
        if op == "read":
            sink = self._workspace("ceiling_sink", 4 * self.lib.cmtfpls_ceiling_max_blocks())
            rc = self.lib.cmtfpls_ceiling_read(_ptr(buf), nbytes, int(row_bytes), int(map), _ptr(sink), int(blocks), self._stream())
        elif op == "rmw":
            rc = self.lib.cmtfpls_ceiling_rmw(_ptr(buf), nbytes, int(row_bytes), int(map), int(blocks), self._stream())
        elif op == "copy":
            assert dst is not None and dst.numel() * dst.element_size() >= nbytes
            rc = self.lib.cmtfpls_ceiling_copy(_ptr(buf), _ptr(dst), nbytes, int(row_bytes), int(map), int(blocks), self._stream())
        else:
            raise ValueError(op)
        if rc == 4:
            return False
        _lib.check(rc, "ceiling_" + op)
        return True

    # -- preprocess: tpls.py:61-71 -----------------------------------------------------------
    def colstats(self, X2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        I, P = X2.shape
        ws = self._workspace("contract", self.lib.cmtfpls_colstats_workspace_bytes(I, P))
        colsum, colcnt = self.empty(P), self.empty(P)
        _lib.check(self._fn("colstats", X2)(_ptr(X2), I, P, _ptr(colsum), _ptr(colcnt), _ptr(ws), ws.numel(), self._stream()), "colstats")
        return colsum, colcnt

    def center(self, X2: torch.Tensor, mean: torch.Tensor, want_rowcnt: bool) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        I, P = X2.shape
        rowcnt = self.empty(I) if want_rowcnt else None
        part = self.empty(self.n_partials)
        _lib.check(self._fn("center", X2)(_ptr(X2), I, P, _ptr(mean), _ptr(rowcnt), _ptr(part), self._stream()), "center")
        return rowcnt, self._close_partials(part)

    # -- K1: tpls.py:83 / missingvals.py:7-20 ------------------------------------------------
    def mode0_contract(self, X2: torch.Tensor, u: torch.Tensor, masked: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        I, P = X2.shape
        ws = self._workspace("contract", self.lib.cmtfpls_mode0_contract_workspace_bytes(I, P))
        Z = out if out is not None else self.empty(P)
        _lib.check(self._fn("mode0_contract", X2)(_ptr(X2), I, P, _ptr(u), _ptr(Z), int(masked), _ptr(ws), ws.numel(), self._stream()), "mode0_contract")
        return Z

    def mode0_contract_yq(self, X2: torch.Tensor, Y: torch.Tensor, q: torch.Tensor, masked: bool,
                          out: torch.Tensor) -> Optional[torch.Tensor]:
        """Z = X x_0 (Y q) with u = Y q (tpls.py:102) formed inside the kernel; None when the shape is outside
        the fused form (caller: rowdot + mode0_contract)."""
        I, P = X2.shape
        ws = self._workspace("contract", self.lib.cmtfpls_mode0_contract_workspace_bytes(I, P))
        rc = self._fn("mode0_contract_yq", X2)(_ptr(X2), I, P, _ptr(Y), Y.stride(0), Y.shape[1], _ptr(q), _ptr(out), int(masked),
                                               _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "mode0_contract_yq")
        return out

    def colscale(self, Z: torch.Tensor, colcnt: torch.Tensor, n_samples: float) -> None:
        _lib.check(self.lib.cmtfpls_colscale_f64(_ptr(Z), Z.numel(), _ptr(colcnt), float(n_samples), self._stream()), "colscale")

    # -- K2: tpls.py:84-90 -------------------------------------------------------------------
    def rank1(self, Z: torch.Tensor, A: int, B: int, wA: torch.Tensor, wB: torch.Tensor,
              info: Optional[torch.Tensor] = None, n_squarings: Optional[int] = None, launches: bool = False) -> None:
        """info (2 doubles, optional): [converged within the budget, squarings computed].  launches: the launch-per-squaring
        form (cmtfpls_rank1_launches_f64) instead of the one-launch chain the entry takes for min(A, B) <= 256 -- same bits."""
        ws = self._workspace("rank1", self.lib.cmtfpls_rank1_workspace_bytes(A, B))
        fn = self.lib.cmtfpls_rank1_launches_f64 if launches else self.lib.cmtfpls_rank1_f64
        _lib.check(fn(_ptr(Z), A, B, _ptr(wA), _ptr(wB), None, _ptr(info), int(n_squarings or self.rank1_squarings),
                      _ptr(ws), ws.numel(), self._stream()), "rank1")

    def rank1_chain_gave_up(self) -> None:
        """An extraction reported info = [0, -1]: a workgroup of the one-launch chain of squarings never became resident (the GPU
        is shared with another process).  From here on every entry of this process takes the launch-per-squaring form."""
        self.lib.cmtfpls_rank1_chain_enable(0)
        self.rank1_chain_side = 0

    def rank1_tensor(self, Z: torch.Tensor, dims, tol: float, factors: torch.Tensor,
                     info: Optional[torch.Tensor] = None, n_squarings: Optional[int] = None) -> None:
        """Rank-1 CP factors of an order-3/4 cross-covariance tensor; factors: (n, ld) f64, row m = mode m."""
        import ctypes
        arr = (ctypes.c_int * len(dims))(*[int(d) for d in dims])
        ws = self._workspace("rank1t", self.lib.cmtfpls_rank1_tensor_workspace_bytes(arr, len(dims)))
        _lib.check(self.lib.cmtfpls_rank1_tensor_f64(_ptr(Z), arr, len(dims), float(tol), _ptr(factors), factors.stride(0), _ptr(info),
                                                     int(n_squarings or self.rank1_squarings), _ptr(ws), ws.numel(), self._stream()),
                   "rank1_tensor")

    def kron(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        _lib.check(self.lib.cmtfpls_kron_f64(_ptr(a), a.numel(), _ptr(b), b.numel(), _ptr(out), self._stream()), "kron")
        return out

    def normalize(self, v: torch.Tensor) -> None:
        _lib.check(self.lib.cmtfpls_normalize_f64(_ptr(v), v.numel(), None, self._stream()), "normalize")

    # -- cross-covariance form (exact re-association of the loop, see include/cmtfpls.h) --------
    def xcov(self, X2: torch.Tensor, Y: torch.Tensor, masked: bool, out: Optional[torch.Tensor] = None,
             mixed: bool = False) -> Optional[torch.Tensor]:
        """S (M, P) = Y^T X_(0) on the matrix cores (f64 MFMA; ``mixed`` = the opt-in f32-MFMA form for
        f32-stored X).  More than 64 responses: tiles of 64, one pass over X each (inside the C entry)."""
        I, P = X2.shape
        M = Y.shape[1]
        ws = self._workspace("contract", self.lib.cmtfpls_xcov_workspace_bytes(I, P, M))
        S = out if out is not None else self.empty(M, P)
        fn = self.lib.cmtfpls_xcov_f32_mixed if (mixed and X2.dtype == torch.float32) else self._fn("xcov", X2)
        _lib.check(fn(_ptr(X2), I, P, _ptr(Y), Y.stride(0), M, _ptr(S), int(masked), _ptr(ws), ws.numel(), self._stream()), "xcov")
        return S

    def xcov_deflate(self, X2: torch.Tensor, A: int, B: int, Y: torch.Tensor, t: torch.Tensor, wA: torch.Tensor, wB: torch.Tensor,
                     out: torch.Tensor) -> Optional[torch.Tensor]:
        """X -= t (x) w in place AND S = Y^T X0 of the deflated block (NaN -> 0) AND |X0|^2, one read + write of X
        (cmtfpls_xcov_deflate_*).  Returns the norm (one-element device tensor); None when the shape is outside the kernel
        (nothing written: deflate, then xcov)."""
        I, P = X2.shape
        M = Y.shape[1]
        if M > 64 or P % 4 != 0 or P != A * B:
            return None
        ws = self._workspace("contract", self.lib.cmtfpls_xcov_ssq_workspace_bytes(I, P, M))
        ssq = self.empty(1)
        rc = self._fn("xcov_deflate", X2)(_ptr(X2), I, A, B, _ptr(Y), Y.stride(0), M, _ptr(t), _ptr(wA), _ptr(wB), _ptr(out), _ptr(ssq),
                                          _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "xcov_deflate")
        return ssq

    def rank1_score(self, Z: torch.Tensor, A: int, B: int, wA: torch.Tensor, wB: torch.Tensor, S: torch.Tensor, tq: torch.Tensor,
                    info: Optional[torch.Tensor] = None, n_squarings: Optional[int] = None) -> torch.Tensor:
        """rank1(Z) -> (wA, wB), then tq = S (wA (x) wB) for the M rows of S, the extraction's last kernel and the score in one
        launch where the shape allows (cmtfpls_rank1_score_f64)."""
        M = S.shape[0]
        assert S.is_contiguous() and S.shape[1] == A * B and tq.numel() >= M
        ws = self._workspace("rank1", self.lib.cmtfpls_rank1_workspace_bytes(A, B))
        _lib.check(self.lib.cmtfpls_rank1_score_f64(_ptr(Z), A, B, _ptr(wA), _ptr(wB), _ptr(info),
                                                    int(n_squarings if n_squarings is not None else self.rank1_squarings),
                                                    _ptr(S), M, _ptr(tq), _ptr(ws), ws.numel(), self._stream()), "rank1_score")
        return tq

    def xcov_ssq(self, X2: torch.Tensor, Y: torch.Tensor, mean: torch.Tensor, out: torch.Tensor):
        """S = Y^T X_(0) AND sum (X - mean)^2 from one read of an uncentred, NaN-free X (cmtfpls_xcov_ssq_*); returns
        (S, ssq as a one-element device tensor).  More than 64 responses: tiles of 64, the norm from the first pass."""
        I, P = X2.shape
        M = Y.shape[1]
        ws = self._workspace("contract", self.lib.cmtfpls_xcov_ssq_workspace_bytes(I, P, M))
        ssq = self.empty(1)
        assert mean.is_contiguous() and mean.numel() == P and mean.dtype == torch.float64
        _lib.check(self._fn("xcov_ssq", X2)(_ptr(X2), I, P, _ptr(Y), Y.stride(0), M, _ptr(out), _ptr(mean), _ptr(ssq), _ptr(ws), ws.numel(),
                                            self._stream()), "xcov_ssq")
        return out, ssq

    def xcov_stats(self, X2: torch.Tensor, Y: torch.Tensor, out: torch.Tensor):
        """S = Y^T X_(0) AND the column sums / sums of squares of an uncentred X from ONE read (cmtfpls_xcov_stats_*): returns
        (S, stats) with stats[:P] the sums and stats[P:] the sums of squares, or None for more than 64 responses."""
        I, P = X2.shape
        M = Y.shape[1]
        if M > 64 or not X2.is_contiguous():
            return None
        ws = self._workspace("contract", self.lib.cmtfpls_xcov_stats_workspace_bytes(I, P, M))
        stats = self.empty(2 * P)
        rc = self._fn("xcov_stats", X2)(_ptr(X2), I, P, _ptr(Y), Y.stride(0), M, _ptr(out), _ptr(stats), _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "xcov_stats")
        return out, stats

    def status_snapshot(self, status: torch.Tensor, slot: int, slots: Optional[dict] = None):
        """Enqueue a copy of a few status words to pinned host memory behind the work issued so far (cmtfpls_status_to_host);
        returns a token for status_wait.  (slot: the caller keeps at most one snapshot per slot in flight.)  `slots`: the
        CALLER's cache of pinned buffers and events -- a fit in flight owns its own (FitRun._pipe), so two fits on one
        backend (other threads, other streams) never share a buffer; None: the backend's own, for single-threaded tools."""
        slots = self._status_slots if slots is None else slots
        ent = slots.get((slot, status.numel()))
        if ent is None:
            host, ev = torch.empty(status.numel(), dtype=torch.float64, pin_memory=True), torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))               # (creates the underlying hipEvent_t)
            ent = slots[(slot, status.numel())] = (host, ev, host.numpy(), host.data_ptr(), ev.cuda_event, status.numel() * 8)
        rc = self.lib.cmtfpls_status_to_host(status.data_ptr(), ent[3], ent[5], ent[4], self._stream())
        if rc:
            _lib.check(rc, "status_to_host")
        return ent

    def status_wait(self, token) -> "np.ndarray":
        token[1].synchronize()
        return token[2]

    def xcov_iterate_plan(self, S: torch.Tensor, A: int, B: int, q_cur: torch.Tensor, Z: torch.Tensor, wA: torch.Tensor,
                          wB: torch.Tensor, status: torch.Tensor, q_new: torch.Tensor, G: torch.Tensor):
        """xcov_iterate on fixed buffers with its arguments marshalled ONCE: returns enqueue(n_squarings, first).  (The inner
        loop on S is ~15 launches of a few microseconds; per-call argument handling in Python was a tenth of an iteration.)"""
        M, P = S.shape
        wc = self._workspace("contract", self.lib.cmtfpls_mode0_contract_workspace_bytes(M, P))
        wr = self._workspace("rank1", self.lib.cmtfpls_rank1_workspace_bytes(A, B))
        keep = (S, q_cur, Z, wA, wB, status, q_new, G, wc, wr)                 # the plan owns references: pointers stay valid
        fn = self.lib.cmtfpls_xcov_iterate_f64
        head = (_ptr(S), M, A, B, _ptr(q_cur), _ptr(Z), _ptr(wA), _ptr(wB), status[1:3].data_ptr())
        tail = (_ptr(wc), wc.numel(), _ptr(wr), wr.numel())
        qn, Gp, du2 = _ptr(q_new), _ptr(G), status[0:1].data_ptr()
        stream = self._stream

        def enqueue(n_squarings: int, first: bool, _keep=keep) -> None:
            rc = fn(*head, n_squarings, qn, Gp, du2, 1 if first else 0, *tail, stream())
            if rc:
                _lib.check(rc, "xcov_iterate")
        return enqueue

    def xcov_iterate(self, S: torch.Tensor, A: int, B: int, q_cur: torch.Tensor, Z: torch.Tensor, wA: torch.Tensor,
                     wB: torch.Tensor, info: torch.Tensor, n_squarings: int, q_new: torch.Tensor, G: torch.Tensor,
                     du2: torch.Tensor, first: bool) -> None:
        """One inner iteration on S (contraction, rank-1, score, norm, |du|^2) issued by one host call."""
        M, P = S.shape
        wc = self._workspace("contract", self.lib.cmtfpls_mode0_contract_workspace_bytes(M, P))
        wr = self._workspace("rank1", self.lib.cmtfpls_rank1_workspace_bytes(A, B))
        _lib.check(self.lib.cmtfpls_xcov_iterate_f64(_ptr(S), M, A, B, _ptr(q_cur), _ptr(Z), _ptr(wA), _ptr(wB), _ptr(info),
                                                     int(n_squarings), _ptr(q_new), _ptr(G), _ptr(du2), int(first),
                                                     _ptr(wc), wc.numel(), _ptr(wr), wr.numel(), self._stream()), "xcov_iterate")

    def xcov_blocks_plan(self, blocks, M: int, q_cur: torch.Tensor, tq: torch.Tensor, q_new: torch.Tensor, G: torch.Tensor,
                         status: torch.Tensor):
        """One inner iteration on S for SEVERAL coupled blocks / blocks with missing values (cmtfpls_xcov_iterate_blocks_f64) on fixed
        buffers, arguments marshalled once.  blocks: dicts with S, S2 (or None), colcnt (or None), n_samples, order (2 | 3), A, B,
        Z, wA, wB; status = [du2, (converged, squarings used) per block].  Returns enqueue(n_squarings per block, first), or
        None when a block is outside the entry (order > 3, M > 64)."""
        nb = len(blocks)
        if M > 64 or any(b["order"] not in (2, 3) for b in blocks) or tq.numel() != nb * M or not tq.is_contiguous():
            return None
        arr = (_lib.XcovBlock * nb)()
        need = 256
        for i, b in enumerate(blocks):
            for name in ("S", "S2", "colcnt", "Z", "wA", "wB"):
                t = b.get(name)
                assert t is None or (t.is_contiguous() and t.dtype == torch.float64 and t.device == self.device), name
                setattr(arr[i], name, _ptr(t))
            arr[i].n_samples, arr[i].order, arr[i].A, arr[i].B = float(b["n_samples"]), int(b["order"]), int(b["A"]), int(b["B"])
            arr[i].info = status[1 + 2 * i: 3 + 2 * i].data_ptr()
            if b["order"] == 3:
                need = max(need, self.lib.cmtfpls_rank1_workspace_bytes(b["A"], b["B"]))
        wr = self._workspace("rank1", need)
        keep = (blocks, q_cur, tq, q_new, G, status, wr, arr)                    # the plan owns references: pointers stay valid
        fn, stream = self.lib.cmtfpls_xcov_iterate_blocks_f64, self._stream
        args = (nb, M, _ptr(q_cur), _ptr(tq), _ptr(q_new), _ptr(G), status[0:1].data_ptr())
        tail = (_ptr(wr), wr.numel())

        def enqueue(n_squarings, first: bool, _keep=keep) -> None:
            for i in range(nb):
                arr[i].n_squarings = int(n_squarings[i])
            rc = fn(arr, *args, 1 if first else 0, *tail, stream())
            if rc:
                _lib.check(rc, "xcov_iterate_blocks")
        return enqueue

    def s_downdate(self, S: torch.Tensor, A: int, B: int, ya: torch.Tensor, wA: torch.Tensor, wB: torch.Tensor,
                   q: torch.Tensor, v: torch.Tensor) -> None:
        """S -= ya w^T + q v^T, w = kron(wA, wB): S = Y^T X_(0) carried across one deflation."""
        assert S.is_contiguous() and S.shape[1] == A * B
        _lib.check(self.lib.cmtfpls_s_downdate_f64(_ptr(S), S.shape[0], A, B, _ptr(ya), _ptr(wA), _ptr(wB), _ptr(q), _ptr(v),
                                                   self._stream()), "s_downdate")

    def kr_axpy(self, v: torch.Tensor, A: int, B: int, WA: torch.Tensor, WB: torch.Tensor, k: int, coef: torch.Tensor) -> torch.Tensor:
        """v (A*B) -= sum_{j<k} coef[j] * kron(WA[:, j], WB[:, j]); WA (A, R), WB (B, R) row-major."""
        assert WA.is_contiguous() and WB.is_contiguous() and WA.shape[1] == WB.shape[1] and coef.numel() >= k
        _lib.check(self.lib.cmtfpls_kr_axpy_f64(_ptr(v), A, B, _ptr(WA), _ptr(WB), WA.shape[1], int(k), _ptr(coef), self._stream()), "kr_axpy")
        return v

    def axpy_scalar(self, y: torch.Tensor, a: torch.Tensor, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y -= a[0] * x (x = None: ones); a is a one-element device tensor."""
        assert y.is_contiguous() and y.dtype == torch.float64 and a.numel() >= 1 and (x is None or (x.is_contiguous() and x.numel() == y.numel()))
        _lib.check(self.lib.cmtfpls_axpy_scalar_f64(_ptr(y), y.numel(), _ptr(a), _ptr(x), self._stream()), "axpy_scalar")
        return y

    def total(self, v: torch.Tensor) -> torch.Tensor:
        """Sum of a contiguous f64 vector as a one-element device tensor (fixed-order, one workgroup)."""
        assert v.is_contiguous() and v.dtype == torch.float64
        return self._close_partials(v)

    def quadform(self, G: torch.Tensor, q: torch.Tensor, q_old: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        _lib.check(self.lib.cmtfpls_quadform_f64(_ptr(G), G.shape[0], _ptr(q), _ptr(q_old), _ptr(out), self._stream()), "quadform")
        return out

    def mttkrp(self, X2: torch.Tensor, A: int, B: int, WA: torch.Tensor, WB: torch.Tensor, out: torch.Tensor,
               mixed: bool = False) -> Optional[torch.Tensor]:
        """out (I, R) = X_(0) (WA (.) WB); None when R > 32 or the loadings do not fit LDS."""
        R = WA.shape[1]
        assert WA.is_contiguous() and WB.is_contiguous() and out.stride(1) == 1
        fn = self.lib.cmtfpls_mttkrp_f32_mixed if (mixed and X2.dtype == torch.float32) else self._fn("mttkrp", X2)
        rc = fn(_ptr(X2), X2.shape[0], A, B, _ptr(WA), _ptr(WB), R, _ptr(out), out.stride(0), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "mttkrp")
        return out

    # -- K3: tpls.py:92-99 / missingvals.py:23-38 --------------------------------------------
    def score(self, X2, A, B, wA, wB, rowcnt, out) -> torch.Tensor:
        _lib.check(self._fn("score", X2)(_ptr(X2), X2.shape[0], A, B, _ptr(wA), _ptr(wB), _ptr(rowcnt), _ptr(out), self._stream()), "score")
        return out

    def score_s(self, S, A, B, wA, wB, out) -> torch.Tensor:
        """out[m] = S[m, :] . kron(wA, wB) for the M rows of a cross-covariance S (or a one-row "tensor" such as the column
        means): cmtfpls_score_s_f64 -- few long rows take one workgroup each, which cmtfpls_score_* never does for samples."""
        assert S.dtype == torch.float64 and S.is_contiguous()
        _lib.check(self.lib.cmtfpls_score_s_f64(_ptr(S), S.shape[0], A, B, _ptr(wA), _ptr(wB), _ptr(out), self._stream()), "score_s")
        return out

    def score_contract(self, X2, A, B, wA, wB, shift, t, Z, sub_own=None, add_other=None, alpha: float = 1.0,
                       csum: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """t = X w - shift - sub_own and Z = X^T c, c = alpha (t + add_other), in one read of X (cmtfpls_score_contract_*); csum[0] =
        sum(c) when given; None when the row fits neither the registers of one workgroup nor those of 16 (the caller then makes the two passes)."""
        I, P = X2.shape
        if not X2.is_contiguous() or P != A * B:
            return None
        for v in (sub_own, add_other):
            assert v is None or (v.is_contiguous() and v.numel() == I and v.dtype == torch.float64)
        nbytes = self.lib.cmtfpls_score_contract_workspace_bytes(I, P)
        ws = self._workspace("score_contract", max(nbytes, 256))
        rc = self._fn("score_contract", X2)(_ptr(X2), I, A, B, _ptr(wA), _ptr(wB), _ptr(shift), _ptr(sub_own), _ptr(add_other), float(alpha),
                                            _ptr(t), _ptr(Z), _ptr(csum), _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "score_contract")
        return Z

    def score_gram(self, X2, A, B, wA, wB, rowcnt, out, Y: torch.Tensor, qpart: torch.Tensor) -> Optional[torch.Tensor]:
        """score + the per-workgroup partial sums of Y^T t into qpart (n_partials x M); None when M > 64."""
        assert qpart.numel() >= self.n_partials * Y.shape[1] and qpart.is_contiguous()
        rc = self._fn("score_gram", X2)(_ptr(X2), X2.shape[0], A, B, _ptr(wA), _ptr(wB), _ptr(rowcnt), _ptr(out),
                                        _ptr(Y), Y.stride(0), Y.shape[1], _ptr(qpart), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "score_gram")
        return out

    def q_update(self, q: torch.Tensor, qpart: Optional[torch.Tensor] = None, normalize: bool = True,
                 G: Optional[torch.Tensor] = None, q_prev: Optional[torch.Tensor] = None,
                 du2: Optional[torch.Tensor] = None, nparts: Optional[int] = None) -> None:
        """The Y-side update in one launch: q = sum of qpart rows (optional), q /= |q| (optional),
        du2 = (q - q_prev)^T G (q - q_prev) (optional).  tpls.py:100-103."""
        M = q.numel()
        nrows = (nparts or self.n_partials) if qpart is not None else 0
        assert qpart is None or (qpart.is_contiguous() and qpart.numel() >= nrows * M)
        _lib.check(self.lib.cmtfpls_q_update_f64(_ptr(qpart), nrows, M, _ptr(q), int(normalize),
                                                 _ptr(G), _ptr(q_prev), _ptr(du2), self._stream()), "q_update")

    # -- K6: tpls.py:109 ----------------------------------------------------------------------
    def deflate(self, X2, A, B, t, wA, wB) -> torch.Tensor:
        part = self.empty(self.n_partials)
        _lib.check(self._fn("deflate", X2)(_ptr(X2), X2.shape[0], A, B, _ptr(t), _ptr(wA), _ptr(wB), _ptr(part), self._stream()), "deflate")
        return self._close_partials(part)

    def deflate_contract_yq(self, X2, A, B, t, wA, wB, Y: torch.Tensor, q: torch.Tensor, masked: bool,
                            out: torch.Tensor) -> Optional[torch.Tensor]:
        """K6 of one component fused with K1 of the next: X -= t (x) w in place, out = X_new x_0 (Y q); returns
        the sum of squares of X_new (1 double) or None when the shape is outside the fused form."""
        I, P = X2.shape
        ws = self._workspace("deflate_contract", self.lib.cmtfpls_deflate_contract_workspace_bytes(I, P))
        ssq = self.empty(1)
        rc = self._fn("deflate_contract_yq", X2)(_ptr(X2), I, A, B, _ptr(t), _ptr(wA), _ptr(wB), _ptr(Y), Y.stride(0), Y.shape[1],
                                                 _ptr(q), _ptr(out), int(masked), _ptr(ssq), _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "deflate_contract_yq")
        return ssq

    def score_deflate(self, X2, A, B, wA, wB, rowcnt, out) -> Optional[torch.Tensor]:
        """Fused K3+K6; returns None when the row does not fit (caller then uses score + deflate)."""
        part = self.empty(self.n_partials)
        rc = self._fn("score_deflate", X2)(_ptr(X2), X2.shape[0], A, B, _ptr(wA), _ptr(wB), _ptr(rowcnt), _ptr(out), _ptr(part), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "score_deflate")
        return self._close_partials(part)

    # -- K4/K5/K7/K11 small algebra ----------------------------------------------------------
    def gram_tn(self, A: torch.Tensor, B: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """C = A^T B for 2-D row-major views (unit inner stride) sharing the leading dimension."""
        if A.dim() == 1:
            A = A.unsqueeze(1)
        if B.dim() == 1:
            B = B.unsqueeze(1)
        assert A.stride(1) == 1 and B.stride(1) == 1 and A.shape[0] == B.shape[0]
        a, b = A.shape[1], B.shape[1]
        ws = self._workspace("small", self.lib.cmtfpls_small_workspace_bytes())
        C = out if out is not None else self.empty(a, b)
        assert C.numel() == a * b and C.is_contiguous()
        _lib.check(self.lib.cmtfpls_gram_tn_f64(_ptr(A), A.stride(0), a, _ptr(B), B.stride(0), b, A.shape[0], _ptr(C),
                                                _ptr(ws), ws.numel(), self._stream()), "gram_tn")
        return C

    def rowdot(self, Y: torch.Tensor, q: torch.Tensor, u_out: torch.Tensor, u_old: Optional[torch.Tensor],
               du2: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        ws = self._workspace("small", self.lib.cmtfpls_small_workspace_bytes())
        if u_old is not None and du2 is None:
            du2 = self.empty(1)
        _lib.check(self.lib.cmtfpls_rowdot_f64(_ptr(Y), Y.stride(0), Y.shape[1], Y.shape[0], _ptr(q), _ptr(u_out), _ptr(u_old),
                                               _ptr(du2) if u_old is not None else None,
                                               _ptr(ws), ws.numel(), self._stream()), "rowdot")
        return du2 if u_old is not None else None

    # -- component epilogue / projection fix-up / reconstruction without a host round trip --------
    def normal_solve(self, G: torch.Tensor, g: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """b = argmin |T b - u| from G = T^T T (k x k), g = T^T u: lstsq(T, u, rcond=-1) of tpls.py:110-112."""
        k = G.shape[0]
        assert G.is_contiguous() and G.shape == (k, k) and g.numel() == k
        b = out if out is not None else self.empty(k)
        if k <= 64:
            _lib.check(self.lib.cmtfpls_normal_solve_f64(_ptr(G), _ptr(g), k, _ptr(b), 1, self._stream()), "normal_solve")
        else:                                                   # more than 64 components: the matrix in a workspace
            ws = self._workspace("normal_solve", self.lib.cmtfpls_normal_solve_workspace_bytes(k))
            _lib.check(self.lib.cmtfpls_normal_solve_ws_f64(_ptr(G), _ptr(g), k, _ptr(b), 1, _ptr(ws), ws.numel(), self._stream()),
                       "normal_solve")
        return b

    def unit_upper_solve_rows(self, M: torch.Tensor, U: torch.Tensor, shift: Optional[torch.Tensor] = None,
                              nan_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rows of M (I x R) overwritten by the rows of T solving T (I + triu(U, 1)) = M - 1 shift^T; nan_flag (one int32,
        zeroed by the caller) is set when M holds a NaN."""
        I, R = M.shape
        assert M.stride(1) == 1 and U.is_contiguous() and U.shape == (R, R)
        assert shift is None or (shift.is_contiguous() and shift.numel() == R and shift.dtype == torch.float64)
        assert nan_flag is None or nan_flag.dtype == torch.int32
        _lib.check(self.lib.cmtfpls_unit_upper_solve_rows_f64(_ptr(M), I, M.stride(0), R, _ptr(U), _ptr(shift), _ptr(nan_flag),
                                                              self._stream()), "unit_upper_solve_rows")
        return M

    def kr_gram_row(self, L: torch.Tensor, a: int, g: torch.Tensor, first: bool) -> torch.Tensor:
        """g[:a] (*)= L[:, :a]^T L[:, a]: row a of the loadings' Gram matrix (cmtfpls_kr_gram_row_f64)."""
        assert L.is_contiguous() and g.is_contiguous() and g.numel() >= a
        _lib.check(self.lib.cmtfpls_kr_gram_row_f64(_ptr(L), L.shape[0], L.shape[1], int(a), _ptr(g), int(first), self._stream()), "kr_gram_row")
        return g

    def kr_gram(self, L: torch.Tensor, G: torch.Tensor, first: bool) -> torch.Tensor:
        """G = L^T L (first) or G .*= L^T L: Gram of a Khatri-Rao product, one mode at a time."""
        n, R = L.shape
        assert L.is_contiguous() and G.is_contiguous() and G.numel() == R * R
        _lib.check(self.lib.cmtfpls_kr_gram_f64(_ptr(L), n, R, _ptr(G), int(first), 1.0, self._stream()), "kr_gram")
        return G

    def khatri_rao(self, Am: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
        na, R = Am.shape
        nb = Bm.shape[0]
        assert Am.is_contiguous() and Bm.is_contiguous() and Bm.shape[1] == R
        out = self.empty(na * nb, R)
        _lib.check(self.lib.cmtfpls_khatri_rao_f64(_ptr(Am), na, _ptr(Bm), nb, R, _ptr(out), self._stream()), "khatri_rao")
        return out

    def project_rows(self, X2: torch.Tensor, A: int, B: int, WA: torch.Tensor, WB: torch.Tensor, mean: Optional[torch.Tensor],
                     out: torch.Tensor, rows: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """out (I, R) = the R sequential masked project-and-deflate steps of every row of the UNCENTRED X2 -- or of the rows
        listed in `rows` (int64 device tensor) only, the others left as they are -- the row kept in registers (one read,
        nothing written); None when the shape is outside that form."""
        I, R = X2.shape[0], WA.shape[1]
        assert WA.is_contiguous() and WB.is_contiguous() and out.stride(1) == 1 and out.shape == (I, R)
        if rows is not None:
            assert rows.dtype == torch.int64 and rows.is_contiguous() and rows.device == self.device
            rc = self._fn("project_rows_idx", X2)(_ptr(X2), _ptr(rows), rows.numel(), A, B, R, _ptr(WA), _ptr(WB), _ptr(mean), _ptr(out),
                                                  out.stride(0), self._stream())
        else:
            rc = self._fn("project_rows", X2)(_ptr(X2), I, A, B, R, _ptr(WA), _ptr(WB), _ptr(mean), _ptr(out), out.stride(0), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "project_rows")
        return out

    def project_rows2(self, X2s, As, Bs, WAs, WBs, means, out: torch.Tensor, rows: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """project_rows for TWO COUPLED blocks (lists of two): the sample's row of both blocks in one workgroup, the score
        of every step the mean of the two masked block scores (cmtf.py:155,206); `rows` as in project_rows; None outside
        that form."""
        I, R = X2s[0].shape[0], WAs[0].shape[1]
        if X2s[0].dtype != X2s[1].dtype or X2s[1].shape[0] != I:
            return None
        assert all(w.is_contiguous() for w in list(WAs) + list(WBs)) and out.stride(1) == 1 and out.shape == (I, R)
        head = (_ptr(X2s[0]), As[0], Bs[0], _ptr(WAs[0]), _ptr(WBs[0]), _ptr(means[0]),
                _ptr(X2s[1]), As[1], Bs[1], _ptr(WAs[1]), _ptr(WBs[1]), _ptr(means[1]))
        if rows is not None:
            assert rows.dtype == torch.int64 and rows.is_contiguous() and rows.device == self.device
            rc = self._fn("project_rows2_idx", X2s[0])(*head, _ptr(rows), rows.numel(), R, _ptr(out), out.stride(0), self._stream())
        else:
            rc = self._fn("project_rows2", X2s[0])(*head, I, R, _ptr(out), out.stride(0), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "project_rows2")
        return out

    def predict_rows(self, S: torch.Tensor, Bm: torch.Tensor, mean: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """out (I, M) = S (I, R) @ Bm (R, M) + mean: the last line of predict (tpls.py:143) on the device-resident scores."""
        I, R = S.shape
        M = Bm.shape[1]
        assert S.stride(1) == 1 and Bm.is_contiguous() and Bm.shape[0] == R
        out = self.empty(I, M)
        rc = self.lib.cmtfpls_predict_rows_f64(_ptr(S), I, S.stride(0), R, _ptr(Bm), M, _ptr(mean), _ptr(out), M, self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "predict_rows")
        return out

    def recon(self, T: torch.Tensor, WA: torch.Tensor, WB: torch.Tensor, mean: Optional[torch.Tensor], out: torch.Tensor) -> Optional[torch.Tensor]:
        """out (I, A*B) = T (WA (.) WB)^T + mean in out's dtype (factors_to_tensor util.py:18-20 + X_mean);
        None when the shape is outside the vector form (caller falls back to the host einsum)."""
        I, R = T.shape
        A, B = WA.shape[0], WB.shape[0]
        assert T.stride(1) == 1 and WA.is_contiguous() and WB.is_contiguous() and out.is_contiguous() and out.numel() == I * A * B
        rc = self._fn("recon", out)(_ptr(T), I, T.stride(0), R, _ptr(WA), _ptr(WB), A, B, _ptr(mean), _ptr(out), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "recon")
        return out

    def recon_r2(self, X2: torch.Tensor, T: torch.Tensor, WA: torch.Tensor, WB: torch.Tensor, mean: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """[sum (xhat - x)^2, sum x^2] over the finite entries of x = X2 - mean: calcR2X's two sums (util.py:7-15) against
        the never-materialised reconstruction; None when the shape has no device form."""
        I, P = X2.shape
        R = T.shape[1]
        A, B = WA.shape[0], WB.shape[0]
        assert X2.is_contiguous() and T.stride(1) == 1 and WA.is_contiguous() and WB.is_contiguous() and P == A * B
        ws = self._workspace("recon_r2", self.lib.cmtfpls_recon_r2_workspace_bytes(I, P))
        out = self.empty(2)
        rc = self._fn("recon_r2", X2)(_ptr(X2), _ptr(T), I, T.stride(0), R, _ptr(WA), _ptr(WB), A, B, _ptr(mean), _ptr(out),
                                     _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "recon_r2")
        return out

    def loo_tpls(self, X2: torch.Tensor, Y: torch.Tensor, A: int, B: int, R: int, tol: float, max_iter: int,
                 max_ws_bytes: Optional[int] = None, forms=("lds", "xcov")) -> Optional[Tuple[torch.Tensor, torch.Tensor, str]]:
        """Leave-one-out predictions of a tPLS model (validate.py:24-33), every fold a workgroup: returns
        (Ypred (I, M), n_iter (I, R), form) or None when the shape is outside both workgroup-per-fold forms.  form:
        "lds" = cmtfpls_loo_tpls_f64 (the fold's vectors in LDS: min(A, B) <= 64), "xcov" = cmtfpls_loo_xcov_f64 (the fold's loop on
        its cross-covariance, Gram squarings on the matrix cores: min(A, B) <= 256).  Folds run in chunks that fit `max_ws_bytes`
        (default: a third of the free HBM) -- the xcov form keeps a centred copy of X per resident fold."""
        I, P = X2.shape
        M = Y.shape[1]
        assert X2.dtype == torch.float64 and Y.dtype == torch.float64 and X2.is_contiguous() and Y.is_contiguous() and P == A * B
        colsum_x, _ = self.colstats(X2)
        colsum_y, _ = self.colstats(Y)
        Ypred = self.empty(I, M)
        n_iter = torch.zeros(I, R, dtype=torch.int32, device=self.device)
        for form, per_fn, fn in (("lds", self.lib.cmtfpls_loo_fold_workspace_bytes, self.lib.cmtfpls_loo_tpls_f64),
                                 ("xcov", self.lib.cmtfpls_loo_xcov_fold_workspace_bytes, self.lib.cmtfpls_loo_xcov_f64)):
            if form not in forms:
                continue
            # probe without a workspace: the shape check comes first (status 4 = this form declines, 2 = it only misses the workspace)
            if fn(_ptr(X2), _ptr(Y), _ptr(colsum_x), _ptr(colsum_y), I, A, B, M, R, float(tol), int(max_iter), 0, 1, _ptr(Ypred),
                  _ptr(n_iter), None, 0, self._stream()) == 4:
                continue
            per = per_fn(I, A, B, M, R)
            budget = max_ws_bytes
            if budget is None:
                budget = (4 << 30) if form == "lds" else max(4 << 30, torch.cuda.mem_get_info(self.device)[0] // 3)
            chunk = max(1, min(I, int(budget // max(per, 1))))
            if form == "xcov":
                chunk = min(chunk, 512)                      # two workgroups' worth of folds per CU is all a launch can overlap
            declined = False
            ws = None
            for f0 in range(0, I, chunk):
                nf = min(chunk, I - f0)
                if ws is None:
                    ws = self._workspace("loo", per * min(chunk, I))
                rc = fn(_ptr(X2), _ptr(Y), _ptr(colsum_x), _ptr(colsum_y), I, A, B, M, R, float(tol), int(max_iter),
                        f0, nf, _ptr(Ypred), _ptr(n_iter), _ptr(ws), ws.numel(), self._stream())
                if rc == 4:
                    declined = True
                    break
                _lib.check(rc, "loo_" + form)
            if not declined:
                return Ypred, n_iter, form
        return None

    def fit_small(self, X2: torch.Tensor, Y: torch.Tensor, A: int, B: int, R: int, tol: float, max_iter: int):
        """The complete tPLS.fit of a small float64 problem without missing values in ONE launch (cmtfpls_fit_small_f64):
        returns a dict of device tensors (T, U, WA, WB, Q, x_mean, y_mean) and host arrays (coef, ssq, n_iter), or None
        when the shape is outside the one-workgroup form or the input holds a non-finite value (nothing usable written)."""
        I, P = X2.shape
        M = Y.shape[1]
        if X2.dtype != torch.float64 or Y.dtype != torch.float64 or not (X2.is_contiguous() and Y.is_contiguous()) or I >= 2 ** 31 or P != A * B:
            return None
        nbytes = self.lib.cmtfpls_fit_small_workspace_bytes(I, A, B, M)
        T, U = self.empty(I, R), self.empty(I, R)
        WA, WB, Q = self.empty(A, R), self.empty(B, R), self.empty(M, R)
        small = self.empty(R * R + 2 * (R + 1))                      # coef | ssq: one device -> host copy
        xm, ym = self.empty(P), self.empty(M)
        ints = torch.zeros(R + 1, dtype=torch.int32, device=self.device)    # n_iter | flag
        ws = self._workspace("fit_small", max(nbytes, 256))
        rc = self.lib.cmtfpls_fit_small_f64(_ptr(X2), _ptr(Y), I, A, B, M, R, float(tol), int(max_iter), _ptr(T), _ptr(U), _ptr(WA), _ptr(WB),
                                            _ptr(Q), _ptr(small), small[R * R:].data_ptr(), _ptr(xm), _ptr(ym), _ptr(ints),
                                            ints[R:].data_ptr(), _ptr(ws), ws.numel(), self._stream())
        if rc == 4:
            return None
        _lib.check(rc, "fit_small")
        ih = ints.cpu().numpy()
        if ih[R] != 0:
            return None                                              # a NaN / inf somewhere: the regular (masked) engine takes it
        sh = small.cpu().numpy()
        return {"T": T, "U": U, "WA": WA, "WB": WB, "Q": Q, "x_mean": xm, "y_mean": ym, "coef": sh[:R * R].reshape(R, R).copy(),
                "ssq": sh[R * R:].reshape(R + 1, 2).copy(), "n_iter": [int(v) for v in ih[:R]]}

    def add_noise(self, X: torch.Tensor, sigma: float, seed: int, offset: int = 0, nan_fraction: float = 0.0) -> torch.Tensor:
        """X += sigma * N(0,1) in place from the counter-based generator (element e of X is global element offset + e
        of the stream keyed by seed), then an i.i.d. NaN mask of density nan_fraction (synthetic.py:71,74)."""
        assert X.is_contiguous() and X.device == self.device
        _lib.check(self._fn("add_noise", X)(_ptr(X), X.numel(), float(sigma), int(seed) & (2 ** 64 - 1), int(offset), float(nan_fraction),
                                            self._stream()), "add_noise")
        return X

    def scores_mean(self, Ts: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        nb, I = Ts.shape
        _lib.check(self.lib.cmtfpls_scores_mean_f64(_ptr(Ts), nb, I, _ptr(out), self._stream()), "scores_mean")
        return out

    def y_deflate(self, Y: torch.Tensor, T: torch.Tensor, ncols: int, b: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
        ws = self._workspace("small", self.lib.cmtfpls_small_workspace_bytes())
        ssq = self.empty(1)
        _lib.check(self.lib.cmtfpls_y_deflate_f64(_ptr(Y), Y.stride(0), Y.shape[1], Y.shape[0], _ptr(T), T.stride(0), ncols, _ptr(b), _ptr(q),
                                                  _ptr(ssq), _ptr(ws), ws.numel(), self._stream()), "y_deflate")
        return ssq
