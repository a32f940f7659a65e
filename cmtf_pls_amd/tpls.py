"""``tPLS``: N-way partial least squares with the reference estimator's surface
(cmtf_pls/tpls.py:15-189 in meyer-lab/cmtf-pls), fitted by the MI355X NIPALS engine.

Same constructor, ``fit / predict / transform / X_reconstructed / copy``, Mapping protocol
(``[0], [1], [2]`` -> X_factors, Y_factors, coef_) and fitted attributes; NumPy in, NumPy out.
Opt-in extras (defaults reproduce the reference): ``dtype`` (storage type of X on the GPU:
"float32" | "float64" | None = follow the input), ``algorithm`` ("direct" = the reference's loop,
"xcov" = the same iteration through the cross-covariance S = X_(0)^T Y, one X read per component),
``graphs`` (replay each iteration as a HIP graph), ``matrix_precision`` ("f64" | "f32": the opt-in
f32-MFMA form of the two matrix-core kernels), ``copy_X`` (False: fit a device-resident X in place),
``device``, ``comm`` (sample-mode sharding: each
rank passes its own rows), ``options`` (an ``EngineOptions``: which exact form of each step to take where the shape
allows it), ``n_iter_`` (inner iterations executed per component), ``fit_report_`` / ``projection_report_`` (which
forms actually ran: algorithm after fallbacks, passes over X per component, centred copy or the caller's tensor,
pipelined, graph replay, projection form; every declined fast form in words) and
``original_X / original_Y`` (which the reference's validate.get_q2y reads, validate.py:18-21).
"""
from __future__ import annotations

from collections.abc import Mapping
from copy import copy
from typing import Optional

import numpy as np
import torch

from .engine import Comm, EngineOptions, NipalsEngine


def _as_torch_dtype(dtype, like) -> torch.dtype:
    """Storage type of X on the GPU: explicit, or float32 only when the input already is float32."""
    if dtype is None:
        return torch.float32 if like.dtype in (np.float32, torch.float32) else torch.float64
    if isinstance(dtype, torch.dtype):
        return dtype
    name = dtype if isinstance(dtype, str) else np.dtype(dtype).name
    return {"float32": torch.float32, "f32": torch.float32, "float64": torch.float64, "f64": torch.float64}[name]


def to_device_copy(X, dtype: torch.dtype, device, copy: bool = True) -> torch.Tensor:
    """A fresh contiguous device tensor of X (inputs are never modified, tpls.py:74,128,151).
    copy=False (opt-in ``copy_X=False``): a device tensor that already has the right type is used in
    place and is centred / deflated by the fit.

    A large HOST array of another type than the storage type (the common case: float64 NumPy input, float32
    storage) is shipped in row blocks in ITS OWN type and cast on the device: casting 8.6 GB on the host first costs
    ~0.75 s at 65536 x 128 x 128, the extra PCIe bytes ~0.08 s (profiles/r02aa_pcie_inclusive.txt)."""
    if isinstance(X, torch.Tensor):
        if not copy and X.is_contiguous() and X.dtype == dtype and X.device == torch.device(device):
            return X
        out = X.to(device=device, dtype=dtype, copy=True)
        return out.contiguous()
    Xn = np.ascontiguousarray(X)
    src = torch.from_numpy(Xn)
    if src.dtype == dtype or Xn.ndim == 0 or Xn.nbytes < (64 << 20):
        return src.to(device=device, dtype=dtype, copy=True).contiguous()
    out = torch.empty(Xn.shape, dtype=dtype, device=device)
    rows = max(1, (256 << 20) // max(Xn[0].nbytes, 1))           # ~256 MB of the source type per block
    for r in range(0, Xn.shape[0], rows):
        out[r:r + rows].copy_(src[r:r + rows].to(device))        # H2D in the source type, cast by the device copy
    return out


def _project_blocks(eng: NipalsEngine, state, Xs, dtypes, mixed: bool) -> torch.Tensor:
    """Scores of new samples (tpls.py:128-142; cmtf.py:143-177).  First the read-only one-pass form on the blocks as
    they are (a device tensor of the storage type is not even copied; a host array is uploaded, not centred); where
    that does not apply (missing values, f32 matrix precision, shapes outside the MTTKRP) the sequential
    project-and-deflate on private copies, which it centres and deflates in place."""
    dev = eng.be.device
    Xd = [to_device_copy(X, dt, dev, copy=False) for X, dt in zip(Xs, dtypes)]
    if not mixed:
        scores = eng.project_readonly(state, Xd)
        if scores is not None:
            return scores
    Xd = [xd.clone() if xd is X else xd for xd, X in zip(Xd, Xs)]        # never touch the caller's tensor
    return eng.project(state, Xd, mixed=mixed)


class _EstimatorBase(Mapping):
    def __init__(self, n_components: int, dtype=None, device=None, comm: Optional[Comm] = None, backend=None,
                 algorithm: str = "direct", graphs: bool = False, matrix_precision: str = "f64", copy_X: bool = True,
                 options: Optional[EngineOptions] = None):
        super().__init__()
        self.n_components = n_components
        self._options = options                   # None: the process default (engine.default_options())
        self._algorithm = algorithm
        self._graphs = graphs                     # replay each iteration's launch sequence as a HIP graph
        if matrix_precision not in ("f64", "f32"):
            raise ValueError("matrix_precision must be 'f64' or 'f32'")
        self._mixed = matrix_precision == "f32"   # f32-MFMA S build / MTTKRP for f32-stored X (opt-in)
        self._copy_X = copy_X                     # False: a device-resident X is fitted in place (and destroyed)
        self._dtype = dtype
        self._device = device
        self._comm = comm
        self._backend = backend
        self._engine = None

    def _get_engine(self) -> NipalsEngine:
        if self._engine is None:
            if self._backend is None:
                from .backend import HipBackend   # raises if the GPU or libcmtfpls.so is missing

                self._backend = HipBackend(self._device)
            self._engine = NipalsEngine(self._backend, self._comm, self._options)
        return self._engine

    def __iter__(self):
        yield self[0]
        yield self[1]
        yield self[2]

    def __len__(self):
        return 3

    @property
    def fit_report_(self) -> dict:
        """Which form of every step the last fit took (fitrun.FitRun.build_report)."""
        return dict(self._state.report)

    @property
    def projection_report_(self) -> dict:
        """Which form the last transform / predict took: one-pass MTTKRP, rows with missing values in registers, or the
        sequential passes -- and why."""
        return dict(self._get_engine().last_projection)

    def copy(self):
        return copy(self)

    def _reconstruct(self, block: int, factors, mean, rows=None, device: bool = False):
        """factors_to_tensor(factors) + mean (tpls.py:188-189, cmtf.py:233-237) through the GPU kernel; `rows` (a
        slice) restricts it to a row block and `device=True` returns the device tensor IN THE STORAGE TYPE (f32 storage:
        entries rounded to f32, ~6e-8 relative) instead of a float64 NumPy array holding the exact f64 reconstruction --
        at 65536 x 128 x 128 the reference's host form is an 8.6 GB float64 einsum."""
        eng = self._get_engine()
        if device:
            rec = eng.reconstruct(self._state, block, rows)
            if rec is not None:
                return rec
        else:
            # the host array is float64 like the reference's (tpls.py:188-189): the kernel writes float64 row blocks
            # (<= 256 MB each on the device) straight from the f64 factors -- no rounding to the storage type on the way
            n = self._state.T.shape[0]
            lo, hi, step = (rows or slice(None)).indices(n)
            if step == 1:
                trailing = tuple(self._state.blocks[block].shape[1:])
                per_row = int(np.prod(trailing)) * 8
                chunk = max(1, (256 << 20) // max(per_row, 1))
                out, ok = np.empty((max(hi - lo, 0),) + trailing, dtype=np.float64), True
                for r in range(lo, hi, chunk):
                    rec = eng.reconstruct(self._state, block, slice(r, min(r + chunk, hi)), dtype=torch.float64)
                    if rec is None:
                        ok = False
                        break
                    out[r - lo: r - lo + rec.shape[0]] = rec.cpu().numpy()
                if ok and hi > lo:
                    return out
        from .util import factors_to_tensor                                  # no device form for this shape / backend
        f = [factors[0] if rows is None else factors[0][rows]] + list(factors[1:])
        out = factors_to_tensor(f) + mean
        return torch.from_numpy(out) if device else out

    def _predict_from_scores(self, scores_dev: torch.Tensor) -> np.ndarray:
        """`X_projection @ coef_ @ Q^T + Y_mean` (tpls.py:143, cmtf.py:177); on the device when the backend has the
        kernel (a 65536 x 10 by 10 x 16 host matmul costs 20 ms of BLAS thread start-up on a 256-core host)."""
        Bm = self.coef_ @ self.Y_factors[1].T                                        # R x M, tiny
        be = self._get_engine().be
        if hasattr(be, "predict_rows") and scores_dev.is_cuda:
            with self._get_engine().device_ctx():
                out = be.predict_rows(scores_dev, torch.from_numpy(np.ascontiguousarray(Bm)).to(scores_dev.device),
                                      torch.from_numpy(np.ascontiguousarray(self.Y_mean, dtype=np.float64)).to(scores_dev.device))
            if out is not None:
                return out.cpu().numpy()
        return scores_dev.cpu().numpy() @ Bm + self.Y_mean

    # shared Y-side epilogue of transform (tpls.py:167-184, cmtf.py:212-229) -- host NumPy, I' x M only
    def _y_scores(self, X_scores, Y) -> np.ndarray:
        """Y-side epilogue of transform (tpls.py:167-184, cmtf.py:212-229): per component Y_scores[:, a] = Y q_a, then
        Y -= (X_scores coef_[:, a]) q_a^T.  Runs through the backend's rowdot / y_deflate kernels on the device-resident
        scores (the host form spends its time starting BLAS threads for I' x 16 products)."""
        Y = Y if isinstance(Y, torch.Tensor) else np.asarray(Y)
        if (Y.ndim != 1) and (Y.ndim != 2):
            raise ValueError("Only a matrix (2-mode tensor) Y is allowed.")
        if Y.ndim == 1:
            Y = Y.reshape((-1, 1))
        if self.Y_shape[1:] != tuple(Y.shape[1:]):
            raise ValueError(f"Training Y has shape {self.Y_shape}, while the new Y has shape {tuple(Y.shape)}")
        eng = self._get_engine()
        be = eng.be
        R = self.n_components
        with eng.device_ctx():
            dev = X_scores.device if isinstance(X_scores, torch.Tensor) else be.device
            Xs = X_scores if isinstance(X_scores, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X_scores)).to(dev)
            Yd = to_device_copy(Y, torch.float64, dev)                                                  # private copy: deflated below
            be.center(Yd, torch.from_numpy(np.ascontiguousarray(self.Y_mean, dtype=np.float64)).to(dev), False)   # Y - Y_mean (tpls.py:178)
            Q = torch.from_numpy(np.ascontiguousarray(self.Y_factors[1].T)).to(dev)                     # row a = q_a
            C = torch.from_numpy(np.ascontiguousarray(self.coef_.T)).to(dev)                            # row a = coef_[:, a]
            Y_scores = be.zeros(Yd.shape[0], R)
            col = be.empty(Yd.shape[0])
            for a in range(R):
                be.rowdot(Yd, Q[a], col, None)                                                          # Y @ q_a
                Y_scores[:, a].copy_(col)
                be.y_deflate(Yd, Xs, R, C[a], Q[a])                                                     # Y -= (X_scores coef_[:, a]) q_a^T
            return Y_scores.cpu().numpy()


class tPLS(_EstimatorBase):
    """Tensor PLS (single X block of order >= 2)."""

    def __getitem__(self, index):
        if index == 0:
            return self.X_factors
        if index == 1:
            return self.Y_factors
        if index == 2:
            return self.coef_
        raise IndexError

    def fit(self, X, Y, tol=1e-8, max_iter=100, verbose=0):
        assert X.shape[0] == Y.shape[0]                                   # tpls.py:46
        assert Y.ndim <= 2, "Only a matrix (2-mode tensor) Y is acceptable."
        eng = self._get_engine()
        dev = eng.be.device
        # kept BY REFERENCE (not copied: X may be tens of GB) for validate.get_q2y (validate.py:18-21) and the lazy
        # X_miss; with copy_X=False the fit centres and deflates the caller's tensor in place, so nothing usable is
        # left to keep and get_q2y's assertion fires
        self.original_X, self.original_Y = (X, Y) if self._copy_X else (None, None)
        Y2 = Y.reshape(-1, 1) if Y.ndim == 1 else Y
        self.X_dim = X.ndim
        self.X_shape = tuple(X.shape)
        self.Y_shape = tuple(Y2.shape)
        # a device tensor of the storage type is handed over as it is; the engine clones it before anything writes it (inputs are
        # never modified, tpls.py:74) -- and not at all when the fit only reads it (algorithm="xcov" without missing values)
        Xd = to_device_copy(X, _as_torch_dtype(self._dtype, X), dev, copy=False)
        Yd = to_device_copy(Y2, torch.float64, dev)
        def notice(blocks):                                               # during preprocess, before the loop: tpls.py:62-63
            if blocks[0].has_miss:
                print("X has missing values")

        st = eng.fit([Xd], Yd, self.n_components, tol, max_iter, coupled=False, verbose=verbose, algorithm=self._algorithm,
                     use_graphs=self._graphs, mixed=self._mixed, on_preprocessed=notice,
                     owned=[(Xd is not X) or not self._copy_X])
        del Xd
        blk = st.blocks[0]
        self._state = st
        self.X_hasMiss = blk.has_miss
        self._X_miss = None                       # X_miss (np.isnan(X), tpls.py:64) is built on first access
        self.X_factors = [st.T.cpu().numpy()] + [L.cpu().numpy() for L in blk.loadings]
        self.Y_factors = [st.U.cpu().numpy(), st.Q.cpu().numpy()]
        self.coef_ = st.coef
        self.R2X = blk.r2x
        self.R2Y = st.r2y
        self.X_mean = blk.mean.cpu().numpy().reshape(self.X_shape[1:])
        self.Y_mean = st.y_mean.cpu().numpy()
        self.n_iter_ = list(st.n_iter)

    @property
    def X_miss(self):
        """Positions of missing values (tpls.py:64); computed lazily: at 65536x128x128 it is a 1 GB array
        nothing on the fit path needs (the kernels read the NaNs in band)."""
        if self._X_miss is None and isinstance(self.original_X, np.ndarray):      # (None after copy_X=False)
            self._X_miss = np.isnan(self.original_X)
        return self._X_miss

    def _project_dev(self, X) -> torch.Tensor:
        if self.X_shape[1:] != tuple(X.shape[1:]):
            raise ValueError(f"Training X has shape {self.X_shape}, while the new X has shape {tuple(X.shape)}")
        eng = self._get_engine()
        return _project_blocks(eng, self._state, [X], [_as_torch_dtype(self._dtype, X)], self._mixed)

    def _project(self, X) -> np.ndarray:
        return self._project_dev(X).cpu().numpy()

    def predict(self, X):
        return self._predict_from_scores(self._project_dev(X))                          # tpls.py:143

    def transform(self, X, Y=None):
        scores = self._project_dev(X)
        X_scores = scores.cpu().numpy()
        if Y is not None:
            return X_scores, self._y_scores(scores, Y)
        return X_scores

    def R2X_literal(self, X):
        """calcR2X(X - X_mean, factors_to_tensor(X_factors)) of the training X (util.py:7-15 as tpls.py:115-117 calls it)
        in one read of X on the GPU; equals R2X[-1], which the fit obtains from the deflation sweep instead."""
        eng = self._get_engine()
        Xd = to_device_copy(X, self._state.blocks[0].dtype or torch.float64, eng.be.device, copy=False)
        r2 = eng.r2x_literal(self._state, Xd, 0)
        if r2 is None:
            from .util import calcR2X, factors_to_tensor
            Xh = X.cpu().numpy() if isinstance(X, torch.Tensor) else np.asarray(X)
            r2 = calcR2X(Xh - self.X_mean, factors_to_tensor(self.X_factors))
        return r2

    def X_reconstructed(self, rows=None, device: bool = False):
        return self._reconstruct(0, self.X_factors, self.X_mean, rows, device)          # tpls.py:188-189
