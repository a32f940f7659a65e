"""``ctPLS``: coupled tensor PLS over a list of X blocks that share the sample mode and one score
matrix (reference cmtf_pls/cmtf.py:15-237), fitted by the MI355X NIPALS engine.

Same surface as the reference: ``fit(Xs, Y) / predict(Xs) / transform(Xs, Y=None) /
Xs_reconstructed / copy``, Mapping ``[0], [1], [2]`` -> Xs_factors, Y_factors, coef_.
``Xs_factors[ti][0]`` is the one shared ``factor_T`` array for every block (cmtf.py:61-65).
"""
from __future__ import annotations

import numpy as np
import torch

from .tpls import _EstimatorBase, _as_torch_dtype, _project_blocks, to_device_copy


class ctPLS(_EstimatorBase):
    """Coupled tensor PLS"""

    def __getitem__(self, index):
        if index == 0:
            return self.Xs_factors
        if index == 1:
            return self.Y_factors
        if index == 2:
            return self.coef_
        raise IndexError

    def fit(self, Xs, Y, tol=1e-8, max_iter=100, verbose=0):
        assert isinstance(Xs, list)                                       # cmtf.py:46
        for X in Xs:
            assert X.shape[0] == Y.shape[0]                               # cmtf.py:49
            assert X.ndim >= 2                                            # cmtf.py:50
        assert Y.ndim <= 2, "Only a matrix (2-mode tensor) Y is acceptable."
        eng = self._get_engine()
        dev = eng.be.device
        Y2 = Y.reshape(-1, 1) if Y.ndim == 1 else Y
        self.Xs_len = len(Xs)
        self.Xs_dim = [X.ndim for X in Xs]
        self.Xs_shape = [tuple(X.shape) for X in Xs]
        self.Y_shape = tuple(Y2.shape)
        Xd = [to_device_copy(X, _as_torch_dtype(self._dtype, X), dev, copy=False) for X in Xs]    # cloned by the engine if written
        Yd = to_device_copy(Y2, torch.float64, dev)
        def notice(blocks):                                               # during preprocess, before the loop: cmtf.py:78-79
            if any(b.has_miss for b in blocks):
                print("At least one X has missing values")

        st = eng.fit(Xd, Yd, self.n_components, tol, max_iter, coupled=True, verbose=verbose, algorithm=self._algorithm,
                     use_graphs=self._graphs, mixed=self._mixed, on_preprocessed=notice,
                     owned=[(xd is not X) or not self._copy_X for xd, X in zip(Xd, Xs)])
        del Xd
        self._state = st
        self.factor_T = st.T.cpu().numpy()
        self.Xs_factors = [[self.factor_T] + [L.cpu().numpy() for L in blk.loadings] for blk in st.blocks]
        self.Y_factors = [st.U.cpu().numpy(), st.Q.cpu().numpy()]
        self.coef_ = st.coef
        self.R2Xs = [blk.r2x for blk in st.blocks]
        self.R2Y = st.r2y
        self.Xs_mean = [blk.mean.cpu().numpy().reshape(shape[1:]) for blk, shape in zip(st.blocks, self.Xs_shape)]
        self.Y_mean = st.y_mean.cpu().numpy()
        self.Xs_hasMiss = [blk.has_miss for blk in st.blocks]
        # kept by reference for the lazy Xs_miss (cmtf.py:80-82); with copy_X=False the blocks were fitted in place
        self._Xs_in, self._Xs_miss = (Xs if self._copy_X else [None] * len(Xs)), None
        self.n_iter_ = list(st.n_iter)

    @property
    def Xs_miss(self):
        if self._Xs_miss is None:
            self._Xs_miss = [np.isnan(X) if isinstance(X, np.ndarray) else None for X in self._Xs_in]
        return self._Xs_miss

    def _project_dev(self, Xs) -> torch.Tensor:
        assert len(Xs) == self.Xs_len                                     # cmtf.py:144,181
        for ti, X in enumerate(Xs):
            if self.Xs_shape[ti][1:] != tuple(X.shape[1:]):
                raise ValueError(
                    f"Training X[{ti}] has shape {self.Xs_shape[ti]}, while the new X has shape {tuple(X.shape)}"
                )
        eng = self._get_engine()
        return _project_blocks(eng, self._state, list(Xs), [_as_torch_dtype(self._dtype, X) for X in Xs], self._mixed)

    def _project(self, Xs) -> np.ndarray:
        return self._project_dev(Xs).cpu().numpy()

    def predict(self, Xs):
        return self._predict_from_scores(self._project_dev(Xs))                         # cmtf.py:177

    def transform(self, Xs, Y=None):
        scores = self._project_dev(Xs)
        X_scores = scores.cpu().numpy()
        if Y is not None:
            return X_scores, self._y_scores(scores, Y)
        return X_scores

    def Xs_reconstructed(self, rows=None, device: bool = False):
        return [self._reconstruct(ti, self.Xs_factors[ti], self.Xs_mean[ti], rows, device) for ti in range(self.Xs_len)]   # cmtf.py:233-237
