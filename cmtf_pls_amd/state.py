"""Shapes, limits and the fitted state of the NIPALS engine (reference attributes: cmtf_pls/tpls.py:15-71, cmtf.py:15-83).

`split_trailing` maps a block's trailing modes onto the factored loading (wA, wB) every kernel takes; `validate_limits` refuses a fit
the kernel set cannot finish BEFORE the first sweep; `BlockState` / `FitState` are what a fit leaves behind (device-resident factors,
host-side R2 arrays, and `report`: which form of every step actually ran)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


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
