"""Restatement of the reference's seeded synthetic-data recipe (TEST INFRASTRUCTURE).

Reference: cmtf_pls/synthetic.py:5-79.  The draw order of the generator is part of the contract
(tests/test_synthetic.py:27-41): sample-mode factor, response factor, remaining mode factors,
X noise, Y noise.  tensorly's ``CPTensor`` / ``cp_to_tensor`` / ``dot`` are replaced by a tiny
record and plain NumPy contractions.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class SyntheticCP:
    """What the reference returns as ``cp_tensor`` (factors + ``y_factor``), synthetic.py:67-68."""

    factors: List[np.ndarray]
    y_factor: np.ndarray

    @property
    def rank(self) -> int:
        return self.factors[0].shape[1]

    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(f.shape[0] for f in self.factors)


def _cp_dense(factors: List[np.ndarray]) -> np.ndarray:
    rank = factors[0].shape[1]
    kr = np.ones((1, rank))
    for f in factors[1:]:
        kr = (kr[:, None, :] * f[None, :, :]).reshape(-1, rank)
    return (factors[0] @ kr.T).reshape([f.shape[0] for f in factors])


def import_synthetic(train_dimensions: tuple, n_response: int, n_latent: int,
                     error: float = 0, seed: int = 215):
    """Reference: ``import_synthetic`` synthetic.py:37-79."""
    rng = np.random.default_rng(seed)                                             # :59
    x_factors = [rng.normal(0, 1, size=(train_dimensions[0], n_latent))]          # :61
    y_factor = rng.normal(0, 1, size=(n_response, n_latent))                      # :62
    for d in train_dimensions[1:]:                                                # :64-65
        x_factors.append(rng.normal(0, 1, size=(d, n_latent)))
    cp = SyntheticCP(x_factors, y_factor)
    x = _cp_dense(x_factors)                                                      # :70
    x += rng.normal(0, error, size=train_dimensions)                              # :71
    y = x_factors[0] @ y_factor.T                                                 # :73
    y += rng.normal(0, error, size=(train_dimensions[0], n_response))             # :74
    if y.shape[1] == 1:                                                           # :76-77
        y = y.flatten()
    return x, y, cp


def make_synthetic_test(cp: SyntheticCP, test_samples: int, error: float = 0, seed: int = 215):
    """Reference: ``make_synthetic_test`` synthetic.py:5-34 (note: like the reference it replaces
    ``cp.factors[0]`` in place, synthetic.py:24-25)."""
    rng = np.random.default_rng(seed)
    test_factors = cp.factors
    test_factors[0] = rng.normal(0, 1, size=(test_samples, cp.rank))
    test_cp = SyntheticCP(test_factors, cp.y_factor)
    x_test = _cp_dense(test_factors)
    x_test += rng.normal(0, error, size=test_cp.shape)
    y_test = test_factors[0] @ cp.y_factor.T
    y_test += rng.normal(0, error, size=y_test.shape)
    return x_test, y_test, test_cp
