"""Seeded synthetic CP-structured data with the reference's recipe and draw order
(cmtf_pls/synthetic.py:5-79): sample factor, response factor, remaining mode factors, X noise,
Y noise, all from one ``np.random.default_rng(seed)``.  Host NumPy (inputs of the small configs);
bench.py forms the large benchmark tensors on the GPU from the same kind of factors."""
from types import SimpleNamespace

import numpy as np

from .util import factors_to_tensor


def _cp_record(factors, y_factor):
    return SimpleNamespace(factors=factors, y_factor=y_factor, rank=factors[0].shape[1],
                           shape=tuple(f.shape[0] for f in factors), weights=None)


def import_synthetic(train_dimensions: tuple, n_response: int, n_latent: int, error: float = 0, seed: int = 215):
    rng = np.random.default_rng(seed)
    sample_factor = rng.normal(0, 1, size=(train_dimensions[0], n_latent))
    y_factor = rng.normal(0, 1, size=(n_response, n_latent))
    factors = [sample_factor] + [rng.normal(0, 1, size=(d, n_latent)) for d in train_dimensions[1:]]
    x = factors_to_tensor(factors) + rng.normal(0, error, size=train_dimensions)
    y = sample_factor @ y_factor.T + rng.normal(0, error, size=(train_dimensions[0], n_response))
    if y.shape[1] == 1:
        y = y.flatten()
    return x, y, _cp_record(factors, y_factor)


def make_synthetic_test(cp_tensor, test_samples: int, error: float = 0, seed: int = 215):
    rng = np.random.default_rng(seed)
    factors = cp_tensor.factors
    factors[0] = rng.normal(0, 1, size=(test_samples, cp_tensor.rank))    # in place, as the reference
    test = _cp_record(factors, cp_tensor.y_factor)
    x_test = factors_to_tensor(factors) + rng.normal(0, error, size=test.shape)
    y_test = factors[0] @ cp_tensor.y_factor.T
    y_test = y_test + rng.normal(0, error, size=y_test.shape)
    return x_test, y_test, test
