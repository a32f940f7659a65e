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


def synthetic_shard_device(train_dimensions: tuple, n_response: int, n_latent: int, error: float = 0,
                           seed: int = 215, row0: int = 0, rows: int = None, device="cuda", dtype=None,
                           matrix_block: int = 0, nan_fraction: float = 0.0):
    """Device-side version of the recipe for tensors that cannot be staged through NumPy
    (65536x128x128 f32 = 4.3 GB, 262144x256x256 = 68.7 GB).

    The factors are drawn on the host exactly as ``import_synthetic`` draws them (same generator, same
    order, synthetic.py:59-65), so a small case equals the host recipe up to the noise stream; the
    dense rows ``[row0, row0 + rows)`` of X and their noise are formed on the GPU block by block
    (noise from a seeded device generator).  Order-3 X only.  Returns (X, Y[, X_matrix]) as device
    tensors: X (rows, J, K) in ``dtype`` (default float32), Y (rows, M) float64.
    ``matrix_block`` > 0 adds a coupled matrix block A0 @ F^T (F ~ N(0,1), seed + 1) sharing the
    sample mode (BASELINE configs[2]); ``nan_fraction`` > 0 plants an i.i.d. NaN mask (configs[3])."""
    import torch

    I_total, J, K = train_dimensions
    rows = I_total - row0 if rows is None else rows
    dtype = dtype or torch.float32
    rng = np.random.default_rng(seed)
    A0 = rng.normal(0, 1, size=(I_total, n_latent))
    C = rng.normal(0, 1, size=(n_response, n_latent))
    BJ = rng.normal(0, 1, size=(J, n_latent))
    BK = rng.normal(0, 1, size=(K, n_latent))
    g = torch.Generator(device=device).manual_seed(1000 + seed + row0)
    A0d = torch.from_numpy(A0[row0:row0 + rows]).to(device)
    KR = (torch.from_numpy(BJ).to(device)[:, None, :] * torch.from_numpy(BK).to(device)[None, :, :]).reshape(J * K, n_latent)
    X = torch.empty(rows, J * K, device=device, dtype=dtype)
    step = max(1, (1 << 26) // (J * K))                 # ~256 MB of f32 per block
    for r in range(0, rows, step):
        blk = (A0d[r:r + step] @ KR.T).to(dtype)
        if error:
            blk += error * torch.randn(blk.shape, device=device, dtype=dtype, generator=g)
        if nan_fraction:
            blk[torch.rand(blk.shape, device=device, generator=g) < nan_fraction] = float("nan")
        X[r:r + step] = blk
    Y = A0d @ torch.from_numpy(C).to(device).T
    if error:
        Y += error * torch.randn(Y.shape, device=device, dtype=torch.float64, generator=g)
    out = [X.view(rows, J, K), Y]
    if matrix_block:
        F = np.random.default_rng(seed + 1).normal(0, 1, size=(matrix_block, n_latent))
        Xm = (A0d @ torch.from_numpy(F).to(device).T).to(dtype)
        if error:
            Xm += error * torch.randn(Xm.shape, device=device, dtype=dtype, generator=g)
        out.append(Xm)
    return tuple(out)
