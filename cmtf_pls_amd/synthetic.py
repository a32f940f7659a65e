"""Seeded synthetic CP-structured data with the reference's recipe and draw order
(cmtf_pls/synthetic.py:5-79): sample factor, response factor, remaining mode factors, X noise,
Y noise, all from one ``np.random.default_rng(seed)``.  Host NumPy (inputs of the small configs) plus
device-side forms (``synthetic_shard_device``, ``make_synthetic_test_device``) that draw the factors with the
same generator and order and form the dense tensors with the library's own HIP kernels."""
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


def _device_backend(device):
    import torch

    from .backend import HipBackend
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return HipBackend(dev)


def _cp_rows_device(be, A0d, mode_factors, dtype, sigma, seed, offset, nan_fraction=0.0):
    """Rows of cp_to_tensor([A0, F1, F2, ...]) + N(0, sigma) (synthetic.py:70-71) formed on the GPU: the dense part is
    cmtfpls_recon_* (Khatri-Rao operand never materialised), the noise cmtfpls_add_noise_* (counter-based generator:
    `offset` = global index of this buffer's first element, so a row shard gets the whole tensor's noise)."""
    import torch

    rows, L = A0d.shape
    fd = [torch.from_numpy(np.ascontiguousarray(F)).to(be.device) for F in mode_factors]
    if len(fd) == 1:
        WA, WB = torch.ones(1, L, dtype=torch.float64, device=be.device), fd[0]
    else:
        WA, WB = fd[0], fd[1]
        for F in fd[2:]:
            WB = be.khatri_rao(WB, F)
    P = WA.shape[0] * WB.shape[0]
    X = torch.empty(rows, P, device=be.device, dtype=dtype)
    if be.recon(A0d, WA, WB, None, X) is None:          # last mode not a multiple of 16 bytes: tiny / odd shapes
        KR = (WA[:, None, :] * WB[None, :, :]).reshape(P, L)
        X.copy_((A0d @ KR.T).to(dtype))
    if sigma or nan_fraction:
        be.add_noise(X, sigma, seed, offset, nan_fraction)
    return X


def synthetic_shard_device(train_dimensions: tuple, n_response: int, n_latent: int, error: float = 0,
                           seed: int = 215, row0: int = 0, rows: int = None, device="cuda", dtype=None,
                           matrix_block: int = 0, nan_fraction: float = 0.0):
    """Device-side ``import_synthetic`` (synthetic.py:37-79) for tensors that cannot be staged through NumPy
    (65536x128x128 f32 = 4.3 GB, 262144x256x256 = 68.7 GB), one row shard at a time.

    The factors are drawn on the host exactly as ``import_synthetic`` draws them (same generator, same order,
    synthetic.py:59-65), so with ``error=0`` the result equals the host recipe; the dense rows
    ``[row0, row0 + rows)`` of X are formed on the GPU by the library's own kernels and the N(0, error) noise
    (synthetic.py:71,74) comes from a counter-based device generator keyed by ``seed`` and indexed by the GLOBAL
    element, so shards are consistent: rank g's rows are bit for bit rows [row0, row0 + rows) of the tensor a single
    GPU would form.  Any order of X.  Returns (X, Y[, X_matrix]) as device tensors: X (rows, *trailing) in ``dtype``
    (default float32), Y (rows, M) float64.  ``matrix_block`` > 0 adds a coupled matrix block A0 @ F^T
    (F ~ N(0,1), seed + 1) sharing the sample mode (BASELINE configs[2]); ``nan_fraction`` > 0 plants an i.i.d.
    NaN mask (configs[3])."""
    import torch

    dims = tuple(int(d) for d in train_dimensions)
    I_total = dims[0]
    rows = I_total - row0 if rows is None else rows
    dtype = dtype or torch.float32
    rng = np.random.default_rng(seed)
    A0 = rng.normal(0, 1, size=(I_total, n_latent))                         # synthetic.py:61
    C = rng.normal(0, 1, size=(n_response, n_latent))                       # synthetic.py:62
    mode_factors = [rng.normal(0, 1, size=(d, n_latent)) for d in dims[1:]]  # synthetic.py:64-65
    be = _device_backend(device)
    with torch.cuda.device(be.device):
        A0d = torch.from_numpy(np.ascontiguousarray(A0[row0:row0 + rows])).to(be.device)
        P = int(np.prod(dims[1:]))
        pad = lambda n: (n + 3) // 4 * 4                                    # each array starts on its own Philox block
        X = _cp_rows_device(be, A0d, mode_factors, dtype, error, seed, row0 * P, nan_fraction)
        # Y = A0 C^T + N(0, error) (synthetic.py:73-74): its noise follows X's in the same counter space
        y_base = pad(I_total * P)
        Y = _cp_rows_device(be, A0d, [C], torch.float64, error, seed, y_base + row0 * n_response)
        out = [X.view((rows,) + dims[1:]), Y]
        if matrix_block:
            F = np.random.default_rng(seed + 1).normal(0, 1, size=(matrix_block, n_latent))
            m_base = y_base + pad(I_total * n_response)
            out.append(_cp_rows_device(be, A0d, [F], dtype, error, seed, m_base + row0 * matrix_block))
    return tuple(out)


def make_synthetic_test_device(cp_tensor, test_samples: int, error: float = 0, seed: int = 215, device="cuda", dtype=None):
    """Device-side ``make_synthetic_test`` (synthetic.py:5-34): a fresh sample-mode factor drawn with the reference's
    generator (it REPLACES ``cp_tensor.factors[0]`` in place, as the reference does, synthetic.py:24-25), the test
    tensor and responses formed on the GPU.  Returns (x_test, y_test, cp_test) with device tensors."""
    import torch

    rng = np.random.default_rng(seed)
    factors = cp_tensor.factors
    factors[0] = rng.normal(0, 1, size=(test_samples, cp_tensor.rank))
    test = _cp_record(factors, cp_tensor.y_factor)
    be = _device_backend(device)
    with torch.cuda.device(be.device):
        A0d = torch.from_numpy(np.ascontiguousarray(factors[0])).to(be.device)
        P = int(np.prod(test.shape[1:]))
        X = _cp_rows_device(be, A0d, factors[1:], dtype or torch.float32, error, seed, 0)
        Y = _cp_rows_device(be, A0d, [cp_tensor.y_factor], torch.float64, error, seed, (test_samples * P + 3) // 4 * 4)
    return X.view(test.shape), Y, test
