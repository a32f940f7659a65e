"""Device-side synthetic inputs: ``synthetic_shard_device`` / ``make_synthetic_test_device``
against the oracle's restatement of the reference recipe (cmtf_pls/synthetic.py:5-79), shard consistency, and the
counter-based noise generator against its NumPy restatement (tests/philox_ref.py, itself checked against the
published Philox4x32-10 known-answer vectors in tests/test_philox_ref.py)."""
import numpy as np
import pytest
import torch

import oracle as O
import philox_ref as PR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims,M", [((48, 10, 8), 4), ((40, 12), 3), ((24, 6, 4, 8), 2), ((37, 7, 9), 5), ((32, 128, 128), 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_device_recipe_equals_the_reference_recipe_without_noise(dims, M, dtype):
    """error = 0: the factors are drawn with the reference's generator and order, so X and Y equal import_synthetic's."""
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    x, y, _ = O.import_synthetic(dims, M, 3, error=0.0, seed=215)
    y = y.reshape(dims[0], -1)
    X, Y = synthetic_shard_device(dims, M, 3, error=0.0, seed=215, device="cuda:0", dtype=dtype)
    assert tuple(X.shape) == dims and X.dtype == dtype and Y.dtype == torch.float64
    tol = 1e-6 if dtype == torch.float32 else 1e-12
    np.testing.assert_allclose(X.cpu().numpy().astype(np.float64), x, rtol=tol, atol=tol * np.abs(x).max())
    np.testing.assert_allclose(Y.cpu().numpy(), y, rtol=1e-12, atol=1e-12 * np.abs(y).max())


@pytest.mark.parametrize("dims", [(96, 16, 12), (50, 7, 9)])
def test_shards_are_rows_of_the_whole_tensor(dims):
    """Rank g's shard (rows [a, b)) equals rows [a, b) of the single-GPU tensor bit for bit, noise and NaN mask
    included, also when a shard starts inside a Philox block (odd row length)."""
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    kw = dict(error=0.3, seed=7, device="cuda:0", matrix_block=20, nan_fraction=0.2)
    X, Y, Xm = synthetic_shard_device(dims, 5, 3, **kw)
    cuts = [0, 13, 14, dims[0] - 9, dims[0]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        Xs, Ys, Xms = synthetic_shard_device(dims, 5, 3, row0=a, rows=b - a, **kw)
        assert torch.equal(Xs.view(torch.int32), X[a:b].contiguous().view(torch.int32))
        assert torch.equal(Ys, Y[a:b]) and torch.equal(Xms.view(torch.int32), Xm[a:b].contiguous().view(torch.int32))
    frac = float(torch.isnan(X).double().mean())
    assert abs(frac - 0.2) < 0.02
    assert not torch.isnan(Y).any()


def test_noise_is_the_published_generator():
    """cmtfpls_add_noise_f64 on zeros == Philox4x32-10 + Box-Muller as restated in NumPy, element by element."""
    from cmtf_pls_amd.backend import HipBackend
    be = HipBackend("cuda:0")
    for first, n in ((0, 4096), (5, 1003), (2 ** 33 + 2, 777)):
        X = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        be.add_noise(X, 1.0, 215, offset=first)
        np.testing.assert_allclose(X.cpu().numpy(), PR.normals(first, n, 215), rtol=0, atol=1e-12)
        Xn = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        be.add_noise(Xn, 0.0, 99, offset=first, nan_fraction=0.3)
        assert np.array_equal(np.isnan(Xn.cpu().numpy()), PR.nan_mask(first, n, 99, 0.3))
    big = torch.zeros(1 << 22, dtype=torch.float32, device="cuda:0")
    be.add_noise(big, 0.5, 1)
    assert abs(float(big.mean())) < 2e-3 and abs(float(big.std()) - 0.5) < 2e-3


def test_noisy_recipe_has_the_reference_statistics():
    """With noise the device stream differs from NumPy's PCG64 draw, so the check is distributional: the residual
    X - cp_to_tensor(factors) is N(0, error), and a fit recovers what a fit of the host recipe recovers."""
    from cmtf_pls_amd import tPLS
    from cmtf_pls_amd.synthetic import synthetic_shard_device
    dims, M, L, err = (400, 12, 16), 4, 3, 0.1
    x0, y0, _ = O.import_synthetic(dims, M, L, error=0.0, seed=215)
    X, Y = synthetic_shard_device(dims, M, L, error=err, seed=215, device="cuda:0", dtype=torch.float64)
    rx, ry = X.cpu().numpy() - x0, Y.cpu().numpy() - y0
    assert abs(rx.std() - err) < 0.002 and abs(rx.mean()) < 0.002 and abs(ry.std() - err) < 0.01
    xh, yh, _ = O.import_synthetic(dims, M, L, error=err, seed=215)
    a, b = tPLS(3), tPLS(3)
    a.fit(X.cpu().numpy(), Y.cpu().numpy())
    b.fit(xh, yh)
    np.testing.assert_allclose(a.R2X, b.R2X, atol=5e-3)
    np.testing.assert_allclose(a.R2Y, b.R2Y, atol=5e-3)


def test_make_synthetic_test_device_matches_the_reference_recipe():
    from cmtf_pls_amd.synthetic import import_synthetic, make_synthetic_test_device
    _, _, cp_dev = import_synthetic((30, 8, 12), 3, 2, error=0.0, seed=5)
    _, _, cp_ora = O.import_synthetic((30, 8, 12), 3, 2, error=0.0, seed=5)
    xt, yt, test = make_synthetic_test_device(cp_dev, 17, error=0.0, seed=7, device="cuda:0", dtype=torch.float64)
    xo, yo, _ = O.make_synthetic_test(cp_ora, 17, 0.0, seed=7)
    np.testing.assert_allclose(xt.cpu().numpy(), xo, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(yt.cpu().numpy(), yo, rtol=1e-12, atol=1e-12)
    assert test.shape == (17, 8, 12) and cp_dev.factors[0].shape == (17, 2)      # replaced in place, as the reference does
