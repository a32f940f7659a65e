"""Randomised shapes through every X kernel against NumPy float64: ragged rows/columns, B not a
multiple of the vector width, P below one wavefront, exact-cover shapes that select the guard-free
specialisations (score_deflate KC/FULL, xcov/mttkrp FAST), both storage types, with and without NaNs.
Seeded: the same 54 cases every run (the last six: rows of 24 000 - 1 048 576 elements, trailing modes up to the
rank-1 kernels' limit of 1024).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TDT = {"f32": torch.float32, "f64": torch.float64}


@pytest.fixture(scope="module")
def be():
    from cmtf_pls_amd.backend import HipBackend
    return HipBackend("cuda:0")


def _cases():
    rng = np.random.default_rng(2024)
    special = [(1, 1, 1), (2, 1, 3), (3, 2, 2), (5, 1, 64), (16, 4, 4), (17, 16, 16), (33, 64, 4), (64, 8, 32),
               (96, 128, 128), (40, 256, 64), (48, 64, 256), (31, 3, 100), (130, 1, 1024), (70, 33, 31), (19, 2, 514),
               (256, 16, 64), (12, 512, 512), (6, 1024, 1024), (9, 1000, 24), (7, 24, 1000), (20, 512, 128), (32, 128, 640)]
    cases = []
    for i, (I, A, B) in enumerate(special):
        cases.append((I, A, B, "f32" if i % 2 == 0 else "f64", i % 3 == 0))
    for _ in range(32):
        I = int(rng.integers(1, 200))
        A = int(rng.integers(1, 40))
        B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 31, 32, 48, 64, 100, 128]))
        cases.append((I, A, B, str(rng.choice(["f32", "f64"])), bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("I,A,B,dt,masked", _cases())
def test_random_shape(be, I, A, B, dt, masked):
    rng = np.random.default_rng(I * 1000003 + A * 1009 + B)
    P = A * B
    x = rng.normal(size=(I, P))
    if dt == "f32":
        x = x.astype(np.float32).astype(np.float64)
    if masked:
        x[rng.random(x.shape) < 0.25] = np.nan
    x0 = np.nan_to_num(x)
    obs = ~np.isnan(x)
    dev = lambda a, t=None: (torch.from_numpy(np.ascontiguousarray(a)).to(t) if t else torch.from_numpy(np.ascontiguousarray(a))).to("cuda:0")
    host = lambda t: t.cpu().numpy()
    u, wa, wb = rng.normal(size=I), rng.normal(size=A), rng.normal(size=B)
    w = np.kron(wa, wb)
    tol = dict(rtol=1e-10, atol=1e-9)

    X = dev(x, TDT[dt])
    # colstats
    cs, cc = be.colstats(X)
    np.testing.assert_allclose(host(cs), x0.sum(0), **tol)
    assert np.array_equal(host(cc), obs.sum(0).astype(float))
    # contraction
    Z = be.mode0_contract(X, dev(u), masked)
    np.testing.assert_allclose(host(Z), (x0 if masked else x).T @ u, **tol)
    # score (masked form uses the per-row counts)
    rowcnt = dev(obs.sum(1).astype(float)) if masked else None
    with np.errstate(all="ignore"):
        t_want = (x0 @ w) / obs.sum(1) * P if masked else x @ w
    t = host(be.score(X, A, B, dev(wa), dev(wb), rowcnt, be.empty(I)))
    assert np.array_equal(np.isnan(t), np.isnan(t_want))
    np.testing.assert_allclose(np.nan_to_num(t), np.nan_to_num(t_want), **tol)
    # xcov and mttkrp (NaN-free operands for mttkrp: it is only used on complete data)
    M = int(rng.integers(1, 40))
    Y = rng.normal(size=(I, M))
    S = be.xcov(X, dev(Y), masked)
    np.testing.assert_allclose(host(S), Y.T @ (x0 if masked else x), **tol)
    if not masked:
        R = int(rng.integers(1, 20))
        WA, WB = rng.normal(size=(A, R)), rng.normal(size=(B, R))
        out = be.mttkrp(X, A, B, dev(WA), dev(WB), be.empty(I, R))
        if out is None:                                        # loadings beyond the LDS: the engine projects sequentially
            assert (A + B) * (R + (-R) % 16) * 8 > 96 * 1024 or R > 32
        else:
            np.testing.assert_allclose(host(out), x @ (WA[:, None, :] * WB[None, :, :]).reshape(P, R), **tol)
    # fused score + deflate, then plain deflate on a second copy
    t_safe = np.nan_to_num(t_want)
    X1 = dev(x, TDT[dt])
    tt = be.empty(I)
    ssq = be.score_deflate(X1, A, B, dev(wa), dev(wb), rowcnt, tt)
    if ssq is not None:
        want = x - np.outer(t_want, w)
        if dt == "f32":
            want = want.astype(np.float32).astype(np.float64)
        got = host(X1).astype(np.float64)
        ok_rows = ~np.isnan(t_want)
        assert np.array_equal(np.isnan(got[ok_rows]), np.isnan(want[ok_rows]))
        np.testing.assert_allclose(np.nan_to_num(got[ok_rows]), np.nan_to_num(want[ok_rows]), rtol=3e-7 if dt == "f32" else 1e-12, atol=1e-9)
    X2 = dev(x, TDT[dt])
    ssq2 = be.deflate(X2, A, B, dev(t_safe), dev(wa), dev(wb))
    want2 = x - np.outer(t_safe, w)
    if dt == "f32":
        want2 = want2.astype(np.float32).astype(np.float64)
    got2 = host(X2).astype(np.float64)
    assert np.array_equal(np.isnan(got2), np.isnan(want2))
    np.testing.assert_allclose(np.nan_to_num(got2), np.nan_to_num(want2), rtol=3e-7 if dt == "f32" else 1e-12, atol=1e-9)
    np.testing.assert_allclose(host(ssq2)[0], np.nansum(got2 ** 2), rtol=1e-10, atol=1e-12)
    # rank-1 of a matrix Z (only when Z has a clear leading pair to compare against LAPACK)
    if A >= 2 and B >= 2 and A <= 256 and B <= 1024:
        Zm = rng.normal(size=(A, B)) + 4.0 * np.outer(rng.normal(size=A), rng.normal(size=B))
        wA, wB = be.empty(A), be.empty(B)
        be.rank1(dev(Zm.ravel()), A, B, wA, wB)
        U, Sv, Vt = np.linalg.svd(Zm, full_matrices=False)
        uu, vv = U[:, 0], Vt[0]
        if vv[np.argmax(np.abs(vv))] < 0:
            uu, vv = -uu, -vv
        np.testing.assert_allclose(host(wA), uu, rtol=0, atol=1e-9)
        np.testing.assert_allclose(host(wB), vv, rtol=0, atol=1e-9)


@pytest.mark.parametrize("A,B", [(1, 16), (4, 12)])
def test_millions_of_short_rows(be, A, B):
    """2.5 million rows of 16-48 elements: 64-bit row indexing in the short-row kernels (a wavefront owns several
    whole rows), the narrow contraction, centring and the fused deflation + contraction."""
    I = 2_500_003
    P = A * B
    rng = np.random.default_rng(7)
    x = rng.normal(size=(I, P)).astype(np.float32)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    w = np.kron(wa, wb)
    u = rng.normal(size=I)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    X = dev(x)
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(be.mode0_contract(X, dev(u), False).cpu().numpy(), x64.T @ u, rtol=1e-9, atol=1e-7)
    t = be.score(X, A, B, dev(wa), dev(wb), None, be.empty(I)).cpu().numpy()
    np.testing.assert_allclose(t, x64 @ w, rtol=1e-10, atol=1e-9)
    tt = be.empty(I)
    ssq = be.score_deflate(X, A, B, dev(wa), dev(wb), None, tt)
    assert ssq is not None
    want = (x64 - np.outer(t, w)).astype(np.float32)
    got = X.cpu().numpy()
    np.testing.assert_allclose(got[[0, 1, I // 2, I - 2, I - 1]], want[[0, 1, I // 2, I - 2, I - 1]], rtol=3e-7, atol=1e-7)
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(ssq.cpu().numpy()[0], float((want.astype(np.float64) ** 2).sum()), rtol=1e-9)
    mean = torch.from_numpy(want.astype(np.float64).mean(0)).to("cuda:0")
    rowcnt, _ = be.center(X, mean, True)
    assert float(rowcnt.min()) == P and float(rowcnt.max()) == P
