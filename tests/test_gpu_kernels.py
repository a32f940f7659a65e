"""Kernel-level parity on a real MI355X: every C-ABI entry point of include/cmtfpls.h against the
CPU oracle / plain NumPy float64 on the same seeded inputs.

Tolerances (written here, per the floating-point bar): every kernel accumulates in f64 from the
stored X, so against a float64 computation on the SAME stored values the sums agree to
rtol 1e-11 (only summation order differs); f32-storage writes (center / deflate) are one rounding
to f32, checked at rtol 2e-7.
"""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

RT = dict(rtol=1e-11, atol=1e-11)


@pytest.fixture(scope="module")
def be():
    from cmtf_pls_amd.backend import HipBackend
    return HipBackend("cuda:0")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to("cuda:0")


def host(t):
    return t.cpu().numpy()


TDT = {"f32": torch.float32, "f64": torch.float64}
# (I, A, B): vector path (B % 4 == 0), scalar path (odd B), matrix block (A == 1), long rows
SHAPES = [(37, 10, 8), (100, 38, 65), (64, 1, 20), (33, 1, 7), (50, 128, 128), (29, 5, 4)]


def make_x(shape, dt, nan_frac=0.0, seed=0):
    rng = np.random.default_rng(seed)
    I, A, B = shape
    x = rng.normal(size=(I, A * B))
    if dt == "f32":
        x = x.astype(np.float32).astype(np.float64)
    if nan_frac:
        x[rng.random(x.shape) < nan_frac] = np.nan
    return x


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape", SHAPES + [(70, 256, 256), (41, 50, 100), (9, 1, 70000)])     # long rows: row-segment centring
def test_colstats_and_center(be, shape, dt):
    x = make_x(shape, dt, nan_frac=0.2, seed=1)
    x[:, 3] = np.nan                                       # a column with no observation
    x[5, :] = np.nan                                       # a row with no observation
    X = dev(x, TDT[dt])
    colsum, colcnt = be.colstats(X)
    obs = ~np.isnan(x)
    np.testing.assert_allclose(host(colsum), np.where(obs, x, 0).sum(0), **RT)
    assert np.array_equal(host(colcnt), obs.sum(0).astype(float))
    with np.errstate(all="ignore"):
        mean = np.where(obs, x, 0).sum(0) / obs.sum(0)
    mean_d = colsum / colcnt
    rowcnt, ssq = be.center(X, mean_d, True)
    want = x - mean
    if dt == "f32":
        want = want.astype(np.float32).astype(np.float64)
    got = host(X).astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=2e-7 if dt == "f32" else 1e-14, atol=1e-12)
    assert np.array_equal(host(rowcnt), (~np.isnan(want)).sum(1).astype(float))
    np.testing.assert_allclose(host(ssq)[0], np.nansum(got ** 2), rtol=1e-11)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape", SHAPES + [(1000, 16, 16)])
def test_mode0_contract(be, shape, dt, masked):
    x = make_x(shape, dt, nan_frac=0.3 if masked else 0.0, seed=2)
    u = np.random.default_rng(3).normal(size=shape[0])
    X = dev(x, TDT[dt])
    Z = be.mode0_contract(X, dev(u), masked)
    if masked:
        cnt = (~np.isnan(x)).sum(0).astype(float)
        be.colscale(Z, dev(cnt), float(shape[0]))
        want = O.masked_mode0_contract(x, u)
    else:
        want = O.mode0_contract(x, u)
    np.testing.assert_allclose(host(Z), want, **RT)


def test_mode0_contract_nan_propagates_when_unmasked(be):
    x = make_x((20, 4, 4), "f64", seed=4)
    x[3, 5] = np.nan
    Z = host(be.mode0_contract(dev(x), dev(np.ones(20)), False))
    assert np.isnan(Z[5]) and not np.isnan(np.delete(Z, 5)).any()      # np.einsum semantics (tpls.py:83)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
# the last four: a FEW LONG rows (one 1024-thread workgroup per row: round 3) -- S of BASELINE configs[1] / [4], a matrix block, a ragged tail
@pytest.mark.parametrize("shape", SHAPES + [(16, 128, 128), (32, 256, 256), (3, 1, 8192), (5, 96, 100)])
def test_score(be, shape, dt, masked):
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.3 if masked else 0.0, seed=5)
    if masked:
        x[2, :] = np.nan                                    # empty row -> NaN score (missingvals.py:37)
    rng = np.random.default_rng(6)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    X = dev(x, TDT[dt])
    rowcnt = dev((~np.isnan(x)).sum(1).astype(float)) if masked else None
    t = host(be.score(X, A, B, dev(wa), dev(wb), rowcnt, be.empty(I)))
    x3 = x.reshape(I, A, B)
    want = O.masked_score(x3, [wa, wb]) if masked else O.score_contract(x3, [wa, wb])
    assert np.array_equal(np.isnan(t), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(t), np.nan_to_num(want), **RT)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape", SHAPES)
def test_deflate(be, shape, dt):
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.1, seed=7)
    rng = np.random.default_rng(8)
    wa, wb, t = rng.normal(size=A), rng.normal(size=B), rng.normal(size=I)
    X = dev(x, TDT[dt])
    ssq = be.deflate(X, A, B, dev(t), dev(wa), dev(wb))
    want = x - np.outer(t, np.kron(wa, wb))
    if dt == "f32":
        want = want.astype(np.float32).astype(np.float64)
    got = host(X).astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=2e-7 if dt == "f32" else 1e-14, atol=1e-12)
    np.testing.assert_allclose(host(ssq)[0], np.nansum(got ** 2), rtol=1e-11)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
# the last four take the 1024-thread variant (half of the row parked in LDS): exact cover, stride % B == 0
# with a ragged tail, the general vector walk, and the scalar walk
@pytest.mark.parametrize("shape", SHAPES + [(9, 64, 64), (3, 200, 75), (5, 256, 256), (4, 100, 256), (4, 150, 200), (3, 101, 99)])
def test_score_deflate_fused(be, shape, dt, masked):
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.2 if masked else 0.0, seed=9)
    rng = np.random.default_rng(10)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    wa /= np.linalg.norm(wa)
    wb /= np.linalg.norm(wb)
    X = dev(x, TDT[dt])
    rowcnt = dev((~np.isnan(x)).sum(1).astype(float)) if masked else None
    t = be.empty(I)
    ssq = be.score_deflate(X, A, B, dev(wa), dev(wb), rowcnt, t)
    if ssq is None:
        assert A * B > 1024 * 16 * (2 if dt == "f64" else 4 if B % 4 == 0 else 1)
        return
    x3 = x.reshape(I, A, B)
    t_want = O.masked_score(x3, [wa, wb]) if masked else O.score_contract(x3, [wa, wb])
    np.testing.assert_allclose(host(t), t_want, **RT)
    want = x - np.outer(t_want, np.kron(wa, wb))
    if dt == "f32":
        want = want.astype(np.float32).astype(np.float64)
    got = host(X).astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=3e-7 if dt == "f32" else 1e-13, atol=1e-10)
    np.testing.assert_allclose(host(ssq)[0], np.nansum(got ** 2), rtol=1e-11)


def _svd_pair(Z):
    U, S, Vt = np.linalg.svd(Z, full_matrices=False)
    u, v = U[:, 0], Vt[0]
    if v[np.argmax(np.abs(v))] < 0:
        u, v = -u, -v
    return u, v, S


@pytest.mark.parametrize("shape", [(10, 8), (8, 10), (128, 128), (38, 65), (200, 7), (1, 9), (9, 1), (256, 256), (17, 300)])
def test_rank1_matches_lapack(be, shape):
    A, B = shape
    rng = np.random.default_rng(11)
    Z = rng.normal(size=shape) + 3.0 * np.outer(rng.normal(size=A), rng.normal(size=B))
    wA, wB = be.empty(A), be.empty(B)
    be.rank1(dev(Z.ravel()), A, B, wA, wB)
    u, v, S = _svd_pair(Z)
    np.testing.assert_allclose(host(wA), u, rtol=0, atol=1e-10)
    np.testing.assert_allclose(host(wB), v, rtol=0, atol=1e-10)
    # same object as the oracle's restatement of parafac(Z, 1, init="svd") (tpls.py:86-88)
    fa, fb = O.rank1_factors(Z)
    np.testing.assert_allclose(host(wA), fa, rtol=0, atol=1e-9)
    np.testing.assert_allclose(host(wB), fb, rtol=0, atol=1e-9)


@pytest.mark.parametrize("shape", [(600, 640), (1024, 1024), (900, 40), (3, 5000), (1040, 1500), (2100, 1100)])
def test_rank1_large(be, shape):
    """Sides beyond 1024 (round 3: the control block of trace / Frobenius partials is sized by the call; the limit is
    min(A, B) <= 4096) and wide / tall shapes whose larger side is streamed in 256-column panels."""
    A, B = shape
    rng = np.random.default_rng(71)
    Z = rng.normal(size=shape) + 6.0 * np.outer(rng.normal(size=A), rng.normal(size=B))
    wA, wB, info = be.empty(A), be.empty(B), be.zeros(2)
    be.rank1(dev(Z.ravel()), A, B, wA, wB, info=info)
    u, v, S = _svd_pair(Z)
    np.testing.assert_allclose(host(wA), u, rtol=0, atol=1e-9)
    np.testing.assert_allclose(host(wB), v, rtol=0, atol=1e-9)
    assert host(info)[0] == 1.0


def test_rank1_refuses_beyond_its_limit_without_launching(be):
    from cmtf_pls_amd import _lib
    wA, wB = be.empty(4100), be.empty(4097)
    with pytest.raises(_lib.CmtfplsError, match="4096"):
        be.rank1(be.empty(16), 4100, 4097, wA, wB)            # refused from the sizes alone: Z is never read


def test_rank1_tensor_larger_modes(be):
    rng = np.random.default_rng(72)
    dims = (40, 30, 20)
    core = rng.normal(size=dims[0])
    for d in dims[1:]:
        core = np.multiply.outer(core, rng.normal(size=d))
    Z = rng.normal(size=dims) + 3.0 * core
    fac = be.zeros(3, 40)
    be.rank1_tensor(dev(Z.ravel()), dims, 1e-8, fac)
    want = O.rank1_factors(Z, 1e-8)
    for m, w in enumerate(want):
        np.testing.assert_allclose(host(fac)[m, : dims[m]], w, rtol=1e-7, atol=1e-9)


def test_rank1_close_singular_values_and_zero_rows(be):
    rng = np.random.default_rng(12)
    A, B = 40, 30
    Qa, _ = np.linalg.qr(rng.normal(size=(A, A)))
    Qb, _ = np.linalg.qr(rng.normal(size=(B, B)))
    s = np.linspace(1.0, 0.1, B)
    s[1] = 0.999                                            # s2/s1 = 0.999: plain power iteration would need ~10^4 steps
    Z = (Qa[:, :B] * s) @ Qb.T
    Z[0, :] = 0.0                                           # constant slice after centring (tests/test_tpls.py:98)
    wA, wB = be.empty(A), be.empty(B)
    be.rank1(dev(Z.ravel()), A, B, wA, wB)
    u, v, S = _svd_pair(Z)
    np.testing.assert_allclose(host(wA), u, rtol=0, atol=1e-8)
    np.testing.assert_allclose(host(wB), v, rtol=0, atol=1e-8)
    assert host(wA)[0] == 0.0                               # exactly zero, not merely small
    np.testing.assert_allclose(np.linalg.norm(host(wA)), 1, rtol=1e-14)
    np.testing.assert_allclose(np.linalg.norm(host(wB)), 1, rtol=1e-14)


@pytest.mark.parametrize("shape", [(128, 128), (256, 256), (96, 160), (200, 200), (160, 72), (16, 4096), (5, 7), (256, 1000), (40, 30)])
@pytest.mark.parametrize("budget", [30, 2, 9])
def test_rank1_chain_of_squarings_in_one_launch_equals_a_launch_per_squaring(be, shape, budget):
    """Round 4: for min(A, B) <= 256 cmtfpls_rank1_f64 runs the Gram matrix and every squaring in ONE launch (syrk_chain_kernel:
    resident workgroups passing the panels of G_s to each other through agent-scope stores whose value is their own flag) --
    bit for bit the loadings, the convergence flag and the squarings used of the launch-per-squaring form
    (cmtfpls_rank1_launches_f64), also when the budget runs out before convergence (2), with an exact zero row / column in Z
    (tests/test_tpls.py:98-104) and with the larger side first (Z transposed inside)."""
    A, B = shape
    rng = np.random.default_rng(A * 7 + B)
    n = min(A, B)
    U, _ = np.linalg.qr(rng.normal(size=(A, n)))
    V, _ = np.linalg.qr(rng.normal(size=(B, n)))
    Z = (U * (0.93 ** np.arange(n))) @ V.T
    Z[A // 2, :] = 0.0
    Z[:, B // 3] = 0.0
    Zd = dev(Z.ravel())
    out = []
    for launches in (True, False):
        wA, wB, info = be.empty(A), be.empty(B), be.zeros(2)
        be.rank1(Zd, A, B, wA, wB, info=info, n_squarings=budget, launches=launches)
        out.append((host(wA).copy(), host(wB).copy(), host(info).copy()))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y), (shape, budget, np.abs(x - y).max())
    assert out[1][0][A // 2] == 0.0 and out[1][1][B // 3] == 0.0
    if budget == 30:
        assert out[1][2][0] == 1.0
        u, v, S = _svd_pair(Z)
        np.testing.assert_allclose(out[1][0], u, rtol=0, atol=1e-9)
        np.testing.assert_allclose(out[1][1], v, rtol=0, atol=1e-9)
    if budget == 2 and n > 8:
        assert out[1][2][0] == 0.0                          # not converged within two squarings: the caller must ask again


def test_small_algebra(be):
    rng = np.random.default_rng(13)
    I, M, R = 1003, 16, 7
    Y, T, t, q, b = rng.normal(size=(I, M)), rng.normal(size=(I, R)), rng.normal(size=I), rng.normal(size=M), rng.normal(size=4)
    Yd, Td = dev(Y), dev(T)
    np.testing.assert_allclose(host(be.gram_tn(Yd, dev(t))).ravel(), Y.T @ t, **RT)
    np.testing.assert_allclose(host(be.gram_tn(Td[:, :4], Td[:, :4])), T[:, :4].T @ T[:, :4], **RT)
    np.testing.assert_allclose(host(be.gram_tn(Yd, Yd)), Y.T @ Y, **RT)
    W = rng.normal(size=(257, 130))                         # wide operand: tiled 64 x 64 inside the call
    np.testing.assert_allclose(host(be.gram_tn(dev(W), dev(W[:, :70].copy()))), W.T @ W[:, :70], **RT)
    u_old = rng.normal(size=I)
    u = be.empty(I)
    du2 = be.rowdot(Yd, dev(q), u, dev(u_old))
    np.testing.assert_allclose(host(u), Y @ q, **RT)
    np.testing.assert_allclose(host(du2)[0], np.sum((u_old - Y @ q) ** 2), rtol=1e-12)
    assert be.rowdot(Yd, dev(q), u, None) is None
    Ts = rng.normal(size=(3, I))
    np.testing.assert_allclose(host(be.scores_mean(dev(Ts), be.empty(I))), np.average(Ts, axis=0), rtol=1e-15)
    ssq = be.y_deflate(Yd, Td, 4, dev(b), dev(q))
    want = Y - np.outer(T[:, :4] @ b, q)
    np.testing.assert_allclose(host(Yd), want, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(host(ssq)[0], np.sum(want ** 2), rtol=1e-12)
    v = rng.normal(size=600)
    vd = dev(v)
    be.normalize(vd)
    np.testing.assert_allclose(host(vd), v / np.linalg.norm(v), rtol=1e-14)


def test_error_statuses(be):
    """C-ABI error behaviour: status codes, no exceptions from the library itself."""
    from cmtf_pls_amd import _lib
    lib = be.lib
    X = dev(np.zeros((4, 8)))
    assert lib.cmtfpls_mode0_contract_f64(X.data_ptr(), 4, 8, None, None, 0, None, 0, None) == 1        # bad argument
    u, Z = dev(np.zeros(4)), dev(np.zeros(8))
    assert lib.cmtfpls_mode0_contract_f64(X.data_ptr(), 4, 8, u.data_ptr(), Z.data_ptr(), 0, None, 0, None) == 2   # workspace
    assert b"workspace" in lib.cmtfpls_last_error()
    with pytest.raises(_lib.CmtfplsError):
        _lib.check(2, "x")


def test_full_size_properties_cfg2(be):
    """BASELINE.json configs[1] at full size (65536 x 128 x 128 f32, 4.29 GB) through
    size-independent properties: linearity of the contraction, |X'|^2 = |X|^2 - |t|^2 for a unit
    loading, and zero score after deflation with the same loading (idempotence)."""
    I, A, B = 65536, 128, 128
    g = torch.Generator(device="cuda:0").manual_seed(0)
    X = torch.randn(I, A * B, device="cuda:0", dtype=torch.float32, generator=g)
    u1 = torch.randn(I, device="cuda:0", dtype=torch.float64, generator=g)
    u2 = torch.randn(I, device="cuda:0", dtype=torch.float64, generator=g)
    Z1 = be.mode0_contract(X, u1, False).clone()
    Z2 = be.mode0_contract(X, u2, False).clone()
    Z12 = be.mode0_contract(X, u1 + u2, False)
    torch.testing.assert_close(Z12, Z1 + Z2, rtol=1e-9, atol=1e-9)
    # a sub-block against torch f64
    torch.testing.assert_close(Z1[:64], (X[:, :64].double() * u1[:, None]).sum(0), rtol=1e-9, atol=1e-8)
    wa = torch.randn(A, device="cuda:0", dtype=torch.float64, generator=g)
    wb = torch.randn(B, device="cuda:0", dtype=torch.float64, generator=g)
    wa /= wa.norm()
    wb /= wb.norm()
    t = be.score(X, A, B, wa, wb, None, be.empty(I))
    w = torch.kron(wa, wb)
    torch.testing.assert_close(t[:256], X[:256].double() @ w, rtol=1e-10, atol=1e-10)
    ssq0 = float((X[:1024].double() ** 2).sum())            # spot value for the norm bookkeeping below
    part = be.empty(be.n_partials)
    _, ssq_all = be.center(X, torch.zeros(A * B, device="cuda:0", dtype=torch.float64), False)
    ssq_new = be.deflate(X, A, B, t, wa, wb)
    tt = float((t ** 2).sum())
    assert abs(float(ssq_new) - (float(ssq_all) - tt)) / float(ssq_all) < 1e-6     # f32 storage rounding
    t2 = be.score(X, A, B, wa, wb, None, be.empty(I))
    assert float(t2.abs().max()) < 1e-4 * float(t.abs().max())
    assert ssq0 > 0 and part.numel() == be.n_partials


@pytest.mark.parametrize("dims", [(9, 8, 7), (5, 4, 3, 2), (38, 12, 10), (3, 70, 5)])
def test_rank1_tensor_matches_oracle(be, dims):
    """Order-3/4 cross-covariance tensor: same restatement of tensorly's parafac as the oracle
    (value-level parity with tensorly itself is unpinned)."""
    rng = np.random.default_rng(14)
    Z = rng.normal(size=dims)
    for v in [rng.normal(size=d) for d in dims][:1]:
        pass
    rank1 = rng.normal(size=dims[0])
    for d in dims[1:]:
        rank1 = np.multiply.outer(rank1, rng.normal(size=d))
    Z = Z + 2.0 * rank1
    fac = be.zeros(len(dims), max(dims))
    info = be.zeros(2)
    be.rank1_tensor(dev(Z.ravel()), dims, 1e-8, fac, info=info)
    want = O.rank1_factors(Z, 1e-8)
    got = host(fac)
    for m, w in enumerate(want):
        np.testing.assert_allclose(got[m, : dims[m]], w, rtol=1e-7, atol=1e-9)
    assert host(info)[0] == 1.0


def test_kron(be):
    rng = np.random.default_rng(15)
    a, b = rng.normal(size=7), rng.normal(size=13)
    np.testing.assert_array_equal(host(be.kron(dev(a), dev(b), be.empty(91))), np.kron(a, b))


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,M", [((37, 10, 8), 4), ((100, 38, 65), 3), ((64, 1, 20), 16), ((300, 16, 16), 17),
                                     ((130, 128, 128), 16), ((257, 24, 12), 33), ((70, 8, 8), 64), ((1000, 16, 16), 16)])
def test_xcov_mfma(be, shape, M, dt, masked):
    """S = Y^T X_(0) on the f64 matrix cores against NumPy float64 (operand layout of
    v_mfma_f64_16x16x4_f64 checked with asymmetric random data, every M-tile count, ragged rows/cols)."""
    I = shape[0]
    x = make_x(shape, dt, nan_frac=0.3 if masked else 0.0, seed=16)
    y = np.random.default_rng(17).normal(size=(I, M))
    S = be.xcov(dev(x, TDT[dt]), dev(y), masked)
    want = y.T @ (np.nan_to_num(x) if masked else x)
    np.testing.assert_allclose(host(S), want, rtol=1e-11, atol=1e-10)


def test_xcov_identities_and_quadform(be):
    """The two identities the xcov loop rests on: einsum(X, Y q) == sum_m q_m S_m and
    Y^T (X w) == S_(0) w; and the quadratic form |Y q - Y q_old|^2."""
    rng = np.random.default_rng(18)
    I, A, B, M = 500, 12, 8, 6
    x, y, q = rng.normal(size=(I, A * B)), rng.normal(size=(I, M)), rng.normal(size=M)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    X, Y = dev(x), dev(y)
    S = be.xcov(X, Y, False)
    Z_direct = be.mode0_contract(X, dev(y @ q), False).clone()
    Z_xcov = be.mode0_contract(S, dev(q), False)
    np.testing.assert_allclose(host(Z_xcov), host(Z_direct), rtol=1e-10, atol=1e-9)
    t = be.score(X, A, B, dev(wa), dev(wb), None, be.empty(I))
    q_direct = host(be.gram_tn(Y, t)).ravel()
    q_xcov = host(be.score(S, A, B, dev(wa), dev(wb), None, be.empty(M)))
    np.testing.assert_allclose(q_xcov, q_direct, rtol=1e-10, atol=1e-9)
    q2 = rng.normal(size=M)
    G = be.gram_tn(Y, Y)
    out = be.quadform(G, dev(q), dev(q2), be.empty(1))
    np.testing.assert_allclose(host(out)[0], np.sum((y @ q - y @ q2) ** 2), rtol=1e-11)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape,R", [((37, 10, 8), 3), ((100, 38, 65), 8), ((64, 1, 20), 5), ((33, 1, 7), 2),
                                     ((50, 128, 128), 10), ((29, 5, 4), 16), ((70, 12, 8), 17), ((300, 16, 16), 32),
                                     # I % 16 == 0 and 512-byte multiples of columns: the coalesced LDS-tile form (1 and 2 component tiles,
                                     # several chunks per row, a matrix block, KronWalk carries)
                                     ((64, 128, 128), 10), ((48, 16, 16), 20), ((32, 1, 512), 3), ((80, 24, 16), 16), ((16, 256, 256), 10),
                                     # round 3, one case per form and instance: k-row on 4x4x4 MFMAs (f32, R <= 12: 1 / 2 / 3 component
                                     # groups; 128- and 64-column passes, two passes at 256 columns, ragged sample count), k-row on
                                     # 16x16x4 (R = 13-16, and every f64 case), j-block (A % 32 != 0 at 64 columns; 48 columns), tile (rest)
                                     ((37, 128, 128), 3), ((21, 32, 128), 7), ((19, 16, 256), 12), ((23, 32, 64), 10), ((9, 64, 64), 4),
                                     ((41, 128, 128), 14), ((17, 16, 384), 16), ((13, 32, 64), 15),
                                     ((11, 48, 64), 9), ((27, 16, 48), 6), ((7, 80, 32), 5)])
def test_mttkrp_mfma(be, shape, R, dt):
    """M = X_(0) (WA (.) WB) on the f64 matrix cores, Khatri-Rao operand formed in LDS, against NumPy
    (asymmetric random data: checks the A/B/D lane maps; ragged rows, odd B, 1 and 2 component tiles)."""
    I, A, B = shape
    x = make_x(shape, dt, seed=19)
    rng = np.random.default_rng(20)
    WA, WB = rng.normal(size=(A, R)), rng.normal(size=(B, R))
    out = be.mttkrp(dev(x, TDT[dt]), A, B, dev(WA), dev(WB), be.empty(I, R))
    W = (WA[:, None, :] * WB[None, :, :]).reshape(A * B, R)
    np.testing.assert_allclose(host(out), x @ W, rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize("shape", [(64, 64, 64), (8, 256, 256)])
@pytest.mark.parametrize("where", ["bit31", "wrap32"])
def test_sweeps_do_not_depend_on_address_bits(be, shape, where):
    """Every X sweep gives bit-identical results when X sits where the low 32 address bits flip sign
    ("bit31": X straddles an address with low word 0x80000000) or carry ("wrap32": X straddles a
    multiple of 2^32).  Regression: a scalar row base rebuilt from two 32-bit halves once sign-extended
    its low half, which faulted for whichever allocations had bit 31 set."""
    I, A, B = shape
    P = A * B
    nbytes = I * P * 4
    pool = torch.empty((1 << 32) + (1 << 29), dtype=torch.uint8, device="cuda:0")      # contains a 2^32 boundary
    base = pool.data_ptr()
    target = (1 << 31) if where == "bit31" else 0
    # first address >= base whose low word is `target`, then step back half of X so that X straddles it
    off = (target - base) % (1 << 32) - (nbytes // 2 // 256) * 256
    if off < 0:
        off += 1 << 32
    assert 0 <= off and off + nbytes <= pool.numel()
    Xs = pool[off:off + nbytes].view(torch.float32).view(I, P)
    lo, hi = Xs.data_ptr(), Xs.data_ptr() + nbytes - 1
    assert (lo >> 31) != (hi >> 31)                                                      # the boundary is inside X
    x = make_x(shape, "f32", seed=77)
    rng = np.random.default_rng(78)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    wa, wb = dev(wa / np.linalg.norm(wa)), dev(wb / np.linalg.norm(wb))
    u = dev(rng.normal(size=I))
    Xn = dev(x, torch.float32)                                                           # an ordinary allocation
    Xs.copy_(Xn)
    outs = []
    for X in (Xn, Xs):
        Z = be.mode0_contract(X, u, False).clone()
        t1 = be.empty(I)
        be.score(X, A, B, wa, wb, None, t1)
        t2 = be.empty(I)
        ssq = be.score_deflate(X, A, B, wa, wb, None, t2)
        assert ssq is not None
        ssq = ssq.clone()
        ssq2 = be.deflate(X, A, B, t1, wa, wb).clone()
        outs.append((Z, t1, t2, ssq, ssq2, X.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    np.testing.assert_allclose(host(outs[0][1]), O.score_contract(x.reshape(I, A, B), [host(wa), host(wb)]), **RT)


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,M", [((37, 10, 8), 4), ((300, 16, 16), 16), ((1000, 16, 16), 17), ((50, 128, 128), 33),
                                     ((129, 4, 64), 64), ((5000, 2, 4), 1)])
def test_mode0_contract_yq(be, shape, M, dt, masked):
    """Contraction with u = Y q formed inside the kernel (tpls.py:80-83 with tpls.py:102 folded in)."""
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.2 if masked else 0.0, seed=91)
    rng = np.random.default_rng(92)
    y, q = rng.normal(size=(I, M)), rng.normal(size=M)
    Z = be.mode0_contract_yq(dev(x, TDT[dt]), dev(y), dev(q), masked, out=be.empty(A * B))
    assert Z is not None
    u = y @ q
    x0 = np.nan_to_num(x) if masked else x
    want = x0.T @ u
    scale = np.abs(x0).T @ np.abs(u) + 1e-300
    assert np.max(np.abs(host(Z) - want) / scale) < 1e-13
    # strided Y (a column block of a wider matrix), as the engine may pass it
    ywide = dev(np.concatenate([y, rng.normal(size=(I, 3))], axis=1))
    Z2 = be.mode0_contract_yq(dev(x, TDT[dt]), ywide[:, :M], dev(q), masked, out=be.empty(A * B))
    assert torch.equal(Z2, Z)


def test_mode0_contract_yq_unsupported_shapes(be):
    rng = np.random.default_rng(93)
    x = dev(rng.normal(size=(40, 7 * 9)), torch.float32)              # P % 4 != 0: scalar shape
    assert be.mode0_contract_yq(x, dev(rng.normal(size=(40, 3))), dev(rng.normal(size=3)), False, out=be.empty(63)) is None
    x = dev(rng.normal(size=(40, 64)), torch.float32)                 # more than 64 responses
    assert be.mode0_contract_yq(x, dev(rng.normal(size=(40, 65))), dev(rng.normal(size=65)), False, out=be.empty(64)) is None


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,M", [((37, 10, 8), 4), ((100, 38, 65), 3), ((300, 16, 16), 16), ((3000, 16, 16), 17),
                                     ((50, 128, 128), 64), ((64, 1, 20), 1)])
def test_score_gram_and_q_update(be, shape, M, dt, masked):
    """score + partial sums of Y^T t (tpls.py:92-100), then the one-launch Y-side update (tpls.py:100-103)."""
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.2 if masked else 0.0, seed=94)
    rng = np.random.default_rng(95)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    y = rng.normal(size=(I, M))
    X, Y = dev(x, TDT[dt]), dev(y)
    rowcnt = dev((~np.isnan(x)).sum(1).astype(float)) if masked else None
    t_ref = be.score(X, A, B, dev(wa), dev(wb), rowcnt, be.empty(I))
    t = be.empty(I)
    qpart = be.empty(be.n_partials * M)
    assert be.score_gram(X, A, B, dev(wa), dev(wb), rowcnt, t, Y, qpart) is not None
    assert torch.equal(t, t_ref)                                        # the score itself is unchanged, whatever the batch size
    if not masked and dt == "f64" and I <= 64 and A * B >= 8192:        # the rows of a cross-covariance S: the workgroup-per-row kernel
        ts = be.score_s(X, A, B, dev(wa), dev(wb), be.empty(I))
        np.testing.assert_allclose(host(ts), host(t_ref), rtol=1e-13, atol=1e-13 * float(t_ref.abs().max()))   # (another summation order)
    th = host(t)
    want_q = y.T @ th
    scale = np.abs(y).T @ np.abs(th) + 1e-300
    got_raw = host(qpart).reshape(be.n_partials, M).sum(0)
    assert np.max(np.abs(got_raw - want_q) / scale) < 1e-13
    # q_update: sum + normalise + quadratic form, all at once and in the two halves a sharded fit uses
    G = y.T @ y
    q_prev = rng.normal(size=M)
    q_prev /= np.linalg.norm(q_prev)
    q1, du1 = be.empty(M), be.empty(1)
    be.q_update(q1, qpart, normalize=True, G=dev(G), q_prev=dev(q_prev), du2=du1)
    qn = want_q / np.linalg.norm(want_q)
    np.testing.assert_allclose(host(q1), qn, rtol=1e-11, atol=1e-13)
    d = host(q1) - q_prev
    np.testing.assert_allclose(host(du1)[0], d @ G @ d, rtol=1e-11)
    np.testing.assert_allclose(host(du1)[0], np.sum((y @ host(q1) - y @ q_prev) ** 2), rtol=1e-9)   # = |du|^2 (tpls.py:103)
    q2, du2 = be.empty(M), be.empty(1)
    be.q_update(q2, qpart, normalize=False)
    np.testing.assert_allclose(host(q2), want_q, rtol=0, atol=1e-13 * np.max(scale))
    be.q_update(q2, None, normalize=True, G=dev(G), q_prev=dev(q_prev), du2=du2)
    assert torch.equal(q2, q1) and torch.equal(du2, du1)


def test_score_gram_more_than_64_responses_is_unsupported(be):
    rng = np.random.default_rng(96)
    X = dev(rng.normal(size=(20, 32)), torch.float32)
    Y = dev(rng.normal(size=(20, 65)))
    assert be.score_gram(X, 4, 8, dev(rng.normal(size=4)), dev(rng.normal(size=8)), None, be.empty(20), Y, be.empty(be.n_partials * 65)) is None


@pytest.mark.parametrize("I,A,B", [(270000, 128, 128), (70000, 256, 256), (2_200_000, 2, 4)])
def test_mode0_contract_yq_row_chunks(be, I, A, B):
    """Workgroups whose row range exceeds one LDS chunk of u = Y q (2048 rows): the wide kernel at
    270000 x 16384 (2110 rows per workgroup) and the narrow kernel at 2.2M x 8; and 70000 x 65536, where
    the 32 column tiles make the entry form u once up front instead.  Data and the f64 reference are
    formed on the device (the tensors are 18 GB / 18 GB / 70 MB)."""
    P, M = A * B, 16
    g = torch.Generator(device="cuda:0").manual_seed(123)
    X = torch.randn(I, P, device="cuda:0", dtype=torch.float32, generator=g)
    Y = torch.randn(I, M, device="cuda:0", dtype=torch.float64, generator=g)
    q = torch.randn(M, device="cuda:0", dtype=torch.float64, generator=g)
    Z = be.mode0_contract_yq(X, Y, q, False, out=be.empty(P))
    assert Z is not None
    u = Y @ q
    want = torch.zeros(P, device="cuda:0", dtype=torch.float64)
    scale = torch.zeros(P, device="cuda:0", dtype=torch.float64)
    step = 4096 if P > 1024 else 262144
    for r in range(0, I, step):
        xb = X[r:r + step].double()
        want += xb.t() @ u[r:r + step]
        scale += xb.abs().t() @ u[r:r + step].abs()
    assert float(((Z - want).abs() / scale).max()) < 1e-13
    assert torch.equal(be.mode0_contract(X, u, False), Z) or float(((be.mode0_contract(X, u, False) - Z).abs() / scale).max()) < 1e-15


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,M", [((37, 10, 8), 4), ((300, 16, 16), 16), ((1000, 16, 64), 17), ((50, 128, 128), 33),
                                     ((2500, 2, 4), 3), ((40, 256, 128), 5),         # 16 column tiles (u formed up front)
                                     # I >= 512 and long rows: the workgroup-per-row-segment form (KC; not KC; 4 segments; ragged)
                                     ((700, 128, 128), 16), ((515, 50, 100), 3), ((513, 256, 256), 32), ((600, 33, 132), 2)])
def test_deflate_contract_yq(be, shape, M, dt, masked):
    """tpls.py:109 fused with tpls.py:80-83 of the next component: X bit-identical to deflate, Z equal to
    mode0_contract_yq on the deflated X up to the summation order over rows; the sum of squares equals the
    deflate kernel's to rounding."""
    I, A, B = shape
    x = make_x(shape, dt, nan_frac=0.2 if masked else 0.0, seed=97)
    rng = np.random.default_rng(98)
    wa, wb = rng.normal(size=A), rng.normal(size=B)
    wa, wb = dev(wa / np.linalg.norm(wa)), dev(wb / np.linalg.norm(wb))
    t, y, q = dev(rng.normal(size=I)), dev(rng.normal(size=(I, M))), dev(rng.normal(size=M))
    X1, X2 = dev(x, TDT[dt]), dev(x, TDT[dt])
    Z1 = be.empty(A * B)
    ssq1 = be.deflate_contract_yq(X1, A, B, t, wa, wb, y, q, masked, out=Z1)
    assert ssq1 is not None
    ssq2 = be.deflate(X2, A, B, t, wa, wb)
    Z2 = be.mode0_contract_yq(X2, y, q, masked, out=be.empty(A * B))
    assert torch.equal(X1.view(torch.uint8), X2.view(torch.uint8))             # same bits, NaNs included
    assert torch.equal(torch.isnan(Z1), torch.isnan(Z2))
    np.testing.assert_allclose(np.nan_to_num(host(Z1)), np.nan_to_num(host(Z2)), rtol=1e-11, atol=1e-12 * float(np.nanmax(np.abs(host(Z2))) + 1e-300))
    np.testing.assert_allclose(host(ssq1)[0], host(ssq2)[0], rtol=1e-12)
    want = x - np.outer(host(t), np.kron(host(wa), host(wb)))
    if dt == "f32":
        want = want.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(np.nan_to_num(host(X1).astype(np.float64)), np.nan_to_num(want), rtol=3e-7 if dt == "f32" else 1e-13, atol=1e-10)


def test_deflate_contract_yq_unsupported(be):
    rng = np.random.default_rng(99)
    X = dev(rng.normal(size=(40, 7 * 9)), torch.float32)
    r = be.deflate_contract_yq(X, 7, 9, dev(rng.normal(size=40)), dev(rng.normal(size=7)), dev(rng.normal(size=9)),
                               dev(rng.normal(size=(40, 3))), dev(rng.normal(size=3)), False, out=be.empty(63))
    assert r is None


@pytest.mark.parametrize("shape", [(8, 1, 12000), (6, 4000, 4), (5, 96, 680)])
def test_deflate_rows_with_large_loadings(be, shape):
    """Workgroup-per-row deflation where the staged loadings take most of the LDS (the one-row-per-CU
    padding must then be dropped, not overflow the 160 KB)."""
    I, A, B = shape
    x = make_x(shape, "f32", seed=101)
    rng = np.random.default_rng(102)
    wa, wb, t = rng.normal(size=A), rng.normal(size=B), rng.normal(size=I)
    X = dev(x, torch.float32)
    ssq = be.deflate(X, A, B, dev(t), dev(wa), dev(wb))
    want = (x - np.outer(t, np.kron(wa, wb))).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(host(X).astype(np.float64), want, rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(host(ssq)[0], np.sum(host(X).astype(np.float64) ** 2), rtol=1e-11)


def test_score_deflate_unsupported_when_lds_cannot_hold_loadings_and_parked_row(be):
    """Rows of > 16 K elements park half of the row in LDS (128 KB); with 40 KB of loadings on top the
    fused kernel must decline (the caller then runs score + deflate), not fail at launch."""
    rng = np.random.default_rng(103)
    I, A, B = 3, 4, 5000                                        # P = 20000, loadings 40 KB
    X = dev(rng.normal(size=(I, A * B)), torch.float32)
    assert be.score_deflate(X, A, B, dev(rng.normal(size=A)), dev(rng.normal(size=B)), None, be.empty(I)) is None
