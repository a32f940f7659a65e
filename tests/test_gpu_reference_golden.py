"""HIP kernels against REFERENCE-GENERATED vectors (VERDICT r1 "close the parity gaps" (a)).

tests/golden/ref_missingvals*.npz hold inputs and outputs of the reference's own
cmtf_pls/missingvals.py (miss_tensordot :7-20, miss_mmodedot :23-38), produced by importing that file in the
build container (tests/golden/make_golden.py).  Here every device form of the two masked contractions
(rows a4 / a7 of SURVEY 8(a)) is run on exactly those inputs and compared with the reference's outputs --
HIP vs reference, no oracle in between.  Tolerance: f64 storage, f64 accumulation, only the summation order
differs from NumPy's: rtol 1e-11 of the column / row scale (atol 1e-11 * max|want|).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ["a", "b", "c", "d", "x128"]


@pytest.fixture(scope="module")
def be():
    from cmtf_pls_amd.backend import HipBackend
    return HipBackend("cuda:0")


def load_case(golden_dir, tag):
    """X (I, d1, ...), u, factors, reference outputs."""
    if tag == "x128":
        from golden.make_golden import decode_x128
        g = np.load(os.path.join(golden_dir, "ref_missingvals_128.npz"))
        return decode_x128(g["code"]), g["u"], [g["w0"], g["w1"]], g["tensordot"], g["mmodedot"]
    g = np.load(os.path.join(golden_dir, "ref_missingvals.npz"))
    X = g[f"{tag}_X"]
    return X, g[f"{tag}_u"], [g[f"{tag}_w{m}"] for m in range(X.ndim - 1)], g[f"{tag}_tensordot"], g[f"{tag}_mmodedot"]


def dev(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to("cuda:0")


def split(X, facs):
    """(A, B, wA, wB) of the factored loading the kernels take: wB = kron of all factors but the first."""
    if X.ndim == 2:
        return 1, X.shape[1], np.ones(1), facs[0]
    wB = facs[1]
    for f in facs[2:]:
        wB = np.kron(wB, f)
    return X.shape[1], wB.size, facs[0], wB


def close(got, want, scale_rtol=1e-11):
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=scale_rtol, atol=scale_rtol * np.abs(want[ok]).max())


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("tag", CASES)
def test_masked_contraction_kernels_match_the_reference(be, golden_dir, tag, dt):
    """miss_tensordot (missingvals.py:7-20) = masked mode-0 contraction + colscale, in its three device forms:
    plain, with u = Y q formed in the kernel, and as one row of the matrix-core cross-covariance S = Y^T X."""
    X, u, facs, want, _ = load_case(golden_dir, tag)
    if dt == "f32" and tag != "x128":
        pytest.skip("only the int8-coded fixture is exactly representable in f32 storage")
    I = X.shape[0]
    Xd = dev(X.reshape(I, -1), torch.float32 if dt == "f32" else torch.float64)
    ud = dev(u)
    _, colcnt = be.colstats(Xd)
    want = want.reshape(-1)
    # (1) the plain masked contraction
    Z = be.mode0_contract(Xd, ud, True)
    be.colscale(Z, colcnt, float(I))
    close(Z.cpu().numpy(), want)
    assert (Z.cpu().numpy()[want == 0.0] == 0.0).all()          # empty column -> exactly 0 (missingvals.py:18)
    # (2) u = Y q inside the kernel: Y = [u, 0], q = [1, 0]
    Y = torch.zeros(I, 2, dtype=torch.float64, device="cuda:0")
    Y[:, 0] = ud
    q = torch.tensor([1.0, 0.0], dtype=torch.float64, device="cuda:0")
    Z2 = torch.empty_like(Z)
    if be.mode0_contract_yq(Xd, Y, q, True, out=Z2) is not None:
        be.colscale(Z2, colcnt, float(I))
        close(Z2.cpu().numpy(), want)
    # (3) S = Y^T X_(0) on the f64 matrix cores, masked: row 0 is the same numerator
    S = be.xcov(Xd, Y, True)
    Z3 = S[0].contiguous()
    be.colscale(Z3, colcnt, float(I))
    close(Z3.cpu().numpy(), want)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("tag", CASES)
def test_masked_score_kernels_match_the_reference(be, golden_dir, tag, dt):
    """miss_mmodedot (missingvals.py:23-38) in its three device forms: score, score + Y^T t partials, and the
    fused score-and-deflate sweep (whose score output is the same quantity)."""
    X, _, facs, _, want = load_case(golden_dir, tag)
    if dt == "f32" and tag != "x128":
        pytest.skip("only the int8-coded fixture is exactly representable in f32 storage")
    I = X.shape[0]
    A, B, wA, wB = split(X, facs)
    tdt = torch.float32 if dt == "f32" else torch.float64
    Xd = dev(X.reshape(I, -1), tdt)
    rowcnt, _ = be.center(Xd, torch.zeros(A * B, dtype=torch.float64, device="cuda:0"), True)   # mean 0: X unchanged
    assert np.array_equal(rowcnt.cpu().numpy(), (~np.isnan(X.reshape(I, -1))).sum(1).astype(float))
    wAd, wBd = dev(wA), dev(wB)
    t = torch.empty(I, dtype=torch.float64, device="cuda:0")
    be.score(Xd, A, B, wAd, wBd, rowcnt, t)
    close(t.cpu().numpy(), want)
    Y = torch.ones(I, 3, dtype=torch.float64, device="cuda:0")
    qpart = torch.empty(be.n_partials * 3, dtype=torch.float64, device="cuda:0")
    t2 = torch.empty_like(t)
    assert be.score_gram(Xd, A, B, wAd, wBd, rowcnt, t2, Y, qpart) is not None
    close(t2.cpu().numpy(), want)
    Xc = Xd.clone()
    t3 = torch.empty_like(t)
    if be.score_deflate(Xc, A, B, wAd, wBd, rowcnt, t3) is not None:
        close(t3.cpu().numpy(), want)
