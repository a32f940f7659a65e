"""The oracle against (i) vectors produced by the real reference and (ii) its own frozen output.

(i)  ref_missingvals.npz was produced by importing /root/reference/cmtf_pls/missingvals.py
     (tests/golden/make_golden.py); it pins oracle.masked_mode0_contract / masked_score.
(ii) BASELINE.md section 2 records what the reference's own tpls.py printed for BASELINE.json
     configs[0] at survey time (iterations 6/35/1 as 0-based break index, R2X, R2Y to 3 decimals).
"""
import os

import numpy as np
import pytest

import oracle as O

CASES = ["a", "b", "c", "d"]


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_missingvals.npz"))


@pytest.mark.parametrize("tag", CASES)
def test_masked_contract_matches_reference(ref, tag):
    X, u = ref[f"{tag}_X"], ref[f"{tag}_u"]
    got = O.masked_mode0_contract(X, u, np.isnan(X))
    np.testing.assert_allclose(got, ref[f"{tag}_tensordot"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("tag", CASES)
def test_masked_score_matches_reference(ref, tag):
    X = ref[f"{tag}_X"]
    facs = [ref[f"{tag}_w{m}"] for m in range(X.ndim - 1)]
    got = O.masked_score(X, facs, np.isnan(X))
    want = ref[f"{tag}_mmodedot"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-12, atol=1e-13)


def test_reference_edge_cases_present(ref):
    # empty column -> exactly 0 (missingvals.py:18), empty row -> NaN (missingvals.py:37)
    assert ref["a_tensordot"][1, 2, 0] == 0.0
    assert np.isnan(ref["a_mmodedot"][3])


def test_cfg1_matches_survey_time_reference_run():
    x, y, _ = O.import_synthetic((200, 10, 8), 4, 3)
    fit = O.fit_tpls(x, y, 3)
    assert [n - 1 for n in fit.n_iter] == [6, 35, 1]
    np.testing.assert_allclose(fit.r2x[0], [0.715, 0.853, 0.940], atol=6e-4)
    np.testing.assert_allclose(fit.r2y, [0.475, 0.800, 1.000], atol=6e-4)


@pytest.mark.parametrize("name", ["oracle_tpls_cfg1", "oracle_tpls_cfg1_noise"])
def test_oracle_regression_cfg1(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    err = 0.0 if name.endswith("cfg1") else 0.1
    x, y, _ = O.import_synthetic((200, 10, 8), 4, 3, error=err)
    fit = O.fit_tpls(x, y, 3)
    np.testing.assert_allclose(fit.T, g["T"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fit.coef, g["coef"], rtol=1e-8, atol=1e-10)
    assert list(fit.n_iter) == list(g["n_iter"])
    np.testing.assert_allclose(O.predict(fit, g["x_test"]), g["pred_test"], rtol=1e-8, atol=1e-9)


def test_inner_loop_is_the_fit_trajectory():
    """nipals_inner_loop (what bench.py's cpu_baseline times) reproduces component 0 of fit_tpls."""
    x, y, _ = O.import_synthetic((60, 7, 5), 3, 2, error=0.1, seed=3)
    fit = O.fit_tpls(x, y, 1)
    xc, yc = x - x.mean(0), y - y.mean(0)
    t, w, q, u, du = O.nipals_inner_loop(xc, yc, fit.n_iter[0])
    np.testing.assert_allclose(t, fit.T[:, 0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(q, fit.Q[:, 0], rtol=1e-12, atol=1e-12)
    assert du < 1e-8


def test_oracle_matches_reference_at_benchmark_trailing_shape(golden_dir):
    """ref_missingvals_128.npz: the reference's two masked contractions at J = K = 128 with 30 % NaN
    (BASELINE configs[3]'s mask density), an empty column and an empty row (tests/golden/make_golden.py)."""
    from golden.make_golden import decode_x128
    g = np.load(os.path.join(golden_dir, "ref_missingvals_128.npz"))
    X = decode_x128(g["code"])
    np.testing.assert_allclose(O.masked_mode0_contract(X, g["u"], np.isnan(X)), g["tensordot"], rtol=1e-12, atol=1e-13)
    got, want = O.masked_score(X, [g["w0"], g["w1"]], np.isnan(X)), g["mmodedot"]
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want[11])
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-12, atol=1e-13)
    assert g["tensordot"][5, 77] == 0.0
