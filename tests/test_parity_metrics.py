"""CPU: the two error metrics of tests/parity_metrics.py mean what DESIGN section 2 says they mean, and the R = 10 golden fixtures
load and describe the inputs their generator regenerates."""
import numpy as np

from golden import make_r10_golden as G10
from parity_metrics import column_errors, paired_sign


def test_normwise_and_elementwise_metrics():
    want = np.array([[100.0, 1.0], [1e-6, -2.0], [50.0, 0.5]])
    got = want.copy()
    got[1, 0] += 1e-4                        # a tiny entry next to entries of order 100: absolute error 1e-4
    got[2, 1] *= 1 + 3e-6                    # a relative perturbation of a mid-size entry
    e = column_errors(got, want)
    # normwise: |d| / (|want| + max|column|): 1e-4 / (1e-6 + 100) = 1e-6; 1.5e-6 / (0.5 + 2) = 6e-7
    np.testing.assert_allclose(e["normwise"], [1e-4 / (1e-6 + 100.0), 1.5e-6 / 2.5], rtol=1e-6)
    # element-wise: the same errors relative to the entry itself: 1e-4 / 1e-6 = 100 (unbounded for cancelling entries), 3e-6
    np.testing.assert_allclose(e["elementwise"], [100.0, 3e-6], rtol=1e-4)
    np.testing.assert_allclose(e["frac_within_1e5"], [2 / 3, 1.0])
    np.testing.assert_allclose(e["worst_rel_size"], [1e-8, 0.25])
    assert column_errors(want, want)["elementwise"].max() == 0.0          # 0 / 0 entries count as exact
    np.testing.assert_array_equal(paired_sign(np.array([[1.0, -1.0], [2.0, -2.0]]), np.array([[1.0, 1.0], [2.0, 2.0]])), [1.0, -1.0])


def test_r10_fixtures_match_their_regenerated_inputs():
    """The smallest configuration end to end on the CPU: the fixture's checksums are those of the inputs `inputs(name)`
    regenerates here, and its first component is what the oracle computes from them (one component: seconds)."""
    import oracle as O
    for name in G10.NAMES:
        assert G10.load(name) is not None, f"tests/golden/oracle_r10_{name}.npz missing"
    blocks, y, coupled = G10.inputs("cfg4")
    fit, sums, head = G10.load("cfg4")
    np.testing.assert_allclose(G10.checksums(blocks, y), sums, rtol=1e-12)
    one = O.fit_tpls(blocks[0], y, 1)
    assert one.n_iter == fit.n_iter[:1]
    np.testing.assert_allclose(one.T[:, 0], fit.T[:, 0], rtol=1e-9, atol=1e-9 * np.abs(fit.T[:, 0]).max())
    np.testing.assert_allclose(one.r2x[0][0], fit.r2x[0][0], rtol=1e-10)
    assert head.shape == (256, 10) and fit.coef.shape == (10, 10)
