"""CPU oracle for the tPLS / ctPLS NIPALS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cmtf_pls_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

Pinning status (see DESIGN.md section "Oracle"):
  * ``masked_mode0_contract`` / ``masked_score`` are pinned against the real
    reference (``/root/reference/cmtf_pls/missingvals.py`` imports with NumPy
    alone) through ``tests/golden/missingvals_*.npz``.
  * the fit loop itself calls ``tensorly==0.9.0`` (``parafac``,
    ``multi_mode_dot``, ``outer``, ``khatri_rao``, ``fold``), which is not
    installed in the build container and cannot be fetched, so
    ``/root/reference/cmtf_pls/tpls.py`` and ``cmtf.py`` are not importable.
    Their arithmetic is restated here from the reference source and from
    tensorly's published algorithm, and is pinned by the reference's own
    known-answer / property tests (ported, seeded, in ``tests/test_oracle_*``).
    Value-level output of ``parafac`` (sign convention, ALS stopping step for
    order >= 3 cross-covariance tensors) is PARITY UNPINNED.
"""
from .nipals_oracle import (  # noqa: F401
    OracleFit,
    calc_r2x,
    cp_factors_to_tensor,
    fit_ctpls,
    fit_tpls,
    masked_mode0_contract,
    masked_score,
    mode0_contract,
    nipals_inner_loop,
    predict,
    rank1_factors,
    score_contract,
    transform,
    reconstruct,
)
from .synthetic_oracle import SyntheticCP, import_synthetic, make_synthetic_test  # noqa: F401
