"""Error metrics of a product fit against an oracle fit, component by component (TEST INFRASTRUCTURE).

Two metrics, both reported for every factor column (VERDICT r2 "What's weak" #3):

* ``normwise``  max_i |got_i - want_i| / (|want_i| + max_j |want_j|): the metric the parity tests ASSERT at 1e-5 --
  an entry is compared relative to its own size plus the column's largest entry, i.e. entries far below the column
  maximum are held to an absolute error of 1e-5 * max|column|.
* ``elementwise``  max_i |got_i - want_i| / |want_i|: the plain rtol of ``np.allclose(rtol=r, atol=0)``.  It is unbounded
  for entries that cancel to ~0 (a score of 1e-7 next to scores of 1e+2), so next to the maximum the table carries the
  fraction of entries that meet 1e-5 and the size (relative to the column maximum) of the worst offender.

Loadings are compared after the per-component paired sign: both trailing-mode loadings of a component
are multiplied by the sign of <got_J, want_J>.
"""
import numpy as np


def column_errors(got, want):
    """Per column: dict(normwise, elementwise, frac_within_1e5, worst_rel_size)."""
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    if got.ndim == 1:
        got, want = got[:, None], want[:, None]
    scale = np.abs(want).max(axis=0, keepdims=True)
    diff = np.abs(got - want)
    normwise = (diff / (np.abs(want) + scale)).max(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        elem = np.where(diff == 0, 0.0, diff / np.abs(want))
    worst = elem.argmax(axis=0)
    cols = np.arange(want.shape[1])
    return {"normwise": normwise, "elementwise": elem.max(axis=0), "frac_within_1e5": (elem <= 1e-5).mean(axis=0),
            "worst_rel_size": np.abs(want)[worst, cols] / scale[0]}


def paired_sign(got_first_mode, want_first_mode):
    s = np.sign(np.sum(got_first_mode * want_first_mode, axis=0))
    s[s == 0] = 1
    return s


def fit_error_table(m, fit, block=0):
    """Rows of the error table of estimator ``m`` (cmtf_pls_amd tPLS / ctPLS) against OracleFit ``fit``:
    one row per (factor, component) with both metrics, plus the R2 / coef / n_iter differences."""
    Xf = m.X_factors if hasattr(m, "X_factors") else m.Xs_factors[block]
    loads = fit.loadings[block]
    s = paired_sign(Xf[1], loads[0]) if len(loads) == 2 else np.ones(fit.n_components)
    factors = [("T", Xf[0], fit.T)]
    for mode, L in enumerate(loads):
        factors.append((f"W{mode + 1}", Xf[1 + mode] * s, L))
    factors += [("Q", m.Y_factors[1], fit.Q), ("U", m.Y_factors[0], fit.U)]
    rows = []
    for name, got, want in factors:
        e = column_errors(got, want)
        for a in range(want.shape[1]):
            rows.append({"factor": name, "component": a, "normwise": float(e["normwise"][a]),
                         "elementwise": float(e["elementwise"][a]), "frac_within_1e5": float(e["frac_within_1e5"][a]),
                         "worst_rel_size": float(e["worst_rel_size"][a])})
    r2x = m.R2X if hasattr(m, "R2X") else m.R2Xs[block]
    extra = {"R2X_abs": np.abs(np.asarray(r2x) - fit.r2x[block]).tolist(), "R2Y_abs": np.abs(np.asarray(m.R2Y) - fit.r2y).tolist(),
             "coef_normwise": float(np.abs(m.coef_ - fit.coef).max() / np.abs(fit.coef).max()),
             "n_iter": list(m.n_iter_), "n_iter_oracle": list(fit.n_iter)}
    return rows, extra


def worst(rows, metric="normwise"):
    r = max(rows, key=lambda r: r[metric])
    return r[metric], r["factor"], r["component"]


def format_table(title, rows, extra):
    R = 1 + max(r["component"] for r in rows)
    names = sorted({r["factor"] for r in rows}, key=lambda n: ["T", "W1", "W2", "W3", "W4", "Q", "U"].index(n))
    by = {(r["factor"], r["component"]): r for r in rows}
    out = [title, "  n_iter product " + str(extra["n_iter"]) + "  oracle " + str(extra["n_iter_oracle"]),
           "  comp | " + " | ".join(f"{n:>5} normwise  elementwise (frac<=1e-5)" for n in names)]
    for a in range(R):
        cells = [f"{by[(n, a)]['normwise']:14.2e}  {by[(n, a)]['elementwise']:11.2e} ({by[(n, a)]['frac_within_1e5']:.4f})" for n in names]
        out.append(f"  {a:4d} | " + " | ".join(cells))
    out.append("  max |dR2X| %.2e   max |dR2Y| %.2e   coef normwise %.2e" % (max(extra["R2X_abs"]), max(extra["R2Y_abs"]), extra["coef_normwise"]))
    return "\n".join(out)
