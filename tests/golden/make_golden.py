#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

Two kinds of fixture, kept apart by file name:

* ``ref_missingvals.npz`` -- REFERENCE-GENERATED.  ``/root/reference/cmtf_pls/missingvals.py``
  needs only NumPy, so it is imported as-is and its two functions (``miss_tensordot``
  missingvals.py:7, ``miss_mmodedot`` missingvals.py:23) are run on seeded inputs, including the
  edge cases the reference defines (a column with no observation -> 0, a row with no
  observation -> NaN).  Inputs and outputs are stored; nothing of the reference's text is.

* ``oracle_*.npz`` -- ORACLE-GENERATED regression vectors.  ``tpls.py`` / ``cmtf.py`` import
  tensorly 0.9.0, which is absent offline, so the reference's fit loop cannot be executed here;
  these files freeze the oracle's own output (inputs reproducible by seed) so that the GPU box,
  where neither the reference nor tensorly exists, checks the HIP path against fixed numbers and
  against the live oracle.  They do NOT pin the oracle to the reference.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def reference_missingvals():
    sys.path.insert(0, "/root/reference")
    from cmtf_pls.missingvals import miss_mmodedot, miss_tensordot  # noqa: E402  (NumPy only)

    out = {}
    rng = np.random.default_rng(20240601)
    cases = {"a": (12, 5, 4, 3, 0.15), "b": (40, 9, 8, 0.30), "c": (16, 24, 0.25), "d": (9, 4, 3, 2, 2, 0.2)}
    for tag, spec in cases.items():
        shape, frac = spec[:-1], spec[-1]
        X = rng.normal(size=shape)
        X[rng.random(shape) < frac] = np.nan
        if tag == "a":
            X[:, 1, 2, 0] = np.nan      # a column with no observation  -> 0   (missingvals.py:18)
            X[3] = np.nan               # a row with no observation     -> NaN (missingvals.py:37)
        u = rng.normal(size=shape[0])
        facs = [rng.normal(size=d) for d in shape[1:]]
        out[f"{tag}_X"] = X
        out[f"{tag}_u"] = u
        for m, f in enumerate(facs):
            out[f"{tag}_w{m}"] = f
        out[f"{tag}_tensordot"] = miss_tensordot(X, u, np.isnan(X))
        with np.errstate(all="ignore"):
            out[f"{tag}_mmodedot"] = miss_mmodedot(X, facs, np.isnan(X))
    np.savez_compressed(os.path.join(HERE, "ref_missingvals.npz"), **out)
    print("ref_missingvals.npz:", sorted(out)[:6], "...")


def decode_x128(code):
    """int8 codes -> float64 X of ref_missingvals_128.npz: value = code / 16, code -128 = missing."""
    x = code.astype(np.float64) / 16.0
    x[code == -128] = np.nan
    return x


def reference_missingvals_bench_shape():
    """The same two reference functions at the benchmark's trailing shape (J = K = 128, the column /
    row lengths every BASELINE config but [4] uses), 30 % NaN as BASELINE configs[3].  X is stored as int8
    codes (value = code / 16, exactly representable in f32 and f64; -128 = NaN) to keep the fixture small;
    u and the loadings are stored as float64; outputs are the reference's."""
    sys.path.insert(0, "/root/reference")
    from cmtf_pls.missingvals import miss_mmodedot, miss_tensordot  # noqa: E402  (NumPy only)

    rng = np.random.default_rng(20241004)
    shape = (24, 128, 128)
    code = rng.integers(-127, 128, size=shape).astype(np.int8)
    code[rng.random(shape) < 0.30] = -128
    code[:, 5, 77] = -128          # a column with no observation -> 0   (missingvals.py:18)
    code[11] = -128                # a row with no observation    -> NaN (missingvals.py:37)
    X = decode_x128(code)
    u = rng.normal(size=shape[0])
    facs = [rng.normal(size=d) for d in shape[1:]]
    with np.errstate(all="ignore"):
        out = dict(code=code, u=u, w0=facs[0], w1=facs[1], tensordot=miss_tensordot(X, u, np.isnan(X)),
                   mmodedot=miss_mmodedot(X, facs, np.isnan(X)))
    np.savez_compressed(os.path.join(HERE, "ref_missingvals_128.npz"), **out)
    print("ref_missingvals_128.npz written")


def _pack(fit, prefix=""):
    d = {prefix + "T": fit.T, prefix + "U": fit.U, prefix + "Q": fit.Q, prefix + "coef": fit.coef,
         prefix + "r2y": fit.r2y, prefix + "y_mean": fit.y_mean, prefix + "n_iter": np.array(fit.n_iter)}
    for b in range(len(fit.loadings)):
        d[f"{prefix}r2x{b}"] = fit.r2x[b]
        d[f"{prefix}x_mean{b}"] = fit.x_means[b]
        for m, L in enumerate(fit.loadings[b]):
            d[f"{prefix}W{b}_{m}"] = L
    return d


def oracle_regression():
    import oracle as O

    # BASELINE.json configs[0]: synthetic (200,10,8), M=4, R=3, noise 0 and 0.1
    for tag, err in (("cfg1", 0.0), ("cfg1_noise", 0.1)):
        x, y, _ = O.import_synthetic((200, 10, 8), 4, 3, error=err)
        fit = O.fit_tpls(x, y, 3)
        xt, yt, _ = O.make_synthetic_test(O.import_synthetic((200, 10, 8), 4, 3, error=err)[2], 17, err, seed=7)
        d = _pack(fit)
        d["x_test"], d["y_test"] = xt, yt
        d["pred_test"] = O.predict(fit, xt)
        d["scores_test"] = O.transform(fit, xt)
        np.savez_compressed(os.path.join(HERE, f"oracle_tpls_{tag}.npz"), **d)

    # fp32-representable inputs (what the f32-storage HIP path is fed), 30 % NaN variant too
    rng = np.random.default_rng(11)
    x, y, _ = O.import_synthetic((96, 12, 16), 5, 4, error=0.1, seed=3)
    x = x.astype(np.float32).astype(np.float64)
    y = y.astype(np.float32).astype(np.float64)
    fit = O.fit_tpls(x, y, 4)
    np.savez_compressed(os.path.join(HERE, "oracle_tpls_f32in.npz"), x=x, y=y, **_pack(fit))
    xm = x.copy()
    xm[rng.random(x.shape) < 0.3] = np.nan
    fitm = O.fit_tpls(xm, y, 4)
    np.savez_compressed(os.path.join(HERE, "oracle_tpls_f32in_nan30.npz"), x=xm, y=y,
                        recon=O.reconstruct(fitm), **_pack(fitm))

    # coupled: tensor block + matrix block sharing the sample mode (BASELINE configs[2] in small)
    x3, y3, cp = O.import_synthetic((64, 8, 12), 3, 4, error=0.05, seed=5)
    xm2 = cp.factors[0] @ np.random.default_rng(216).normal(size=(20, 4)).T
    fitc = O.fit_ctpls([x3, xm2], y3, 4)
    np.savez_compressed(os.path.join(HERE, "oracle_ctpls_small.npz"), x0=x3, x1=xm2, y=y3, **_pack(fitc))
    print("oracle_*.npz written")


if __name__ == "__main__":
    reference_missingvals()
    reference_missingvals_bench_shape()
    oracle_regression()
