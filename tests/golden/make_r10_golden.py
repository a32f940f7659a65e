#!/usr/bin/env python3
"""Oracle outputs of the R = 10 replicas of BASELINE configs[1..4] -> tests/golden/oracle_r10_*.npz.

ORACLE-generated regression vectors (they do not pin the oracle): the float64 oracle needs 1-4 minutes
per configuration on 8 cores (its literal R2Y re-projects the training set once per component, tpls.py:118-120), too
long to repeat inside the GPU suite, so it is run here once and tests/test_gpu_r10_parity.py compares the HIP fits with
the stored outputs.  Inputs are NOT stored: `inputs(name)` below regenerates them from their seeds and the fixture
carries float64 checksums of them, which the test verifies before comparing.

Run from the repository root:  python tests/golden/make_r10_golden.py [name ...]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R = 10
NAMES = ("cfg2", "cfg5", "cfg3", "cfg4")


def _f32(a):
    return a.astype(np.float32).astype(np.float64)


def inputs(name):
    """(blocks, y, coupled): f32-representable float64 arrays, from seeds only."""
    import oracle as O
    if name == "cfg2":                                   # BASELINE configs[1] with 1/16 of the rows
        x, y, _ = O.import_synthetic((4096, 128, 128), 16, 10, error=0.1, seed=215)
        return [_f32(x)], _f32(y), False
    if name == "cfg5":                                   # BASELINE configs[4] with 1/256 of the rows
        x, y, _ = O.import_synthetic((1024, 256, 256), 32, 10, error=0.1, seed=215)
        return [_f32(x)], _f32(y), False
    if name == "cfg3":                                   # BASELINE configs[2]: tensor + 512-column matrix, coupled
        x, y, cp = O.import_synthetic((1024, 128, 128), 16, 10, error=0.1, seed=215)
        xm = cp.factors[0] @ np.random.default_rng(216).normal(size=(512, 10)).T + 0.1 * np.random.default_rng(5).normal(size=(1024, 512))
        return [_f32(x), _f32(xm)], _f32(y), True
    if name == "cfg4":                                   # BASELINE configs[3]: 30 % NaN
        x, y, _ = O.import_synthetic((1024, 128, 128), 16, 10, error=0.1, seed=215)
        x, y = _f32(x), _f32(y)
        x[np.random.default_rng(217).random(x.shape) < 0.3] = np.nan
        return [x], y, False
    raise KeyError(name)


def checksums(blocks, y):
    return np.array([float(np.nansum(b)) for b in blocks] + [float(np.nansum(np.abs(b))) for b in blocks] + [float(y.sum())])


def path(name):
    return os.path.join(HERE, f"oracle_r10_{name}.npz")


def load(name):
    """OracleFit of the fixture (the fields the parity tests read) or None when it has not been generated."""
    import oracle as O
    if not os.path.exists(path(name)):
        return None
    d = np.load(path(name))
    nb = int(d["n_blocks"])
    loadings = [[d[f"L{b}_{m}"] for m in range(int(d[f"n_modes{b}"]))] for b in range(nb)]
    fit = O.OracleFit(coupled=bool(d["coupled"]), n_components=R, block_shapes=[tuple(d[f"shape{b}"]) for b in range(nb)],
                      y_shape=tuple(d["y_shape"]), T=d["T"], loadings=loadings, U=d["U"], Q=d["Q"], coef=d["coef"],
                      r2x=[d[f"r2x{b}"] for b in range(nb)], r2y=d["r2y"], x_means=[d[f"xmean{b}"] for b in range(nb)],
                      y_mean=d["y_mean"], has_miss=[bool(v) for v in d["has_miss"]], n_iter=[int(v) for v in d["n_iter"]])
    return fit, d["checksums"], d["transform_head"]


def main():
    import oracle as O
    for name in (sys.argv[1:] or NAMES):
        blocks, y, coupled = inputs(name)
        t0 = time.time()
        fit = O.fit_ctpls(blocks, y, R) if coupled else O.fit_tpls(blocks[0], y, R)
        head = O.transform(fit, [b[:256] for b in blocks] if coupled else blocks[0][:256])
        out = {"n_blocks": len(blocks), "coupled": coupled, "y_shape": np.array(fit.y_shape), "T": fit.T, "U": fit.U, "Q": fit.Q,
               "coef": fit.coef, "r2y": fit.r2y, "y_mean": fit.y_mean, "has_miss": np.array(fit.has_miss), "n_iter": np.array(fit.n_iter),
               "checksums": checksums(blocks, y), "transform_head": head}
        for b in range(len(blocks)):
            out[f"shape{b}"] = np.array(fit.block_shapes[b])
            out[f"n_modes{b}"] = len(fit.loadings[b])
            out[f"r2x{b}"] = fit.r2x[b]
            out[f"xmean{b}"] = fit.x_means[b]
            for m, L in enumerate(fit.loadings[b]):
                out[f"L{b}_{m}"] = L
        np.savez_compressed(path(name), **out)
        print(f"{name}: n_iter {fit.n_iter}  {time.time() - t0:.0f} s  -> {path(name)} ({os.path.getsize(path(name)) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    main()
