#!/usr/bin/env python3
"""Leave-one-out Q2Y (validate.get_q2y, validate.py:24-37): the workgroup-per-fold kernels against one refit per fold on the regular
engine.  Small shapes take cmtfpls_loo_tpls_f64 (vectors in LDS); min(J, K) > 64 takes cmtfpls_loo_xcov_f64 (round 4).
Usage: python tools/loo_time.py [small|big|all] [--refits N]   (big: (512,128,128) R = 4; the refit loop is timed on N folds and scaled)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.validate import get_q2y, loo_predictions
from cmtf_pls_amd.synthetic import import_synthetic

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
n_refits = int(sys.argv[sys.argv.index("--refits") + 1]) if "--refits" in sys.argv else 32
if which in ("small", "all"):
    for shape, M, R in (((200, 10, 8), 4, 3), ((100, 38, 65), 3, 4)):
        x, y, _ = import_synthetic(shape, M, R, error=0.1, seed=3)
        m = tPLS(R)
        m.fit(x, y)
        for dev_folds in (False, True):
            get_q2y(m, device_folds=dev_folds) if dev_folds else None      # warm (module load, workspaces)
            t0 = time.perf_counter()
            q = get_q2y(m, device_folds=dev_folds)
            dt = time.perf_counter() - t0
            print(shape, m.q2y_report_["form"], "q2y", round(q, 8), "LOO seconds", round(dt, 4), "per fold ms", round(dt / shape[0] * 1e3, 4), flush=True)
if which in ("big", "all"):
    for shape, M, R in (((512, 128, 128), 16, 4), ((256, 96, 160), 8, 3)):
        x, y, _ = import_synthetic(shape, M, R + 2, error=0.3, seed=3)
        m = tPLS(R)
        m.fit(x, y)
        get_q2y(m)                                                           # warm
        t0 = time.perf_counter()
        q = get_q2y(m)
        dt = time.perf_counter() - t0
        rep = dict(m.q2y_report_)
        pred = loo_predictions(m)
        # the refit loop on a sample of folds (every fold costs the same): literal tPLS.fit + predict without sample i
        idx = np.linspace(0, shape[0] - 1, n_refits).astype(int)
        r = tPLS(R)
        keep = np.ones(shape[0], dtype=bool)
        keep[0] = False
        r.fit(x[keep], y[keep])                                              # warm
        worst, t1 = 0.0, time.perf_counter()
        for i in idx:
            keep[:] = True
            keep[i] = False
            r.fit(x[keep], y[keep])
            want = r.predict(x[i:i + 1]).reshape(-1)
            worst = max(worst, float(np.abs(pred[i].reshape(-1) - want).max() / max(1.0, np.abs(want).max())))
        per_refit = (time.perf_counter() - t1) / len(idx)
        print(f"{shape} M={M} R={R}: {rep['form']}: q2y {q:.10f}, {dt:.3f} s for {shape[0]} folds ({dt / shape[0] * 1e3:.3f} ms per fold, "
              f"{rep['n_iter_total']} inner iterations in all) | one refit per fold on the regular engine: {per_refit * 1e3:.2f} ms per fold "
              f"(timed on {len(idx)} folds) = {per_refit * shape[0]:.2f} s for all | speed-up {per_refit * shape[0] / dt:.1f}x | "
              f"max |prediction - literal refit| = {worst:.1e} (relative)", flush=True)
