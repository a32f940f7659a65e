#!/usr/bin/env python3
"""Leave-one-out Q2Y (validate.get_q2y, one refit per sample) timed on small problems."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.validate import get_q2y
from cmtf_pls_amd.synthetic import import_synthetic
for shape, M, R in (((200, 10, 8), 4, 3), ((100, 38, 65), 3, 4)):
    x, y, _ = import_synthetic(shape, M, R, error=0.1, seed=3)
    for algo in ("direct", "xcov"):
        m = tPLS(R, algorithm=algo)
        m.fit(x, y)
        for dev_folds in (False, True):
            get_q2y(m, device_folds=dev_folds) if dev_folds else None      # warm (module load, workspaces)
            t0 = time.perf_counter()
            q = get_q2y(m, device_folds=dev_folds)
            dt = time.perf_counter() - t0
            print(shape, algo, "all folds in one launch" if dev_folds else "one refit per fold", "q2y", round(q, 8), "LOO seconds", round(dt, 4),
                  "per fold ms", round(dt / shape[0] * 1e3, 4), flush=True)
