#!/usr/bin/env python3
"""transform / predict of samples WITH missing values at BASELINE configs[3] (65536 x 128 x 128 f32, 30 % NaN):
the row-in-registers form (cmtfpls_project_rows_*: one read of the raw X) against the passes it replaces
(clone + centre + R x fused score_deflate)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.synthetic import synthetic_shard_device
from cmtf_pls_amd.tpls import to_device_copy


def clock(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


X, Y = synthetic_shard_device((65536, 128, 128), 16, 10, error=0.1, device="cuda:0", nan_fraction=0.3, seed=217)
m = tPLS(10, dtype="float32", algorithm="xcov")
m.fit(X, Y, max_iter=30)
eng = m._get_engine()
t_new, s_new = clock(lambda: m.transform(X))
t_ro, _ = clock(lambda: eng.project_readonly(m._state, [X]))
t_old, s_old = clock(lambda: eng.project(m._state, [to_device_copy(X, torch.float32, "cuda:0")]).cpu().numpy())
import numpy as np
d = np.nanmax(np.abs(s_new - s_old)) / np.nanmax(np.abs(s_old))
print(f"transform of 65536 x 128 x 128 f32 with 30 % NaN, R = 10: API {t_new*1e3:.2f} ms (engine.project_readonly {t_ro*1e3:.2f} ms: "
      f"MTTKRP attempt + NaN flag + row-in-registers kernel) | clone + centre + 10 score_deflate passes {t_old*1e3:.2f} ms | "
      f"max |diff| / max|scores| = {d:.1e}")
t_p, _ = clock(lambda: m.predict(X))
print(f"predict: {t_p*1e3:.2f} ms")
