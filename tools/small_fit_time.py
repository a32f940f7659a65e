#!/usr/bin/env python3
"""tPLS.fit of small float64 problems: the one-launch form (cmtfpls_fit_small_f64, one workgroup runs the whole fit) against
the regular multi-launch engine, wall time of the estimator call (host arrays in, host arrays out)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from cmtf_pls_amd import tPLS
from cmtf_pls_amd.engine import EngineOptions


def clock(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


for shape, M, R in [((200, 10, 8), 4, 3), ((100, 38, 65), 4, 8), ((500, 16, 16), 8, 5), ((2000, 8, 8), 4, 4), ((64, 32, 32), 4, 6)]:
    x, y, _ = O.import_synthetic(shape, M, min(R, 4), error=0.1, seed=215)
    m = tPLS(R, options=EngineOptions(small_fit=True, small_fit_elements=1 << 30))
    t_one = clock(lambda: m.fit(x, y))
    n1 = list(m.n_iter_)
    be = m._get_engine().be
    xd, yd = torch.from_numpy(x).cuda().view(shape[0], -1), torch.from_numpy(y.reshape(shape[0], -1)).cuda()
    A, B = (1, shape[1]) if len(shape) == 2 else (shape[1], int(np.prod(shape[2:])))
    t_kernel = clock(lambda: be.fit_small(xd, yd, A, B, R, 1e-8, 100)) if be.fit_small(xd, yd, A, B, R, 1e-8, 100) is not None else float("nan")
    m = tPLS(R, options=EngineOptions(small_fit=False))
    t_reg = clock(lambda: m.fit(x, y))
    print(f"{str(shape):16s} M={M} R={R} ({int(np.prod(shape))} elements, {sum(n1)} iterations): one launch {t_one*1e3:7.3f} ms "
          f"(kernel + read-back {t_kernel*1e3:7.3f} ms) | regular engine {t_reg*1e3:7.3f} ms | n_iter equal: {n1 == list(m.n_iter_)}", flush=True)
